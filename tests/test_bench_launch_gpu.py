"""GPU: the N > 1 code path of bench.py on a ONE-GPU box -- `python bench.py --gpus 2` self-launches two ranks under
torch.distributed.run exactly as the driver does; with TT_BENCH_ONE_GPU=1 both ranks use cuda:0 and talk over gloo, so the
launcher, the flat-buffer weight broadcast from rank 0, the barrier-bracketed timing, the max over ranks and the rank-0 JSON
line all run (only the numbers mean nothing: two replicas share one GPU).  Config 4 (8 x MI355X over RCCL / xGMI) itself is
unmeasured on hardware until the driver has an 8-GPU node."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_broadcast_weights_and_report_one_line():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, TT_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-kernel-profile"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 prints ONE line, the other rank nothing
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    assert cfg["finite_output"] is True
    assert cfg["weights_identical_on_all_ranks"] is True     # rank 1 was never filled: it holds rank 0's bytes after the broadcast
    assert cfg["weight_broadcast_s"] is not None and cfg["weight_broadcast_s"] > 0
    assert len(cfg["ms_per_step_per_rank"]) == 2 and all(v > 0 for v in cfg["ms_per_step_per_rank"])
    assert abs(d["ms_per_step"] - max(cfg["ms_per_step_per_rank"])) < 1e-6 * d["ms_per_step"] + 1e-9    # max over ranks
    assert abs(d["value"] - 2 * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * d["value"]   # whole-job aggregate
