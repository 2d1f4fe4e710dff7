"""`python bench.py --gpus N` must launch its own ranks: run the N = 2 launcher path on CPU with the
stubbed loop (gloo) and check the single rank-0 JSON line."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_prints_one_json_line():
    out = _run(["--gpus", "2", "--steps", "5", "--warmup", "1", "--stub-cpu"])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["scaling"] == "weak"
    # max over ranks: rank 1 sleeps 2x as long as rank 0
    assert out["ms_per_step"] >= 2 * 2.0 * 0.9
    for key in ("metric", "value", "unit", "higher_is_better", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out


def test_single_rank_needs_no_launcher():
    out = _run(["--steps", "3", "--stub-cpu"])
    assert out["n_gpus"] == 1
