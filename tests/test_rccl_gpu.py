"""GPU: the RCCL transport itself on the ONE-GPU box.  The N > 1 tests elsewhere run over gloo (two ranks cannot share a GPU under
RCCL: "duplicate GPU detected"), so until the driver has a multi-GPU node the "nccl" backend would never have been initialised at all.
This test brings up a ONE-rank "nccl" process group on cuda:0 in a child process -- bootstrap, topology discovery and communicator
creation are what fail on a misconfigured host (HSA_ENABLE_IPC_MODE_LEGACY, missing librccl) -- and issues on it exactly the collectives
`this_and_that_vdm_amd.dist` and `bench.py` use for N > 1: the broadcast of the flat weight buffer, the MAX all-reduce of the window
time, the all-gather of the per-rank numbers and the barrier.  It proves that the transport starts on this image and that the calls
are well-formed for device tensors; it says nothing about xGMI bandwidth (SCALE is the driver's to measure)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import datetime, json, os, sys
sys.path.insert(0, os.environ["TT_REPO"])
import torch
import torch.distributed as dist
from this_and_that_vdm_amd.dist import flat_param_buffer
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120), device_id=dev)
out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
model = torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.Linear(96, 32)).to(dev, torch.bfloat16)
before = [p.detach().clone() for p in model.parameters()]
flat = flat_param_buffer(model)                       # one buffer, parameters re-homed as views of it
dist.broadcast(flat, src=0)                           # what broadcast_model_ issues for N > 1
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)              # max_over_ranks
ones = torch.ones(1, dtype=torch.float32, device=dev)
dist.all_reduce(ones)                                 # the rank count bench.py reports as config.rccl_ranks
mine = torch.tensor([3.0, 4.0], dtype=torch.float64, device=dev)
allv = [torch.zeros_like(mine)]
dist.all_gather(allv, mine)                           # gather_floats
dist.barrier()
torch.cuda.synchronize()
out["params_kept"] = all(torch.equal(a, b) for a, b in zip(before, model.parameters()))
out["params_are_views"] = all(p.data_ptr() >= flat.data_ptr() and p.data_ptr() < flat.data_ptr() + flat.numel() * 2 for p in model.parameters())
out["max"] = float(t.item()); out["ranks"] = float(ones.item()); out["gathered"] = [float(x) for x in allv[0]]
dist.destroy_process_group()
print(json.dumps(out), flush=True)
"""


def test_one_rank_rccl_group_runs_the_collectives_of_the_n_gpu_path():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import socket
    with socket.socket() as sk:                               # a free rendezvous port (the suite may share the host with other jobs)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, TT_REPO=REPO, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["world"] == 1
    assert d["params_kept"] and d["params_are_views"]
    assert d["max"] == 1.25 and d["ranks"] == 1.0 and d["gathered"] == [3.0, 4.0]
