#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 counter passes).  python tools/one_gemm.py cfg m n k [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops, _lib
cfg, m, n, k = map(int, sys.argv[1:5]); iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
lib = _lib.load(); lib.tt_gemm_set_tile_override(cfg)
a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16); w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
for _ in range(iters): ops.gemm(a, w, out=out)
torch.cuda.synchronize()
