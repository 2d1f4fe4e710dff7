#!/usr/bin/env python3
"""Which hipBLASLt kernels (macro tile, wave tiling, prefetch knobs in the name) serve the step's linear shapes -- yardstick only.
rocprofv3 --kernel-trace --stats -- python tools/lib_kernels.py"""
import torch
for m, n, k in ((12544, 5120, 640), (3136, 10240, 1280), (12544, 640, 2560), (3136, 1280, 5120), (50176, 320, 1280), (50176, 2560, 320), (50176, 320, 320), (12544, 640, 640)):
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16); w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
