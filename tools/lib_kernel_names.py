#!/usr/bin/env python3
"""Yardstick helper: which vendor-library kernels torch.matmul picks for the model's GEMM shapes (run under rocprofv3 --kernel-trace)."""
import torch
SHAPES = [(8192, 8192, 8192), (50176, 2560, 320), (12544, 5120, 640), (3136, 10240, 1280), (50176, 320, 1280), (12544, 640, 2560), (3136, 1280, 5120)]
for m, n, k in SHAPES:
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16); w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
