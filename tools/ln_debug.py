#!/usr/bin/env python3
"""debug: what 1/sigma does the fused-LayerNorm GEMM apply?  out = rstd_kernel * (x W^T); compare with the true rstd."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops
torch.manual_seed(0)
for dtype in (torch.bfloat16, torch.float16, torch.float32):
    for m, c, n in ((128, 64, 128), (333, 320, 192), (9000, 640, 1280)):
        x = (torch.randn(m, c) * 1.5 + torch.randn(m, 1)).to(dtype)
        w = (torch.randn(n, c) * c ** -0.5).to(dtype)
        raw = ops.gemm(x.cuda(), w.cuda(), out_f32=True).float().cpu()
        got = ops.gemm(x.cuda(), w.cuda(), out_f32=True, ln_fold=1, ln_eps=1e-5).float().cpu()
        xf = x.float()
        rstd = 1 / torch.sqrt(xf.var(1, unbiased=False) + 1e-5)
        ratio = (got / raw).median(dim=1).values          # per row: the scale the kernel applied
        # candidates
        s, q = xf.sum(1), (xf * xf).sum(1)
        print(dtype, (m, c, n), "ratio/rstd: min %.4f med %.4f max %.4f" % tuple((ratio / rstd).quantile(torch.tensor([0., .5, 1.])).tolist()),
              "| rows 0..3 ratio", ratio[:4].tolist(), "rstd", rstd[:4].tolist(),
              "| 1/sqrt(q/c)", (1 / torch.sqrt(q / c + 1e-5))[:4].tolist())
