#!/usr/bin/env python3
"""Probe for a split-K route of gemm_w320h_kernel at the two coarsest levels: per-tile time of the 128 x 320 kernel as a function
of K on problems it does not fill the chip with (TT_W320_FORCE=2), against the planner's tiled route (GPU box).
    python tools/w320_split_probe.py tiled | TT_W320_FORCE=2 python tools/w320_split_probe.py w320h"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from this_and_that_vdm_amd import _lib, ops
from w320_bench import graph_time


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "?"
    lib = _lib.load()
    lib.tt_gemm_set_big_tile(0 if tag == "tiled" else 3)
    dt, dev = torch.bfloat16, "cuda"
    r = lambda *s: torch.randn(*s, device=dev, dtype=dt)
    cases = []
    for m, n, k in ((3136, 1280, 1280), (3136, 1280, 2560), (3136, 1280, 5120), (784, 1280, 1280), (784, 1280, 5120)):
        a, w, x = r(m, k), r(n, k) * k ** -0.5, r(m, n)
        cases.append((f"linear {m}x{n}x{k} +res", 2.0 * m * n * k, (a, w), dict(bias=torch.randn(n, device=dev), residual=x, out=torch.empty(m, n, device=dev, dtype=dt))))
    for nimg, h, w_, c0, c1, co in ((28, 8, 14, 1280, 0, 1280), (28, 8, 14, 1280, 1280, 1280), (28, 8, 14, 1280, 640, 1280), (28, 4, 7, 1280, 0, 1280), (28, 4, 7, 1280, 1280, 1280)):
        m = nimg * h * w_
        x0, x1 = r(m, c0), (r(m, c1) if c1 else None)
        wt = r(co, 9 * (c0 + c1)) * (9 * (c0 + c1)) ** -0.5
        cases.append((f"conv3x3 M={m} cin={c0}+{c1} cout={co} +film", 2.0 * m * co * 9 * (c0 + c1), (x0, wt),
                      dict(a1=x1, mode=1, conv=(nimg, h, w_, h, w_, 1, 0), bias=torch.randn(co, device=dev), rowvec=torch.randn(2, co, device=dev),
                           rowvec_rows=14 * h * w_, out=torch.empty(m, co, device=dev, dtype=dt))))
    for name, flops, a, kw in cases:
        ops.PROFILE = []
        ops.gemm(*a, **kw)
        torch.cuda.synchronize()
        kn = ops.PROFILE[0][0]
        ops.PROFILE = None
        t = graph_time(lambda: ops.gemm(*a, **kw))
        print(f"{tag:6s} {name:52s} {t * 1e6:8.1f} us {flops / t / 1e12:6.0f} TF/s  [{kn}]")


if __name__ == "__main__":
    main()
