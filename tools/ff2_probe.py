#!/usr/bin/env python3
"""Probe (round 6): FF2 of the second / third level (12544 x 640 x 2560, 3136 x 1280 x 5120, + residual) on COLD operands -- in the step its 4C input
was just written by the GEGLU projection and its weight is read once per step -- across tile shapes of the tiled template for ONE split factor per
process (TT_GEMM_SPLITK is read once).  The tiled template's K loops run at the chip-wide LDS fill rate (~10-11 TB/s: 128 x 128 x 64 bf16 tiles = 65
flop per filled byte -> ~690 TFLOP/s at best); 256 x 256 tiles halve the fill per flop but leave too few tiles, hence the split factors.
    for s in 0 1 2 3 4 6; do TT_GEMM_SPLITK=$s python tools/ff2_probe.py; done        (0 = the planner's own choice)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import _lib, ops


def main():
    lib = _lib.load()
    dt, dev = torch.bfloat16, "cuda"
    split = os.environ.get("TT_GEMM_SPLITK", "plan")
    for m, n, k in ((12544, 640, 2560), (3136, 1280, 5120), (784, 1280, 5120), (3136, 1280, 1280), (12544, 640, 640)):
        copies = max(3, int(500e6 // ((m + n) * k * 2)) + 1)
        a = [torch.randn(m, k, device=dev, dtype=dt) for _ in range(copies)]
        w = [torch.randn(n, k, device=dev, dtype=dt) * 0.02 for _ in range(copies)]
        res = torch.randn(m, n, device=dev, dtype=dt)
        bias = torch.randn(n, device=dev)
        out = torch.empty(m, n, device=dev, dtype=dt)
        row = []
        for cfg in (-1, 16, 11, 10, 3, 13, 9, 17, 18, 6):
            lib.tt_gemm_set_tile_override(cfg)
            try:
                run = lambda i: ops.gemm(a[i % copies], w[i % copies], bias=bias, residual=res, out=out)
                for i in range(copies):
                    run(i)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for i in range(copies):
                        run(i)
                g.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    g.replay()
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / (5 * copies) * 1e3
                row.append(f"cfg {cfg:3d}: {us:6.1f}")
            except Exception:
                row.append(f"cfg {cfg:3d}:   n/a ")
        lib.tt_gemm_set_tile_override(-1)
        print(f"split {split:>4s}  {m:5d} x {n:4d} x {k:4d} + residual ({2.0 * m * n * k / 1e9:5.1f} GFLOP): " + " | ".join(row) + "   us (cold operands)")


if __name__ == "__main__":
    main()
