#!/usr/bin/env python3
"""Probe for the verdict's item 5 (round 6): is ~47 us the floor of the coarsest level's 3x3 convs (784 rows x 1280 x 11 520 / 23 040)?
Sweeps the tiled template's tile shape (tt_gemm_set_tile_override) for ONE split factor per process (TT_GEMM_SPLITK is read once) on COLD
weights -- the step reads every weight from HBM once, so the probe cycles through enough weight copies to exceed the 256 MB MALL -- and
prints us per launch including the split-K reduction pass.
    for s in 1 2 3 4 6 8 12 16; do TT_GEMM_SPLITK=$s python tools/coarse_conv_probe.py; done"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import _lib, ops


def main():
    lib = _lib.load()
    dt, dev = torch.bfloat16, "cuda"
    split = os.environ.get("TT_GEMM_SPLITK", "plan")
    for nimg, h, w_, cin, cout in ((28, 4, 7, 1280, 1280), (28, 4, 7, 2560, 1280), (28, 8, 14, 1280, 1280)):
        m = nimg * h * w_
        x = torch.randn(m, cin, device=dev, dtype=dt)
        copies = max(2, int(400e6 // (cout * 9 * cin * 2)) + 1)
        wts = [torch.randn(cout, 9 * cin, device=dev, dtype=dt) * 0.02 for _ in range(copies)]
        out = torch.empty(m, cout, device=dev, dtype=dt)
        row = []
        for cfg in (-1, 16, 10, 11, 15, 3, 13, 19, 17, 9):
            lib.tt_gemm_set_tile_override(cfg)
            try:
                run = lambda i: ops.gemm(x, wts[i % copies], mode=1, conv=(nimg, h, w_, h, w_, 1, 0), out=out)
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
                plan = (torch.zeros(7, dtype=torch.int32)).tolist()
                row.append(f"cfg {cfg:3d}: {us:6.1f}")
            except Exception as e:                                   # a configuration the template does not build for this mode
                row.append(f"cfg {cfg:3d}:   n/a ")
        lib.tt_gemm_set_tile_override(-1)
        print(f"split {split:>4s}  conv M={m:5d} K=9x{cin:4d} N={cout}: " + " | ".join(row) + "   us per launch (cold weights, graph of launches)")


if __name__ == "__main__":
    main()
