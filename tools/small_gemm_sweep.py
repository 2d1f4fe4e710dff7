#!/usr/bin/env python3
"""Tile configurations of the tiled template on the step's SMALL linears (bias + residual, in place): the half-row output projections of
the cross-attentions (live context class only), the K = C projections of the second and third level and the 784-row problems of the
coarsest one -- launches that sit at the ~8.5 us floor of a launch.  Graph replays of 10 launches.     python tools/small_gemm_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops, _lib
from w320_bench import graph_time

SHAPES = [(25088, 320, 320), (6272, 640, 640), (1568, 1280, 1280), (12544, 640, 640), (3136, 1280, 1280), (784, 1280, 1280), (392, 1280, 1280),
          (784, 1280, 2560), (784, 1280, 5120), (784, 3840, 1280), (3136, 640, 640)]
CFGS = [-1, 2, 1, 0, 5, 11, 16, 15, 10]


def main():
    lib = _lib.load()
    dt, dev = torch.bfloat16, "cuda"
    print(f"{'shape (bias + residual in place)':36s}" + "".join(f"  cfg{c:>3d}" for c in CFGS) + "   (us per launch; * = best)")
    for m, n, k in SHAPES:
        a = torch.randn(m, k, device=dev, dtype=dt); w = torch.randn(n, k, device=dev, dtype=dt) * k ** -0.5
        x = torch.randn(m, n, device=dev, dtype=dt); bias = torch.randn(n, device=dev)
        r = []
        for c in CFGS:
            lib.tt_gemm_set_tile_override(c)
            try:
                r.append(graph_time(lambda: ops.gemm(a, w, bias=bias, residual=x, out=x)) * 1e6)
            except Exception as e:
                r.append(float("nan"))
        lib.tt_gemm_set_tile_override(-1)
        best = min(v for v in r if v == v)
        print(f"lin {m:6d}x{n:5d}x{k:5d}".ljust(36) + "".join(f" {v:7.1f}{'*' if v == best else ' '}" for v in r))


if __name__ == "__main__":
    main()
