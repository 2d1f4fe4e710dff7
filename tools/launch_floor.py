#!/usr/bin/env python3
"""What one launch costs inside a hipGraph, whatever it computes: chains of dependent launches of (a) a one-block elementwise kernel,
(b) a one-tile GEMM, (c) full-chip GEMMs with 1 / 2 / 5 / 10 / 20 K steps per tile, replayed; per-launch time = replay / launches.
The intercept of (c) over the K steps is the fixed cost of a one-tile-per-CU launch (dispatch, first fill, epilogue, drain).
python tools/launch_floor.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops


def graph_time(fn, reps=50, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3      # us per launch


def main():
    dt, dev = torch.bfloat16, "cuda"
    r = lambda *s: torch.randn(*s, device=dev, dtype=dt)
    x = r(64, 320); rv = torch.randn(1, 320, device=dev)
    print(f"one-block elementwise (tt_add_rowvec 64x320, in place): {graph_time(lambda: ops.add_rowvec(x, rv, 64, 1, out=x)):6.2f} us / launch")
    a, w = r(128, 64), r(128, 64); o = torch.empty(128, 128, device=dev, dtype=dt)
    print(f"one-tile GEMM 128x128x64:                               {graph_time(lambda: ops.gemm(a, w, out=o)):6.2f} us / launch")
    for m, n in ((12544, 640), (3136, 1280), (50176, 320), (25088, 320), (784, 1280)):
        line = []
        for k in (64, 128, 320, 640, 1280, 2560):
            a, w = r(m, k), r(n, k) * k ** -0.5
            res = r(m, n); o = torch.empty(m, n, device=dev, dtype=dt)
            bias = torch.randn(n, device=dev)
            t = graph_time(lambda: ops.gemm(a, w, bias=bias, residual=res, out=o), reps=20)
            line.append(f"K={k}: {t:6.1f}")
        print(f"GEMM {m}x{n} + bias + residual, us / launch:  " + "  ".join(line))
    # the same with NO dependency between consecutive launches' data (different buffers): is it the dependency or the launch?
    m, n, k = 12544, 640, 640
    bufs = [(r(m, k), torch.empty(m, n, device=dev, dtype=dt)) for _ in range(4)]
    w = r(n, k) * k ** -0.5
    i = [0]
    def rot():
        a, o = bufs[i[0] % 4]; i[0] += 1
        ops.gemm(a, w, out=o)
    print(f"GEMM {m}x{n}x{k} plain, rotating buffers:               {graph_time(rot, reps=20):6.1f} us / launch")


if __name__ == "__main__":
    main()
