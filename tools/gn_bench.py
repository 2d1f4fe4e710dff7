#!/usr/bin/env python3
"""ops.groupnorm on the step's per-image shapes (GPU box), us per call in a hipGraph of 20 back-to-back calls.
TT_GN_GROUPED=0|1 python tools/gn_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops

def graph_time(fn, n=20, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3

def main():
    dt, nimg = torch.bfloat16, 28
    print("TT_GN_GROUPED =", os.environ.get("TT_GN_GROUPED", "1"))
    for hw, c0, c1 in ((7168, 320, 0), (7168, 320, 320), (1792, 320, 0), (1792, 320, 320), (1792, 640, 320), (448, 640, 0), (448, 640, 640), (448, 1280, 640), (448, 320, 0),
                       (112, 1280, 0), (112, 1280, 1280), (112, 640, 0), (28, 1280, 0), (28, 1280, 1280)):
        x0 = torch.randn(nimg * hw, c0, device="cuda").to(dt)
        x1 = torch.randn(nimg * hw, c1, device="cuda").to(dt) if c1 else None
        c = c0 + c1
        g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
        t = graph_time(lambda: ops.groupnorm(x0, x1, nimg, hw, 1, g, b, 1e-5, True))
        print(f"hw {hw:5d} C {c0:4d}+{c1:4d}: {t:7.1f} us   ({nimg * hw * c * 4 / 1e6 / t * 1e6 / 1e6:5.2f} TB/s in+out)")

if __name__ == "__main__":
    main()
