#!/usr/bin/env python3
"""Coarsest UNet level (784 rows = CFG batch 2 x 14 frames x 4 x 7): a chain of its launches on ONE stream with both CFG halves in a launch, against
the two halves as separate chains on TWO streams (392 rows per launch), both captured in a hipGraph.  Tells whether the fixed per-launch cost that
dominates this level (<= 70 tiles on 256 CUs) overlaps when the halves run side by side.          python tools/half_batch_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops

dt, dev = torch.bfloat16, "cuda"
r = lambda *s: torch.randn(*s, device=dev, dtype=dt)


def chain(nimg, h, w, c, x, wc, wt, wl, bias, reps):
    """reps x [conv3x3 -> temporal conv -> linear + residual] on nimg images"""
    m = nimg * h * w
    for _ in range(reps):
        y = ops.gemm(x, wc, mode=1, conv=(nimg, h, w, h, w, 1, 0), bias=bias)
        z = ops.gemm(y, wt, mode=2, tconv=(14, h * w), bias=bias, residual=y, blend=y, alpha=0.3)
        x = ops.gemm(z, wl, bias=bias, residual=z)
    return x


def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    reps = 6
    for (h, w, c) in ((4, 7, 1280), (8, 14, 1280)):
        wc, wt, wl = r(c, 9 * c) * (9 * c) ** -0.5, r(c, 3 * c) * (3 * c) ** -0.5, r(c, c) * c ** -0.5
        bias = torch.randn(c, device=dev)
        x = r(28 * h * w, c)
        one = timed(lambda: chain(28, h, w, c, x, wc, wt, wl, bias, reps))
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        half = 14 * h * w

        def two():
            cur = torch.cuda.current_stream()
            ev = torch.cuda.Event(); ev.record(cur)
            outs = []
            for s, lo in ((s1, 0), (s2, half)):
                s.wait_event(ev)
                with torch.cuda.stream(s):
                    outs.append(chain(14, h, w, c, x[lo:lo + half], wc, wt, wl, bias, reps))
                    e = torch.cuda.Event(); e.record(s)
                cur.wait_event(e)
            return outs
        both = timed(two)
        print(f"{28 * h * w} rows x {c}: {3 * reps} launches on one stream {one:7.1f} us | the two halves on two streams {both:7.1f} us  ({one / both:4.2f}x)")


if __name__ == "__main__":
    main()
