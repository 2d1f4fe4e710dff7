#!/usr/bin/env python3
"""The N = 320 t problems of the finest UNet level on the 256 x 320 big-tile kernel (gemm_w320.hip) against the tiled template,
same operands, epilogues as in the step (GPU box).  Each problem is captured in a hipGraph of `reps` launches and replayed, so
the numbers are back-to-back kernel times without launch gaps.        python tools/w320_bench.py [--hi]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from this_and_that_vdm_amd import _lib, ops


def graph_time(fn, reps=10, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e-3


def main():
    hi = "--hi" in sys.argv
    lib = _lib.load()
    dt, dev = torch.bfloat16, "cuda"
    nimg, h, w = 28, (64 if hi else 32), (112 if hi else 56)
    m = nimg * h * w
    r = lambda *s: torch.randn(*s, device=dev, dtype=dt)
    cases = []
    for n, k, res, ln in ((320, 1280, True, 0), (320, 320, True, 0), (320, 320, False, 0), (320, 320, False, 1), (960, 320, False, 1),
                          (640, 320, False, 0), (320, 640, False, 0), (320, 960, False, 0)):
        a, wt, x = r(m, k), r(n, k) * k ** -0.5, r(m, n)
        bias = torch.randn(n, device=dev)
        kw = dict(bias=bias, ln_fold=ln, ln_eps=1e-5)
        if res:
            kw["residual"] = x
        out = torch.empty(m, n, device=dev, dtype=dt)
        cases.append((f"linear {m}x{n}x{k}{' +res' if res else ''}{' LN' if ln else ''}", 2.0 * m * n * k, (a, wt), dict(kw, out=out)))
    for cin0, cin1, cout in ((320, 0, 320), (320, 320, 320), (640, 320, 320)):
        x0, x1 = r(m, cin0), (r(m, cin1) if cin1 else None)
        cin = cin0 + cin1
        wt = r(cout, 9 * cin) * (9 * cin) ** -0.5
        film = torch.randn(2, cout, device=dev)
        out = torch.empty(m, cout, device=dev, dtype=dt)
        cases.append((f"conv3x3 M={m} cin={cin0}+{cin1} cout={cout} +film", 2.0 * m * cout * 9 * cin, (x0, wt),
                      dict(a1=x1, mode=1, conv=(nimg, h, w, h, w, 1, 0), bias=torch.randn(cout, device=dev), rowvec=film,
                           rowvec_rows=14 * h * w, out=out)))
    x, wt, res = r(m, 320), r(320, 960) * 960 ** -0.5, r(m, 320)
    cases.append((f"tconv M={m} 320->320 +blend", 2.0 * m * 320 * 960, (x, wt),
                  dict(mode=2, tconv=(14, h * w), bias=torch.randn(320, device=dev), residual=res, blend=res, alpha=0.3,
                       out=torch.empty(m, 320, device=dev, dtype=dt))))
    # the next level (half the height and width, C = 640): 128-row tiles at 32x56 latents; and the live rows of the finest level
    h2, w2 = h // 2, w // 2
    m2 = nimg * h2 * w2
    for mm, n, k, res, ln in ((m2, 640, 2560, True, 0), (m2, 640, 640, True, 0), (m2, 1920, 640, False, 1), (m // 2, 320, 320, True, 0), (m // 2, 320, 320, False, 1)):
        a, wt, x = r(mm, k), r(n, k) * k ** -0.5, r(mm, n)
        kw = dict(bias=torch.randn(n, device=dev), ln_fold=ln, ln_eps=1e-5)
        if res:
            kw["residual"] = x
        cases.append((f"linear {mm}x{n}x{k}{' +res' if res else ''}{' LN' if ln else ''}", 2.0 * mm * n * k, (a, wt),
                      dict(kw, out=torch.empty(mm, n, device=dev, dtype=dt))))
    for cin0, cin1, cout in ((640, 0, 640), (640, 640, 640), (1280, 640, 640)):
        x0, x1 = r(m2, cin0), (r(m2, cin1) if cin1 else None)
        cin = cin0 + cin1
        wt = r(cout, 9 * cin) * (9 * cin) ** -0.5
        cases.append((f"conv3x3 M={m2} cin={cin0}+{cin1} cout={cout} +film", 2.0 * m2 * cout * 9 * cin, (x0, wt),
                      dict(a1=x1, mode=1, conv=(nimg, h2, w2, h2, w2, 1, 0), bias=torch.randn(cout, device=dev), rowvec=torch.randn(2, cout, device=dev),
                           rowvec_rows=14 * h2 * w2, out=torch.empty(m2, cout, device=dev, dtype=dt))))
    x, wt, res = r(m2, 640), r(640, 1920) * 1920 ** -0.5, r(m2, 640)
    cases.append((f"tconv M={m2} 640->640 +blend", 2.0 * m2 * 640 * 1920, (x, wt),
                  dict(mode=2, tconv=(14, h2 * w2), bias=torch.randn(640, device=dev), residual=res, blend=res, alpha=0.3,
                       out=torch.empty(m2, 640, device=dev, dtype=dt))))
    print(f"{'problem':58s} {'tiled us':>9s} {'TF/s':>6s} | {'w320 us':>8s} {'TF/s':>6s} | speed-up")
    for name, flops, a, kw in cases:
        t = {}
        for route in ("tiled", "w320"):
            lib.tt_gemm_set_big_tile(0 if route == "tiled" else 1)       # tiled: the planner's own choice among the tiled kernels
            try:
                ops.PROFILE = []
                ops.gemm(*a, **kw)
                torch.cuda.synchronize()
                kn = ops.PROFILE[0][0]
                ops.PROFILE = None
                t[route] = (graph_time(lambda: ops.gemm(*a, **kw)), kn)
            finally:
                lib.tt_gemm_set_big_tile(1)
        tt, tw = t["tiled"][0], t["w320"][0]
        print(f"{name:58s} {tt * 1e6:9.1f} {flops / tt / 1e12:6.0f} | {tw * 1e6:8.1f} {flops / tw / 1e12:6.0f} | {tt / tw:5.2f}x   [{t['tiled'][1]} -> {t['w320'][1]}]")


if __name__ == "__main__":
    main()
