#!/usr/bin/env python3
"""Does a big-tile launch slow down when more CUs work?  The finest level has 196 tiles of 256 x 320 on 256 CUs (50176 = 2^10 * 7^2 rows);
this runs the same problems with 196 / 224 / 252 tiles (28 / 32 / 36 images of 32 x 56) and 98 / 112 / 126 of the next level's 128 x 320 x 2
tiles.  Equal times = the CUs do not share a power / bandwidth budget at this load and a 224-row tile (224 tiles, 87.5 % of the CUs instead
of 76.6 %) would be worth its rows; times growing with the tile count = it would not.       python tools/fill_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops
from w320_bench import graph_time

dt, dev = torch.bfloat16, "cuda"
r = lambda *s: torch.randn(*s, device=dev, dtype=dt)


def main():
    print(f"{'problem':46s} " + " ".join(f"{'img ' + str(n):>16s}" for n in (28, 32, 36)))
    for name, lvl, kind, k in (("conv3x3 320->320 +film (256x320 tiles)", 0, "conv", 320), ("linear K=1280 +res (FF2)", 0, "lin", 1280), ("linear K=320 +res", 0, "lin", 320),
                               ("conv3x3 640->640 +film (128x320 tiles)", 1, "conv", 640), ("linear 640x2560 +res (tiled FF2)", 1, "lin", 2560)):
        row = []
        for nimg in (28, 32, 36):
            h, w = 32 >> lvl, 56 >> lvl
            c = 320 << lvl
            m = nimg * h * w
            if kind == "conv":
                x, wt = r(m, k), r(c, 9 * k) * (9 * k) ** -0.5
                kw = dict(mode=1, conv=(nimg, h, w, h, w, 1, 0), bias=torch.randn(c, device=dev), rowvec=torch.randn(2, c, device=dev), rowvec_rows=(nimg // 2) * h * w,
                          out=torch.empty(m, c, device=dev, dtype=dt))
                fl = 2.0 * m * c * 9 * k
            else:
                x, wt = r(m, k), r(c, k) * k ** -0.5
                kw = dict(bias=torch.randn(c, device=dev), residual=r(m, c), out=torch.empty(m, c, device=dev, dtype=dt))
                fl = 2.0 * m * c * k
            ops.PROFILE = []
            ops.gemm(x, wt, **kw); torch.cuda.synchronize()
            kn = ops.PROFILE[0][0]; ops.PROFILE = None
            t = graph_time(lambda: ops.gemm(x, wt, **kw))
            row.append(f"{t * 1e6:7.1f} us {fl / t / 1e12:5.0f}")
        print(f"{name:46s} " + " ".join(f"{x:>16s}" for x in row) + f"   [{kn}]")


if __name__ == "__main__":
    main()
