#!/usr/bin/env python3
"""GroupNorm from the producer's tile sums (tt_groupnorm_tiles) against the statistics-pass kernels on the step's shapes, per image and
across frames; the producer's extra cost (tt_gemm with / without stats_out); and how many GroupNorms of one denoise step take which route.
python tools/gn_tiles_bench.py [--count]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops, _lib
from tools.gn_bench import graph_time


def main():
    dt, nimg, frames = torch.bfloat16, 28, 14
    lib = _lib.load()
    for hw, c, r in ((1792, 320, 256), (448, 640, 128), (112, 1280, 128), (1536, 320, 256), (7168, 320, 256)):
        rows = nimg * hw
        x = torch.randn(rows, c, device="cuda").to(dt)
        g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
        xf = x.float()
        sb = torch.stack([xf.view(rows // r, r, c).sum(1), (xf * xf).view(rows // r, r, c).sum(1)], 1).contiguous() if rows % r == 0 else None
        for fpg in (1, frames):
            seg = fpg * hw
            old = graph_time(lambda: ops.groupnorm(x, None, nimg, hw, fpg, g, b, 1e-5, True))
            line = f"hw {hw:5d} C {c:4d} {'per image ' if fpg == 1 else 'cross-frame'}: statistics pass {old:6.1f} us"
            if sb is not None and lib.tt_groupnorm_tiles_supported(seg, c, r, ops._code(dt)):
                x._tt_stats = (sb, r)
                new = graph_time(lambda: ops.groupnorm(x, None, nimg, hw, fpg, g, b, 1e-5, True))
                del x._tt_stats
                line += f" | from tile sums {new:6.1f} us ({rows * c * 4 / new / 1e6:5.2f} TB/s in+out)"
            sc, sh = torch.ones(nimg, c, device="cuda"), torch.zeros(nimg, c, device="cuda")
            ap = graph_time(lambda: ops.groupnorm_apply(x, None, nimg, hw, sc, sh, True))
            print(line + f" | apply alone {ap:6.1f} us")
    # producer side: conv3x3 320 -> 320 at 32x56 and the temporal conv, with and without the statistics epilogue
    m = 28 * 1792
    a = torch.randn(m, 320, device="cuda").to(dt)
    for name, w, kw in (("conv3x3 320->320", torch.randn(320, 2880, device="cuda").to(dt) * 0.02, dict(mode=1, conv=(28, 32, 56, 32, 56, 1, 0))),
                        ("tconv 320->320 + blend", torch.randn(320, 960, device="cuda").to(dt) * 0.03, dict(mode=2, tconv=(14, 1792), residual=a, blend=a, alpha=0.3)),
                        ("linear 320x320 + res", torch.randn(320, 320, device="cuda").to(dt) * 0.05, dict(residual=a))):
        o = torch.empty(m, 320, device="cuda", dtype=dt)
        t0 = graph_time(lambda: ops.gemm(a, w, out=o, **kw))
        t1 = graph_time(lambda: ops.gemm(a, w, out=o, stats=14 * 1792, **kw))
        print(f"producer {name:24s}: {t0:6.1f} us without, {t1:6.1f} us with stats_out")
    if "--count" in sys.argv:
        import bench, collections
        dev = torch.device("cuda", 0)
        unet, cn, _, _ = bench.build_models("vgl", dt, dev, 0, 1)
        loop, args = bench.make_loop(unet, cn, "lo", dev, 0)
        loop.use_graph = False
        loop.step(); torch.cuda.synchronize()
        cnt = collections.Counter()
        real = ops.groupnorm
        def counting(x0, x1, nimg, hw, fpg, *a, **k):
            st = getattr(x0, "_tt_stats", None)
            route = "tiles" if (st is not None and x1 is None and x0.is_contiguous() and lib.tt_groupnorm_tiles_supported(fpg * hw, x0.shape[1], st[1], ops._code(x0.dtype))) else \
                ("two sources" if x1 is not None else ("no sums attached" if st is None else f"segment {fpg * hw} not a multiple of {st[1]}"))
            cnt[(hw, x0.shape[1] + (x1.shape[1] if x1 is not None else 0), "cross-frame" if fpg > 1 else "per image", route)] += 1
            return real(x0, x1, nimg, hw, fpg, *a, **k)
        ops.groupnorm = counting
        loop.step(); torch.cuda.synchronize()
        ops.groupnorm = real
        for k, v in sorted(cnt.items()):
            print(f"  {v:3d} x GroupNorm hw {k[0]:5d} C {k[1]:5d} {k[2]:11s} -> {k[3]}")


if __name__ == "__main__":
    main()
