#!/usr/bin/env python3
"""K scan of the persistent ping-pong GEMM (gemm_pp.hip) against the tiled kernel and the vendor library (GPU box):
time per tile round = a + b * (K / 64) separates the main-loop rate from the per-tile overhead (epilogue, pipeline restart).
python tools/pp_scan.py [M N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops, _lib


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    a = [int(x) for x in sys.argv[1:]]
    m, n = (a + [50176, 2560])[:2]
    lib = _lib.load()
    dt, dev = torch.bfloat16, "cuda"
    tiles = -(-m // 256) * -(-n // 256)
    rounds = -(-tiles // 256)
    print(f"M={m} N={n}: {tiles} tiles of 256x256 = {tiles / 256:.2f} rounds ({rounds} for the slowest CU)")
    print(f"{'K':>6s} {'mode':>9s} {'pp us':>8s} {'pp TF':>7s} {'us/tile':>8s} {'tiled TF':>9s} {'lib TF':>7s}")
    for mode in ("plain", "ln", "ln+geglu"):
        pts = []
        for k in (128, 320, 640, 1280, 2560, 5120):
            x = torch.randn(m, k, device=dev, dtype=dt); w = (torch.randn(n, k, device=dev) * k ** -0.5).to(dt)
            b = torch.randn(n, device=dev)
            args = dict(bias=b, geglu=mode.endswith("geglu"), ln_fold=1 if mode.startswith("ln") else 0, ln_eps=1e-5)
            out = torch.empty(m, n // 2 if args["geglu"] else n, device=dev, dtype=dt)
            fl = 2.0 * m * n * k
            lib.tt_gemm_set_tile_override(-1)
            tp = timeit(lambda: ops.gemm(x, w, out=out, **args))
            lib.tt_gemm_set_tile_override(11)
            tt = timeit(lambda: ops.gemm(x, w, out=out, **args))
            lib.tt_gemm_set_tile_override(-1)
            tl = None
            if mode == "plain":
                tl = timeit(lambda: torch.addmm(b.to(dt), x, w.t(), out=out))
            pts.append((k / 64, tp * 1e6 / rounds))
            print(f"{k:6d} {mode:>9s} {tp * 1e6:8.1f} {fl / tp / 1e12:7.0f} {tp * 1e6 / rounds:8.2f} {fl / tt / 1e12:9.0f} " + (f"{fl / tl / 1e12:7.0f}" if tl else "      -"))
        # least-squares line through the points with K >= 320
        xs = [p[0] for p in pts[1:]]; ys = [p[1] for p in pts[1:]]
        mx, my = sum(xs) / len(xs), sum(ys) / len(ys)
        slope = sum((u - mx) * (v - my) for u, v in zip(xs, ys)) / sum((u - mx) ** 2 for u in xs)
        icpt = my - slope * mx
        print(f"   {mode}: per tile {icpt:.2f} us + {slope:.3f} us per 64-deep slab  (main loop alone = {2 * 256 * 256 * 64 * 256 / slope / 1e6:.0f} TFLOP/s on 256 CUs)")


if __name__ == "__main__":
    main()
