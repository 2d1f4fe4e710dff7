#!/usr/bin/env python3
"""Per-shape census of one denoise step with the BINDING FLOOR of every MFMA launch next to its time (round 6, the method of DESIGN.md 6.C applied
to the whole step): MFMA floor at the sustained clock (1 900 TFLOP/s), HBM floor at 8 TB/s on the algorithmic bytes (operands + result + residual
once, 2 bytes per element), the larger of the two, and measured / floor -- sorted by the time ABOVE the floor (where the step's headroom is).
    python tools/floor_census.py [vgl|vl] [lo|ref|hi]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from this_and_that_vdm_amd import ops


def main():
    mode, res = (sys.argv[1:] + ["vgl", "lo"])[:2]
    dev = torch.device("cuda", 0)
    unet, cn, _, _ = bench.build_models(mode, torch.bfloat16, dev, 0, 1)
    loop, args = bench.make_loop(unet, cn, res, dev, 0)
    loop.use_graph = False
    loop.overlap_branches = False
    loop.step(); torch.cuda.synchronize()
    ops.PROFILE = []
    loop.step(); torch.cuda.synchronize()
    rec, ops.PROFILE = ops.PROFILE, None
    agg = {}
    for name, fl, e0, e1, shape in rec:
        a = agg.setdefault((shape, name), [0, 0.0, 0.0])
        a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1) * 1e-3
    rows = []
    for (shape, name), (n, fl, t) in agg.items():
        if shape[0] == "attn":
            _, bh, lq, lk, mask, _ = shape
            byts = 2.0 * 64 * (2 * bh * lq + 2 * bh * lk) if mask == 0 else 2.0 * 320 * 0 + 2.0 * 64 * 2 * bh * lq     # q + out (+ k, v for self-attention)
        else:
            md, m, nn, k, geglu, has_res = shape
            a_bytes = 2.0 * m * (k // (9 if md == 1 else 3 if md == 2 else 1))            # the conv / temporal conv reads its input once (taps come from cache)
            byts = a_bytes + 2.0 * nn * k + 2.0 * m * (nn // 2 if geglu else nn) * (2 if has_res else 1)
        f_mfma, f_hbm = fl / n / 1.9e15, byts / 8e12
        floor = max(f_mfma, f_hbm)
        rows.append((t - n * floor, shape, name, n, fl / n, t / n, f_mfma, f_hbm, floor))
    tot_t = sum(r[5] * r[3] for r in rows); tot_floor = sum(r[8] * r[3] for r in rows)
    print(f"{sum(r[3] for r in rows)} MFMA launches, {tot_t * 1e3:.2f} ms event-timed (eager, serial); sum of binding floors {tot_floor * 1e3:.2f} ms = {100 * tot_floor / tot_t:.0f} %")
    print("shape (mode, M, N, K, geglu, res | attn, BH, Lq, Lk, mask)            n   GFLOP     us   MFMA-floor  HBM-floor   x floor   ms above floor   kernel")
    for above, shape, name, n, fl, t, fm, fh, floor in sorted(rows, key=lambda r: -r[0]):
        print(f"{str(shape):58s} {n:3d} {fl / 1e9:7.1f} {t * 1e6:6.1f} {fm * 1e6:9.1f} {fh * 1e6:10.1f} {t / floor:8.1f} {above * 1e3:12.2f}      {name[:60]}")


if __name__ == "__main__":
    main()
