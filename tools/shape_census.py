#!/usr/bin/env python3
"""Per-shape census of the MFMA launches of one denoise step (GPU box): count, GFLOP, time, TFLOP/s."""
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
        a = agg.setdefault(shape, [0, 0.0, 0.0, name])
        a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1) * 1e-3
    tot_t = sum(v[2] for v in agg.values()); tot_f = sum(v[1] for v in agg.values())
    print(f"total {tot_f/1e12:.2f} TFLOP in {tot_t*1e3:.1f} ms (event-timed, eager)")
    print("mode      M      N      K gg res |  n   GFLOP/launch   us/launch   TF/s   ms/step  %time")
    for shape, (n, fl, t, name) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        print(f"{str(shape[0]):>4} {shape[1]:6d} {shape[2]:6d} {shape[3]:6d} {shape[4]:2d} {shape[5]:2d} | {n:3d} {fl/n/1e9:10.1f} {t/n*1e6:10.1f} {fl/t/1e12:7.1f} {t*1e3:8.2f} {100*t/tot_t:5.1f}  {name.split('<')[1][:-1] if '<' in name else ''}")

if __name__ == "__main__":
    main()
