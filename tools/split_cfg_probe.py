#!/usr/bin/env python3
"""A/B of DenoiseLoop(split_cfg=...) on the bench workload (GPU box): CFG halves as separate graph branches vs one joint launch."""
import os
import sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
unet, cn, _, _ = bench.build_models("vgl", torch.bfloat16, dev, 0, 1)
from this_and_that_vdm_amd.svd.denoise import DenoiseLoop
for split in (False, True, False, True):
    loop, args = bench.make_loop(unet, cn, "lo", dev, 0)
    loop = DenoiseLoop(unet, cn, use_graph=True, split_cfg=split)
    loop.begin(**args)
    bench.advance(loop, args, 5); torch.cuda.synchronize()
    t0 = time.perf_counter(); bench.advance(loop, args, 20); torch.cuda.synchronize()
    print("split_cfg", split, round((time.perf_counter() - t0) / 20 * 1e3, 3), "ms/step")
