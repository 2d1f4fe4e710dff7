#!/usr/bin/env python3
"""Two eager denoise steps (no graph) for rocprofv3 counter passes; use with --kernel-include-regex."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
mode, res = (sys.argv[1:] + ["vgl", "lo"])[:2]
dev = torch.device("cuda", 0)
unet, cn, _, _ = bench.build_models(mode, torch.bfloat16, dev, 0, 1)
loop, args = bench.make_loop(unet, cn, res, dev, 0)
loop.use_graph = False
loop.overlap_branches = False
loop.step(); loop.step()
torch.cuda.synchronize()
