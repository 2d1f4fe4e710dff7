#!/usr/bin/env python3
"""Timing experiment of round 5: what would the step cost if EVERY GroupNorm were an apply-only consumer (one launch, no
statistics)?  Replaces ops.groupnorm by groupnorm_apply with scale 1 / shift 0 -- WRONG NUMBERS BY DESIGN, timing only -- and prints
ms / step next to the real path.  Lived inside ops.groupnorm behind TT_GN_EMULATE until round 6 (advisor: a leaked environment variable
would have corrupted every GroupNorm silently); it is a monkeypatch in this script now and the library refuses the variable.
    python tools/gn_emulate.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from this_and_that_vdm_amd import ops                          # noqa: E402


def ms_per_step(loop, args, steps=25):
    bench.advance(loop, args, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.advance(loop, args, steps, fresh=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    dev = torch.device("cuda", 0)
    unet, cn, _, _ = bench.build_models("vgl", torch.bfloat16, dev, 0, 1)
    loop, args = bench.make_loop(unet, cn, "lo", dev, 0)
    print(f"real GroupNorms: {ms_per_step(loop, args):.3f} ms / step")
    cache = {}

    def emulated(x0, x1, nimg, hw, fpg, gamma, beta, eps, silu):
        c = x0.shape[-1] + (0 if x1 is None else x1.shape[-1])
        ss = cache.get((nimg, c))
        if ss is None:
            ss = cache[(nimg, c)] = (torch.ones((nimg, c), dtype=torch.float32, device=x0.device), torch.zeros((nimg, c), dtype=torch.float32, device=x0.device))
        return ops.groupnorm_apply(x0, x1, nimg, hw, ss[0], ss[1], silu)
    real, ops.groupnorm = ops.groupnorm, emulated
    try:
        loop2, args2 = bench.make_loop(unet, cn, "lo", dev, 0)   # a new loop: a new captured graph with the patched launches
        print(f"apply-only GroupNorms (wrong numbers, timing only): {ms_per_step(loop2, args2):.3f} ms / step")
    finally:
        ops.groupnorm = real


if __name__ == "__main__":
    main()
