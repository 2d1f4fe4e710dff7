#!/usr/bin/env python3
"""Which route every GroupNorm of one denoise step takes (GPU box): from the producer's tile sums (one pass, one launch), two sources
(the up blocks' skip concat: statistics-pass kernels), normalised by the producer's split-K reduction pass (no launch), or no sums attached (the producer's route / tile height does not serve the
segment).          python tools/gn_census.py [vgl|vl] [lo|ref|hi]"""
import os, sys
from collections import Counter
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
    seen, real = Counter(), ops.groupnorm

    def spy(x0, x1, nimg, hw, fpg, *a, **k):
        st = getattr(x0, "_tt_stats", None)
        fz = getattr(x0, "_tt_gn", None) if x1 is None else None
        how = "by the producer's split-K reduction pass" if fz is not None and fz[5] == fpg * hw else "two sources" if x1 is not None else (f"tiles of {st[1]} rows" if st is not None and (fpg * hw) % st[1] == 0 and ops._handoff_valid(x0, *st[2:]) else "no sums attached")
        seen[(hw, x0.shape[-1] + (0 if x1 is None else x1.shape[-1]), "cross-frame" if fpg > 1 else "per image", how)] += 1
        return real(x0, x1, nimg, hw, fpg, *a, **k)
    ops.groupnorm = spy
    for m in list(sys.modules.values()):                     # modules that imported the function by name
        if getattr(m, "groupnorm", None) is real:
            m.groupnorm = spy
    loop.step(); torch.cuda.synchronize()
    for (hw, c, kind, how), n in sorted(seen.items()):
        print(f"{n:5d} x GroupNorm hw {hw:5d} C {c:5d} {kind:11s} -> {how}")
    print(f"{sum(seen.values())} GroupNorms, {sum(n for k, n in seen.items() if k[3].startswith('tiles'))} from tile sums, "
          f"{sum(n for k, n in seen.items() if k[3].startswith('by the'))} inside the producer's reduction pass")


if __name__ == "__main__":
    main()
