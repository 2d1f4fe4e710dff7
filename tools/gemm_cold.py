#!/usr/bin/env python3
"""tt_gemm tile configurations on the model's linear shapes with COLD operands: activations, residuals and outputs
rotate through a pool larger than L2 + Infinity Cache, as they are in the model where the previous kernel produced them.
python tools/gemm_cold.py cfg [cfg ...]      (-1 = planner)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops, _lib

POOL_BYTES = 1 << 30
# (M, N, K, geglu, residual)
SHAPES = [(50176, 2560, 320, 1, 0), (12544, 5120, 640, 1, 0), (3136, 10240, 1280, 1, 0), (50176, 320, 1280, 0, 1),
          (3136, 1280, 5120, 0, 1), (50176, 320, 320, 0, 1), (12544, 640, 2560, 0, 1), (3136, 1280, 1280, 0, 1),
          (12544, 640, 640, 0, 1), (50176, 960, 320, 0, 0), (12544, 1920, 640, 0, 0), (3136, 3840, 1280, 0, 0)]

def main():
    cfgs = [int(x) for x in sys.argv[1:]] or [-1]
    lib = _lib.load()
    dt, dev = torch.bfloat16, "cuda"
    print(f"{'M':>6s} {'N':>6s} {'K':>6s} gg res" + "".join(f"  cfg{c:>3d}" for c in cfgs) + "   (TFLOP/s, cold operands)")
    for m, n, k, gg, res in SHAPES:
        nout = n // 2 if gg else n
        per = (m * k + m * nout * (2 if res else 1)) * 2
        nb = max(2, min(24, POOL_BYTES // per + 1))
        A = [torch.randn(m, k, device=dev, dtype=dt) for _ in range(nb)]
        O = [torch.empty(m, nout, device=dev, dtype=dt) for _ in range(nb)]
        R = [torch.randn(m, nout, device=dev, dtype=dt) for _ in range(nb)] if res else None
        w = torch.randn(n, k, device=dev, dtype=dt)
        if gg:
            from this_and_that_vdm_amd import packing
        bias = torch.randn(n, device=dev, dtype=torch.float32)
        r = []
        for c in cfgs:
            lib.tt_gemm_set_tile_override(c)
            def run(i):
                ops.gemm(A[i % nb], w, bias=bias, geglu=bool(gg), residual=R[i % nb] if res else None, out=O[i % nb])
            for i in range(nb): run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 2 * nb
            e0.record()
            for i in range(iters): run(i)
            e1.record(); torch.cuda.synchronize()
            s = e0.elapsed_time(e1) / iters * 1e-3
            r.append(2 * m * n * k / s / 1e12)
        lib.tt_gemm_set_tile_override(-1)
        best = max(r)
        print(f"{m:6d} {n:6d} {k:6d} {gg:2d} {res:3d}" + "".join(f" {v:7.0f}{'*' if v == best else ' '}" for v in r))
        del A, O, R

if __name__ == "__main__":
    main()
