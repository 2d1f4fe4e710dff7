#!/usr/bin/env python3
"""What the residual / bias epilogue terms cost on the HBM-side-bound linears (GPU box): same GEMM with and without them."""
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
    lib = _lib.load()
    dt, dev = torch.bfloat16, "cuda"
    shapes = [(50176, 320, 1280), (50176, 320, 320), (12544, 640, 2560), (12544, 640, 640), (3136, 1280, 5120), (3136, 1280, 1280)]
    cfgs = [int(x) for x in sys.argv[1:]] or [-1]
    print(f"{'shape':24s} cfg   plain   +bias  +bias+res  in-place   us")
    for m, n, k in shapes:
        a = torch.randn(m, k, device=dev, dtype=dt); w = torch.randn(n, k, device=dev, dtype=dt) * k ** -0.5
        bias = torch.randn(n, device=dev); res = torch.randn(m, n, device=dev, dtype=dt); out = torch.empty_like(res)
        for c in cfgs:
            lib.tt_gemm_set_tile_override(c)
            t0 = timeit(lambda: ops.gemm(a, w, out=out))
            t1 = timeit(lambda: ops.gemm(a, w, bias=bias, out=out))
            t2 = timeit(lambda: ops.gemm(a, w, bias=bias, residual=res, out=out))
            t3 = timeit(lambda: ops.gemm(a, w, bias=bias, residual=res, out=res))
            print(f"{m:6d}x{n:5d}x{k:5d}      {c:3d} {t0*1e6:7.1f} {t1*1e6:7.1f} {t2*1e6:9.1f} {t3*1e6:9.1f}")
    lib.tt_gemm_set_tile_override(-1)

if __name__ == "__main__":
    main()
