#!/usr/bin/env python3
"""Micro-benchmark of tt_gemm tile configurations / tt_attention on the model's shapes (GPU box).
python tools/gemm_bench.py [cfg ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops, _lib

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

LIN = [(8192, 8192, 8192), (50176, 960, 320), (50176, 640, 320), (50176, 2560, 320), (12544, 5120, 640), (3136, 10240, 1280), (50176, 320, 1280), (12544, 640, 2560),
       (3136, 1280, 5120), (50176, 320, 320), (12544, 640, 640), (3136, 1280, 1280), (784, 1280, 5120), (784, 10240, 1280)]
CONV = [(28, 32, 56, 320, 320), (28, 32, 56, 960, 320), (28, 16, 28, 640, 640), (28, 16, 28, 1920, 640), (28, 8, 14, 1280, 1280),
        (28, 8, 14, 2560, 1280), (28, 4, 7, 1280, 1280), (28, 4, 7, 2560, 1280)]

def main():
    cfgs = [int(x) for x in sys.argv[1:] if not x.startswith("--")] or [-1]
    lib = _lib.load()
    dt, dev = torch.bfloat16, "cuda"
    rows = []
    for m, n, k in LIN:
        a = torch.randn(m, k, device=dev, dtype=dt); w = torch.randn(n, k, device=dev, dtype=dt)
        out = torch.empty(m, n, device=dev, dtype=dt)
        r = []
        for c in cfgs:
            lib.tt_gemm_set_tile_override(c)
            s = timeit(lambda: ops.gemm(a, w, out=out))
            r.append(2 * m * n * k / s / 1e12)
        if "--lib" in sys.argv:          # yardstick only: the vendor library (hipBLASLt through torch) on the same operands
            r.append(2 * m * n * k / timeit(lambda: torch.matmul(a, w.t(), out=out)) / 1e12)
        rows.append((f"lin  {m:6d}x{n:6d}x{k:6d}", r))
    for nimg, h, w_, cin, cout in CONV:
        x = torch.randn(nimg * h * w_, cin, device=dev, dtype=dt); wt = torch.randn(cout, 9 * cin, device=dev, dtype=dt)
        out = torch.empty(nimg * h * w_, cout, device=dev, dtype=dt)
        r = []
        for c in cfgs:
            lib.tt_gemm_set_tile_override(c)
            s = timeit(lambda: ops.gemm(x, wt, mode=1, conv=(nimg, h, w_, h, w_, 1, 0), out=out))
            r.append(2 * nimg * h * w_ * cout * 9 * cin / s / 1e12)
        if "--lib" in sys.argv:          # yardstick only: MIOpen through torch, channels-last
            xi = x.view(nimg, h, w_, cin).permute(0, 3, 1, 2)
            wi = wt.view(cout, 3, 3, cin).permute(0, 3, 1, 2)
            r.append(2 * nimg * h * w_ * cout * 9 * cin / timeit(lambda: torch.nn.functional.conv2d(xi, wi, padding=1)) / 1e12)
        rows.append((f"conv M={nimg*h*w_:6d} cin={cin:5d} cout={cout:5d}", r))
    lib.tt_gemm_set_tile_override(-1)
    print(f"{'shape':38s}" + "".join(f"  cfg{c:>3d}" for c in cfgs) + ("     lib" if "--lib" in sys.argv else "") + "   (TFLOP/s)")
    for name, r in rows:
        best = max(r)
        print(f"{name:38s}" + "".join(f" {v:7.0f}{'*' if v == best else ' '}" for v in r))
    if "--attn" in sys.argv:
        pass

if __name__ == "__main__":
    main()
