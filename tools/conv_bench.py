#!/usr/bin/env python3
"""tt_conv3x3 (GroupNorm+SiLU fused, LDS patch) against tt_groupnorm_apply + tt_gemm mode 1 on the ResBlock shapes (GPU box).
python tools/conv_bench.py [lo|hi]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

def main():
    hi = len(sys.argv) > 1 and sys.argv[1] == "hi"
    s = 2 if hi else 1
    shapes = [(28, 32 * s, 56 * s, 320, 320), (28, 32 * s, 56 * s, 640, 320), (28, 32 * s, 56 * s, 960, 320), (28, 16 * s, 28 * s, 640, 640),
              (28, 16 * s, 28 * s, 1280, 640), (28, 16 * s, 28 * s, 1920, 640), (28, 8 * s, 14 * s, 1280, 1280), (28, 8 * s, 14 * s, 2560, 1280)]
    dt, dev = torch.bfloat16, "cuda"
    print(f"{'shape':44s} {'conv3x3 raw':>12s} {'conv3x3+gn':>12s} {'gn+gemm':>9s} {'gemm':>9s}   us (TFLOP/s)")
    for nimg, h, w, cin, cout in shapes:
        if not ops.conv3x3_supported(h, w, cin, 0, cout, dt):
            print(f"{nimg}x{h}x{w} {cin}->{cout}: unsupported"); continue
        m = nimg * h * w
        x = torch.randn(m, cin, device=dev, dtype=dt)
        wt = (torch.randn(cout, 9 * cin, device=dev, dtype=dt) * 0.02)
        bias = torch.randn(cout, device=dev, dtype=torch.float32)
        gamma, beta = torch.randn(cin, device=dev), torch.randn(cin, device=dev)
        out = torch.empty(m, cout, device=dev, dtype=dt)
        xn = torch.empty_like(x)
        gn = ops.groupnorm_stats(x, None, nimg, h * w, 1, gamma, beta, 1e-5)
        fl = 2.0 * m * cout * 9 * cin
        t_conv = timeit(lambda: ops.conv3x3(x, None, wt, nimg, h, w, gn=gn, silu=True, bias=bias, out=out))
        t_raw = timeit(lambda: ops.conv3x3(x, None, wt, nimg, h, w, gn=None, silu=False, bias=bias, out=out))
        t_gemm = timeit(lambda: ops.gemm(x, wt, mode=1, conv=(nimg, h, w, h, w, 1, 0), bias=bias, out=out))
        t_both = timeit(lambda: (ops.groupnorm_apply(x, None, nimg, h * w, gn[0], gn[1], True, out=xn), ops.gemm(xn, wt, mode=1, conv=(nimg, h, w, h, w, 1, 0), bias=bias, out=out)))
        print(f"{nimg}x{h}x{w} {cin:5d}->{cout:5d} (M={m:6d})".ljust(44) +
              f" {t_raw*1e6:7.1f}({fl/t_raw/1e12:4.0f}) {t_conv*1e6:7.1f}({fl/t_conv/1e12:4.0f}) {t_both*1e6:7.1f}({fl/t_both/1e12:4.0f}) {t_gemm*1e6:7.1f}({fl/t_gemm/1e12:4.0f})")

if __name__ == "__main__":
    main()
