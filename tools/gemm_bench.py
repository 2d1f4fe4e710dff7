#!/usr/bin/env python3
"""Micro-benchmark of tt_gemm / tt_attention on representative shapes (GPU box).  python tools/gemm_bench.py"""
import sys, os, time
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
    dt = torch.bfloat16
    dev = "cuda"
    shapes = [(8192, 8192, 8192), (4096, 4096, 4096), (50176, 320, 320), (50176, 320, 1280), (50176, 2560, 320), (50176, 960, 320),
              (50176, 320, 2880), (12544, 640, 5760), (12544, 5120, 640), (3136, 1280, 11520), (3136, 10240, 1280), (784, 1280, 11520), (784, 1280, 23040)]
    for m, n, k in shapes:
        a = torch.randn(m, k, device=dev, dtype=dt)
        w = torch.randn(n, k, device=dev, dtype=dt)
        out = torch.empty(m, n, device=dev, dtype=dt)
        s = timeit(lambda: ops.gemm(a, w, out=out))
        print(f"linear m={m:6d} n={n:6d} k={k:6d}: {s*1e6:9.1f} us  {2*m*n*k/s/1e12:7.1f} TF   bytes(min) {(m*k+n*k+m*n)*2/s/1e12:5.2f} TB/s")
    # conv3x3 L0
    for (nimg, h, w_, cin, cout) in [(28, 32, 56, 320, 320), (28, 32, 56, 960, 320), (28, 16, 28, 640, 640), (28, 8, 14, 1280, 1280), (28, 4, 7, 1280, 1280)]:
        x = torch.randn(nimg * h * w_, cin, device=dev, dtype=dt)
        wt = torch.randn(cout, 9 * cin, device=dev, dtype=dt)
        out = torch.empty(nimg * h * w_, cout, device=dev, dtype=dt)
        s = timeit(lambda: ops.gemm(x, wt, mode=1, conv=(nimg, h, w_, h, w_, 1, 0), out=out))
        fl = 2 * nimg * h * w_ * cout * 9 * cin
        print(f"conv3x3 n={nimg} {h}x{w_} cin={cin} cout={cout}: {s*1e6:9.1f} us  {fl/s/1e12:7.1f} TF")
    for (nseq, l, heads, d) in [(28, 1792, 5, 64), (28, 448, 10, 64), (28, 112, 20, 64), (28, 7168, 5, 64)]:
        c = heads * d
        q = torch.randn(nseq * l, c, device=dev, dtype=dt); k = torch.randn(nseq * l, c, device=dev, dtype=dt)
        vt = torch.randn(c, nseq * l, device=dev, dtype=dt); out = torch.empty(nseq * l, c, device=dev, dtype=dt)
        s = timeit(lambda: ops.attention(q, k, vt, out, nseq=nseq, lq=l, heads=heads, head_dim=d, mask=0, lk=l, k_seq_stride=l, v_seq_stride=l), iters=5)
        print(f"attn nseq={nseq} l={l} heads={heads}: {s*1e6:9.1f} us  {4*nseq*heads*l*l*d/s/1e12:7.1f} TF")

if __name__ == "__main__":
    main()
