#!/usr/bin/env python3
"""tt_attention (spatial self-attention, mask 0) at the model's shapes: TFLOP/s = 4*nseq*heads*L*L*d / time.
python tools/attn_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops

def bench(nseq, heads, L, d, dtype=torch.bfloat16, iters=10):
    c = heads * d
    q = torch.randn(nseq * L, c, device="cuda", dtype=dtype); k = torch.randn_like(q)
    vt = torch.randn(c, nseq * L, device="cuda", dtype=dtype)
    out = torch.empty_like(q)
    f = lambda: ops.attention(q, k, vt, out, nseq=nseq, lq=L, heads=heads, head_dim=d, mask=0, lk=L, k_seq_stride=L, v_seq_stride=L)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    s = e0.elapsed_time(e1) / iters * 1e-3
    fl = 4.0 * nseq * heads * L * L * d
    print(f"nseq {nseq:3d} heads {heads:2d} L {L:5d} d {d:3d} {str(dtype)[6:]:9s}: {s*1e6:9.1f} us  {fl/s/1e12:7.1f} TFLOP/s  ({fl/s/2.5e15*100:4.1f} % of 2.5 PF)")

for args in ((28, 5, 1792, 64), (28, 10, 448, 64), (28, 20, 112, 64), (28, 5, 7168, 64), (28, 10, 1792, 64), (28, 20, 448, 64), (28, 10, 1792, 128)):
    bench(*args)
bench(28, 5, 7168, 64, torch.float16)

def bench8(nseq, heads, L, d, iters=10):
    c = heads * d
    q = torch.randn(nseq * L, c, device="cuda").to(ops.FP8); k = torch.randn(nseq * L, c, device="cuda").to(ops.FP8)
    vt = torch.randn(c, nseq * L, device="cuda").to(ops.FP8)
    out = torch.empty(nseq * L, c, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.attention(q, k, vt, out, nseq=nseq, lq=L, heads=heads, head_dim=d, mask=0, lk=L, k_seq_stride=L, v_seq_stride=L)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    s = e0.elapsed_time(e1) / iters * 1e-3
    fl = 4.0 * nseq * heads * L * L * d
    print(f"fp8  nseq {nseq:3d} heads {heads:2d} L {L:5d} d {d:3d}: {s*1e6:9.1f} us  {fl/s/1e12:7.1f} TFLOP/s  ({fl/s/2.5e15*100:4.1f} % of the 2.5 PF bf16/fp8 (non-MX) MFMA peak)")

for args in ((28, 5, 1792, 64), (28, 5, 7168, 64), (28, 10, 1792, 64)):
    bench8(*args)
