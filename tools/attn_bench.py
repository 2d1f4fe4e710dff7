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


def bench_vr(nseq, heads, L, d=64, dtype=torch.bfloat16, iters=10):
    """the same problem with V read as rows of a fused Q | K | V tensor (TtAttnArgs.v_rows: transposed on the way out of LDS)"""
    c = heads * d
    qkv = torch.randn(nseq * L, 3 * c, device="cuda", dtype=dtype)
    out = torch.empty(nseq * L, c, device="cuda", dtype=dtype)
    f = lambda: ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], out, nseq=nseq, lq=L, heads=heads, head_dim=d, mask=0, lk=L,
                              k_seq_stride=L, v_seq_stride=L, v_rows=True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    s = e0.elapsed_time(e1) / iters * 1e-3
    fl = 4.0 * nseq * heads * L * L * d
    print(f"nseq {nseq:3d} heads {heads:2d} L {L:5d} d {d:3d} V as rows: {s*1e6:9.1f} us  {fl/s/1e12:7.1f} TFLOP/s  ({fl/s/2.5e15*100:4.1f} % of 2.5 PF)")


for args in ((28, 5, 1792), (28, 10, 448), (28, 20, 112), (28, 5, 7168)):
    bench_vr(*args)
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


def bench_cross(b, f, hw, heads, s=78, d=64, dtype=torch.bfloat16, iters=10):
    """spatial cross-attention (mask 1) with the query projection fused in (TtAttnArgs.qx) at the model's live-row shapes:
    14 frames of one batch element, 78 context tokens.  flops = scores + P V + the projection the kernel computes itself."""
    from this_and_that_vdm_amd.packing import permute_q_rows
    c = heads * d
    sp = (s + 7) // 8 * 8
    rows = b * f * hw
    x = torch.randn(rows, c, device="cuda", dtype=dtype)
    wq = permute_q_rows((torch.randn(c, c, device="cuda") * c ** -0.5).to(dtype)); bq = permute_q_rows(torch.randn(c, device="cuda"))
    k = torch.randn(b * sp, c, device="cuda", dtype=dtype); vt = torch.randn(c, b * sp, device="cuda", dtype=dtype)
    out = torch.empty(rows, c, device="cuda", dtype=dtype)
    fn = lambda: ops.attention(None, k, vt, out, qx=x, wq=wq, bq=bq, ln_eps=1e-5, nseq=b * f, lq=hw, heads=heads, head_dim=d, mask=1, lk=s,
                               k_seq_stride=sp, v_seq_stride=sp, frames=f, ctx_batches=b)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    fl = 4.0 * rows * heads * s * d + 2.0 * rows * c * c
    print(f"cross+qproj rows {rows:6d} heads {heads:2d} C {c:4d} S {s}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TFLOP/s  (x read + out write {2*rows*c*2/t/1e9:6.0f} GB/s)")

for args in ((1, 14, 1792, 5), (1, 14, 448, 10), (1, 14, 112, 20), (1, 14, 7168, 5)):
    bench_cross(*args)
