"""GPU box: temporal self-attention (tt_temporal_attention) at the four UNet levels of the 256x448 workload,
CFG batch 2, 14 frames, head_dim 64.  Prints us per launch and the HBM-side rate (Q|K|V read once, O written once)."""
import sys, torch
sys.path.insert(0, ".")
from this_and_that_vdm_amd import ops

for hw, heads in ((1792, 5), (448, 10), (112, 20), (28, 20)):
    b, f, c = 2, 14, heads * 64
    qkv = torch.randn(b * f * hw, 3 * c, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(b * f * hw, c, device="cuda", dtype=torch.bfloat16)
    spoil = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        ops.temporal_attention(qkv, out, batch=b, frames=f, hw=hw, heads=heads, head_dim=64)
    n, tot = 50, 0.0
    for _ in range(n):
        spoil.fill_(1)                              # cold operands, as in the model
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.temporal_attention(qkv, out, batch=b, frames=f, hw=hw, heads=heads, head_dim=64)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    us = tot / n * 1e3
    print(f"hw {hw:5d} heads {heads:2d}: {us:7.1f} us  {qkv.numel() * 2 * 4 / 3 / us / 1e6:6.2f} TB/s")
