import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from oracle import models as om
def build(ctor):
    with torch.device("meta"):
        m = ctor()
    m = m.to_empty(device="cpu").eval()
    for name, p in m.named_parameters():
        p.data.fill_(1.0 if (p.dim() == 1 and name.endswith("weight")) else 0.01)
    return m
torch.set_flush_denormal(True)
with torch.no_grad():
    unet = build(lambda: om.UNetSpatioTemporalConditionModel(num_attention_heads=(5, 10, 20, 20), num_frames=14))
    for f in (1, 2):
        x = torch.randn(1, f, 8, 32, 56); ehs = torch.randn(1, 78, 1024); ati = torch.tensor([[6.0, 200.0, 0.1]])
        for th in (16, 32, 64, 128, 256):
            torch.set_num_threads(th)
            t0 = time.perf_counter(); unet(x, 1.0, ehs, ati); dt = time.perf_counter() - t0
            print(f"frames={f} threads={th}: UNet forward {dt:.1f} s", flush=True)
