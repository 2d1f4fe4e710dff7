#!/usr/bin/env python3
"""decode_latents through the native temporal VAE decoder at the reference's inference setting (14 frames, 256x448 output,
decode_chunk_size = 8 as test_code/inference.py:258 passes): ms per request, next to the 25-step denoise loop it follows.
python tools/decode_bench.py [--dtype bf16|fp16] [--chunk 8]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops
from this_and_that_vdm_amd.svd.autoencoder_kl_temporal_decoder import AutoencoderKLTemporalDecoder
from this_and_that_vdm_amd.utils.synthetic import fill_parameters_

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--chunk", type=int, default=8)
ap.add_argument("--frames", type=int, default=14)
a = ap.parse_args()
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32}[a.dtype]
with torch.device("cuda"):
    vae = AutoencoderKLTemporalDecoder().to(dt).eval()
fill_parameters_(vae, "vae.")
if dt == torch.float32:
    vae.compute_dtype = torch.float32
vae.prepare()
lat = torch.randn(a.frames, 4, 32, 56, device="cuda")

def decode():
    out = []
    for i in range(0, a.frames, a.chunk):
        out.append(vae.decode(lat[i:i + a.chunk] / vae.config.scaling_factor, num_frames=min(a.chunk, a.frames - i)).sample)
    return torch.cat(out, 0)

for _ in range(2):
    y = decode()
torch.cuda.synchronize()
ops.PROFILE = []
decode()
torch.cuda.synchronize()
rec, ops.PROFILE = ops.PROFILE, None
flops = sum(r[1] for r in rec)
t0 = time.perf_counter()
n = 5
for _ in range(n):
    y = decode()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"temporal VAE decode, {a.frames} frames -> {tuple(y.shape)}, {a.dtype}, chunks of {a.chunk}: {ms:.1f} ms per request "
      f"({flops / 1e12:.2f} TFLOP in tt_gemm launches = {flops / (ms * 1e-3) / 1e12:.0f} TFLOP/s), finite {bool(torch.isfinite(y).all())}")
