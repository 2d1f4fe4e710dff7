"""Deterministic, RNG-free synthetic weights and inputs.

Weights are a pure function of (parameter name, element index) through a 32-bit integer hash,
so the build container, the GPU box, the oracle and the HIP path all see bit-identical values
without shipping checkpoints.  Inputs follow BASELINE.md section 3 (CPU-seeded torch generator).
This is data plumbing (torch ops on whatever device the tensor lives on), not product compute.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Optional

import torch

_M32 = 0xFFFFFFFF


def hash_uniform(n: int, seed: int, device=None) -> torch.Tensor:
    """n values in [-1, 1), exactly representable in fp32; identical on CPU and GPU."""
    x = torch.arange(n, dtype=torch.int64, device=device)
    x = (x * 0x9E3779B1 + (seed & _M32)) & _M32
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & _M32
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & _M32
    x = x ^ (x >> 16)
    return (x >> 8).to(torch.float32) * (2.0 / (1 << 24)) - 1.0


def _bound_for(name: str, shape) -> tuple:
    """(offset, amplitude) per parameter kind; amplitudes follow PyTorch's default init scale."""
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "mix_factor":
        return 0.5, 0.3
    if len(shape) == 1:
        if leaf == "weight":            # norm gamma
            return 1.0, 0.1
        return 0.0, 0.05                # biases / norm beta
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return 0.0, 1.0 / math.sqrt(fan_in)


@torch.no_grad()
def fill_parameters_(module: torch.nn.Module, salt: str = "", round_to: Optional[torch.dtype] = None) -> None:
    """In-place deterministic fill of every parameter (zero-convs included: they get non-zero values,
    SURVEY.md 8(d)).  ``round_to`` rounds the values through a storage dtype (fp16/bf16) so an fp32
    oracle and a low-precision product hold the same numbers."""
    for name, p in module.named_parameters():
        seed = zlib.crc32((salt + name).encode())
        off, amp = _bound_for(name, tuple(p.shape))
        v = hash_uniform(p.numel(), seed, device=p.device).mul_(amp).add_(off).reshape(p.shape)
        if round_to is not None:
            v = v.to(round_to)
        p.copy_(v.to(p.dtype))


def synthetic_inputs(batch: int, frames: int, h: int, w: int, ctx_tokens: int = 78, ctx_dim: int = 1024,
                     seed: int = 0, sigma_max: float = 700.0) -> Dict[str, torch.Tensor]:
    """CPU fp32 inputs of one denoise request (CFG batch = 2 when ``batch`` == 2: row 0 is the uncond half)."""
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(1, frames, 4, h, w, generator=g) * math.sqrt(sigma_max ** 2 + 1)
    img = torch.randn(1, frames, 4, h, w, generator=g)
    ctx = torch.randn(1, ctx_tokens, ctx_dim, generator=g)
    ctx = torch.nn.functional.layer_norm(ctx, (ctx_tokens, ctx_dim))
    ges = torch.zeros(frames, 4, h, w)
    for fr in (4, 10):
        if fr < frames:
            ges[fr] = torch.randn(4, h, w, generator=g)
    if batch == 2:
        image_latents = torch.cat([torch.zeros_like(img), img])
        ehs = torch.cat([torch.zeros_like(ctx), ctx])
    else:
        image_latents = img.repeat(batch, 1, 1, 1, 1)
        ehs = ctx.repeat(batch, 1, 1)
    return dict(
        latents=lat, image_latents=image_latents, encoder_hidden_states=ehs,
        added_time_ids=torch.tensor([[6.0, 200.0, 0.1]]).repeat(batch, 1),
        gesture_latents=ges, guidance_scale=torch.linspace(1.0, 3.0, frames).reshape(1, frames, 1, 1, 1),
    )
