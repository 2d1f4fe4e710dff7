"""Tiny stand-ins for the third-party models a pipeline wraps (VAE, CLIP vision/text): deterministic, no weights files."""
from types import SimpleNamespace

import torch
import torch.nn as nn


class _Dist:
    def __init__(self, m):
        self._m = m

    def mode(self):
        return self._m


class StubVAE(nn.Module):
    """8x spatial reduction to 4 latent channels and back (shape contract of AutoencoderKLTemporalDecoder)."""

    def __init__(self):
        super().__init__()
        self.config = SimpleNamespace(block_out_channels=(1, 1, 1, 1), scaling_factor=0.18215, force_upcast=False)
        self.enc = nn.Conv2d(3, 4, 8, stride=8)
        self.dec = nn.ConvTranspose2d(4, 3, 8, stride=8)
        g = torch.Generator().manual_seed(7)
        for p in self.parameters():
            p.data = torch.randn(p.shape, generator=g) * 0.05

    @property
    def dtype(self):
        return self.enc.weight.dtype

    def encode(self, x):
        return SimpleNamespace(latent_dist=_Dist(self.enc(x)))

    def decode(self, z, num_frames=None):
        return SimpleNamespace(sample=torch.tanh(self.dec(z)))

    def forward(self, x, num_frames=None):
        return self.decode(self.encode(x).latent_dist.mode(), num_frames)


class StubCLIPVision(nn.Module):
    def __init__(self, dim=64):
        super().__init__()
        self.proj = nn.Linear(3 * 8 * 8, dim)
        g = torch.Generator().manual_seed(8)
        for p in self.parameters():
            p.data = torch.randn(p.shape, generator=g) * 0.1

    def forward(self, image):
        x = torch.nn.functional.adaptive_avg_pool2d(image, 8).flatten(1)
        return SimpleNamespace(image_embeds=self.proj(x))


class StubTextEncoder(nn.Module):
    def __init__(self, dim=64, tokens=4):
        super().__init__()
        self.emb = nn.Embedding(100, dim)
        self.tokens = tokens
        g = torch.Generator().manual_seed(9)
        self.emb.weight.data = torch.randn(self.emb.weight.shape, generator=g)

    def forward(self, ids):
        return (self.emb(ids[:, :self.tokens]),)
