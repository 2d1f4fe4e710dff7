"""TEST INFRASTRUCTURE ONLY (imported by tests/ and tools/decode_bench.py's check; never by the product package).

CPU fp32 restatement of the temporal VAE *decoder* the reference's pipelines call after the denoise loop
(`self.vae.decode(latents, num_frames=...)`, svd/pipeline_stable_video_diffusion_controlnet.py:257-283, :722-726).
The class lives in the third-party dependency diffusers==0.25.1 (requirements.txt:23), absent from /root/reference and
from this image: what follows restates its published algorithm (models/autoencoder_kl_temporal_decoder.py `TemporalDecoder`
/ `AutoencoderKLTemporalDecoder.decode`, models/unet_3d_blocks.py `MidBlockTemporalDecoder` / `UpBlockTemporalDecoder`,
models/attention_processor.py `Attention` with group_norm + residual_connection under AttnProcessor2_0) with the same
module / parameter names, so a diffusers checkpoint's `decoder.*` keys load by name.  PARITY UNPINNED: there is no
diffusers here to check the restatement against, nor does the reference hold golden vectors for the VAE; the product decoder
is tested against this file on identical weights, and this file's building blocks (SpatioTemporalResBlock, Upsample2D,
AlphaBlender) are the ones oracle/leaves.py already uses for the UNet, which the reference's own model files exercise."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .leaves import SpatioTemporalResBlock, Upsample2D


class VaeAttention(nn.Module):
    """diffusers Attention(query_dim=C, heads=C // dim_head, dim_head, eps=1e-6, norm_num_groups=32, bias=True,
    residual_connection=True) on a [N, C, h, w] input: GroupNorm over the tokens' channels, q/k/v with bias, SDPA, to_out,
    + input."""

    def __init__(self, query_dim: int, heads: int, dim_head: int, eps: float = 1e-6, norm_num_groups: int = 32):
        super().__init__()
        self.heads = heads
        inner = heads * dim_head
        self.group_norm = nn.GroupNorm(num_channels=query_dim, num_groups=norm_num_groups, eps=eps, affine=True)
        self.to_q = nn.Linear(query_dim, inner, bias=True)
        self.to_k = nn.Linear(query_dim, inner, bias=True)
        self.to_v = nn.Linear(query_dim, inner, bias=True)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])

    def forward(self, hidden_states):
        residual = hidden_states
        b, c, h, w = hidden_states.shape
        x = hidden_states.view(b, c, h * w).transpose(1, 2)
        x = self.group_norm(x.transpose(1, 2)).transpose(1, 2)
        q, k, v = self.to_q(x), self.to_k(x), self.to_v(x)
        d = q.shape[-1] // self.heads
        sp = lambda t: t.view(b, -1, self.heads, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), attn_mask=None, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, -1, self.heads * d)
        o = self.to_out[1](self.to_out[0](o))
        return o.transpose(-1, -2).reshape(b, c, h, w) + residual


def _res(cin, cout):
    return SpatioTemporalResBlock(in_channels=cin, out_channels=cout, temb_channels=None, eps=1e-6, temporal_eps=1e-5,
                                  merge_factor=0.0, merge_strategy="learned", switch_spatial_to_temporal_mix=True)


class MidBlockTemporalDecoder(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, attention_head_dim: int = 512, num_layers: int = 1):
        super().__init__()
        self.resnets = nn.ModuleList([_res(in_channels if i == 0 else out_channels, out_channels) for i in range(num_layers)])
        self.attentions = nn.ModuleList([VaeAttention(in_channels, in_channels // attention_head_dim, attention_head_dim)])

    def forward(self, hidden_states, image_only_indicator):
        hidden_states = self.resnets[0](hidden_states, image_only_indicator=image_only_indicator)
        for resnet, attn in zip(self.resnets[1:], self.attentions):
            hidden_states = attn(hidden_states)
            hidden_states = resnet(hidden_states, image_only_indicator=image_only_indicator)
        return hidden_states


class UpBlockTemporalDecoder(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, num_layers: int = 1, add_upsample: bool = True):
        super().__init__()
        self.resnets = nn.ModuleList([_res(in_channels if i == 0 else out_channels, out_channels) for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)]) if add_upsample else None

    def forward(self, hidden_states, image_only_indicator):
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, image_only_indicator=image_only_indicator)
        if self.upsamplers is not None:
            for up in self.upsamplers:
                hidden_states = up(hidden_states)
        return hidden_states


class TemporalDecoder(nn.Module):
    def __init__(self, in_channels: int = 4, out_channels: int = 3, block_out_channels: Tuple[int, ...] = (128, 256, 512, 512),
                 layers_per_block: int = 2):
        super().__init__()
        self.layers_per_block = layers_per_block
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[-1], kernel_size=3, stride=1, padding=1)
        self.mid_block = MidBlockTemporalDecoder(num_layers=layers_per_block, in_channels=block_out_channels[-1],
                                                 out_channels=block_out_channels[-1], attention_head_dim=block_out_channels[-1])
        self.up_blocks = nn.ModuleList([])
        rev = list(reversed(block_out_channels))
        out_ch = rev[0]
        for i in range(len(block_out_channels)):
            prev, out_ch = out_ch, rev[i]
            self.up_blocks.append(UpBlockTemporalDecoder(num_layers=layers_per_block + 1, in_channels=prev, out_channels=out_ch,
                                                         add_upsample=i != len(block_out_channels) - 1))
        self.conv_norm_out = nn.GroupNorm(num_channels=block_out_channels[0], num_groups=32, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, kernel_size=3, padding=1)
        self.time_conv_out = nn.Conv3d(out_channels, out_channels, kernel_size=(3, 1, 1), padding=(1, 0, 0))

    def forward(self, sample, image_only_indicator, num_frames: int = 1):
        sample = self.conv_in(sample)
        sample = self.mid_block(sample, image_only_indicator=image_only_indicator)
        for up in self.up_blocks:
            sample = up(sample, image_only_indicator=image_only_indicator)
        sample = self.conv_out(self.conv_act(self.conv_norm_out(sample)))
        bf, c, h, w = sample.shape
        b = bf // num_frames
        sample = sample[None, :].reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        sample = self.time_conv_out(sample)
        return sample.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


class AutoencoderKLTemporalDecoder(nn.Module):
    """decode() only (the encoder is not on this path's "next" row: the pipelines encode one image per request with the
    caller's stock module)."""

    def __init__(self, latent_channels: int = 4, out_channels: int = 3, block_out_channels: Tuple[int, ...] = (128, 256, 512, 512),
                 layers_per_block: int = 2, scaling_factor: float = 0.18215):
        super().__init__()
        self.scaling_factor = scaling_factor
        self.decoder = TemporalDecoder(latent_channels, out_channels, block_out_channels, layers_per_block)

    def decode(self, z, num_frames: int = 1):
        b = z.shape[0] // num_frames
        return self.decoder(z, num_frames=num_frames, image_only_indicator=torch.zeros(b, num_frames, dtype=z.dtype, device=z.device))
