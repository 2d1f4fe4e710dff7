"""Oracle composition: SVD spatio-temporal UNet + GestureNet ControlNet on CPU fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Own restatement (not a copy) of the
reference's wiring; every class cites the reference lines it follows.  Attribute
names equal the reference's so one state dict feeds reference, oracle and product.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from .leaves import (AlphaBlender, BasicTransformerBlock, Downsample2D, SpatioTemporalResBlock,
                     TemporalBasicTransformerBlock, TimestepEmbedding, Timesteps, Upsample2D)


class TransformerSpatioTemporalModel(nn.Module):
    """svd/diffusion_arch/transformer_temporal.py:201-381."""

    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=320, num_layers=1,
                 cross_attention_dim=None):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6)                                  # :234
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim=cross_attention_dim)
             for _ in range(num_layers)])
        self.temporal_transformer_blocks = nn.ModuleList(
            [TemporalBasicTransformerBlock(inner, inner, num_attention_heads, attention_head_dim,
                                           cross_attention_dim=cross_attention_dim) for _ in range(num_layers)])
        self.time_pos_embed = TimestepEmbedding(in_channels, in_channels * 4, out_dim=in_channels)  # :265
        self.time_proj = Timesteps(in_channels, True, 0)
        self.time_mixer = AlphaBlender(alpha=0.5, merge_strategy="learned_with_images")
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, hidden_states, encoder_hidden_states=None, image_only_indicator=None):
        n, c, h, w = hidden_states.shape
        f = image_only_indicator.shape[-1]
        b = n // f
        # :309-319 -- context of frame 0 per batch element, flattened (hw, B): quirk Q3.
        ctx0 = encoder_hidden_states.reshape(b, f, -1, encoder_hidden_states.shape[-1])[:, 0]   # [B,S,D]
        time_context = ctx0[None].expand(h * w, b, ctx0.shape[1], ctx0.shape[2]).reshape(h * w * b, ctx0.shape[1], ctx0.shape[2])

        residual = hidden_states
        x = self.norm(hidden_states).permute(0, 2, 3, 1).reshape(n, h * w, c)
        x = self.proj_in(x)
        frame_idx = torch.arange(f, device=x.device).repeat(b)                                 # :328-330
        emb = self.time_pos_embed(self.time_proj(frame_idx).to(x.dtype))[:, None, :]
        for blk, tblk in zip(self.transformer_blocks, self.temporal_transformer_blocks):
            x = blk(x, encoder_hidden_states=encoder_hidden_states)
            mix = tblk(x + emb, num_frames=f, encoder_hidden_states=time_context)
            x = self.time_mixer(x_spatial=x, x_temporal=mix, image_only_indicator=image_only_indicator)
        x = self.proj_out(x).reshape(n, h, w, c).permute(0, 3, 1, 2).contiguous()
        return x + residual


def _tfm(heads, channels, cross_dim, layers=1):
    return TransformerSpatioTemporalModel(heads, channels // heads, in_channels=channels, num_layers=layers,
                                          cross_attention_dim=cross_dim)


class UNetMidBlockSpatioTemporal(nn.Module):
    """unet_3d_blocks.py:1870-1977: STRes(1e-5) -> Tfm -> STRes(1e-5)."""

    def __init__(self, in_channels, temb_channels, num_layers=1, transformer_layers_per_block=1,
                 num_attention_heads=1, cross_attention_dim=1280):
        super().__init__()
        self.has_cross_attention = True
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(in_channels, in_channels, temb_channels, eps=1e-5)
                                      for _ in range(num_layers + 1)])
        self.attentions = nn.ModuleList([_tfm(num_attention_heads, in_channels, cross_attention_dim, transformer_layers_per_block)
                                         for _ in range(num_layers)])

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, image_only_indicator=None):
        hidden_states = self.resnets[0](hidden_states, temb, image_only_indicator=image_only_indicator)
        for attn, res in zip(self.attentions, self.resnets[1:]):
            hidden_states = attn(hidden_states, encoder_hidden_states, image_only_indicator)
            hidden_states = res(hidden_states, temb, image_only_indicator=image_only_indicator)
        return hidden_states


class DownBlockSpatioTemporal(nn.Module):
    """unet_3d_blocks.py:1980-2067 (eps 1e-5)."""

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, add_downsample=True):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels,
                                                             temb_channels, eps=1e-5) for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels, name="op")]) \
            if add_downsample else None

    def forward(self, hidden_states, temb=None, image_only_indicator=None):
        outs = ()
        for res in self.resnets:
            hidden_states = res(hidden_states, temb, image_only_indicator=image_only_indicator)
            outs += (hidden_states,)
        if self.downsamplers is not None:
            hidden_states = self.downsamplers[0](hidden_states)
            outs += (hidden_states,)
        return hidden_states, outs


class CrossAttnDownBlockSpatioTemporal(nn.Module):
    """unet_3d_blocks.py:2070-2189 (ResBlock eps hard-coded 1e-6, :2098)."""

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, transformer_layers_per_block=1,
                 num_attention_heads=1, cross_attention_dim=1280, add_downsample=True):
        super().__init__()
        self.has_cross_attention = True
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels,
                                                             temb_channels, eps=1e-6) for i in range(num_layers)])
        self.attentions = nn.ModuleList([_tfm(num_attention_heads, out_channels, cross_attention_dim, transformer_layers_per_block)
                                         for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                        padding=1, name="op")]) if add_downsample else None

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, image_only_indicator=None):
        outs = ()
        for res, attn in zip(self.resnets, self.attentions):
            hidden_states = res(hidden_states, temb, image_only_indicator=image_only_indicator)
            hidden_states = attn(hidden_states, encoder_hidden_states, image_only_indicator)
            outs += (hidden_states,)
        if self.downsamplers is not None:
            hidden_states = self.downsamplers[0](hidden_states)
            outs += (hidden_states,)
        return hidden_states, outs


class UpBlockSpatioTemporal(nn.Module):
    """unet_3d_blocks.py:2192-2278 (eps default 1e-6; the factory :277-285 drops resnet_eps)."""

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers=1,
                 resnet_eps=1e-6, add_upsample=True):
        super().__init__()
        res = []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            cin = prev_output_channel if i == 0 else out_channels
            res.append(SpatioTemporalResBlock(cin + skip, out_channels, temb_channels, eps=resnet_eps))
        self.resnets = nn.ModuleList(res)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)]) \
            if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, image_only_indicator=None):
        for res in self.resnets:
            skip, res_hidden_states_tuple = res_hidden_states_tuple[-1], res_hidden_states_tuple[:-1]
            hidden_states = res(torch.cat([hidden_states, skip], dim=1), temb, image_only_indicator=image_only_indicator)
        if self.upsamplers is not None:
            hidden_states = self.upsamplers[0](hidden_states)
        return hidden_states


class CrossAttnUpBlockSpatioTemporal(nn.Module):
    """unet_3d_blocks.py:2281-2396."""

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers=1,
                 transformer_layers_per_block=1, resnet_eps=1e-6, num_attention_heads=1, cross_attention_dim=1280,
                 add_upsample=True):
        super().__init__()
        self.has_cross_attention = True
        res, att = [], []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            cin = prev_output_channel if i == 0 else out_channels
            res.append(SpatioTemporalResBlock(cin + skip, out_channels, temb_channels, eps=resnet_eps))
            att.append(_tfm(num_attention_heads, out_channels, cross_attention_dim, transformer_layers_per_block))
        self.resnets, self.attentions = nn.ModuleList(res), nn.ModuleList(att)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)]) \
            if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None,
                image_only_indicator=None):
        for res, attn in zip(self.resnets, self.attentions):
            skip, res_hidden_states_tuple = res_hidden_states_tuple[-1], res_hidden_states_tuple[:-1]
            hidden_states = res(torch.cat([hidden_states, skip], dim=1), temb, image_only_indicator=image_only_indicator)
            hidden_states = attn(hidden_states, encoder_hidden_states, image_only_indicator)
        if self.upsamplers is not None:
            hidden_states = self.upsamplers[0](hidden_states)
        return hidden_states


def _down(kind, n_layers, cin, cout, temb, add_ds, cross, heads, tl):
    """unet_3d_blocks.py:140-162 (SVD branch of get_down_block)."""
    if kind == "DownBlockSpatioTemporal":
        return DownBlockSpatioTemporal(cin, cout, temb, num_layers=n_layers, add_downsample=add_ds)
    if kind == "CrossAttnDownBlockSpatioTemporal":
        return CrossAttnDownBlockSpatioTemporal(cin, cout, temb, num_layers=n_layers, transformer_layers_per_block=tl,
                                                num_attention_heads=heads, cross_attention_dim=cross, add_downsample=add_ds)
    raise ValueError(f"{kind} does not exist.")


def _up(kind, n_layers, cin, cout, prev, temb, add_us, cross, heads, tl):
    """unet_3d_blocks.py:275-301 (SVD branch of get_up_block; resnet_eps is NOT forwarded)."""
    if kind == "UpBlockSpatioTemporal":
        return UpBlockSpatioTemporal(cin, prev, cout, temb, num_layers=n_layers, add_upsample=add_us)
    if kind == "CrossAttnUpBlockSpatioTemporal":
        return CrossAttnUpBlockSpatioTemporal(cin, cout, prev, temb, num_layers=n_layers, transformer_layers_per_block=tl,
                                              num_attention_heads=heads, cross_attention_dim=cross, add_upsample=add_us)
    raise ValueError(f"{kind} does not exist.")


def _tup(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def _time_embed(self, sample, timestep, added_time_ids):
    """unet...:399-432 == temporal_controlnet.py:527-560."""
    b = sample.shape[0]
    t = timestep
    if not torch.is_tensor(t):
        t = torch.tensor([t], dtype=torch.float64 if isinstance(t, float) else torch.int64, device=sample.device)
    elif t.dim() == 0:
        t = t[None].to(sample.device)
    t = t.expand(b)
    emb = self.time_embedding(self.time_proj(t).to(sample.dtype))
    te = self.add_time_proj(added_time_ids.flatten()).reshape(b, -1).to(emb.dtype)
    return emb + self.add_embedding(te)


class UNetSpatioTemporalConditionModel(nn.Module):
    """svd/unet_spatio_temporal_condition.py:38-536."""

    def __init__(self, sample_size=None, in_channels=8, out_channels=4,
                 down_block_types=("CrossAttnDownBlockSpatioTemporal",) * 3 + ("DownBlockSpatioTemporal",),
                 up_block_types=("UpBlockSpatioTemporal",) + ("CrossAttnUpBlockSpatioTemporal",) * 3,
                 block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim=256,
                 projection_class_embeddings_input_dim=768, layers_per_block=2, cross_attention_dim=1024,
                 transformer_layers_per_block=1, num_attention_heads=(5, 10, 10, 20), num_frames=25):
        super().__init__()
        nb = len(down_block_types)
        heads, cross = _tup(num_attention_heads, nb), _tup(cross_attention_dim, nb)
        lpb, tl = _tup(layers_per_block, nb), _tup(transformer_layers_per_block, nb)
        ch = block_out_channels
        temb = ch[0] * 4
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        self.time_proj = Timesteps(ch[0], True, 0)
        self.time_embedding = TimestepEmbedding(ch[0], temb)
        self.add_time_proj = Timesteps(addition_time_embed_dim, True, 0)
        self.add_embedding = TimestepEmbedding(projection_class_embeddings_input_dim, temb)
        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i, kind in enumerate(down_block_types):
            cin, out = out, ch[i]
            self.down_blocks.append(_down(kind, lpb[i], cin, out, temb, i != nb - 1, cross[i], heads[i], tl[i]))
        self.mid_block = UNetMidBlockSpatioTemporal(ch[-1], temb, transformer_layers_per_block=tl[-1],
                                                    cross_attention_dim=cross[-1], num_attention_heads=heads[-1])
        rch, rheads, rlpb, rcross, rtl = [list(reversed(x)) for x in (ch, heads, lpb, cross, tl)]
        self.up_blocks = nn.ModuleList()
        out = rch[0]
        for i, kind in enumerate(up_block_types):
            prev, out = out, rch[i]
            cin = rch[min(i + 1, nb - 1)]
            self.up_blocks.append(_up(kind, rlpb[i] + 1, cin, out, prev, temb, i != nb - 1, rcross[i], rheads[i], rtl[i]))
        self.conv_norm_out = nn.GroupNorm(32, ch[0], eps=1e-5)
        self.conv_out = nn.Conv2d(ch[0], out_channels, 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, added_positions=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None):
        b, f = sample.shape[:2]
        emb = _time_embed(self, sample, timestep, added_time_ids)
        sample = sample.flatten(0, 1)
        emb = emb.repeat_interleave(f, dim=0)
        ehs = encoder_hidden_states.repeat_interleave(f, dim=0)
        sample = self.conv_in(sample)
        ioi = torch.zeros(b, f, dtype=sample.dtype, device=sample.device)
        skips = (sample,)
        for blk in self.down_blocks:
            if getattr(blk, "has_cross_attention", False):
                sample, res = blk(sample, emb, ehs, ioi)
            else:
                sample, res = blk(sample, emb, ioi)
            skips += res
        is_cn = mid_block_additional_residual is not None and down_block_additional_residuals is not None
        if is_cn:                                                                             # :481-491 (Q4)
            skips = tuple(s + r for s, r in zip(skips, down_block_additional_residuals))
        sample = self.mid_block(sample, emb, ehs, ioi)
        if is_cn:
            sample = sample + mid_block_additional_residual
        for blk in self.up_blocks:
            k = len(blk.resnets)
            res, skips = skips[-k:], skips[:-k]
            if getattr(blk, "has_cross_attention", False):
                sample = blk(sample, res, emb, ehs, ioi)
            else:
                sample = blk(sample, res, emb, ioi)
        sample = self.conv_out(F.silu(self.conv_norm_out(sample)))
        return sample.reshape(b, f, *sample.shape[1:])


class ControlNetModel(nn.Module):
    """svd/temporal_controlnet.py:75-641 (GestureNet)."""

    def __init__(self, in_channels=8, conditioning_channels=3,
                 down_block_types=("CrossAttnDownBlockSpatioTemporal",) * 3 + ("DownBlockSpatioTemporal",),
                 block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim=256, layers_per_block=2,
                 cross_attention_dim=1024, projection_class_embeddings_input_dim=768,
                 transformer_layers_per_block=1, num_attention_heads=(5, 10, 20, 20)):
        super().__init__()
        nb = len(down_block_types)
        heads, cross = _tup(num_attention_heads, nb), _tup(cross_attention_dim, nb)
        lpb, tl = _tup(layers_per_block, nb), _tup(transformer_layers_per_block, nb)
        ch = block_out_channels
        temb = ch[0] * 4
        self.conv_in_concat = nn.Conv2d(12, ch[0], 3, padding=1)                               # :203-205
        self.time_proj = Timesteps(ch[0], True, 0)
        self.time_embedding = TimestepEmbedding(ch[0], temb)
        self.add_time_proj = Timesteps(addition_time_embed_dim, True, 0)
        self.add_embedding = TimestepEmbedding(projection_class_embeddings_input_dim, temb)
        self.down_blocks = nn.ModuleList()
        self.controlnet_down_blocks = nn.ModuleList([nn.Conv2d(ch[0], ch[0], 1)])
        out = ch[0]
        for i, kind in enumerate(down_block_types):
            cin, out = out, ch[i]
            self.down_blocks.append(_down(kind, lpb[i], cin, out, temb, i != nb - 1, cross[i], heads[i], tl[i]))
            for _ in range(lpb[0] + (0 if i == nb - 1 else 1)):                                # :281-289
                self.controlnet_down_blocks.append(nn.Conv2d(out, out, 1))
        self.controlnet_mid_block = nn.Conv2d(ch[-1], ch[-1], 1)
        self.mid_block = UNetMidBlockSpatioTemporal(ch[-1], temb, transformer_layers_per_block=tl[-1],
                                                    cross_attention_dim=cross[-1], num_attention_heads=heads[-1])
        for m in [self.conv_in_concat, self.controlnet_mid_block, *self.controlnet_down_blocks]:
            for p in m.parameters():
                nn.init.zeros_(p)

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, added_positions=None,
                controlnet_cond=None, conditioning_scale=1.0, guess_mode=False):
        b, f = sample.shape[:2]
        emb = _time_embed(self, sample, timestep, added_time_ids)
        sample = sample.flatten(0, 1)
        emb = emb.repeat_interleave(f, dim=0)
        ehs = encoder_hidden_states.repeat_interleave(f, dim=0)
        ioi = torch.zeros(b, f, dtype=sample.dtype, device=sample.device)
        sample = self.conv_in_concat(torch.cat([sample, controlnet_cond], dim=1))              # :576-580
        skips = (sample,)
        for blk in self.down_blocks:
            if getattr(blk, "has_cross_attention", False):
                sample, res = blk(sample, emb, ehs, ioi)
            else:
                sample, res = blk(sample, emb, ioi)
            skips += res
        sample = self.mid_block(sample, emb, ehs, ioi)
        down = [z(s) for s, z in zip(skips, self.controlnet_down_blocks)]
        mid = self.controlnet_mid_block(sample)
        if guess_mode:                                                                         # :626-630
            scales = torch.logspace(-1, 0, len(down) + 1) * conditioning_scale
            down = [d * s for d, s in zip(down, scales)]
            mid = mid * scales[-1]
        else:
            down = [d * conditioning_scale for d in down]
            mid = mid * conditioning_scale
        return down, mid
