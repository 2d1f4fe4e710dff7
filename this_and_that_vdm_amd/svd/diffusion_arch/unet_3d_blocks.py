"""Spatio-temporal UNet blocks on libttvdm kernels (reference: svd/diffusion_arch/unet_3d_blocks.py:1870-2396,
factories :39-303).  Same class names, constructor arguments and parameter names; GroupNorm eps per block type
follows the reference exactly.  Skip-concats (:2242,2352) are never materialised: the
two tensors are passed as the two channel sources of the GroupNorm / GEMM kernels."""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch.nn as nn

from ..layers import Downsample2D, Geom, SpatioTemporalResBlock, StepContext, Upsample2D, _Packable
from .transformer_temporal import TransformerSpatioTemporalModel


def _tfm(heads, channels, cross, layers):
    return TransformerSpatioTemporalModel(heads, channels // heads, in_channels=channels, num_layers=layers,
                                          cross_attention_dim=cross)


class _Block(_Packable):
    def pack(self, reg, dtype):
        for name in ("resnets", "attentions", "downsamplers", "upsamplers"):
            mods = getattr(self, name, None)
            if mods is not None:
                for m in mods:
                    m.pack(reg, dtype)


class UNetMidBlockSpatioTemporal(_Block):
    """:1870-1977 -- STRes(eps 1e-5) -> Tfm -> STRes(eps 1e-5)."""

    def __init__(self, in_channels: int, temb_channels: int, num_layers: int = 1,
                 transformer_layers_per_block: Union[int, Tuple[int]] = 1, num_attention_heads: int = 1,
                 cross_attention_dim: int = 1280):
        super().__init__()
        self.has_cross_attention = True
        self.num_attention_heads = num_attention_heads
        tl = [transformer_layers_per_block] * num_layers if isinstance(transformer_layers_per_block, int) else transformer_layers_per_block
        self.attentions = nn.ModuleList([_tfm(num_attention_heads, in_channels, cross_attention_dim, tl[i]) for i in range(num_layers)])
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(in_channels, in_channels, temb_channels, eps=1e-5)
                                      for _ in range(num_layers + 1)])

    def forward(self, x, g: Geom, ctx: StepContext):
        x = self.resnets[0](x, None, g, ctx)
        for attn, res in zip(self.attentions, self.resnets[1:]):
            x = res(attn(x, g, ctx), None, g, ctx)
        return x


class DownBlockSpatioTemporal(_Block):
    """:1980-2067 (eps 1e-5)."""

    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, num_layers: int = 1, add_downsample: bool = True):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels,
                                                             temb_channels, eps=1e-5) for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels, name="op")]) \
            if add_downsample else None

    def forward(self, x, g: Geom, ctx: StepContext):
        outs = []
        for res in self.resnets:
            x = res(x, None, g, ctx)
            outs.append((x, g))
        if self.downsamplers is not None:
            x, g = self.downsamplers[0](x, g)
            outs.append((x, g))
        return x, g, outs


class CrossAttnDownBlockSpatioTemporal(_Block):
    """:2070-2189 (ResBlock eps hard-coded 1e-6 at :2098)."""

    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, num_layers: int = 1,
                 transformer_layers_per_block: Union[int, Tuple[int]] = 1, num_attention_heads: int = 1,
                 cross_attention_dim: int = 1280, add_downsample: bool = True):
        super().__init__()
        self.has_cross_attention = True
        self.num_attention_heads = num_attention_heads
        tl = [transformer_layers_per_block] * num_layers if isinstance(transformer_layers_per_block, int) else transformer_layers_per_block
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels,
                                                             temb_channels, eps=1e-6) for i in range(num_layers)])
        self.attentions = nn.ModuleList([_tfm(num_attention_heads, out_channels, cross_attention_dim, tl[i]) for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels, padding=1, name="op")]) \
            if add_downsample else None

    def forward(self, x, g: Geom, ctx: StepContext):
        outs = []
        for res, attn in zip(self.resnets, self.attentions):
            x = attn(res(x, None, g, ctx), g, ctx)
            outs.append((x, g))
        if self.downsamplers is not None:
            x, g = self.downsamplers[0](x, g)
            outs.append((x, g))
        return x, g, outs


class UpBlockSpatioTemporal(_Block):
    """:2192-2278 (eps 1e-6: the factory does not forward resnet_eps, :277-285)."""

    def __init__(self, in_channels: int, prev_output_channel: int, out_channels: int, temb_channels: int,
                 resolution_idx: Optional[int] = None, num_layers: int = 1, resnet_eps: float = 1e-6, add_upsample: bool = True):
        super().__init__()
        res = []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            cin = prev_output_channel if i == 0 else out_channels
            res.append(SpatioTemporalResBlock(cin + skip, out_channels, temb_channels, eps=resnet_eps))
        self.resnets = nn.ModuleList(res)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)]) if add_upsample else None
        self.resolution_idx = resolution_idx

    def forward(self, x, skips, g: Geom, ctx: StepContext):
        for res in self.resnets:
            x = res(x, skips.pop()[0], g, ctx)
        if self.upsamplers is not None:
            x, g = self.upsamplers[0](x, g)
        return x, g


class CrossAttnUpBlockSpatioTemporal(_Block):
    """:2281-2396."""

    def __init__(self, in_channels: int, out_channels: int, prev_output_channel: int, temb_channels: int,
                 resolution_idx: Optional[int] = None, num_layers: int = 1,
                 transformer_layers_per_block: Union[int, Tuple[int]] = 1, resnet_eps: float = 1e-6,
                 num_attention_heads: int = 1, cross_attention_dim: int = 1280, add_upsample: bool = True):
        super().__init__()
        self.has_cross_attention = True
        self.num_attention_heads = num_attention_heads
        tl = [transformer_layers_per_block] * num_layers if isinstance(transformer_layers_per_block, int) else transformer_layers_per_block
        res, att = [], []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            cin = prev_output_channel if i == 0 else out_channels
            res.append(SpatioTemporalResBlock(cin + skip, out_channels, temb_channels, eps=resnet_eps))
            att.append(_tfm(num_attention_heads, out_channels, cross_attention_dim, tl[i]))
        self.resnets, self.attentions = nn.ModuleList(res), nn.ModuleList(att)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)]) if add_upsample else None
        self.resolution_idx = resolution_idx

    def forward(self, x, skips, g: Geom, ctx: StepContext):
        for res, attn in zip(self.resnets, self.attentions):
            x = attn(res(x, skips.pop()[0], g, ctx), g, ctx)
        if self.upsamplers is not None:
            x, g = self.upsamplers[0](x, g)
        return x, g


def get_down_block(down_block_type: str, num_layers: int, in_channels: int, out_channels: int, temb_channels: int,
                   add_downsample: bool, resnet_eps: float = 1e-5, resnet_act_fn: str = "silu", num_attention_heads: int = 1,
                   cross_attention_dim: Optional[int] = None, transformer_layers_per_block: int = 1, **unused):
    """SVD branches of the reference factory (:140-162); ``resnet_eps`` is accepted and ignored, as there."""
    if down_block_type == "DownBlockSpatioTemporal":
        return DownBlockSpatioTemporal(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
                                       temb_channels=temb_channels, add_downsample=add_downsample)
    if down_block_type == "CrossAttnDownBlockSpatioTemporal":
        if cross_attention_dim is None:
            raise ValueError("cross_attention_dim must be specified for CrossAttnDownBlockSpatioTemporal")
        return CrossAttnDownBlockSpatioTemporal(in_channels=in_channels, out_channels=out_channels, temb_channels=temb_channels,
                                                num_layers=num_layers, transformer_layers_per_block=transformer_layers_per_block,
                                                add_downsample=add_downsample, cross_attention_dim=cross_attention_dim,
                                                num_attention_heads=num_attention_heads)
    raise ValueError(f"{down_block_type} does not exist.")


def get_up_block(up_block_type: str, num_layers: int, in_channels: int, out_channels: int, prev_output_channel: int,
                 temb_channels: int, add_upsample: bool, resnet_eps: float = 1e-5, resnet_act_fn: str = "silu",
                 resolution_idx: Optional[int] = None, num_attention_heads: int = 1, cross_attention_dim: Optional[int] = None,
                 transformer_layers_per_block: int = 1, **unused):
    """SVD branches of the reference factory (:275-301)."""
    if up_block_type == "UpBlockSpatioTemporal":
        return UpBlockSpatioTemporal(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
                                     prev_output_channel=prev_output_channel, temb_channels=temb_channels,
                                     resolution_idx=resolution_idx, add_upsample=add_upsample)
    if up_block_type == "CrossAttnUpBlockSpatioTemporal":
        if cross_attention_dim is None:
            raise ValueError("cross_attention_dim must be specified for CrossAttnUpBlockSpatioTemporal")
        return CrossAttnUpBlockSpatioTemporal(in_channels=in_channels, out_channels=out_channels,
                                              prev_output_channel=prev_output_channel, temb_channels=temb_channels,
                                              num_layers=num_layers, transformer_layers_per_block=transformer_layers_per_block,
                                              add_upsample=add_upsample, cross_attention_dim=cross_attention_dim,
                                              num_attention_heads=num_attention_heads, resolution_idx=resolution_idx)
    raise ValueError(f"{up_block_type} does not exist.")
