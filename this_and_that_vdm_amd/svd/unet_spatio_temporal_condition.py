"""UNetSpatioTemporalConditionModel, MI355X-native drop-in for svd/unet_spatio_temporal_condition.py:38-536.

Same constructor, config, parameter names and ``forward`` signature/return types; the forward is a sequence of
libttvdm (gfx950 HIP) launches.  There is no CPU path: calling it off-device raises."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import ops
from ..packing import pack_conv3x3
from .denoiser_base import DenoiserBase, as_nchw_view, as_tokens
from .diffusion_arch.unet_3d_blocks import UNetMidBlockSpatioTemporal, get_down_block, get_up_block
from .layers import Geom, TimestepEmbedding, Timesteps, _f32, _gn
from .modeling_utils import BaseOutput, ConfigMixin, register_to_config


@dataclass
class UNetSpatioTemporalConditionOutput(BaseOutput):
    sample: torch.FloatTensor = None


class UNetSpatioTemporalConditionModel(DenoiserBase, ConfigMixin):
    _supports_gradient_checkpointing = False

    @register_to_config
    def __init__(
        self,
        sample_size: Optional[int] = None,
        in_channels: int = 8,
        out_channels: int = 4,
        down_block_types: Tuple[str] = ("CrossAttnDownBlockSpatioTemporal", "CrossAttnDownBlockSpatioTemporal",
                                        "CrossAttnDownBlockSpatioTemporal", "DownBlockSpatioTemporal"),
        up_block_types: Tuple[str] = ("UpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal",
                                      "CrossAttnUpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal"),
        block_out_channels: Tuple[int] = (320, 640, 1280, 1280),
        addition_time_embed_dim: int = 256,
        projection_class_embeddings_input_dim: int = 768,
        layers_per_block: Union[int, Tuple[int]] = 2,
        cross_attention_dim: Union[int, Tuple[int]] = 1024,
        transformer_layers_per_block: Union[int, Tuple[int], Tuple[Tuple]] = 1,
        num_attention_heads: Union[int, Tuple[int]] = (5, 10, 10, 20),
        num_frames: int = 25,
    ):
        super().__init__()
        self.sample_size = sample_size
        nb = len(down_block_types)
        # the reference's argument checks (:107-130), same messages' intent
        if len(down_block_types) != len(up_block_types):
            raise ValueError(f"Must provide the same number of `down_block_types` as `up_block_types`. `down_block_types`: {down_block_types}. `up_block_types`: {up_block_types}.")
        if len(block_out_channels) != nb:
            raise ValueError(f"Must provide the same number of `block_out_channels` as `down_block_types`. `block_out_channels`: {block_out_channels}. `down_block_types`: {down_block_types}.")
        for nm, v in (("num_attention_heads", num_attention_heads), ("layers_per_block", layers_per_block)):
            if not isinstance(v, int) and len(v) != nb:
                raise ValueError(f"Must provide the same number of `{nm}` as `down_block_types`. `{nm}`: {v}. `down_block_types`: {down_block_types}.")
        if isinstance(cross_attention_dim, list) and len(cross_attention_dim) != nb:
            raise ValueError(f"Must provide the same number of `cross_attention_dim` as `down_block_types`. `cross_attention_dim`: {cross_attention_dim}. `down_block_types`: {down_block_types}.")
        tup = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v,) * nb
        heads, cross, lpb, tl = tup(num_attention_heads), tup(cross_attention_dim), tup(layers_per_block), tup(transformer_layers_per_block)
        ch = tuple(block_out_channels)
        temb = ch[0] * 4

        self.conv_in = nn.Conv2d(in_channels, ch[0], kernel_size=3, padding=1)
        self.time_proj = Timesteps(ch[0], True, downscale_freq_shift=0)
        self.time_embedding = TimestepEmbedding(ch[0], temb)
        self.add_time_proj = Timesteps(addition_time_embed_dim, True, downscale_freq_shift=0)
        self.add_embedding = TimestepEmbedding(projection_class_embeddings_input_dim, temb)

        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i, kind in enumerate(down_block_types):
            cin, out = out, ch[i]
            self.down_blocks.append(get_down_block(kind, num_layers=lpb[i], transformer_layers_per_block=tl[i], in_channels=cin,
                                                   out_channels=out, temb_channels=temb, add_downsample=i != nb - 1,
                                                   resnet_eps=1e-5, cross_attention_dim=cross[i],
                                                   num_attention_heads=heads[i], resnet_act_fn="silu"))
        self.mid_block = UNetMidBlockSpatioTemporal(ch[-1], temb_channels=temb, transformer_layers_per_block=tl[-1],
                                                    cross_attention_dim=cross[-1], num_attention_heads=heads[-1])
        self.num_upsamplers = 0
        self.up_blocks = nn.ModuleList()
        rch, rheads, rlpb, rcross, rtl = [list(reversed(v)) for v in (ch, heads, lpb, cross, tl)]
        out = rch[0]
        for i, kind in enumerate(up_block_types):
            prev, out = out, rch[i]
            cin = rch[min(i + 1, nb - 1)]
            last = i == nb - 1
            self.num_upsamplers += 0 if last else 1
            self.up_blocks.append(get_up_block(kind, num_layers=rlpb[i] + 1, transformer_layers_per_block=rtl[i], in_channels=cin,
                                               out_channels=out, prev_output_channel=prev, temb_channels=temb,
                                               add_upsample=not last, resnet_eps=1e-5, resolution_idx=i,
                                               cross_attention_dim=rcross[i], num_attention_heads=rheads[i],
                                               resnet_act_fn="silu"))
        self.conv_norm_out = nn.GroupNorm(num_channels=ch[0], num_groups=32, eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(ch[0], out_channels, kernel_size=3, padding=1)

    # ---- packing
    def _pack_modules(self, reg, dtype):
        cin = self.conv_in.in_channels
        self._cin_pad = (cin + 7) // 8 * 8
        w = self.conv_in.weight.detach().to(dtype)
        if self._cin_pad != cin:
            w = torch.cat([w, w.new_zeros(w.shape[0], self._cin_pad - cin, 3, 3)], 1)
        self._w_in, self._b_in = pack_conv3x3(w), _f32(self.conv_in.bias)
        for m in (self.time_embedding, self.add_embedding, *self.down_blocks, self.mid_block, *self.up_blocks):
            m.pack(reg, dtype)
        self._gn_out = (_f32(self.conv_norm_out.weight), _f32(self.conv_norm_out.bias))
        self._w_out, self._b_out = pack_conv3x3(self.conv_out.weight.detach().to(dtype)), _f32(self.conv_out.bias)

    # ---- forward
    def encode_tokens(self, x_tok, g: Geom, ctx):
        """conv_in + 4 down blocks + mid block -> (x_mid, geom_mid, skips[12]) ; no ControlNet terms yet."""
        x = ops.gemm(x_tok, self._w_in, mode=1, conv=(g.n, g.h, g.w, g.h, g.w, 1, 0), bias=self._b_in, stats=g.hw)
        x, gm, skips = self._encode(x, g, ctx)
        return self.mid_block(x, gm, ctx), gm, skips

    def decode_tokens(self, x, gm: Geom, skips, ctx, eps_out=None):
        """4 up blocks + GN/SiLU/conv_out -> eps fp32 [M, out_channels] (written into ``eps_out`` if given)."""
        skips = list(skips)
        for blk in self.up_blocks:
            x, gm = blk(x, skips, gm, ctx)
        a = _gn(x, None, gm, 1, self._gn_out[0], self._gn_out[1], 1e-5, True)
        return ops.gemm(a, self._w_out, mode=1, conv=(gm.n, gm.h, gm.w, gm.h, gm.w, 1, 0), bias=self._b_out, out_f32=True,
                        out=eps_out)

    def forward_tokens(self, x_tok, g: Geom, emb, context, down_res_tok=None, mid_res_tok=None):
        """Token-level core: x_tok [M, cin_pad] -> eps fp32 [M, out_channels].  The reference adds the ControlNet
        residuals to all 12 skips after the encoder and to the mid output (:481-502, quirk Q4); the mid block
        itself never sees them, so encode+mid can run before the residuals exist."""
        ctx = self._step_context(emb, context)
        x, gm, skips = self.encode_tokens(x_tok, g, ctx)
        if down_res_tok is not None:
            if len(down_res_tok) != len(skips):
                raise ValueError(f"expected {len(skips)} down-block residuals, got {len(down_res_tok)}")
            skips = [(ops.add_scaled(s, r), sg) for (s, sg), r in zip(skips, down_res_tok)]
        if mid_res_tok is not None:
            x = ops.add_scaled(x, mid_res_tok)
        return self.decode_tokens(x, gm, skips, ctx)

    def forward(
        self,
        sample: torch.FloatTensor,
        timestep: Union[torch.Tensor, float, int],
        encoder_hidden_states: torch.Tensor,
        added_time_ids: torch.Tensor,
        added_positions: torch.Tensor = None,
        down_block_additional_residuals: Optional[Tuple[torch.Tensor]] = None,
        mid_block_additional_residual: Optional[torch.Tensor] = None,
        return_dict: bool = True,
        _context=None,
    ) -> Union[UNetSpatioTemporalConditionOutput, Tuple]:
        """sample [B,F,C_in,h,w]; encoder_hidden_states [B,S,D]; added_time_ids [B,3] -> [B,F,C_out,h,w] in sample.dtype.
        ``added_positions`` is accepted and ignored, as in the reference (:369,435-440)."""
        if not sample.is_cuda:
            raise RuntimeError("UNetSpatioTemporalConditionModel.forward: inputs must be on the HIP device (no CPU fallback)")
        self.prepare()
        dtype = self._run_dtype()
        b, f, cin, h, w = sample.shape
        g = Geom(b, f, h, w)
        emb = self._embed(timestep, added_time_ids, b, sample.device)
        context = _context if _context is not None else self.project_context(encoder_hidden_states)
        x_tok = ops.nchw_to_tokens(sample.reshape(b * f, cin, h, w), dtype, ld=self._cin_pad)
        is_controlnet = mid_block_additional_residual is not None and down_block_additional_residuals is not None
        down_tok = [as_tokens(r, dtype) for r in down_block_additional_residuals] if is_controlnet else None
        mid_tok = as_tokens(mid_block_additional_residual, dtype) if is_controlnet else None
        eps = self.forward_tokens(x_tok, g, emb, context, down_tok, mid_tok)
        cout = self.conv_out.out_channels
        out_dtype = sample.dtype if sample.dtype in (torch.float32, dtype) else torch.float32
        out = ops.tokens_to_nchw(eps, b * f, cout, h, w, out_dtype).reshape(b, f, cout, h, w).to(sample.dtype)
        if not return_dict:
            return (out,)
        return UNetSpatioTemporalConditionOutput(sample=out)
