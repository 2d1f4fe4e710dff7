"""GestureNet ``ControlNetModel``, MI355X-native drop-in for svd/temporal_controlnet.py:75-641.

UNet encoder + mid block over a 12-channel input (8 latent + 4 gesture-latent channels, conv_in_concat),
12 + 1 zero-initialised 1x1 convolutions on the skips, scaled by ``conditioning_scale``.  The scale is the
``acc_scale`` of the zero-conv GEMM epilogue; the returned residuals are channels-last views of the token
buffers, so the UNet consumes them without a layout change."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import ops
from ..packing import pack_conv1x1, pack_conv3x3
from .denoiser_base import DenoiserBase, as_nchw_view
from .diffusion_arch.unet_3d_blocks import UNetMidBlockSpatioTemporal, get_down_block
from .layers import Geom, TimestepEmbedding, Timesteps, _f32
from .modeling_utils import BaseOutput, ConfigMixin, register_to_config
from .unet_spatio_temporal_condition import UNetSpatioTemporalConditionModel


def zero_module(module):
    for p in module.parameters():
        nn.init.zeros_(p)
    return module


@dataclass
class ControlNetOutput(BaseOutput):
    down_block_res_samples: Tuple[torch.Tensor] = None
    mid_block_res_sample: torch.Tensor = None


class ControlNetModel(DenoiserBase, ConfigMixin):
    _supports_gradient_checkpointing = False

    @register_to_config
    def __init__(
        self,
        in_channels: int = 8,
        conditioning_channels: int = 3,
        flip_sin_to_cos: bool = True,
        freq_shift: int = 0,
        down_block_types: Tuple[str, ...] = ("CrossAttnDownBlockSpatioTemporal", "CrossAttnDownBlockSpatioTemporal",
                                             "CrossAttnDownBlockSpatioTemporal", "DownBlockSpatioTemporal"),
        mid_block_type: Optional[str] = "UNetMidBlockSpatioTemporal",
        block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280),
        addition_time_embed_dim: int = 256,
        layers_per_block: int = 2,
        act_fn: str = "silu",
        cross_attention_dim: int = 1024,
        projection_class_embeddings_input_dim: Optional[int] = 768,
        conditioning_embedding_out_channels: Optional[Tuple[int, ...]] = (16, 32, 96, 256),
        transformer_layers_per_block: Union[int, Tuple[int], Tuple[Tuple]] = 1,
        num_attention_heads: Union[int, Tuple[int]] = (5, 10, 20, 20),
        encoder_hid_dim: Optional[int] = None,
        encoder_hid_dim_type: Optional[str] = None,
        controlnet_conditioning_channel_order="rgb",
    ):
        super().__init__()
        self.controlnet_conditioning_channel_order = controlnet_conditioning_channel_order
        nb = len(down_block_types)
        if len(block_out_channels) != nb:
            raise ValueError(f"Must provide the same number of `block_out_channels` as `down_block_types`. `block_out_channels`: {block_out_channels}. `down_block_types`: {down_block_types}.")
        if not isinstance(num_attention_heads, int) and len(num_attention_heads) != nb:
            raise ValueError(f"Must provide the same number of `num_attention_heads` as `down_block_types`. `num_attention_heads`: {num_attention_heads}. `down_block_types`: {down_block_types}.")
        if encoder_hid_dim is None and encoder_hid_dim_type is not None:
            raise ValueError(f"`encoder_hid_dim` has to be defined when `encoder_hid_dim_type` is set to {encoder_hid_dim_type}.")
        tup = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v,) * nb
        heads, cross, lpb, tl = tup(num_attention_heads), tup(cross_attention_dim), tup(layers_per_block), tup(transformer_layers_per_block)
        ch = tuple(block_out_channels)
        temb = ch[0] * 4

        self.conv_in_concat = zero_module(nn.Conv2d(12, ch[0], kernel_size=3, padding=1))          # :203-205
        self.time_proj = Timesteps(ch[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(ch[0], temb, act_fn=act_fn)
        self.add_time_proj = Timesteps(addition_time_embed_dim, True, downscale_freq_shift=0)
        self.add_embedding = TimestepEmbedding(projection_class_embeddings_input_dim, temb)

        self.down_blocks = nn.ModuleList()
        self.controlnet_down_blocks = nn.ModuleList([zero_module(nn.Conv2d(ch[0], ch[0], kernel_size=1))])
        out = ch[0]
        for i, kind in enumerate(down_block_types):
            cin, out = out, ch[i]
            last = i == nb - 1
            self.down_blocks.append(get_down_block(kind, num_layers=lpb[i], transformer_layers_per_block=tl[i], in_channels=cin,
                                                   out_channels=out, temb_channels=temb, add_downsample=not last,
                                                   resnet_eps=1e-5, cross_attention_dim=cross[i],
                                                   num_attention_heads=heads[i], resnet_act_fn="silu"))
            for _ in range(lpb[0] + (0 if last else 1)):                                            # :281-289
                self.controlnet_down_blocks.append(zero_module(nn.Conv2d(out, out, kernel_size=1)))
        self.controlnet_mid_block = zero_module(nn.Conv2d(ch[-1], ch[-1], kernel_size=1))
        if mid_block_type != "UNetMidBlockSpatioTemporal":
            raise ValueError(f"unknown mid_block_type : {mid_block_type}")
        self.mid_block = UNetMidBlockSpatioTemporal(ch[-1], temb_channels=temb, transformer_layers_per_block=tl[-1],
                                                    cross_attention_dim=cross[-1], num_attention_heads=heads[-1])

    @classmethod
    def from_unet(cls, unet: UNetSpatioTemporalConditionModel, conditioning_channels: int = 3, load_weights_from_unet: bool = True):
        """:311-339 -- built with the ControlNet's OWN defaults (quirk Q2); add_embedding is NOT copied."""
        controlnet = cls(conditioning_channels=conditioning_channels)
        if load_weights_from_unet:
            controlnet.time_proj.load_state_dict(unet.time_proj.state_dict())
            controlnet.time_embedding.load_state_dict(unet.time_embedding.state_dict())
            controlnet.down_blocks.load_state_dict(unet.down_blocks.state_dict())
            controlnet.mid_block.load_state_dict(unet.mid_block.state_dict())
        return controlnet

    # ---- packing
    def _pack_modules(self, reg, dtype):
        cin = self.conv_in_concat.in_channels
        self._cin_pad = (cin + 7) // 8 * 8
        w = self.conv_in_concat.weight.detach().to(dtype)
        if self._cin_pad != cin:
            w = torch.cat([w, w.new_zeros(w.shape[0], self._cin_pad - cin, 3, 3)], 1)
        self._w_in, self._b_in = pack_conv3x3(w), _f32(self.conv_in_concat.bias)
        for m in (self.time_embedding, self.add_embedding, *self.down_blocks, self.mid_block):
            m.pack(reg, dtype)
        self._zero = [(pack_conv1x1(z.weight.detach().to(dtype)), _f32(z.bias)) for z in self.controlnet_down_blocks]
        self._zero_mid = (pack_conv1x1(self.controlnet_mid_block.weight.detach().to(dtype)), _f32(self.controlnet_mid_block.bias))

    # ---- forward
    def encode_tokens(self, x_tok, g: Geom, ctx):
        """conv_in_concat + 4 down blocks + mid block -> (12 skips [(tok, geom)], mid tokens, mid geom)."""
        x = ops.gemm(x_tok, self._w_in, mode=1, conv=(g.n, g.h, g.w, g.h, g.w, 1, 0), bias=self._b_in, stats=g.hw)
        x, gm, skips = self._encode(x, g, ctx)
        return skips, self.mid_block(x, gm, ctx), gm

    def zero_convs(self, skips, mid, gm: Geom, scales: List[float], add_to=None):
        """12 + 1 zero-initialised 1x1 convs (:614-622) with the conditioning scale as acc_scale; with ``add_to`` =
        (unet_skips, unet_mid) the UNet tensors are added in the epilogue (fused `skip += residual`)."""
        down = []
        for i, ((s, sg), (wz, bz)) in enumerate(zip(skips, self._zero)):
            res = add_to[0][i] if add_to is not None else None
            down.append((ops.gemm(s, wz, bias=bz, acc_scale=scales[i], residual=res), sg))
        mid_out = ops.gemm(mid, self._zero_mid[0], bias=self._zero_mid[1], acc_scale=scales[-1],
                           residual=add_to[1] if add_to is not None else None)
        return down, (mid_out, gm)

    def forward_tokens(self, x_tok, g: Geom, emb, context, scales: List[float], add_to=None):
        """x_tok [M, cin_pad] (latent | image-latent | gesture-latent channels) -> 12 down residuals + mid residual."""
        ctx = self._step_context(emb, context)
        skips, mid, gm = self.encode_tokens(x_tok, g, ctx)
        return self.zero_convs(skips, mid, gm, scales, add_to)

    def _scales(self, conditioning_scale: float, guess_mode: bool, n_down: int) -> List[float]:
        if guess_mode:                                                   # :626-630 logspace(-1, 0, 13) * scale
            ls = torch.logspace(-1, 0, n_down + 1).tolist()
            return [v * conditioning_scale for v in ls]
        return [float(conditioning_scale)] * (n_down + 1)

    def forward(
        self,
        sample: torch.FloatTensor,
        timestep: Union[torch.Tensor, float, int],
        encoder_hidden_states: torch.Tensor,
        added_time_ids: torch.Tensor,
        added_positions: torch.Tensor = None,
        controlnet_cond: torch.FloatTensor = None,
        conditioning_scale: float = 1.0,
        inner_conditioning_scale: float = 1.0,
        timestep_cond: Optional[torch.Tensor] = None,
        attention_mask: Optional[torch.Tensor] = None,
        guess_mode: bool = False,
        return_dict: bool = True,
        _context=None,
    ) -> Union[ControlNetOutput, Tuple[Tuple[torch.FloatTensor, ...], torch.FloatTensor]]:
        """sample [B,F,8,h,w]; controlnet_cond [B*F,4,h,w] (VAE-encoded gesture map, already flattened over B*F).
        ``added_positions``, ``inner_conditioning_scale``, ``timestep_cond``, ``attention_mask`` are accepted and
        ignored, as in the reference (:461-467,522-524).  Returns 12 tensors [B*F,C,h',w'] + mid [B*F,C,h/8,w/8]."""
        if not sample.is_cuda:
            raise RuntimeError("ControlNetModel.forward: inputs must be on the HIP device (no CPU fallback)")
        if controlnet_cond is None:
            raise ValueError("controlnet_cond is required")
        self.prepare()
        dtype = self._run_dtype()
        b, f, cin, h, w = sample.shape
        g = Geom(b, f, h, w)
        emb = self._embed(timestep, added_time_ids, b, sample.device)
        context = _context if _context is not None else self.project_context(encoder_hidden_states)
        x_tok = torch.zeros((g.m, self._cin_pad), dtype=dtype, device=sample.device)
        ops.nchw_to_tokens(sample.reshape(b * f, cin, h, w), dtype, out=x_tok[:, :cin])
        cc = controlnet_cond.shape[1]
        ops.nchw_to_tokens(controlnet_cond.reshape(b * f, cc, h, w), dtype, out=x_tok[:, cin:cin + cc])
        scales = self._scales(float(conditioning_scale), guess_mode, len(self.controlnet_down_blocks))
        down, (mid, gm) = self.forward_tokens(x_tok, g, emb, context, scales)
        down_out = [as_nchw_view(t, sg) for t, sg in down]
        mid_out = as_nchw_view(mid, gm)
        if sample.dtype != dtype:
            down_out = [d.to(sample.dtype) for d in down_out]
            mid_out = mid_out.to(sample.dtype)
        if not return_dict:
            return (down_out, mid_out)
        return ControlNetOutput(down_block_res_samples=down_out, mid_block_res_sample=mid_out)
