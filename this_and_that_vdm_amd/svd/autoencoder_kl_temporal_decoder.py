"""MI355X-native temporal VAE *decoder*: what `decode_latents` runs after the denoise loop
(reference svd/pipeline_stable_video_diffusion_controlnet.py:257-283, called at :722-726;
`self.vae.decode(latents[i:i+chunk], num_frames=chunk).sample`).

The reference takes the class from diffusers==0.25.1 (`AutoencoderKLTemporalDecoder`, not vendored); this module keeps its
decoder-side surface -- `decode(z, num_frames).sample`, `config.scaling_factor`, the `decoder.*` state-dict keys of a diffusers
checkpoint -- and runs the arithmetic on the libttvdm kernels the UNet path already uses (token-major activations):

  conv_in 3x3 -> mid block [SpatioTemporalResBlock, single-head attention over h*w tokens (d = 512), SpatioTemporalResBlock]
  -> 4 up blocks of 3 SpatioTemporalResBlocks (+ nearest x2 upsample fused into the following 3x3 conv)
  -> GroupNorm + SiLU -> conv_out 3x3 -> 3-tap conv along the frame axis.

SpatioTemporalResBlock here = the UNet's (layers.py) without a time embedding, AlphaBlender("learned",
switch_spatial_to_temporal_mix=True).  The mid-block attention has ONE head of 512 channels, outside tt_attention's 64 / 128:
per frame  scores = tt_gemm(Q, K, fp32 out, scale d^-1/2) -> tt_softmax_rows -> tt_gemm(P, V^T)  (V's bias rides on to_out's
bias: softmax rows sum to 1).  The encoder side (one image + the gesture frames per request, reference :168-188,:652) is not on
the hot path and stays with the caller's stock module: ``AutoencoderKLTemporalDecoder(encoder=stock_vae)`` /
``from_pretrained(folder, encoder=stock_vae)`` / ``.with_encoder(stock_vae)`` make ``encode()`` delegate to it, so ONE object
serves the pipeline's ``vae=`` argument for both directions (test_code/inference.py:169-176 passes one ``vae``).
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..packing import pack_conv3x3, pack_tconv3
from .layers import Geom, PackRegistry, SpatioTemporalResBlock, Upsample2D, _f32, _Packable
from .modeling_utils import ConfigMixin, ModelMixin, register_to_config


def _res(cin: int, cout: int) -> SpatioTemporalResBlock:
    return SpatioTemporalResBlock(cin, cout, temb_channels=None, eps=1e-6, temporal_eps=1e-5, merge_factor=0.0,
                                  merge_strategy="learned", switch_spatial_to_temporal_mix=True)


class VaeAttention(_Packable):
    """GroupNorm -> q / k / v (bias) -> softmax(q k^T / sqrt(d)) v -> to_out -> + input, one sequence per frame."""

    def __init__(self, query_dim: int, heads: int, dim_head: int, eps: float = 1e-6, norm_num_groups: int = 32):
        super().__init__()
        if heads != 1:
            raise NotImplementedError("the temporal VAE decoder uses one attention head (attention_head_dim = channels)")
        if norm_num_groups != 32:
            raise NotImplementedError("GroupNorm kernels are built for 32 groups")
        self.dim, self.eps = query_dim, eps
        self.group_norm = nn.GroupNorm(norm_num_groups, query_dim, eps=eps)
        self.to_q = nn.Linear(query_dim, query_dim, bias=True)
        self.to_k = nn.Linear(query_dim, query_dim, bias=True)
        self.to_v = nn.Linear(query_dim, query_dim, bias=True)
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim, bias=True), nn.Dropout(0.0)])

    def pack(self, reg, dtype):
        cv = lambda t: t.detach().to(dtype).contiguous()
        self.gamma, self.beta = _f32(self.group_norm.weight), _f32(self.group_norm.bias)
        self.wq, self.bq = cv(self.to_q.weight), _f32(self.to_q.bias)
        self.wk, self.bk = cv(self.to_k.weight), _f32(self.to_k.bias)
        self.wv = cv(self.to_v.weight)
        self.wo = cv(self.to_out[0].weight)
        # P (V + 1 bv^T) = P V + bv^T  (rows of P sum to 1): V's bias becomes part of to_out's
        self.bo = (_f32(self.to_out[0].bias) + self.wo.float() @ _f32(self.to_v.bias)).contiguous()
        self._scratch = {}

    def _scratch_for(self, like: torch.Tensor, n: int, l: int, lp: int):
        key = (like.device, like.dtype, n, l, torch.cuda.current_stream().cuda_stream)
        buf = self._scratch.get(key)
        if buf is None:
            buf = self._scratch[key] = (torch.zeros((self.dim, n * lp), dtype=like.dtype, device=like.device),     # V^T, padding stays 0
                                        torch.empty((l, lp), dtype=torch.float32, device=like.device))           # scores of one frame
        return buf

    def forward(self, x: torch.Tensor, g: Geom) -> torch.Tensor:
        c, l = self.dim, g.hw
        if l % 4:
            raise NotImplementedError(f"VAE attention over {l} tokens per frame: h*w must be a multiple of 4 (latents are multiples of 8)")
        lp = (l + 7) // 8 * 8
        xn = ops.groupnorm(x, None, g.n, l, 1, self.gamma, self.beta, self.eps, False)
        q = ops.gemm(xn, self.wq, bias=self.bq)
        k = ops.gemm(xn, self.wk, bias=self.bk)
        vt, scores = self._scratch_for(x, g.n, l, lp)
        ops.gemm(self.wv, xn, out=vt, out_col_pad=(l, lp) if lp != l else None)
        o = torch.empty((g.m, c), dtype=x.dtype, device=x.device)
        scale = 1.0 / math.sqrt(c)
        for f in range(g.n):
            rows = slice(f * l, (f + 1) * l)
            ops.gemm(q[rows], k[rows], acc_scale=scale, out=scores[:, :l], out_f32=x.dtype != torch.float32)
            p = ops.softmax_rows(scores, x.dtype, cols=l)                     # [l, lp], padding columns zero
            ops.gemm(p, vt[:, f * lp:(f + 1) * lp], out=o[rows])
        return ops.gemm(o, self.wo, bias=self.bo, residual=x)


class MidBlockTemporalDecoder(_Packable):
    def __init__(self, in_channels: int, out_channels: int, attention_head_dim: int = 512, num_layers: int = 1):
        super().__init__()
        self.resnets = nn.ModuleList([_res(in_channels if i == 0 else out_channels, out_channels) for i in range(num_layers)])
        self.attentions = nn.ModuleList([VaeAttention(in_channels, in_channels // attention_head_dim, attention_head_dim)])

    def pack(self, reg, dtype):
        for m in list(self.resnets) + list(self.attentions):
            m.pack(reg, dtype)

    def forward(self, x, g: Geom):
        x = self.resnets[0](x, None, g, None)
        for resnet, attn in zip(list(self.resnets)[1:], self.attentions):
            x = attn(x, g)
            x = resnet(x, None, g, None)
        return x


class UpBlockTemporalDecoder(_Packable):
    def __init__(self, in_channels: int, out_channels: int, num_layers: int = 1, add_upsample: bool = True):
        super().__init__()
        self.resnets = nn.ModuleList([_res(in_channels if i == 0 else out_channels, out_channels) for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)]) if add_upsample else None

    def pack(self, reg, dtype):
        for m in self.resnets:
            m.pack(reg, dtype)
        if self.upsamplers is not None:
            self.upsamplers[0].pack(reg, dtype)

    def forward(self, x, g: Geom):
        for resnet in self.resnets:
            x = resnet(x, None, g, None)
        if self.upsamplers is not None:
            x, g = self.upsamplers[0](x, g)
        return x, g


class TemporalDecoder(_Packable):
    CIN_PAD = 8        # latent channels (4) padded to one 16-byte chunk; conv_out's 3 channels likewise (input of the frame conv)

    def __init__(self, in_channels: int = 4, out_channels: int = 3, block_out_channels: Tuple[int, ...] = (128, 256, 512, 512),
                 layers_per_block: int = 2):
        super().__init__()
        if in_channels > self.CIN_PAD or out_channels > 4:
            raise NotImplementedError("latent channels <= 8 and image channels <= 4")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[-1], 3, padding=1)
        self.mid_block = MidBlockTemporalDecoder(num_layers=layers_per_block, in_channels=block_out_channels[-1],
                                                 out_channels=block_out_channels[-1], attention_head_dim=block_out_channels[-1])
        self.up_blocks = nn.ModuleList([])
        rev = list(reversed(block_out_channels))
        out_ch = rev[0]
        for i in range(len(block_out_channels)):
            prev, out_ch = out_ch, rev[i]
            self.up_blocks.append(UpBlockTemporalDecoder(num_layers=layers_per_block + 1, in_channels=prev, out_channels=out_ch,
                                                         add_upsample=i != len(block_out_channels) - 1))
        self.conv_norm_out = nn.GroupNorm(32, block_out_channels[0], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, 3, padding=1)
        self.time_conv_out = nn.Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))

    def pack(self, reg, dtype):
        w = self.conv_in.weight.detach()
        wp = torch.zeros((w.shape[0], self.CIN_PAD, 3, 3), dtype=w.dtype, device=w.device)
        wp[:, :self.in_channels] = w
        self.w_in, self.b_in = pack_conv3x3(wp.to(dtype)), _f32(self.conv_in.bias)
        self.mid_block.pack(reg, dtype)
        for b in self.up_blocks:
            b.pack(reg, dtype)
        self.g_out, self.be_out = _f32(self.conv_norm_out.weight), _f32(self.conv_norm_out.bias)
        # conv_out: 3 output channels padded to 8 (zero rows), so its output is the 8-channel input of the frame conv
        w = self.conv_out.weight.detach()
        wo = torch.zeros((self.CIN_PAD,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
        wo[:self.out_channels] = w
        bo = torch.zeros(self.CIN_PAD, dtype=torch.float32, device=w.device)
        bo[:self.out_channels] = self.conv_out.bias.detach().float()
        self.w_out, self.b_out = pack_conv3x3(wo.to(dtype)), bo
        w = self.time_conv_out.weight.detach()                             # [3, 3, 3, 1, 1] -> [4, 8, 3, 1, 1]
        wt = torch.zeros((4, self.CIN_PAD, 3, 1, 1), dtype=w.dtype, device=w.device)
        wt[:self.out_channels, :self.out_channels] = w
        bt = torch.zeros(4, dtype=torch.float32, device=w.device)
        bt[:self.out_channels] = self.time_conv_out.bias.detach().float()
        self.w_t, self.b_t = pack_tconv3(wt.to(dtype)), bt

    def forward(self, z: torch.Tensor, num_frames: int, dtype: torch.dtype) -> torch.Tensor:
        """z [N, 4, h, w] (N = batch * num_frames) -> fp32 [N, 3, 8h, 8w]."""
        n, _, h, w = z.shape
        if n % num_frames:
            raise ValueError(f"{n} latent frames are not a multiple of num_frames = {num_frames}")
        g = Geom(n // num_frames, num_frames, h, w)
        x = ops.nchw_to_tokens(z, dtype, ld=self.CIN_PAD)
        x = ops.gemm(x, self.w_in, mode=1, conv=(n, h, w, h, w, 1, 0), bias=self.b_in)
        x = self.mid_block(x, g)
        for blk in self.up_blocks:
            x, g = blk(x, g)
        x = ops.groupnorm(x, None, g.n, g.hw, 1, self.g_out, self.be_out, 1e-6, True)
        x = ops.gemm(x, self.w_out, mode=1, conv=(n, g.h, g.w, g.h, g.w, 1, 0), bias=self.b_out)            # [M, 8]
        x = ops.gemm(x, self.w_t, mode=2, tconv=(g.frames, g.hw), bias=self.b_t)                             # [M, 4]
        return ops.tokens_to_nchw(x, n, self.out_channels, g.h, g.w, torch.float32)


class AutoencoderKLTemporalDecoder(ModelMixin, ConfigMixin):
    """diffusers' AutoencoderKLTemporalDecoder with the reference call surface: ``decode(z, num_frames)`` returns an object with
    ``.sample`` and runs on the libttvdm kernels; ``config.scaling_factor`` / ``force_upcast``; ``decoder.*`` parameter names.
    ``encode(x)`` is delegated to the caller's stock module given as ``encoder=`` (any object whose ``encode(x)`` returns
    ``.latent_dist`` -- diffusers' own AutoencoderKLTemporalDecoder); without one it raises.  The stock module is NOT a
    sub-module (no parameters of it appear in ``state_dict()`` / ``parameters()``), but ``.to()`` / ``.half()`` / ``.float()``
    reach it too, so the pipeline's force_upcast round trip (reference :556-571: vae.to(fp32) -> encode -> vae.to(fp16)) encodes
    in fp32 exactly as with the stock class."""

    _config_exclude = ("encoder",)          # a module, not a config entry (save_pretrained writes config.json from self.config)

    @register_to_config
    def __init__(self, in_channels: int = 3, out_channels: int = 3, block_out_channels: Tuple[int, ...] = (128, 256, 512, 512),
                 layers_per_block: int = 2, latent_channels: int = 4, sample_size: int = 768, scaling_factor: float = 0.18215,
                 force_upcast: bool = True, encoder=None):
        super().__init__()
        self.decoder = TemporalDecoder(latent_channels, out_channels, tuple(block_out_channels), layers_per_block)
        self.compute_dtype: Optional[torch.dtype] = None      # None: the parameter dtype if 16-bit, else bf16; float32 = TT_F32 mode
        self._packed_key = None
        self.__dict__["_stock_encoder"] = None
        self.with_encoder(encoder)

    def with_encoder(self, encoder):
        """``encoder``: the caller's stock VAE (or any object with ``encode(x) -> .latent_dist``); None removes it."""
        if encoder is not None and not callable(getattr(encoder, "encode", None)):
            raise TypeError("encoder must provide encode(x) returning an object with .latent_dist (diffusers' AutoencoderKLTemporalDecoder)")
        self.__dict__["_stock_encoder"] = encoder              # outside nn.Module's registry on purpose (see the class docstring)
        return self

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *args, encoder=None, **kwargs):
        """diffusers-format folder (vae/ of an SVD checkpoint: encoder.* / quant_conv.* entries are ignored) + the stock encoder."""
        model = super().from_pretrained(pretrained_model_name_or_path, *args, **kwargs)
        return model.with_encoder(encoder)

    def _run_dtype(self) -> torch.dtype:
        if self.compute_dtype is not None:
            return self.compute_dtype
        dt = next(self.parameters()).dtype
        return dt if dt in (torch.float16, torch.bfloat16) else torch.bfloat16

    def prepare(self, force: bool = False):
        p0 = next(self.parameters())
        key = (p0.device, self._run_dtype(), sum(p._version for p in self.parameters()), p0.data_ptr())
        if force or self._packed_key != key:
            if p0.device.type != "cuda":
                raise RuntimeError("AutoencoderKLTemporalDecoder: the decoder runs on the MI355X only (no CPU fallback)")
            self.decoder.pack(PackRegistry(), key[1])
            self._packed_key = key
        return self

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Accepts a full diffusers AutoencoderKLTemporalDecoder state dict: encoder.* / quant_conv.* entries are ignored."""
        sd = {k: v for k, v in state_dict.items() if k.startswith("decoder.")}
        self._packed_key = None
        return super().load_state_dict(sd, strict=strict, **kw)

    def _apply(self, fn, *a, **k):
        self._packed_key = None
        stock = self.__dict__.get("_stock_encoder")
        if isinstance(stock, nn.Module):
            stock._apply(fn, *a, **k)                          # .to() / .half() / .float() / .cuda() reach the stock encoder too
        return super()._apply(fn, *a, **k)

    def encode(self, x, *a, **k):
        """Delegated to the stock module given as ``encoder=`` (one image + the gesture frames per request, off the hot path)."""
        stock = self.__dict__.get("_stock_encoder")
        if stock is None:
            raise NotImplementedError("AutoencoderKLTemporalDecoder.encode: no stock encoder attached -- build the model with "
                                      "encoder=<the diffusers VAE> (or call .with_encoder(vae)); the native kernels cover the decoder side "
                                      "(decode_latents, reference :257-283), the encoder runs once per request in the caller's module")
        return stock.encode(x, *a, **k)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, num_frames: int = 1, return_dict: bool = True):
        self.prepare()
        out = self.decoder(z.to(next(self.parameters()).device), num_frames, self._run_dtype())
        return SimpleNamespace(sample=out) if return_dict else (out,)

    def forward(self, sample, num_frames: int = 1):
        # (signature carries num_frames: decode_latents, reference :264, inspects vae.forward for it)
        return self.decode(sample, num_frames=num_frames)
