"""Leaf modules of the SVD denoise path, MI355X-native.

Each class mirrors a diffusers==0.25.1 leaf the reference imports (names, constructor arguments and
state-dict keys identical -- SURVEY.md Appendix A/B; import sites
svd/diffusion_arch/unet_3d_blocks.py:20-31, transformer_temporal.py:19-24) but its compute is a
sequence of libttvdm launches on token-major activations.  The nn.Linear / nn.Conv / nn.*Norm children
are parameter containers only (so reference checkpoints load by name); they are never called.

Activations: 2-D token tensors ``[N*h*w, C]`` (C contiguous) in fp16/bf16 plus a ``Geom``.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..packing import fold_layernorm, pack_conv1x1, pack_conv3x3, pack_geglu, pack_tconv3, permute_q_rows, zero_sum_round


ZERO_CTX_TEMPORAL = os.environ.get("TT_ZERO_CTX_T", "1") != "0"      # A/B switch of the temporal zero-context shortcut
ATTN_V_ROWS = os.environ.get("TT_ATTN_V_ROWS", "1") != "0"            # spatial self-attention from ONE Q | K | V launch, V not transposed (tt_attention v_rows)
MERGE_FRAMES = os.environ.get("TT_XATTN_MERGE_FRAMES", "1") != "0"    # spatial cross-attention: the frames of a batch element as one sequence (A/B)


@dataclass
class Geom:
    batch: int       # B (CFG-expanded) covered by this launch
    frames: int      # F
    h: int
    w: int
    batch0: int = 0          # index of its first batch element inside the request ...
    batch_total: int = 0     # ... and the request's full batch (0 = same as `batch`): the fused loop runs CFG halves apart

    @property
    def n(self):
        return self.batch * self.frames

    @property
    def hw(self):
        return self.h * self.w

    @property
    def m(self):
        return self.n * self.hw

    @property
    def ctx_batches(self):
        return self.batch_total or self.batch

    def film(self, film: torch.Tensor, off: int, c: int) -> torch.Tensor:
        """FiLM rows of this launch's batch elements, columns [off, off+c)."""
        return film[self.batch0:self.batch0 + self.batch, off:off + c]


class StepContext:
    """Per-forward state shared by all blocks: FiLM rows of every ResBlock (one batched GEMV) and the
    cross-attention K / V^T of every attention layer (two GEMMs per request, step-invariant)."""

    def __init__(self, film: torch.Tensor, k_all: torch.Tensor, vt_all: torch.Tensor, s_ctx: int, s_pad: int,
                 attn_fp8: bool = False, zero_mask: int = 0):
        self.film, self.k_all, self.vt_all, self.s_ctx, self.s_pad = film, k_all, vt_all, s_ctx, s_pad
        self.attn_fp8 = attn_fp8            # spatial self-attention on e4m3 operands (BASELINE config 5)
        self.zero_mask = zero_mask          # bit b: the context of batch element b is all zeros (denoiser_base.project_context)

    def live_batches(self, g: "Geom"):
        """Which batch elements of launch `g` need real cross-attention.  None: all of them (general path).  Otherwise
        (first, count): elements [first, first+count) of the launch have a non-zero context and every other one an all-zero
        context, whose cross-attention output is exactly 0 (count may be 0)."""
        flags = [(self.zero_mask >> (g.batch0 + i)) & 1 for i in range(g.batch)]
        if not any(flags):
            return None
        live = [i for i, z in enumerate(flags) if not z]
        if not live:
            return (0, 0)
        if live[-1] - live[0] + 1 != len(live):
            return None                     # live elements not contiguous in the row order: keep the general path
        return (live[0], len(live))

    def live_classes(self, g: "Geom"):
        """Temporal cross-attention (reference quirk Q3, transformer_temporal.py:316-319): query pixel p of batch element b sees
        context (b*hw + p) % CB.  When hw is a multiple of CB that is p % CB = (token row) % CB for every row of the launch: the
        rows of residue class c all use context c.  Returns None when every context is live (or the classes do not align with
        the rows): general path.  Otherwise the list of classes whose context is NOT all-zero; the rows of the other classes
        get exactly 0 from the attention (K = V = 0) and therefore only to_out's bias."""
        cb = g.ctx_batches
        if not self.zero_mask or cb < 2 or g.hw % cb or not ZERO_CTX_TEMPORAL:
            return None
        live = [c for c in range(cb) if not (self.zero_mask >> c) & 1]
        return None if len(live) == cb else live

    _VT_CACHE: Dict[tuple, torch.Tensor] = {}

    def vt_buffer(self, c: int, cols: int, like: torch.Tensor, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """Scratch for the transposed V projection of self-attention.  Persistent per shape: padding columns
        (sequence length not a multiple of 8) are zeroed once and never written, and launches on one stream
        are ordered, so layers can share it."""
        # one scratch per (stream, shape): the fused loop runs GestureNet and UNet encoders on two streams
        dtype = like.dtype if dtype is None else dtype
        key = (like.device, dtype, c, cols, torch.cuda.current_stream().cuda_stream)
        buf = StepContext._VT_CACHE.get(key)
        if buf is None:
            buf = StepContext._VT_CACHE[key] = torch.zeros((c, cols), dtype=torch.uint8 if dtype == ops.FP8 else dtype,
                                                           device=like.device).view(dtype)
        return buf


class _Side:
    """Run launches that are independent of the caller's next launches on a second stream of the same branch (fork / join with
    events: parallel branches of the captured hipGraph).  Used where two under-filled GEMMs of a transformer block have no
    dependency on each other: the swapped V^T projection next to the Q|K projection (both read the block input), and the
    output projection of the zero-context residue class next to the live class's query projection -> attention -> output
    projection chain.  Buffers the side launch writes are allocated by the CALLER on its own stream before the fork (the caching
    allocator's per-stream pools never see the side stream), and the caller joins before it reads them or drops them.
    OFF by default: measured in the step (one gpurun call, interleaved) 30.99 / 30.94 ms without against 31.10 / 31.08 ms with
    -- the forked launches compete with the neighbours they were meant to fill in; TT_SIDE_STREAM=1 enables it (A/B)."""
    ENABLED = os.environ.get("TT_SIDE_STREAM", "0") == "1"
    _streams: Dict[int, "torch.cuda.Stream"] = {}
    _spare: List["torch.cuda.Stream"] = []
    origin: Optional[int] = None        # raw handle of the stream a capture started on (set by DenoiseLoop._launch_step)
    # events stay alive until long after the capture that recorded them has ended (destroying an event while the stream
    # capture that recorded it is still open crashed hipStreamEndCapture); a bounded ring, refilled by later launches
    _events: List["torch.cuda.Event"] = []

    @staticmethod
    def _event():
        ev = torch.cuda.Event()
        _Side._events.append(ev)
        # trim the ring only OUTSIDE a capture: the oldest entries may have been recorded by the capture that is still open
        if len(_Side._events) > 8192 and not torch.cuda.is_current_stream_capturing():
            del _Side._events[:4096]
        return ev

    def __init__(self):
        self.main = torch.cuda.current_stream()
        self.side = None
        capturing = torch.cuda.is_current_stream_capturing()
        # Under stream capture only the ORIGIN stream of the capture may fork: a fork from a stream that is itself a forked branch
        # (GestureNet's encoder in DenoiseLoop) makes hipStreamEndCapture crash on ROCm 7.2 (tools/capture_fork_probe.py, pattern c)
        if _Side.ENABLED and (not capturing or self.main.cuda_stream == _Side.origin):
            key = self.main.cuda_stream
            self.side = _Side._streams.get(key)
            if self.side is None:
                # no stream is created while a capture is open: a spare one made outside of it (the warm-up pass of
                # DenoiseLoop._capture runs the same code eagerly) is handed to the capturing stream
                if capturing:
                    if not _Side._spare:
                        return
                    self.side = _Side._spare.pop()
                else:
                    self.side = torch.cuda.Stream()
                    while len(_Side._spare) < 4:
                        _Side._spare.append(torch.cuda.Stream())
                _Side._streams[key] = self.side
            ev = _Side._event()
            ev.record(self.main)
            self.side.wait_event(ev)

    def __enter__(self):
        if self.side is not None:
            self._guard = torch.cuda.stream(self.side)
            self._guard.__enter__()
        return self

    def __exit__(self, *exc):
        if self.side is not None:
            self._guard.__exit__(*exc)
        return False

    def join(self):
        if self.side is not None:
            ev = _Side._event()
            ev.record(self.side)
            self.main.wait_event(ev)


class PackRegistry:
    """Collects, at pack time, the weights that are batched across layers (FiLM projections, context K/V)."""

    def __init__(self):
        self.film_w: List[torch.Tensor] = []
        self.film_b: List[torch.Tensor] = []
        self.film_cols = 0
        self.k_w: List[torch.Tensor] = []
        self.v_w: List[torch.Tensor] = []
        self.kv_cols = 0

    def add_film(self, lin: nn.Linear) -> Tuple[int, int]:
        off = self.film_cols
        self.film_w.append(lin.weight.detach())
        self.film_b.append(lin.bias.detach())
        self.film_cols += lin.out_features
        return off, lin.out_features

    def add_kv(self, to_k: nn.Linear, to_v: nn.Linear) -> Tuple[int, int]:
        off = self.kv_cols
        self.k_w.append(to_k.weight.detach())
        self.v_w.append(to_v.weight.detach())
        self.kv_cols += to_k.out_features
        return off, to_k.out_features


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().float().contiguous()


class _Packable(nn.Module):
    """Mixin: ``pack(reg, dtype)`` converts parameters into kernel-ready buffers (plain attributes, not
    registered, so state_dict() is unchanged)."""

    def pack(self, reg: PackRegistry, dtype: torch.dtype):
        raise NotImplementedError


# --------------------------------------------------------------------------- embeddings (parameter containers)
class Timesteps(nn.Module):
    def __init__(self, num_channels: int, flip_sin_to_cos: bool, downscale_freq_shift: float):
        super().__init__()
        if not flip_sin_to_cos or downscale_freq_shift != 0:
            raise NotImplementedError("the SVD path uses Timesteps(dim, True, 0) only")
        self.num_channels = num_channels

    def forward(self, timesteps: torch.Tensor) -> torch.Tensor:
        return ops.timestep_embedding(timesteps.float().contiguous(), self.num_channels)


class TimestepEmbedding(_Packable):
    def __init__(self, in_channels: int, time_embed_dim: int, act_fn: str = "silu", out_dim: Optional[int] = None):
        super().__init__()
        if act_fn != "silu":
            raise NotImplementedError(act_fn)
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim)

    def pack(self, reg, dtype):
        self.w1, self.b1 = self.linear_1.weight.detach().to(dtype).contiguous(), _f32(self.linear_1.bias)
        self.w2, self.b2 = self.linear_2.weight.detach().to(dtype).contiguous(), _f32(self.linear_2.bias)

    def forward(self, sample: torch.Tensor) -> torch.Tensor:          # fp32 [rows<=32, in] -> fp32 [rows, out]
        hid = ops.small_linear(sample, self.w1, self.b1, act_out=True)
        return ops.small_linear(hid, self.w2, self.b2)


# --------------------------------------------------------------------------- resnets
def _gn(x0, x1, g: Geom, frames_per_group, gamma, beta, eps, silu):
    return ops.groupnorm(x0, x1, g.n, g.hw, frames_per_group, gamma, beta, eps, silu)


class ResnetBlock2D(_Packable):
    """GN+SiLU -> conv3x3 (+FiLM) -> GN+SiLU -> conv3x3 (+shortcut) ; diffusers ResnetBlock2D (Appendix A.3)."""

    def __init__(self, *, in_channels: int, out_channels: Optional[int] = None, temb_channels: int = 512,
                 groups: int = 32, eps: float = 1e-6):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        if groups != 32:
            raise NotImplementedError("GroupNorm kernels are built for 32 groups")
        self.in_channels, self.out_channels, self.eps = in_channels, out_channels, eps
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        # temb_channels=None: no FiLM term (the temporal VAE decoder's blocks, diffusers autoencoder_kl_temporal_decoder.py)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def pack(self, reg, dtype):
        self.g1, self.be1 = _f32(self.norm1.weight), _f32(self.norm1.bias)
        self.g2, self.be2 = _f32(self.norm2.weight), _f32(self.norm2.bias)
        self.w1, self.b1 = pack_conv3x3(self.conv1.weight.detach().to(dtype)), _f32(self.conv1.bias)
        self.w2, self.b2 = pack_conv3x3(self.conv2.weight.detach().to(dtype)), _f32(self.conv2.bias)
        if self.conv_shortcut is not None:
            self.ws, self.bs = pack_conv1x1(self.conv_shortcut.weight.detach().to(dtype)), _f32(self.conv_shortcut.bias)
        self.film = reg.add_film(self.time_emb_proj) if self.time_emb_proj is not None else None

    def forward(self, x0, x1, g: Geom, ctx: StepContext, next_gn=None):
        """next_gn = (gamma, beta, eps, silu) of the cross-frame GroupNorm that reads this block's output (TemporalResnetBlock.norm1)"""
        film = None
        if self.film is not None:
            off, c = self.film
            film = g.film(ctx.film, off, c)
        c0, c1 = x0.shape[1], (x1.shape[1] if x1 is not None else 0)
        # tt_conv3x3 (GroupNorm + SiLU applied while the input patch is staged in LDS, no normalised copy) is opt-in
        # (TT_CONV3X3=1): every column tile re-does the SiLU of its patch, and at 5..10 column tiles per conv that costs more
        # than the one tt_groupnorm_apply pass it removes (465-635 vs 770-950 TFLOP/s) -- the default is
        # groupnorm_apply + the implicit-GEMM conv
        fused = ops.CONV3X3_FUSED and ops.conv3x3_supported(g.h, g.w, c0, c1, self.out_channels, x0.dtype) and \
            ops.conv3x3_supported(g.h, g.w, self.out_channels, 0, self.out_channels, x0.dtype)
        conv = (g.n, g.h, g.w, g.h, g.w, 1, 0)
        if fused:
            st1 = ops.groupnorm_stats(x0, x1, g.n, g.hw, 1, self.g1, self.be1, self.eps)
            hmid = ops.conv3x3(x0, x1, self.w1, g.n, g.h, g.w, gn=st1, silu=True, bias=self.b1, rowvec=film,
                               rowvec_rows=g.frames * g.hw if film is not None else 0)
        else:
            a = ops.groupnorm(x0, x1, g.n, g.hw, 1, self.g1, self.be1, self.eps, True)
            hmid = ops.gemm(a, self.w1, mode=1, conv=conv, bias=self.b1, rowvec=film, rowvec_rows=g.frames * g.hw if film is not None else 0,
                            stats=g.hw, gn=(self.g2, self.be2, self.eps, True))      # norm2 (per image) reads it: per-tile column sums from the
                                                                                         # epilogue, or normalised by the split-K reduction pass
        if self.conv_shortcut is not None:
            xs = ops.gemm(x0, self.ws, a1=x1, bias=self.bs)
        else:
            if x1 is not None:
                raise RuntimeError("identity shortcut with a concatenated input")
            xs = x0
        if fused:
            st2 = ops.groupnorm_stats(hmid, None, g.n, g.hw, 1, self.g2, self.be2, self.eps)
            return ops.conv3x3(hmid, None, self.w2, g.n, g.h, g.w, gn=st2, silu=True, bias=self.b2, residual=xs)
        a = ops.groupnorm(hmid, None, g.n, g.hw, 1, self.g2, self.be2, self.eps, True)
        # (read by the temporal block's norm1: statistics over the frames x h x w rows of a video)
        return ops.gemm(a, self.w2, mode=1, conv=conv, bias=self.b2, residual=xs, stats=g.frames * g.hw, gn=next_gn)


class TemporalResnetBlock(_Packable):
    """GroupNorm over (F,h,w) + 3-tap frame conv, twice (Appendix A.4).  The AlphaBlender that follows it
    inside SpatioTemporalResBlock is folded into conv2's epilogue."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, temb_channels: int = 512, eps: float = 1e-6):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        if in_channels != out_channels:
            raise NotImplementedError("SVD only uses channel-preserving temporal blocks")
        self.eps = eps
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps)
        self.conv1 = nn.Conv3d(in_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps)
        self.conv2 = nn.Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))

    def pack(self, reg, dtype):
        self.g1, self.be1 = _f32(self.norm1.weight), _f32(self.norm1.bias)
        self.g2, self.be2 = _f32(self.norm2.weight), _f32(self.norm2.bias)
        self.w1, self.b1 = pack_tconv3(self.conv1.weight.detach().to(dtype)), _f32(self.conv1.bias)
        self.w2, self.b2 = pack_tconv3(self.conv2.weight.detach().to(dtype)), _f32(self.conv2.bias)
        self.film = reg.add_film(self.time_emb_proj) if self.time_emb_proj is not None else None

    def forward(self, s, g: Geom, ctx: StepContext, alpha: float):
        film = g.film(ctx.film, *self.film) if self.film is not None else None
        a = _gn(s, None, g, g.frames, self.g1, self.be1, self.eps, True)
        t = ops.gemm(a, self.w1, mode=2, tconv=(g.frames, g.hw), bias=self.b1, rowvec=film,
                     rowvec_rows=g.frames * g.hw if film is not None else 0, stats=g.frames * g.hw, gn=(self.g2, self.be2, self.eps, True))
        a = _gn(t, None, g, g.frames, self.g2, self.be2, self.eps, True)
        # x_temporal = s + conv2(...);  out = alpha*s + (1-alpha)*x_temporal
        # (the block's output: the transformer's GroupNorm or the next ResBlock's norm1 reads it)
        return ops.gemm(a, self.w2, mode=2, tconv=(g.frames, g.hw), bias=self.b2, residual=s, blend=s, alpha=alpha, stats=g.hw)


class AlphaBlender(nn.Module):
    """learned_with_images blender; on this path image_only_indicator is all zeros
    (unet_spatio_temporal_condition.py:457) so alpha = sigmoid(mix_factor) (Appendix A.6)."""

    def __init__(self, alpha: float, merge_strategy: str = "learned_with_images", switch_spatial_to_temporal_mix: bool = False):
        super().__init__()
        if merge_strategy not in ("learned_with_images", "learned"):
            raise NotImplementedError(merge_strategy)
        # "learned" (the temporal VAE decoder) and "learned_with_images" with an all-zero indicator both give sigmoid(mix_factor);
        # switch_spatial_to_temporal_mix (decoder) blends with 1 - alpha
        self.switch = bool(switch_spatial_to_temporal_mix)
        self.mix_factor = nn.Parameter(torch.tensor([float(alpha)]))

    def alpha_value(self) -> float:
        a = float(torch.sigmoid(self.mix_factor.detach().float()).item())
        return 1.0 - a if self.switch else a


class SpatioTemporalResBlock(_Packable):
    def __init__(self, in_channels: int, out_channels: Optional[int] = None, temb_channels: int = 512, eps: float = 1e-6,
                 temporal_eps: Optional[float] = None, merge_factor: float = 0.5, merge_strategy="learned_with_images",
                 switch_spatial_to_temporal_mix: bool = False):
        super().__init__()
        oc = out_channels if out_channels is not None else in_channels
        self.spatial_res_block = ResnetBlock2D(in_channels=in_channels, out_channels=oc, temb_channels=temb_channels, eps=eps)
        self.temporal_res_block = TemporalResnetBlock(oc, oc, temb_channels=temb_channels,
                                                      eps=temporal_eps if temporal_eps is not None else eps)
        self.time_mixer = AlphaBlender(merge_factor, merge_strategy, switch_spatial_to_temporal_mix)

    def pack(self, reg, dtype):
        self.spatial_res_block.pack(reg, dtype)
        self.temporal_res_block.pack(reg, dtype)
        self.alpha = self.time_mixer.alpha_value()

    def forward(self, x0, x1, g: Geom, ctx: StepContext):
        t = self.temporal_res_block
        s = self.spatial_res_block(x0, x1, g, ctx, next_gn=(t.g1, t.be1, t.eps, True))
        return t(s, g, ctx, self.alpha)


class Downsample2D(_Packable):
    def __init__(self, channels: int, use_conv: bool = False, out_channels: Optional[int] = None, padding: int = 1, name: str = "conv"):
        super().__init__()
        if not use_conv or padding != 1:
            raise NotImplementedError("SVD uses Downsample2D(use_conv=True, padding=1)")
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=1)

    def pack(self, reg, dtype):
        self.w, self.b = pack_conv3x3(self.conv.weight.detach().to(dtype)), _f32(self.conv.bias)

    def forward(self, x, g: Geom):
        ho, wo = (g.h + 2 - 3) // 2 + 1, (g.w + 2 - 3) // 2 + 1
        out = ops.gemm(x, self.w, mode=1, conv=(g.n, g.h, g.w, ho, wo, 2, 0), bias=self.b, stats=ho * wo)
        return out, Geom(g.batch, g.frames, ho, wo, g.batch0, g.batch_total)


class Upsample2D(_Packable):
    def __init__(self, channels: int, use_conv: bool = False, use_conv_transpose: bool = False,
                 out_channels: Optional[int] = None, name: str = "conv"):
        super().__init__()
        if not use_conv or use_conv_transpose:
            raise NotImplementedError("SVD uses Upsample2D(use_conv=True)")
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)

    def pack(self, reg, dtype):
        self.w, self.b = pack_conv3x3(self.conv.weight.detach().to(dtype)), _f32(self.conv.bias)

    def forward(self, x, g: Geom):
        # nearest x2 is an index map inside the conv gather: never materialised
        out = ops.gemm(x, self.w, mode=1, conv=(g.n, g.h, g.w, 2 * g.h, 2 * g.w, 1, 1), bias=self.b, stats=4 * g.hw)
        return out, Geom(g.batch, g.frames, 2 * g.h, 2 * g.w, g.batch0, g.batch_total)


# --------------------------------------------------------------------------- attention / feed-forward
class Attention(nn.Module):
    """Parameter container with the diffusers names (to_q/to_k/to_v no bias, to_out.0 with bias)."""

    def __init__(self, query_dim: int, cross_attention_dim: Optional[int] = None, heads: int = 8, dim_head: int = 64,
                 dropout: float = 0.0, bias: bool = False, out_bias: bool = True):
        super().__init__()
        self.inner_dim, self.heads, self.dim_head = heads * dim_head, heads, dim_head
        self.is_cross = cross_attention_dim is not None
        cross = cross_attention_dim if self.is_cross else query_dim
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(cross, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(cross, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        if dim_head not in (64, 128):
            raise NotImplementedError(f"attention head_dim {dim_head}: kernels are built for 64 and 128")


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(_Packable):
    def __init__(self, dim: int, dim_out: Optional[int] = None, mult: int = 4, dropout: float = 0.0, activation_fn: str = "geglu"):
        super().__init__()
        if activation_fn != "geglu":
            raise NotImplementedError(activation_fn)
        inner = int(dim * mult)
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out if dim_out is not None else dim)])

    def pack(self, reg, dtype, norm: Optional[nn.LayerNorm] = None):
        """``norm``: the LayerNorm in front of this feed-forward; it is folded into the GEGLU projection (tt_gemm ln_fold)."""
        w, b = self.net[0].proj.weight.detach(), self.net[0].proj.bias
        self.ln_fold, self.ln_eps = 0, 1e-5
        if norm is not None:
            w, b = fold_layernorm(w, b, norm.weight, norm.bias)
            w = zero_sum_round(w, dtype)
            self.ln_fold, self.ln_eps = 1, norm.eps
        self.wg, self.bg = pack_geglu(w.to(dtype), _f32(b))
        self.w2, self.b2 = self.net[2].weight.detach().to(dtype).contiguous(), _f32(self.net[2].bias)

    def forward(self, x, residual, blend=None, alpha=0.0, rowvec=None, rowvec_rows: int = 0):
        """x: the UN-normalised hidden states when a LayerNorm was folded in at pack().
        rowvec fp32 [groups, C]: row vector added to rows [i * rowvec_rows, (i+1) * rowvec_rows) before residual and blend."""
        hid = ops.gemm(x, self.wg, bias=self.bg, geglu=True, ln_fold=self.ln_fold, ln_eps=self.ln_eps)   # the 8C tensor never exists
        return ops.gemm(hid, self.w2, bias=self.b2, residual=residual, blend=blend, alpha=alpha, rowvec=rowvec, rowvec_rows=rowvec_rows)


def _self_attention(x, attn: Attention, wqk, bqk, wv, eps, g: Geom, ctx: StepContext, wqkv=None, bqkv=None):
    """spatial self-attention over hw tokens per frame on the UN-normalised hidden states: norm1 is folded into the QK
    projection (rows) and into the swapped V^T projection (columns); flash kernel.
    wqkv / bqkv (16-bit storage, head dimension 64, not the fp8 path): ONE Q | K | V projection launch and the attention kernel reads V
    as it is (tt_attention v_rows: transposed on the way out of LDS) -- no V^T projection launch."""
    c = attn.inner_dim
    fp8 = ctx.attn_fp8 and x.dtype != torch.float32
    if wqkv is not None and not fp8 and ATTN_V_ROWS and attn.dim_head == 64 and x.dtype in (torch.float16, torch.bfloat16):
        qkv = ops.gemm(x, wqkv, bias=bqkv, ln_fold=1, ln_eps=eps)         # [M, 3C]
        out = torch.empty((g.m, c), dtype=x.dtype, device=x.device)
        return ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], out, nseq=g.n, lq=g.hw, heads=attn.heads, head_dim=attn.dim_head,
                             mask=0, lk=g.hw, k_seq_stride=g.hw, v_seq_stride=g.hw, v_rows=True)
    pad = 16 if fp8 else 8                                                # V^T sequences start on 16-byte chunks
    hwp = (g.hw + pad - 1) // pad * pad
    vt = ctx.vt_buffer(c, g.n * hwp, x, ops.FP8 if fp8 else None)
    side = _Side()                                                        # V^T projection next to the Q | K projection
    with side:
        ops.gemm(wv, x, out=vt, out_col_pad=(g.hw, hwp) if hwp != g.hw else None, ln_fold=2, ln_eps=eps, out_fp8=fp8)
    qk = ops.gemm(x, wqk, bias=bqk, ln_fold=1, ln_eps=eps, out_fp8=fp8)   # [M, 2C]  (e4m3 bytes on the fp8 path)
    side.join()
    x_norm = x
    out = torch.empty((g.m, c), dtype=x_norm.dtype, device=x_norm.device)
    return ops.attention(qk[:, :c], qk[:, c:], vt, out, nseq=g.n, lq=g.hw, heads=attn.heads, head_dim=attn.dim_head,
                         mask=0, lk=g.hw, k_seq_stride=g.hw, v_seq_stride=hwp)


class _QProj:
    """The LayerNorm-folded query projection of a cross-attention layer, kept ONCE: row-permuted for the attention kernel's own
    projection (packing.permute_q_rows) when that path can serve the layer (d = 64, 16-bit storage, TT_ATTN_QPROJ != 0), in the
    plain order otherwise.  The other order is derived on demand (the permutation is an involution) -- e.g. for a row view whose
    stride the fused path does not take -- so the default path does not hold a second copy of every to_q (~60 MB over both models)."""

    def __init__(self, w: torch.Tensor, b: torch.Tensor, dim_head: int):
        self.is_fused = ops.ATTN_QPROJ and dim_head == 64 and w.dtype in (torch.float16, torch.bfloat16)
        self.w, self.b = (permute_q_rows(w), permute_q_rows(b.contiguous())) if self.is_fused else (w, b.contiguous())
        self._other = None

    def fused(self):
        return (self.w, self.b) if self.is_fused else None

    def plain(self):
        if not self.is_fused:
            return self.w, self.b
        if self._other is None:
            self._other = (permute_q_rows(self.w), permute_q_rows(self.b))
        return self._other


def _cross_attention(x, attn: Attention, qp: _QProj, eps, kv, g: Geom, ctx: StepContext, temporal: bool):
    """cross-attention on the UN-normalised hidden states (norm2 folded into the query projection ``qp``).  With the row-permuted
    weights (qp.fused()) the attention kernel projects the queries itself: no Q tensor, no projection launch."""
    off, c = kv
    mask = 2 if temporal else 1
    out = torch.empty((g.m, c), dtype=x.dtype, device=x.device)
    kw = dict(nseq=g.n, lq=g.hw, heads=attn.heads, head_dim=attn.dim_head, mask=mask, lk=ctx.s_ctx, k_seq_stride=ctx.s_pad,
              v_seq_stride=ctx.s_pad, frames=g.frames, ctx_batches=g.ctx_batches, batch0=g.batch0)
    if not temporal and MERGE_FRAMES and g.n % g.frames == 0:
        # the frames of a batch element share its context and are contiguous rows: ONE sequence of frames * hw queries per batch element.
        # Same result row by row (a query's work does not depend on its block); the 128-query blocks are then full -- at the coarsest
        # level an image has 28 tokens, i.e. 78 % of every block's query projection and attention was padding (560 -> 80 blocks)
        kw.update(nseq=g.n // g.frames, lq=g.frames * g.hw, frames=1)
    fused = qp.fused()
    if fused is not None and ops.attention_qproj_supported(x, attn.dim_head, mask):
        return ops.attention(None, ctx.k_all[:, off:off + c], ctx.vt_all[off:off + c], out, qx=x, wq=fused[0], bq=fused[1], ln_eps=eps, **kw)
    wq, bq = qp.plain()
    q = ops.gemm(x, wq, bias=bq, ln_fold=1, ln_eps=eps)
    return ops.attention(q, ctx.k_all[:, off:off + c], ctx.vt_all[off:off + c], out, **kw)


class BasicTransformerBlock(_Packable):
    """LN -> self-attn -> LN -> cross-attn -> LN -> GEGLU FF, residual after each (Appendix A.8)."""

    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int, cross_attention_dim: Optional[int] = None):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads=num_attention_heads, dim_head=attention_head_dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, cross_attention_dim=cross_attention_dim, heads=num_attention_heads, dim_head=attention_head_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def pack(self, reg, dtype):
        """The three LayerNorms are folded into the GEMMs that consume them (packing.fold_layernorm, tt_gemm ln_fold): no
        LayerNorm kernel and no normalised copy of the hidden states exist.  norm1's beta reaches V as the constant vector
        Wv beta added to every key's value; softmax rows sum to 1, so it passes through attention unchanged and is added to
        to_out's bias (Wo (Wv beta))."""
        cv = lambda t: t.detach().to(dtype).contiguous()
        zr = lambda t: zero_sum_round(t, dtype)              # folded weights: rounded with exact zero row sums
        n1, n2 = self.norm1, self.norm2
        wq, bq = fold_layernorm(self.attn1.to_q.weight, None, n1.weight, n1.bias)
        wk, bk = fold_layernorm(self.attn1.to_k.weight, None, n1.weight, n1.bias)
        wv, bv = fold_layernorm(self.attn1.to_v.weight, None, n1.weight, n1.bias)
        # one [3C, C] matrix: Q | K rows first (the two-launch route reads them as a view), V rows last; V has no bias of its own (below)
        self.wqkv = zr(torch.cat([wq, wk, wv], 0))
        self.bqkv = torch.cat([bq, bk, torch.zeros_like(bq)], 0).contiguous()
        c2 = 2 * wq.shape[0]
        self.wqk, self.bqk, self.wv = self.wqkv[:c2], self.bqkv[:c2], self.wqkv[c2:]
        self.wo1 = cv(self.attn1.to_out[0].weight)
        self.bo1 = (_f32(self.attn1.to_out[0].bias) + self.wo1.float() @ bv).contiguous()
        wq2, bq2 = fold_layernorm(self.attn2.to_q.weight, None, n2.weight, n2.bias)
        self.q2 = _QProj(zr(wq2), bq2, self.attn2.dim_head)
        self.wo2, self.bo2 = cv(self.attn2.to_out[0].weight), _f32(self.attn2.to_out[0].bias)
        self.kv = reg.add_kv(self.attn2.to_k, self.attn2.to_v)
        self.ff.pack(reg, dtype, norm=self.norm3)
        # rows built from the previous pack's bo2 must not survive a repack: the caching allocator can hand the new bo2 the
        # address of an old one, so a data_ptr key alone would return the previous checkpoint's bias
        self.__dict__.pop("_zrows", None)

    def _zero_ctx_rows(self, ctx: StepContext, g: Geom) -> torch.Tensor:
        """fp32 [g.batch, C]: to_out's bias of the cross-attention for batch elements with an all-zero context, 0 for the others
        (cached per mask: built once, outside graph capture)."""
        cache = self.__dict__.setdefault("_zrows", {})
        key = (ctx.zero_mask, g.batch0, g.batch, self.bo2.data_ptr())
        rows = cache.get(key)
        if rows is None:
            flags = [float((ctx.zero_mask >> (g.batch0 + i)) & 1) for i in range(g.batch)]
            rows = cache[key] = (torch.tensor(flags, dtype=torch.float32, device=self.bo2.device)[:, None] * self.bo2[None, :]).contiguous()
        return rows

    def forward(self, x, g: Geom, ctx: StepContext, out_rowvec=None):
        """out_rowvec fp32 [N, C] (one row per frame of the batch): added to the block's output -- the frame-position embedding the
        temporal block that follows would otherwise add in a pass of its own (TransformerSpatioTemporalModel.forward)."""
        ff_rv = dict(rowvec=out_rowvec, rowvec_rows=g.hw) if out_rowvec is not None else {}
        a = _self_attention(x, self.attn1, self.wqk, self.bqk, self.wv, self.norm1.eps, g, ctx, self.wqkv, self.bqkv)
        live = ctx.live_batches(g)
        if live is None:
            x = ops.gemm(a, self.wo1, bias=self.bo1, residual=x)
            a = _cross_attention(x, self.attn2, self.q2, self.norm2.eps, self.kv, g, ctx, temporal=False)
            x = ops.gemm(a, self.wo2, bias=self.bo2, residual=x)
            return self.ff(x, residual=x, **ff_rv)
        # Batch elements with an all-zero context (the CFG uncond half): K = V = 0, so their cross-attention output is exactly 0
        # and the layer adds to_out's bias only -- that bias rides on the self-attention output projection as a per-batch row
        # vector, and query projection / attention / output projection run on the rows of the live elements alone (in place:
        # every lane of the GEMM epilogue reads the residual element it overwrites).
        first, count = live
        x = ops.gemm(a, self.wo1, bias=self.bo1, residual=x, rowvec=self._zero_ctx_rows(ctx, g), rowvec_rows=g.frames * g.hw)
        if count:
            rows = g.frames * g.hw
            xs = x[first * rows:(first + count) * rows]
            gl = Geom(count, g.frames, g.h, g.w, g.batch0 + first, g.ctx_batches)
            a = _cross_attention(xs, self.attn2, self.q2, self.norm2.eps, self.kv, gl, ctx, temporal=False)
            ops.gemm(a, self.wo2, bias=self.bo2, residual=xs, out=xs)
        return self.ff(x, residual=x, **ff_rv)


class TemporalBasicTransformerBlock(_Packable):
    """Frame-axis transformer block.  The reference permutes [(B F),hw,C] <-> [(B hw),F,C] around it; here the
    token-major layout is kept and the frame axis is addressed by stride inside the kernels."""

    def __init__(self, dim: int, time_mix_inner_dim: int, num_attention_heads: int, attention_head_dim: int,
                 cross_attention_dim: Optional[int] = None):
        super().__init__()
        if dim != time_mix_inner_dim or cross_attention_dim is None:
            raise NotImplementedError("SVD uses dim == time_mix_inner_dim with cross-attention")
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = FeedForward(dim, dim_out=time_mix_inner_dim)
        self.norm1 = nn.LayerNorm(time_mix_inner_dim)
        self.attn1 = Attention(time_mix_inner_dim, heads=num_attention_heads, dim_head=attention_head_dim)
        self.norm2 = nn.LayerNorm(time_mix_inner_dim)
        self.attn2 = Attention(time_mix_inner_dim, cross_attention_dim=cross_attention_dim, heads=num_attention_heads,
                               dim_head=attention_head_dim)
        self.norm3 = nn.LayerNorm(time_mix_inner_dim)
        self.ff = FeedForward(time_mix_inner_dim)

    def pack(self, reg, dtype):
        """All four LayerNorms are folded into their consumer GEMMs (see BasicTransformerBlock.pack)."""
        cv = lambda t: t.detach().to(dtype).contiguous()
        zr = lambda t: zero_sum_round(t, dtype)
        n1, n2 = self.norm1, self.norm2
        folded = [fold_layernorm(w.weight, None, n1.weight, n1.bias) for w in (self.attn1.to_q, self.attn1.to_k, self.attn1.to_v)]
        self.wqkv = zr(torch.cat([w for w, _ in folded], 0))
        self.bqkv = torch.cat([b for _, b in folded], 0).contiguous()
        self.wo1, self.bo1 = cv(self.attn1.to_out[0].weight), _f32(self.attn1.to_out[0].bias)
        wq2, bq2 = fold_layernorm(self.attn2.to_q.weight, None, n2.weight, n2.bias)
        self.q2 = _QProj(zr(wq2), bq2, self.attn2.dim_head)
        self.wo2, self.bo2 = cv(self.attn2.to_out[0].weight), _f32(self.attn2.to_out[0].bias)
        self.__dict__.pop("_crows", None)                   # rows built from the previous pack's bo2 (see _class_rows)
        self.kv = reg.add_kv(self.attn2.to_k, self.attn2.to_v)
        self.ff_in.pack(reg, dtype, norm=self.norm_in)
        self.ff.pack(reg, dtype, norm=self.norm3)

    def _class_rows(self, live, cb: int) -> torch.Tensor:
        """fp32 [cb, C]: row c = the cross-attention's to_out bias for a residue class whose context is all zeros (its attention
        output is exactly 0), zeros for a live class (cached per live set: built once, outside graph capture)."""
        cache = self.__dict__.setdefault("_crows", {})
        key = (tuple(sorted(live)), cb, self.bo2.data_ptr())
        rows = cache.get(key)
        if rows is None:
            flags = [0.0 if c in live else 1.0 for c in range(cb)]
            rows = cache[key] = (torch.tensor(flags, dtype=torch.float32, device=self.bo2.device)[:, None] * self.bo2[None, :]).contiguous()
        return rows

    def forward(self, x_spatial, pos_emb, g: Geom, ctx: StepContext, alpha: float, blend_fix=None):
        """x_spatial [M,C]; pos_emb fp32 [F,C].  Returns alpha*x_spatial + (1-alpha)*temporal(x_spatial + pos_emb).
        blend_fix fp32 [N, C]: x_spatial ALREADY carries the frame-position embedding e (the spatial block added it in its last
        epilogue), and blend_fix = -alpha / (1 - alpha) * e per frame row: the final blend alpha * (x + e) + (1 - alpha) * (v +
        blend_fix) equals alpha * x + (1 - alpha) * v -- no pass that materialises x + e, no second copy of x."""
        c = x_spatial.shape[1]
        if blend_fix is None:
            xs = ops.add_rowvec(x_spatial, pos_emb, rows_per_vec=g.hw, nvec=g.frames)        # + frame-position embedding
        else:
            xs = x_spatial
        t = self.ff_in(xs, residual=xs)
        qkv = ops.gemm(t, self.wqkv, bias=self.bqkv, ln_fold=1, ln_eps=self.norm1.eps)
        a = torch.empty((g.m, c), dtype=t.dtype, device=t.device)
        ops.temporal_attention(qkv, a, batch=g.batch, frames=g.frames, hw=g.hw, heads=self.attn1.heads,
                               head_dim=self.attn1.dim_head)
        live = ctx.live_classes(g)
        if live is None:
            t = ops.gemm(a, self.wo1, bias=self.bo1, residual=t)
            a = _cross_attention(t, self.attn2, self.q2, self.norm2.eps, self.kv, g, ctx, temporal=True)
            t = ops.gemm(a, self.wo2, bias=self.bo2, residual=t)
        else:
            # Rows of a residue class whose context is all zeros (the CFG uncond context: every other pixel, quirk Q3) get
            # exactly 0 from the cross-attention, i.e. only to_out's bias.  Query projection, attention and output projection
            # run on the live classes alone, as strided row views t[c::CB] (row stride CB*C; the GEMM and the attention take
            # row strides, the output projection writes in place over its residual).  Exact.  The dead classes' "+ bias" rides
            # on the self-attention output projection as a per-class (periodic) row vector: no separate pass over half of the rows.
            cb, off, cc = g.ctx_batches, self.kv[0], self.kv[1]
            # the self-attention's output projection for ALL classes in one launch: the dead classes' extra bias (the cross-attention's
            # to_out bias bo2) is a periodic row vector -- row r takes class_rows[r % cb] (tt_gemm rowvec_mod; cb = 2: even / odd rows)
            t = ops.gemm(a, self.wo1, bias=self.bo1, residual=t, rowvec=self._class_rows(live, cb), rowvec_rows=1, rowvec_mod=cb)
            for cls in range(cb):
                tv = t[cls::cb]
                if cls in live:
                    # every sequence of this class uses context `cls`: mask 1 with one "batch" spanning all sequences
                    akw = dict(nseq=g.n, lq=g.hw // cb, heads=self.attn2.heads, head_dim=self.attn2.dim_head, mask=1, lk=ctx.s_ctx,
                               k_seq_stride=ctx.s_pad, v_seq_stride=ctx.s_pad, frames=g.n, ctx_batches=cb, batch0=cls)
                    if MERGE_FRAMES:                         # ... i.e. ONE sequence of all the class's rows: full 128-query blocks (an image has
                        akw.update(nseq=1, lq=g.n * (g.hw // cb), frames=1)      # 56 / 14 rows of a class at the two coarsest levels)
                    a = torch.empty((tv.shape[0], c), dtype=tv.dtype, device=tv.device)
                    fused = self.q2.fused()
                    if fused is not None and ops.attention_qproj_supported(tv, self.attn2.dim_head, 1):
                        ops.attention(None, ctx.k_all[:, off:off + cc], ctx.vt_all[off:off + cc], a, qx=tv, wq=fused[0],
                                      bq=fused[1], ln_eps=self.norm2.eps, **akw)
                    else:
                        wq2, bq2 = self.q2.plain()
                        q = ops.gemm(tv, wq2, bias=bq2, ln_fold=1, ln_eps=self.norm2.eps)
                        ops.attention(q, ctx.k_all[:, off:off + cc], ctx.vt_all[off:off + cc], a, **akw)
                    ops.gemm(a, self.wo2, bias=self.bo2, residual=tv, out=tv)
        if blend_fix is not None:
            return self.ff(t, residual=t, blend=xs, alpha=alpha, rowvec=blend_fix, rowvec_rows=g.hw)
        return self.ff(t, residual=t, blend=x_spatial, alpha=alpha)
