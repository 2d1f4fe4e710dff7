"""Oracle leaves: restatement of the diffusers==0.25.1 modules the reference imports.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference does not own this
arithmetic: it imports it (``svd/diffusion_arch/unet_3d_blocks.py:20-31``,
``svd/diffusion_arch/transformer_temporal.py:19-24``,
``svd/unet_spatio_temporal_condition.py:7-12``).  diffusers is absent from this
image, so every class below restates the published v0.25.1 algorithm
(SURVEY.md Appendix A); constructor surfaces are corroborated by the reference's
own call sites, cited per class.  **Leaf-level parity is unpinned.**

Parameter names equal the diffusers names so a reference
state dict loads unchanged.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- embeddings
def get_timestep_embedding(timesteps: torch.Tensor, embedding_dim: int, flip_sin_to_cos: bool = False,
                           downscale_freq_shift: float = 1, scale: float = 1, max_period: int = 10000):
    """diffusers.models.embeddings.get_timestep_embedding (A.1)."""
    assert timesteps.dim() == 1
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half_dim, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    """call sites: unet_spatio_temporal_condition.py:143,148; transformer_temporal.py:266."""

    def __init__(self, num_channels: int, flip_sin_to_cos: bool, downscale_freq_shift: float):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, flip_sin_to_cos=self.flip_sin_to_cos,
                                      downscale_freq_shift=self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    """linear_2(silu(linear_1(x))) (A.2). call sites: unet...:146,149; transformer_temporal.py:265."""

    def __init__(self, in_channels: int, time_embed_dim: int, act_fn: str = "silu", out_dim: Optional[int] = None):
        super().__init__()
        assert act_fn == "silu"
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


# --------------------------------------------------------------------------- resnets
class ResnetBlock2D(nn.Module):
    """A.3; ctor kwargs corroborated by unet_3d_blocks.py:333-344."""

    def __init__(self, *, in_channels: int, out_channels: Optional[int] = None, temb_channels: int = 512,
                 groups: int = 32, eps: float = 1e-6, output_scale_factor: float = 1.0):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.output_scale_factor = output_scale_factor
        self.norm1 = nn.GroupNorm(num_groups=groups, num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(num_groups=groups, num_channels=out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.conv_shortcut = None
        if in_channels != out_channels:
            self.conv_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0, bias=True)

    def forward(self, input_tensor, temb):
        h = self.conv1(F.silu(self.norm1(input_tensor)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + h) / self.output_scale_factor


class TemporalResnetBlock(nn.Module):
    """A.4: GroupNorm over [B,C,F,h,w] + 3-tap conv along the frame axis."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, temb_channels: int = 512, eps: float = 1e-6):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv3d(in_channels, out_channels, kernel_size=(3, 1, 1), stride=1, padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(num_groups=32, num_channels=out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv3d(out_channels, out_channels, kernel_size=(3, 1, 1), stride=1, padding=(1, 0, 0))
        self.conv_shortcut = None
        if in_channels != out_channels:
            self.conv_shortcut = nn.Conv3d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, input_tensor, temb):
        h = self.conv1(F.silu(self.norm1(input_tensor)))
        if self.time_emb_proj is not None:
            t = self.time_emb_proj(F.silu(temb))[:, :, :, None, None]      # [B,F,C,1,1]
            h = h + t.permute(0, 2, 1, 3, 4)
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return input_tensor + h


class AlphaBlender(nn.Module):
    """A.6; call site transformer_temporal.py:267."""

    def __init__(self, alpha: float, merge_strategy: str = "learned_with_images",
                 switch_spatial_to_temporal_mix: bool = False):
        super().__init__()
        self.merge_strategy = merge_strategy
        self.switch_spatial_to_temporal_mix = switch_spatial_to_temporal_mix
        if merge_strategy == "fixed":
            self.register_buffer("mix_factor", torch.Tensor([alpha]))
        elif merge_strategy in ("learned", "learned_with_images"):
            self.register_parameter("mix_factor", nn.Parameter(torch.Tensor([alpha])))
        else:
            raise ValueError(f"unknown merge strategy {merge_strategy}")

    def get_alpha(self, image_only_indicator, ndims):
        if self.merge_strategy == "fixed":
            alpha = self.mix_factor
        elif self.merge_strategy == "learned":
            alpha = torch.sigmoid(self.mix_factor)
        else:
            if image_only_indicator is None:
                raise ValueError("Please provide image_only_indicator to use learned_with_images merge strategy")
            alpha = torch.where(image_only_indicator.bool(),
                                torch.ones(1, 1, device=image_only_indicator.device),
                                torch.sigmoid(self.mix_factor)[..., None])
            if ndims == 5:
                alpha = alpha[:, None, :, None, None]
            elif ndims == 3:
                alpha = alpha.reshape(-1)[:, None, None]
            else:
                raise ValueError(f"Unexpected ndims {ndims}")
        return alpha

    def forward(self, x_spatial, x_temporal, image_only_indicator=None):
        alpha = self.get_alpha(image_only_indicator, x_spatial.ndim).to(x_spatial.dtype)
        if self.switch_spatial_to_temporal_mix:
            alpha = 1.0 - alpha
        return alpha * x_spatial + (1.0 - alpha) * x_temporal


class SpatioTemporalResBlock(nn.Module):
    """A.5; ctor kwargs at unet_3d_blocks.py:1891-1896, 2094-2099, 2212-2217."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, temb_channels: int = 512,
                 eps: float = 1e-6, temporal_eps: Optional[float] = None, merge_factor: float = 0.5,
                 merge_strategy="learned_with_images", switch_spatial_to_temporal_mix: bool = False):
        super().__init__()
        self.spatial_res_block = ResnetBlock2D(in_channels=in_channels, out_channels=out_channels,
                                               temb_channels=temb_channels, eps=eps)
        oc = out_channels if out_channels is not None else in_channels
        self.temporal_res_block = TemporalResnetBlock(in_channels=oc, out_channels=oc, temb_channels=temb_channels,
                                                      eps=temporal_eps if temporal_eps is not None else eps)
        self.time_mixer = AlphaBlender(alpha=merge_factor, merge_strategy=merge_strategy,
                                       switch_spatial_to_temporal_mix=switch_spatial_to_temporal_mix)

    def forward(self, hidden_states, temb=None, image_only_indicator=None):
        num_frames = image_only_indicator.shape[-1]
        hidden_states = self.spatial_res_block(hidden_states, temb)
        batch_frames, channels, height, width = hidden_states.shape
        batch_size = batch_frames // num_frames
        hidden_states_mix = hidden_states[None, :].reshape(batch_size, num_frames, channels, height, width).permute(0, 2, 1, 3, 4)
        hidden_states = hidden_states[None, :].reshape(batch_size, num_frames, channels, height, width).permute(0, 2, 1, 3, 4)
        if temb is not None:
            temb = temb.reshape(batch_size, num_frames, -1)
        hidden_states = self.temporal_res_block(hidden_states, temb)
        hidden_states = self.time_mixer(x_spatial=hidden_states_mix, x_temporal=hidden_states,
                                        image_only_indicator=image_only_indicator)
        return hidden_states.permute(0, 2, 1, 3, 4).reshape(batch_frames, channels, height, width)


class Downsample2D(nn.Module):
    """A.7; call site unet_3d_blocks.py:2117-2123 (name="op" -> attribute ``conv``)."""

    def __init__(self, channels: int, use_conv: bool = False, out_channels: Optional[int] = None,
                 padding: int = 1, name: str = "conv"):
        super().__init__()
        assert use_conv
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(self, hidden_states):
        return self.conv(hidden_states)


class Upsample2D(nn.Module):
    """A.7; call site unet_3d_blocks.py:2223,2332."""

    def __init__(self, channels: int, use_conv: bool = False, use_conv_transpose: bool = False,
                 out_channels: Optional[int] = None, name: str = "conv"):
        super().__init__()
        assert use_conv and not use_conv_transpose
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)

    def forward(self, hidden_states, output_size=None):
        hidden_states = F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")
        return self.conv(hidden_states)


# --------------------------------------------------------------------------- attention
class Attention(nn.Module):
    """A.8 with AttnProcessor2_0: q/k/v (no bias) -> SDPA(scale d^-1/2) -> to_out[0] (bias)."""

    def __init__(self, query_dim: int, cross_attention_dim: Optional[int] = None, heads: int = 8,
                 dim_head: int = 64, dropout: float = 0.0, bias: bool = False, out_bias: bool = True):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.heads = heads
        cross = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(cross, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(cross, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])

    def forward(self, hidden_states, encoder_hidden_states=None):
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        b = hidden_states.shape[0]
        q, k, v = self.to_q(hidden_states), self.to_k(ctx), self.to_v(ctx)
        d = self.inner_dim // self.heads
        q = q.view(b, -1, self.heads, d).transpose(1, 2)
        k = k.view(b, -1, self.heads, d).transpose(1, 2)
        v = v.view(b, -1, self.heads, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, -1, self.heads * d).to(q.dtype)
        return self.to_out[1](self.to_out[0](o))


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, hidden_states):
        hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)          # exact erf GELU


class FeedForward(nn.Module):
    def __init__(self, dim: int, dim_out: Optional[int] = None, mult: int = 4, dropout: float = 0.0,
                 activation_fn: str = "geglu"):
        super().__init__()
        assert activation_fn == "geglu"
        inner_dim = int(dim * mult)
        dim_out = dim_out if dim_out is not None else dim
        self.net = nn.ModuleList([GEGLU(dim, inner_dim), nn.Dropout(dropout), nn.Linear(inner_dim, dim_out)])

    def forward(self, hidden_states):
        for m in self.net:
            hidden_states = m(hidden_states)
        return hidden_states


class BasicTransformerBlock(nn.Module):
    """A.8; ctor call transformer_temporal.py:240-245."""

    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int,
                 cross_attention_dim: Optional[int] = None):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_attention_dim,
                               heads=num_attention_heads, dim_head=attention_head_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None):
        hidden_states = self.attn1(self.norm1(hidden_states)) + hidden_states
        hidden_states = self.attn2(self.norm2(hidden_states), encoder_hidden_states=encoder_hidden_states) + hidden_states
        hidden_states = self.ff(self.norm3(hidden_states)) + hidden_states
        return hidden_states


class TemporalBasicTransformerBlock(nn.Module):
    """A.8; ctor call transformer_temporal.py:253-259; call :361-365."""

    def __init__(self, dim: int, time_mix_inner_dim: int, num_attention_heads: int, attention_head_dim: int,
                 cross_attention_dim: Optional[int] = None):
        super().__init__()
        self.is_res = dim == time_mix_inner_dim
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = FeedForward(dim, dim_out=time_mix_inner_dim)
        self.norm1 = nn.LayerNorm(time_mix_inner_dim)
        self.attn1 = Attention(query_dim=time_mix_inner_dim, heads=num_attention_heads, dim_head=attention_head_dim)
        if cross_attention_dim is not None:
            self.norm2 = nn.LayerNorm(time_mix_inner_dim)
            self.attn2 = Attention(query_dim=time_mix_inner_dim, cross_attention_dim=cross_attention_dim,
                                   heads=num_attention_heads, dim_head=attention_head_dim)
        else:
            self.norm2, self.attn2 = None, None
        self.norm3 = nn.LayerNorm(time_mix_inner_dim)
        self.ff = FeedForward(time_mix_inner_dim)

    def forward(self, hidden_states, num_frames: int, encoder_hidden_states=None):
        batch_frames, seq_length, channels = hidden_states.shape
        batch_size = batch_frames // num_frames
        hidden_states = hidden_states[None, :].reshape(batch_size, num_frames, seq_length, channels)
        hidden_states = hidden_states.permute(0, 2, 1, 3).reshape(batch_size * seq_length, num_frames, channels)
        residual = hidden_states
        hidden_states = self.ff_in(self.norm_in(hidden_states))
        if self.is_res:
            hidden_states = hidden_states + residual
        hidden_states = self.attn1(self.norm1(hidden_states)) + hidden_states
        if self.attn2 is not None:
            hidden_states = self.attn2(self.norm2(hidden_states), encoder_hidden_states=encoder_hidden_states) + hidden_states
        ff_output = self.ff(self.norm3(hidden_states))
        hidden_states = ff_output + hidden_states if self.is_res else ff_output
        hidden_states = hidden_states[None, :].reshape(batch_size, seq_length, num_frames, channels)
        return hidden_states.permute(0, 2, 1, 3).reshape(batch_size * num_frames, seq_length, channels)
