"""What UNetSpatioTemporalConditionModel and ControlNetModel share: weight packing, the time/FiLM embedding
path, the per-request context K/V projection and the encoder walk."""
from __future__ import annotations

import os

from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from .layers import Geom, PackRegistry, StepContext, _f32
from .modeling_utils import ModelMixin
from ..packing import pack_conv3x3


def as_tokens(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """logical [N,C,h,w] -> token view [N*h*w, C]; zero-copy when the tensor is already channels-last `dtype`."""
    n, c, h, w = t.shape
    if t.dtype == dtype and t.permute(0, 2, 3, 1).is_contiguous():
        return t.permute(0, 2, 3, 1).reshape(n * h * w, c)
    return ops.nchw_to_tokens(t, dtype)


def as_nchw_view(tok: torch.Tensor, g: Geom) -> torch.Tensor:
    """token buffer -> logical [N,C,h,w] tensor (channels-last strides, no copy)."""
    return tok.view(g.n, g.h, g.w, tok.shape[1]).permute(0, 3, 1, 2)


# split16 (TT_F32 with split-fp16 products): every packed GEMM weight is a constant of the request, so it is split into its fp16
# (h, l) pairs ONCE here instead of in every launch (packing.presplit_f32; the kernel then converts the activation operand only --
# half of its conversion VALU).  Attribute names per leaf class; views are re-derived from their converted base.
_PRESPLIT_ATTRS = {
    "ResnetBlock2D": ("w1", "w2", "ws"), "TemporalResnetBlock": ("w1", "w2"), "Downsample2D": ("w",), "Upsample2D": ("w",),
    "FeedForward": ("wg", "w2"), "BasicTransformerBlock": ("wqkv", "wo1", "wo2"), "TemporalBasicTransformerBlock": ("wqkv", "wo1", "wo2"),
    "TransformerSpatioTemporalModel": ("w_in", "w_out"),
}


def presplit_packed_weights(root: nn.Module) -> None:
    from ..packing import presplit_f32
    for m in root.modules():
        d = m.__dict__
        for name in _PRESPLIT_ATTRS.get(type(m).__name__, ()):
            t = d.get(name)
            if torch.is_tensor(t) and t.dtype == torch.float32 and t.dim() == 2:
                d[name] = presplit_f32(t)
        if type(m).__name__ == "BasicTransformerBlock" and "wqk" in d:       # views of wqkv: Q | K rows, V rows
            c2 = d["wqk"].shape[0]
            d["wqk"], d["wv"] = d["wqkv"][:c2], d["wqkv"][c2:]
        q2 = d.get("q2")
        if q2 is not None and not q2.is_fused and q2.w.dtype == torch.float32:
            q2.w, q2._other = presplit_f32(q2.w), None
    for name in ("_w_in", "_w_out", "_k_w", "_v_w"):
        t = root.__dict__.get(name)
        if torch.is_tensor(t) and t.dtype == torch.float32 and t.dim() == 2:
            root.__dict__[name] = presplit_f32(t)
    for name in ("_zero", "_zero_mid"):                                     # ControlNet zero-convs: (weight, bias) pairs
        z = root.__dict__.get(name)
        if isinstance(z, list):
            root.__dict__[name] = [(presplit_f32(w), b) if torch.is_tensor(w) and w.dtype == torch.float32 and w.dim() == 2 else (w, b) for w, b in z]
        elif isinstance(z, tuple) and torch.is_tensor(z[0]) and z[0].dtype == torch.float32 and z[0].dim() == 2:
            root.__dict__[name] = (presplit_f32(z[0]),) + tuple(z[1:])


class DenoiserBase(ModelMixin):
    # None: the parameter dtype if it is 16-bit, else bf16.  torch.float32 selects the reference-precision mode (TT_F32:
    # the same launch sequence on fp32 storage with the exact-fp32 MFMA, ~1/16 of the bf16 rate) used by the parity tests.
    compute_dtype: Optional[torch.dtype] = None
    # True: the spatial self-attention (the O(L^2) kernel) runs on OCP e4m3 operands with fp8 MFMA (BASELINE config 5:
    # "fp8 MFMA attention path"); projections write e4m3 directly, softmax / accumulation stay fp32.  Lower precision than the
    # default 16-bit path (3 mantissa bits on Q, K, V, P): tests/test_ops_gpu.py::test_attention_fp8 states the tolerance.
    attention_fp8: bool = False
    # Skip the cross-attention arithmetic of batch elements whose context is all zeros (see project_context): exact, on by
    # default; TT_ZERO_CTX=0 or setting this to False keeps the general path (A/B measurements, tests).
    zero_context_shortcut: bool = os.environ.get("TT_ZERO_CTX", "1") != "0"

    # ---- packing
    _pack_gen = 0          # bumped by every (re)pack: consumers holding raw pointers to packed buffers (captured
                           # hipGraphs in DenoiseLoop) compare it to know their pointers went stale

    def _pack_key(self):
        # _version catches in-place updates and load_state_dict; data_ptr catches `.data` re-homing (dist.flat_param_buffer),
        # whose later writes through the flat buffer do NOT bump _version -- dist.broadcast_model_ also invalidates explicitly
        # (the parameter LIST is cached: walking the module tree of a 1.5 B-parameter UNet costs ~0.9 ms per call and begin() / step()
        # ask several times per request -- 8 ms of host time per begin() before round 5.  _apply / load_state_dict / invalidate_packs and
        # every repack drop the list.  What the key does NOT see: a parameter REPLACED by assignment (`mod.weight = nn.Parameter(..)`,
        # an adapter merge that swaps tensors) -- the cached list still holds the old object.  Call invalidate_packs() after replacing
        # parameter objects; in-place updates (`.copy_`, optimizer steps, load_state_dict) need nothing.)
        pl = self.__dict__.get("_plist")
        if pl is None:
            pl = self.__dict__["_plist"] = list(self.parameters())
        p0 = pl[0]
        # (the TT_F32 product mode decides whether the packed weights are pre-split: toggling it repacks)
        return (p0.device, self._run_dtype(), sum(p._version for p in pl), p0.data_ptr(), ops.f32_split())

    def invalidate_packs(self):
        """Force the next prepare() to repack.  REQUIRED after replacing parameter objects by assignment (`mod.weight = nn.Parameter(..)`)
        and after writing parameters through ``.data`` or an aliasing buffer: neither bumps a `_version` the pack key can see."""
        self._packed_key = None
        self.__dict__.pop("_plist", None)

    def _run_dtype(self) -> torch.dtype:
        if self.compute_dtype is not None:
            return self.compute_dtype
        dt = self.dtype
        return dt if dt in (torch.float16, torch.bfloat16) else torch.bfloat16

    def prepare(self, force: bool = False):
        """Pack weights for the kernels (lazy; repeated when parameters move or change)."""
        key = self._pack_key()
        if not force and getattr(self, "_packed_key", None) == key:
            return self
        if next(self.parameters()).device.type != "cuda":
            raise RuntimeError(f"{type(self).__name__}: the denoise path runs on the MI355X only; move the model to the "
                               "HIP device first (there is no CPU fallback)")
        dtype = key[1]
        self.__dict__.pop("_plist", None)                   # a repack re-reads the module tree (parameters registered since the last one)
        reg = PackRegistry()
        self._pack_modules(reg, dtype)
        self._film_w = torch.cat(reg.film_w, 0).to(dtype).contiguous()
        self._film_b = torch.cat(reg.film_b, 0).float().contiguous()
        self._k_w = torch.cat(reg.k_w, 0).to(dtype).contiguous()
        self._v_w = torch.cat(reg.v_w, 0).to(dtype).contiguous()
        if dtype == torch.float32 and ops.f32_split():
            presplit_packed_weights(self)
        self._packed_key = self._pack_key()
        self._pack_gen = self._pack_gen + 1
        return self

    def _pack_modules(self, reg: PackRegistry, dtype):
        raise NotImplementedError

    # ---- embeddings (unet_spatio_temporal_condition.py:399-432 == temporal_controlnet.py:527-560)
    def _embed(self, timestep, added_time_ids, batch: int, device) -> torch.Tensor:
        if not torch.is_tensor(timestep):
            t = torch.full((batch,), float(timestep), dtype=torch.float32, device=device)
        else:
            t = timestep.to(device=device, dtype=torch.float32).reshape(-1).expand(batch).contiguous()
        emb = self.time_embedding(self.time_proj(t))
        ids = added_time_ids.to(device=device, dtype=torch.float32).reshape(-1).contiguous()
        te = self.add_time_proj(ids).reshape(batch, -1)
        ae = self.add_embedding
        hid = ops.small_linear(te, ae.w1, ae.b1, act_out=True)
        return ops.small_linear(hid, ae.w2, ae.b2, out=emb, accumulate=True)     # emb + aug_emb, fp32 [B, 1280]

    def film_table(self, timesteps: torch.Tensor, added_time_ids: torch.Tensor, batch: int) -> torch.Tensor:
        """FiLM rows of every ResBlock for EVERY step of a request: fp32 [steps, batch, sum C].  The time embedding and the
        ResBlocks' time_emb_proj depend on the schedule only (timestep t_i and the request's added_time_ids), not on the
        latents, so the fused loop evaluates them once per request for all steps -- `steps * batch` rows through the same
        kernels, <= 32 rows per launch -- instead of re-reading the ~100 MB of FiLM weights in ~20 launch-bound kernels at
        the head of every step (reference: unet_spatio_temporal_condition.py:399-432 + ResnetBlock2D.time_emb_proj, per
        forward).  Row r of a launch is computed exactly as a single-row launch would: the values are the per-step ones bit for bit."""
        self.prepare()
        dev = self._film_w.device
        t = timesteps.to(device=dev, dtype=torch.float32).reshape(-1)
        ids = added_time_ids.to(device=dev, dtype=torch.float32).reshape(batch, -1)
        steps = t.numel()
        out = torch.empty((steps, batch, self._film_w.shape[0]), dtype=torch.float32, device=dev)
        per = max(1, 32 // batch)                           # steps per launch
        for s0 in range(0, steps, per):
            s1 = min(steps, s0 + per)
            rows = (s1 - s0) * batch
            emb = self.time_embedding(self.time_proj(t[s0:s1].repeat_interleave(batch).contiguous()))
            te = self.add_time_proj(ids.repeat(s1 - s0, 1).reshape(-1).contiguous()).reshape(rows, -1)
            ae = self.add_embedding
            hid = ops.small_linear(te, ae.w1, ae.b1, act_out=True)
            ops.small_linear(hid, ae.w2, ae.b2, out=emb, accumulate=True)
            ops.small_linear(emb, self._film_w, self._film_b, act_in=True, out=out[s0:s1].view(rows, -1))
        return out

    def project_context(self, encoder_hidden_states: torch.Tensor):
        """K / V^T of every cross-attention layer from the request's context: two GEMMs, step-invariant.
        Returns an opaque tuple accepted by forward(..., _context=...)."""
        self.prepare()
        dtype = self._run_dtype()
        b, s, d = encoder_hidden_states.shape
        sp = (s + 7) // 8 * 8
        pad = torch.zeros((b, sp, d), dtype=dtype, device=encoder_hidden_states.device)
        pad[:, :s] = encoder_hidden_states.to(dtype)
        pad = pad.view(b * sp, d)
        k_all = ops.gemm(pad, self._k_w)                              # [B*Sp, sumC]
        vt_all = ops.gemm(self._v_w, pad)                             # [sumC, B*Sp]
        # Batch elements whose context is ALL zeros (the CFG uncond half: `negative_image_embeddings = zeros_like`, reference
        # pipeline :145-152, quirk Q8): to_k / to_v have no bias, so K = V = 0, every score is 0, softmax is uniform over
        # zero values and the cross-attention output is exactly 0 -- the layer reduces to to_out's bias.  The spatial
        # transformer blocks skip the query projection / attention / output projection for those rows (layers.py).
        # One host read per request (step-invariant); bit i = batch element i.
        zero_mask = 0
        # (the check reads the flags on the host: skipped while the caller is capturing a graph around forward())
        if self.zero_context_shortcut and not torch.cuda.is_current_stream_capturing():
            flags = (pad.view(b, -1) == 0).all(1).tolist()
            zero_mask = sum(1 << i for i, z in enumerate(flags) if z)
        return (k_all, vt_all, s, sp, zero_mask)

    def _step_context(self, emb: Optional[torch.Tensor], context, film: Optional[torch.Tensor] = None) -> StepContext:
        """``film``: this step's FiLM rows when the caller holds them already (DenoiseLoop: one row block of film_table)."""
        if film is None:
            film = ops.small_linear(emb, self._film_w, self._film_b, act_in=True)     # every ResBlock's FiLM row at once
        k_all, vt_all, s, sp, zero_mask = context
        return StepContext(film, k_all, vt_all, s, sp, attn_fp8=bool(self.attention_fp8), zero_mask=zero_mask)

    # ---- encoder walk shared by both models
    def _encode(self, x, g: Geom, ctx: StepContext):
        skips: List[Tuple[torch.Tensor, Geom]] = [(x, g)]
        for blk in self.down_blocks:
            x, g, outs = blk(x, g, ctx)
            skips.extend(outs)
        return x, g, skips

    def _apply(self, fn, *a, **k):
        self._packed_key = None
        self.__dict__.pop("_plist", None)
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed_key = None
        self.__dict__.pop("_plist", None)
        return super().load_state_dict(*a, **k)
