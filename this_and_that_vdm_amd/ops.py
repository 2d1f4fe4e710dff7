"""Thin torch-tensor front end over the C ABI (include/ttvdm.h).  torch supplies device memory and the
current HIP stream only; every op below is a libttvdm kernel launch and raises if the library is
missing or the tensors are not on a HIP device."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import TT_BF16, TT_F16, TT_F32, TtAttnArgs, TtConvArgs, TtGemmArgs, check
from .packing import PreSplitF32


# Optional launch profiler (bench.py): when set to a list, gemm()/attention() append
# (kernel instance name, algorithmic flops, start event, end event) around their launch on the current stream.
PROFILE = None


def _prof_begin():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(start, name, flops, shape=None):
    if start is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        PROFILE.append((name, flops, start, e, shape))


# ---- producer -> GroupNorm hand-off (gemm(..., stats=) / gemm(..., gn=) -> groupnorm()).  The producer's per-tile column sums (or the
# tensor its reduction pass already normalised) travel with the OUTPUT TENSOR OBJECT as `_tt_stats` / `_tt_gn`, because producer and
# consumer sit in different modules (a ResBlock's conv feeds the next block's norm).  What makes that safe is the write ledger below:
# every ops.* function that writes a tensor through a raw pointer records a serial number for the tensor's STORAGE, the hand-off stores
# the serial of the launch that produced the sums, and groupnorm() uses them only if no ops.* launch has written that storage since
# (a write through ANY view of the buffer -- gemm(out=view), attention(out=), add_rowvec(out=), add_scaled(out=), nchw_to_tokens(out=) --
# bumps the serial) and the tensor still has the pointer and shape it had then.  Stale sums are ignored (statistics pass), never used.
import itertools

_SERIAL = itertools.count(1)
_LAST_WRITE = {}


def _wrote(t: torch.Tensor) -> int:
    """record a raw-pointer write to (a view of) t's storage; returns the write's serial number"""
    n = next(_SERIAL)
    _LAST_WRITE[t.untyped_storage().data_ptr()] = n
    for name in ("_tt_stats", "_tt_gn"):
        if hasattr(t, name):
            delattr(t, name)
    return n


def _handoff_valid(t: torch.Tensor, serial: int, ptr: int, shape) -> bool:
    return _LAST_WRITE.get(t.untyped_storage().data_ptr()) == serial and t.data_ptr() == ptr and tuple(t.shape) == shape


_WS = {}
_WS_RETIRED = []          # outgrown buffers stay allocated: a captured hipGraph may still hold their address
WS_FLOOR_BYTES = 64 << 20


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Persistent split-K scratch per (device, stream), grown on demand.  Launches on one stream are ordered, so every
    GEMM on it can share the buffer.  A buffer that has been handed out is NEVER freed: a hipGraph captured earlier keeps
    its raw pointer, so an outgrown buffer is retired (kept alive) instead of dropped."""
    key = (device, torch.cuda.current_stream().cuda_stream)      # per stream: concurrent branches must not share slabs
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _WS_RETIRED.append(buf)
        buf = _WS[key] = torch.empty(max(nbytes, WS_FLOOR_BYTES), dtype=torch.uint8, device=device)
    return buf


def _code(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return TT_BF16
    if dt == torch.float16:
        return TT_F16
    if dt == torch.float32:
        return TT_F32          # reference-precision mode (exact-fp32 MFMA): parity runs, not the benchmarked path
    raise RuntimeError(f"libttvdm activations/weights must be bfloat16, float16 or float32, got {dt}")


# TT_F32 products: exact-fp32 MFMA (default) or "split16" -- fp32 operands split on the fly into fp16 hi + lo, three 16-bit MFMAs per
# product block (tt_gemm_set_f32_split, include/ttvdm.h): the mode that meets the north-star tolerance at about a third of the time.
_F32_SPLIT = os.environ.get("TT_F32_SPLIT", "0") not in ("", "0")


def set_f32_split(on: bool) -> None:
    """Process-wide: TT_F32 launches of gemm() use split-fp16 products (True) or the exact-fp32 MFMA (False).  Part of DenoiseLoop's
    graph key, so a captured step is never replayed in the other mode."""
    global _F32_SPLIT
    check(_lib.load().tt_gemm_set_f32_split(int(bool(on))), "tt_gemm_set_f32_split")
    _F32_SPLIT = bool(on)


def f32_split() -> bool:
    return _F32_SPLIT


_TAG = {TT_BF16: "bf16_tag", TT_F16: "f16_tag", TT_F32: "f32_tag"}
FP8 = torch.float8_e4m3fn          # OCP e4m3 (gfx950's fp8; not MI300's fnuz)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libttvdm ops need tensors on the HIP device (no CPU fallback)")
    return t.data_ptr()


def _rows2d(t: torch.Tensor) -> Tuple[int, int]:
    """(row stride, cols) of a 2-D row view whose last dim is contiguous."""
    assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
    return t.stride(0), t.shape[1]


def gemm(a0: torch.Tensor, w: torch.Tensor, *, a1: Optional[torch.Tensor] = None, mode: int = 0, conv=None, tconv=None,
         bias=None, acc_scale: float = 1.0, rowvec=None, rowvec_rows: int = 0, rowvec_mod: int = 0, geglu: bool = False, residual=None,
         blend=None, alpha: float = 0.0, out: Optional[torch.Tensor] = None, out_f32: bool = False, m: Optional[int] = None,
         out_col_pad: Optional[Tuple[int, int]] = None, ln_fold: int = 0, ln_eps: float = 1e-5,
         out_fp8: bool = False, stats: int = 0, gn=None) -> torch.Tensor:
    """out[m, n] = epilogue(gather(a0|a1) @ w.T); see TtGemmArgs in include/ttvdm.h.
    stats = S > 0: the output will be read by a GroupNorm whose segments are S rows (h*w for per-image statistics, frames*h*w for the
    temporal ResBlock) -- ask the launch for its per-tile column sums (TtGemmArgs.stats_out).  When the route has a statistics
    epilogue AND S is a whole number of its tiles they are attached to the returned tensor (``out._tt_stats = (buffer, rows per
    tile)``) and groupnorm() normalises in one pass from them; otherwise the launch runs without them (no cost).
    gn = (gamma, beta, eps, silu) next to stats = S: the parameters of THAT GroupNorm.  Where the launch ends in a split-K reduction pass
    (the two coarsest UNet levels) the pass also normalises (TtGemmArgs.gn_out): the result is attached (``out._tt_gn``) and groupnorm()
    with the same parameters returns it without a launch; elsewhere `gn` is ignored and the statistics route above applies.
    conv = (nimg, hin, win, hout, wout, stride, upsample); tconv = (frames, hw).
    ln_fold: 1 = rows of a0 / 2 = rows of w are LayerNorm inputs (weights pre-folded by packing.fold_layernorm).
    out_fp8: the output is stored as OCP e4m3 (torch.float8_e4m3fn), the operand format of attention(..., fp8 path)."""
    lib = _lib.load()
    g = TtGemmArgs()
    lda0, k0 = _rows2d(a0)
    g.a0, g.k0, g.lda0 = _p(a0), k0, lda0
    if a1 is not None:
        lda1, k1 = _rows2d(a1)
        g.a1, g.k1, g.lda1 = _p(a1), k1, lda1
    ldw, _ = _rows2d(w)
    n = w.shape[0]
    g.w, g.ldw, g.n = _p(w), ldw, n
    g.mode = mode
    if mode == 1:
        g.nimg, g.hin, g.win, g.hout, g.wout, g.stride, g.upsample = conv
        mm = conv[0] * conv[3] * conv[4]
    elif mode == 2:
        g.frames, g.hw = tconv
        mm = a0.shape[0]
    else:
        mm = a0.shape[0]
    g.m = mm if m is None else m
    g.bias, g.acc_scale = _p(bias), acc_scale
    if rowvec is not None:
        g.rowvec, g.rowvec_rows, g.ld_rowvec, g.rowvec_mod = _p(rowvec), rowvec_rows, rowvec.stride(0), int(rowvec_mod)
        need_rows = rowvec_mod if rowvec_mod > 0 else (((mm if m is None else m) - 1) // rowvec_rows + 1)
        if rowvec.dtype != torch.float32 or rowvec.stride(1) != 1 or rowvec.shape[0] < need_rows or rowvec.shape[1] < n:
            raise RuntimeError(f"gemm: rowvec must be fp32 [>= {need_rows}, >= {n}] with contiguous rows, got {tuple(rowvec.shape)} {rowvec.dtype}")
    g.geglu = int(geglu)
    if residual is not None:
        g.residual, g.ld_res = _p(residual), residual.stride(0)
    if blend is not None:
        g.blend, g.ld_blend, g.alpha = _p(blend), blend.stride(0), alpha
    n_out = n // 2 if geglu else n
    if out is None:
        out = torch.empty((g.m, n_out), dtype=FP8 if out_fp8 else (torch.float32 if out_f32 else a0.dtype), device=a0.device)
    if out_fp8 and out.dtype != FP8:
        raise RuntimeError("out_fp8 needs a torch.float8_e4m3fn output tensor")
    g.out, g.ldo, g.out_f32, g.out_fp8 = _p(out), out.stride(0), int(out_f32), int(out_fp8)
    if out_col_pad is not None:
        g.out_col_hw, g.out_col_hwp = out_col_pad
    g.dtype = _code(a0.dtype)
    g.ln_fold, g.ln_eps = int(ln_fold), float(ln_eps)
    if g.dtype == TT_F32 and _F32_SPLIT:       # operands packed as fp16 (h, l) pairs (packing.presplit_f32): no conversion in the kernel
        g.presplit = (1 if isinstance(a0, PreSplitF32) else 0) | (2 if isinstance(w, PreSplitF32) else 0)
    elif isinstance(a0, PreSplitF32) or isinstance(w, PreSplitF32):
        raise RuntimeError("gemm: a pre-split operand outside the split16 mode (set_f32_split changed after the weights were packed: "
                           "call prepare() / begin() again)")
    need = lib.tt_gemm_ws_bytes(C.byref(g))
    if need:
        ws = _workspace(need, a0.device)
        g.ws, g.ws_bytes = ws.data_ptr(), ws.numel()
    sbuf = gnbuf = None
    if stats and gn is not None and GN_FUSED:
        g.stats_seg = int(stats)
        if lib.tt_gemm_gn_fused(C.byref(g)):
            gnbuf = torch.empty((g.m, n), dtype=out.dtype, device=a0.device)
            g.gn_out, g.ld_gn, g.gn_gamma, g.gn_beta, g.gn_eps, g.gn_silu = gnbuf.data_ptr(), n, _p(gn[0]), _p(gn[1]), float(gn[2]), int(bool(gn[3]))
    # (only for outputs up to the finest level's size at 256x448: the statistics epilogue costs 1-2.6 us PER TILE of a workgroup, so a
    # launch that walks 3-4 tiles per CU -- 512x896 latents -- pays more than the consumer's saved pass: 107.5 -> 108.5 ms/step, one call)
    if stats and gnbuf is None and GN_TILES and g.m * n <= GN_TILES_MAX_ELEMS:
        g.stats_seg = int(stats) if GN_TILES_SEG else 0        # (lets the split-K routes and the tiled template pick a tile height that divides the segment)
        srows = lib.tt_gemm_stats_rows(C.byref(g))
        if srows > 0 and lib.tt_groupnorm_tiles_supported(int(stats), n, srows, g.dtype):
            sbuf = torch.empty(((g.m + srows - 1) // srows, 2, n), dtype=torch.float32, device=a0.device)
            g.stats_out = sbuf.data_ptr()
    ev = _prof_begin()
    check(lib.tt_gemm(C.byref(g), _stream()), "tt_gemm")
    serial = _wrote(out)                                    # `out` was overwritten: sums attached by an earlier launch are dropped
    if sbuf is not None:
        out._tt_stats = (sbuf, srows, serial, out.data_ptr(), tuple(out.shape))
    if gnbuf is not None:
        out._tt_gn = (gnbuf, gn[0].data_ptr(), gn[1].data_ptr(), float(gn[2]), bool(gn[3]), int(stats), serial, out.data_ptr(), tuple(out.shape))
    if ev is not None:
        cfg = (C.c_int32 * 7)()
        lib.tt_gemm_plan(C.byref(g), cfg)
        taps = 9 if mode == 1 else (3 if mode == 2 else 1)
        tag = _TAG[g.dtype]
        if cfg[3] == 0 and cfg[1] == 320:            # the 256 x 320 big-tile kernel (gemm_w320.hip)
            kname = f"gemm_w320{'h' if cfg[0] == 128 else ''}_kernel<{tag}, {mode}, {int(bool(ln_fold))}>"
        elif cfg[3] == 0:                            # the persistent ping-pong kernel (gemm_pp.hip)
            kname = f"gemm_pp_kernel<{tag}, {int(bool(ln_fold))}, {int(bool(geglu))}>"
        elif cfg[0] == 32 and cfg[1] == 320:        # the opt-in streaming kernel for the 320 x 320 linears
            kname = f"sq320_kernel<{tag}, {'true' if residual is not None else 'false'}>"
        else:
            kmode = (mode if not ln_fold else 2 + ln_fold) + (8 if (g.dtype == TT_F32 and _F32_SPLIT) else 0)     # + 8: split-fp16 products
            kname = f"gemm_kernel<{tag}, {', '.join(str(v) for v in cfg[:6])}, {kmode}>"
        _prof_end(ev, kname, 2.0 * g.m * n * taps * (g.k0 + g.k1),
                  shape=(mode, g.m, n, taps * (g.k0 + g.k1), int(geglu), int(residual is not None)))
    return out


# ResnetBlock2D routes its 3x3 convs through tt_conv3x3 only when asked to (measured slower than groupnorm_apply + tt_gemm mode 1)
CONV3X3_FUSED = os.environ.get("TT_CONV3X3", "0") == "1"


def conv3x3_supported(h: int, w: int, c0: int, c1: int, n: int, dtype: torch.dtype) -> bool:
    """can tt_conv3x3 (fused GroupNorm + LDS-patch 3x3 conv) serve this problem?  Otherwise: groupnorm_apply + gemm(mode=1)."""
    if dtype not in (torch.bfloat16, torch.float16):
        return False
    return bool(_lib.load().tt_conv3x3_supported(h, w, c0, c1, n, _code(dtype)))


def conv3x3(x0, x1, w, nimg: int, h: int, wd: int, *, gn=None, silu: bool = True, bias=None, rowvec=None, rowvec_rows: int = 0,
            residual=None, out=None):
    """3x3 / stride 1 / pad 1 conv over token-major x0 | x1 with the input's GroupNorm (+SiLU) applied on the fly:
    gn = (scale, shift) fp32 [nimg, C] from groupnorm_stats (None: raw input).  See TtConvArgs in include/ttvdm.h."""
    lib = _lib.load()
    a = TtConvArgs()
    a.x0, a.c0, a.ld0 = _p(x0), x0.shape[1], x0.stride(0)
    if x1 is not None:
        a.x1, a.c1, a.ld1 = _p(x1), x1.shape[1], x1.stride(0)
    n = w.shape[0]
    a.w, a.ldw = _p(w), w.stride(0)
    a.nimg, a.h, a.w_img, a.n = nimg, h, wd, n
    if gn is not None:
        a.gn_scale, a.gn_shift, a.silu = _p(gn[0]), _p(gn[1]), int(silu)
    a.bias = _p(bias)
    if rowvec is not None:
        a.rowvec, a.rowvec_rows, a.ld_rowvec = _p(rowvec), rowvec_rows, rowvec.stride(0)
    if residual is not None:
        a.residual, a.ld_res = _p(residual), residual.stride(0)
    if out is None:
        out = torch.empty((nimg * h * wd, n), dtype=x0.dtype, device=x0.device)
    a.out, a.ldo, a.dtype = _p(out), out.stride(0), _code(x0.dtype)
    ev = _prof_begin()
    check(lib.tt_conv3x3(C.byref(a), _stream()), "tt_conv3x3")
    _wrote(out)
    if ev is not None:
        k = 9 * (a.c0 + a.c1)
        _prof_end(ev, f"conv_patch_kernel<{_TAG[a.dtype]}>", 2.0 * nimg * h * wd * n * k, shape=(1, nimg * h * wd, n, k, 0, int(residual is not None)))
    return out


# cross-attention with the query projection computed by the attention kernel itself (tt_attention, TtAttnArgs.qx): on by default,
# TT_ATTN_QPROJ=0 keeps the separate LayerNorm-folded projection GEMM (A/B)
ATTN_QPROJ = os.environ.get("TT_ATTN_QPROJ", "1") != "0"


def attention_qproj_supported(x, head_dim: int, mask: int) -> bool:
    return ATTN_QPROJ and mask != 0 and head_dim == 64 and x.dtype in (torch.float16, torch.bfloat16) and x.shape[-1] % 64 == 0 and \
        x.stride(0) % 8 == 0 and x.stride(1) == 1


def attention(q, k, vt, out, *, nseq, lq, heads, head_dim, mask, lk, k_seq_stride, v_seq_stride, frames=1, ctx_batches=1,
              batch0=0, qx=None, wq=None, bq=None, ln_eps: float = 1e-5, v_rows: bool = False):
    """q [nseq*lq, heads*d] -- or q=None with qx / wq / bq: the kernel computes Q = LN(qx rows) wq^T + bq itself (cross-attention,
    d = 64, 16-bit; wq / bq LayerNorm-folded and row-permuted by packing.permute_q_rows).
    v_rows: `vt` is V itself, [key rows, heads*d] (e.g. a column slice of a fused Q | K | V projection): mask 0, d = 64, 16-bit."""
    lib = _lib.load()
    a = TtAttnArgs()
    if qx is not None:
        # what the C entry point cannot see through raw pointers: element types and the weight's extent (heads * 64 rows of qc)
        if wq.dtype != qx.dtype or bq.dtype != torch.float32 or not bq.is_contiguous() or wq.stride(1) != 1 or \
                tuple(wq.shape) != (heads * head_dim, qx.shape[1]) or bq.numel() != heads * head_dim:
            raise RuntimeError(f"attention: fused query projection needs wq [{heads * head_dim}, {qx.shape[1]}] in {qx.dtype} and a contiguous "
                               f"fp32 bq [{heads * head_dim}]; got wq {tuple(wq.shape)} {wq.dtype}, bq {tuple(bq.shape)} {bq.dtype}")
        a.qx, a.ldqx, a.wq, a.ldwq, a.bq, a.qc, a.ln_eps = _p(qx), qx.stride(0), _p(wq), wq.stride(0), _p(bq), qx.shape[1], float(ln_eps)
        q = qx                                     # dtype / profiling below; a.q stays NULL
    else:
        a.q, a.ldq = _p(q), q.stride(0)
    a.k, a.ldk = _p(k), k.stride(0)
    a.vt, a.ldvt = _p(vt), vt.stride(0)
    a.out, a.ldo = _p(out), out.stride(0)
    a.nseq, a.lq, a.heads, a.head_dim = nseq, lq, heads, head_dim
    a.mask, a.lk, a.k_seq_stride, a.v_seq_stride = mask, lk, k_seq_stride, v_seq_stride
    fp8 = q.dtype == FP8                       # e4m3 operands (gemm(..., out_fp8=True)); the output type names the kernel
    if fp8 and not (k.dtype == FP8 and vt.dtype == FP8):
        raise RuntimeError("fp8 attention needs q, k and vt in torch.float8_e4m3fn")
    a.frames, a.ctx_batches, a.dtype, a.batch0, a.fp8 = frames, ctx_batches, _code(out.dtype if fp8 else q.dtype), batch0, int(fp8)
    a.v_rows = int(bool(v_rows))
    ev = _prof_begin()
    check(lib.tt_attention(C.byref(a), _stream()), "tt_attention")
    _wrote(out)
    if ev is not None:
        tag = _TAG[a.dtype]
        if fp8:
            kname = f"attn8_kernel<{tag}, {head_dim}>"
        elif v_rows and lk >= 128 and lk % 64 == 0 and os.environ.get("TT_ATTN_PIPE", "1") != "0":     # (launch_attn in attention.hip)
            kname = f"attn_pipe_kernel<{tag}>"
        else:
            kname = f"attn_kernel<{tag}, {head_dim}, {mask}{', true' if qx is not None else ''}{', false, true' if v_rows else ''}>"
        flops = 4.0 * nseq * heads * lq * lk * head_dim + (2.0 * nseq * lq * heads * head_dim * qx.shape[1] if qx is not None else 0.0)
        _prof_end(ev, kname, flops, shape=("attn", nseq * heads, lq, lk, mask, 0))
    return out


def temporal_attention(qkv, out, *, batch, frames, hw, heads, head_dim):
    lib = _lib.load()
    check(lib.tt_temporal_attention(_p(qkv), qkv.stride(0), _p(out), out.stride(0), batch, frames, hw, heads, head_dim,
                                    _code(qkv.dtype), _stream()), "tt_temporal_attention")
    _wrote(out)
    return out


def groupnorm_stats(x0, x1, nimg, hw, frames_per_group, gamma, beta, eps):
    """-> (scale, shift) fp32 [nimg, C] with y = x*scale + shift."""
    lib = _lib.load()
    c0 = x0.shape[-1]
    c1 = 0 if x1 is None else x1.shape[-1]
    c = c0 + c1
    ws_bytes = lib.tt_groupnorm_ws_bytes(nimg, hw, c)
    ws = torch.empty(ws_bytes // 8, dtype=torch.float64, device=x0.device)
    scale = torch.empty((nimg, c), dtype=torch.float32, device=x0.device)
    shift = torch.empty_like(scale)
    check(lib.tt_groupnorm_stats(_p(x0), c0, _p(x1), c1, nimg, hw, frames_per_group, _p(gamma), _p(beta), eps,
                                 _p(scale), _p(shift), _p(ws), ws_bytes, _code(x0.dtype), _stream()), "tt_groupnorm_stats")
    return scale, shift


def groupnorm_apply(x0, x1, nimg, hw, scale, shift, silu: bool, out=None):
    lib = _lib.load()
    c0 = x0.shape[-1]
    c1 = 0 if x1 is None else x1.shape[-1]
    if out is None:
        out = torch.empty((nimg * hw, c0 + c1), dtype=x0.dtype, device=x0.device)
    check(lib.tt_groupnorm_apply(_p(x0), c0, _p(x1), c1, nimg, hw, _p(scale), _p(shift), int(silu), _p(out), out.stride(0),
                                 _code(x0.dtype), _stream()), "tt_groupnorm_apply")
    _wrote(out)
    return out


GN_SMALL = os.environ.get("TT_GN_SMALL", "1") != "0"      # A/B switch for the one-launch GroupNorm
# cross-frame statistics (TemporalResnetBlock: one GroupNorm per video over frames x h x w): the frames of a video are contiguous
# rows, so a video is one "image" of frames * hw rows for the one-launch kernel (a launch then has batch x 16 working blocks).
# OFF by default (0 rows): one launch instead of three (-44 launches per step at the coarsest level, 14 x 28 = 392 rows per video),
# but the step is not faster -- 30.40 / 30.42 ms without, 30.46 / 30.47 with a bound of 512 rows, 30.62 / 30.60 with 2048 (one
# gpurun call, interleaved): the three short launches hide behind the other branch.  TT_GN_CROSS_ROWS=512 enables it (A/B, tests).
GN_CROSS_MAX_ROWS = int(os.environ.get("TT_GN_CROSS_ROWS", "0"))


# GroupNorm from the producer's tile sums (tt_gemm stats_out -> tt_groupnorm_tiles): on by default, TT_GN_TILES=0 keeps every
# GroupNorm on the statistics-pass kernels (A/B)
GN_TILES = os.environ.get("TT_GN_TILES", "1") != "0"
GN_TILES_SEG = os.environ.get("TT_GN_TILES_SEG", "1") != "0"       # ... also on the split-K routes (coarse levels) and with per-wave-row sums of the tiled template (A/B)
GN_FUSED = os.environ.get("TT_GN_FUSED", "1") != "0"        # GroupNorm inside the split-K reduction pass (TtGemmArgs.gn_out), A/B
GN_TILES_MAX_ELEMS = int(os.environ.get("TT_GN_TILES_MAX_ELEMS", str(17 << 20)))
# (the "apply-only consumer" timing experiment of round 5 -- groupnorm() with scale 1 / shift 0, wrong numbers -- lives in
# tools/gn_emulate.py as a monkeypatch now: a leaked TT_GN_EMULATE=1 can no longer corrupt every GroupNorm silently)
if os.environ.get("TT_GN_EMULATE", "0") == "1":
    raise RuntimeError("TT_GN_EMULATE is no longer read by the library (it produced wrong numbers by design): use tools/gn_emulate.py")


def groupnorm(x0, x1, nimg, hw, frames_per_group, gamma, beta, eps, silu: bool):
    """act(group_norm(x0 | x1)) -> new tensor.  Small per-image problems take ONE launch (tt_groupnorm_small), everything else
    tt_groupnorm_stats + tt_groupnorm_apply."""
    lib = _lib.load()
    c0 = x0.shape[-1]
    c1 = 0 if x1 is None else x1.shape[-1]
    fz = getattr(x0, "_tt_gn", None) if x1 is None else None
    if fz is not None and fz[1:6] == (gamma.data_ptr(), beta.data_ptr(), float(eps), bool(silu), frames_per_group * hw) and \
            _handoff_valid(x0, *fz[6:]):
        return fz[0]                                         # the producer's reduction pass already normalised (gemm(..., gn=))
    st = getattr(x0, "_tt_stats", None) if (GN_TILES and x1 is None) else None
    if st is not None and not _handoff_valid(x0, *st[2:]):
        st = None                                            # something wrote the buffer after the producer: statistics pass instead
    if st is not None and nimg % frames_per_group == 0 and x0.is_contiguous() and st[0].shape[2] == c0 and \
            lib.tt_groupnorm_tiles_supported(frames_per_group * hw, c0, st[1], _code(x0.dtype)):
        # the producer of x0 left per-tile column sums: one pass over x0, one launch, per-image and cross-frame statistics alike
        out = torch.empty((nimg * hw, c0), dtype=x0.dtype, device=x0.device)
        check(lib.tt_groupnorm_tiles(_p(x0), c0, _p(st[0]), st[1], nimg // frames_per_group, frames_per_group * hw, _p(gamma), _p(beta), eps,
                                     int(silu), _p(out), out.stride(0), _code(x0.dtype), _stream()), "tt_groupnorm_tiles")
        return out
    if GN_SMALL and frames_per_group > 1 and nimg % frames_per_group == 0 and frames_per_group * hw <= GN_CROSS_MAX_ROWS and \
            lib.tt_groupnorm_small_supported(frames_per_group * hw, c0 + c1, _code(x0.dtype)):
        nimg, hw, frames_per_group = nimg // frames_per_group, frames_per_group * hw, 1
    if GN_SMALL and frames_per_group == 1 and lib.tt_groupnorm_small_supported(hw, c0 + c1, _code(x0.dtype)):
        out = torch.empty((nimg * hw, c0 + c1), dtype=x0.dtype, device=x0.device)
        check(lib.tt_groupnorm_small(_p(x0), c0, _p(x1), c1, nimg, hw, _p(gamma), _p(beta), eps, int(silu), _p(out), out.stride(0),
                                     _code(x0.dtype), _stream()), "tt_groupnorm_small")
        return out
    sc, sh = groupnorm_stats(x0, x1, nimg, hw, frames_per_group, gamma, beta, eps)
    return groupnorm_apply(x0, x1, nimg, hw, sc, sh, silu)


def layernorm(x, gamma, beta, eps=1e-5, rowvec=None, rows_per_vec=0, nvec=0):
    """-> y, or (x + rowvec, y) when a row vector is fused in."""
    lib = _lib.load()
    rows, c = x.shape
    y = torch.empty((rows, c), dtype=x.dtype, device=x.device)
    xs = torch.empty((rows, c), dtype=x.dtype, device=x.device) if rowvec is not None else None
    if xs is not None:
        assert x.stride(0) == c
    check(lib.tt_layernorm(_p(x), x.stride(0), rows, c, _p(gamma), _p(beta), eps, _p(rowvec), rows_per_vec, nvec, _p(xs),
                           _p(y), y.stride(0), _code(x.dtype), _stream()), "tt_layernorm")
    return (xs, y) if xs is not None else y


def add_rowvec(x, rowvec, rows_per_vec: int, nvec: int, out=None):
    """y[r] = x[r] + rowvec[(r // rows_per_vec) % nvec]  (frame-position embedding, transformer_temporal.py:358-359).
    ``out`` may be ``x`` itself (in place) and both may be strided row views."""
    lib = _lib.load()
    rows, c = x.shape
    assert x.stride(1) == 1 and rowvec.dtype == torch.float32 and rowvec.stride(1) == 1
    y = torch.empty((rows, c), dtype=x.dtype, device=x.device) if out is None else out
    assert y.shape == x.shape and y.stride(1) == 1 and y.dtype == x.dtype
    check(lib.tt_add_rowvec(_p(x), x.stride(0), rows, c, _p(rowvec), rowvec.stride(0), rows_per_vec, nvec, _p(y), y.stride(0),
                            _code(x.dtype), _stream()), "tt_add_rowvec")
    _wrote(y)
    return y


def softmax_rows(x, dtype, cols: Optional[int] = None, out=None):
    """row softmax of fp32 scores x[:, :cols] -> `dtype`, columns cols..out.shape[1] zeroed (padding for a following GEMM).
    Contract of tt_softmax_rows: the kernel reads whole float4 quads up to round_up(cols, 8), so the score rows must be
    at least that long in memory (x.stride(0) >= round_up(cols, 8); the values beyond `cols` are ignored) and the output
    has round_up(cols, 8) or more columns."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.stride(1) == 1
    rows = x.shape[0]
    cols = x.shape[1] if cols is None else cols
    cols_pad = (cols + 7) // 8 * 8
    if x.stride(0) < cols_pad:
        raise ValueError(f"softmax_rows: score rows of {cols} columns need a row stride >= {cols_pad} floats (got {x.stride(0)}): "
                         "allocate the score buffer with its column count rounded up to 8")
    if out is None:
        out = torch.empty((rows, cols_pad), dtype=dtype, device=x.device)
    if out.shape[1] < cols_pad:
        raise ValueError(f"softmax_rows: output needs >= {cols_pad} columns, got {out.shape[1]}")
    check(lib.tt_softmax_rows(_p(x), x.stride(0), rows, cols, _p(out), out.stride(0), out.shape[1], _code(dtype), _stream()),
          "tt_softmax_rows")
    _wrote(out)
    return out


def small_linear(x, w, bias=None, act_in=False, act_out=False, out=None, accumulate=False):
    lib = _lib.load()
    rows, k = x.shape
    n = w.shape[0]
    if out is None:
        out = torch.empty((rows, n), dtype=torch.float32, device=x.device)
    assert x.dtype == torch.float32 and out.dtype == torch.float32
    check(lib.tt_small_linear(_p(x), x.stride(0), rows, k, _p(w), w.stride(0), n, _p(bias), int(act_in), int(act_out),
                              int(accumulate), _p(out), out.stride(0), _code(w.dtype), _stream()), "tt_small_linear")
    _wrote(out)
    return out


def timestep_embedding(t, dim):
    lib = _lib.load()
    assert t.dtype == torch.float32 and t.dim() == 1
    out = torch.empty((t.shape[0], dim), dtype=torch.float32, device=t.device)
    check(lib.tt_timestep_embedding(_p(t), t.shape[0], dim, _p(out), out.stride(0), _stream()), "tt_timestep_embedding")
    return out


def prep_model_input(latents, image_latents, cond, sigmas, step, batch, frames, h, w, cpad, dtype):
    lib = _lib.load()
    x = torch.empty((batch * frames * h * w, cpad), dtype=dtype, device=latents.device)
    check(lib.tt_prep_model_input(_p(latents), _p(image_latents), _p(cond), _p(sigmas), step, batch, frames, h, w, cpad,
                                  _p(x), _code(dtype), _stream()), "tt_prep_model_input")
    return x


def cfg_euler_step(eps, latents, guidance, sigmas, step, batch, frames, h, w, image_guidance_scale=None):
    """batch 1 (no CFG), 2 (uncond, cond) or 3 (use_instructpix2pix: first-frame, cond, uncond + image_guidance_scale)."""
    lib = _lib.load()
    if batch == 3:
        if image_guidance_scale is None or guidance is None:
            raise ValueError("a CFG batch of 3 needs guidance and image_guidance_scale")
        check(lib.tt_cfg3_euler_step(_p(eps), eps.stride(0), _p(latents), _p(guidance), float(image_guidance_scale), _p(sigmas),
                                     step, frames, h, w, _stream()), "tt_cfg3_euler_step")
        return latents
    check(lib.tt_cfg_euler_step(_p(eps), eps.stride(0), _p(latents), _p(guidance), _p(sigmas), step, batch, frames, h, w,
                                _stream()), "tt_cfg_euler_step")
    return latents


def nchw_to_tokens(src, dtype, ld=None, out=None):
    """[N,C,H,W] contiguous (fp32 or `dtype`) -> token-major `dtype` [N*H*W, ld>=C] (extra columns zero), or into
    ``out`` (a [N*H*W, C] column window of a wider token buffer)."""
    lib = _lib.load()
    n, c, h, w = src.shape
    src = src.contiguous()
    if src.dtype not in (torch.float32, dtype):
        src = src.float()
    if out is None:
        ld = c if ld is None else ld
        out = (torch.zeros if ld != c else torch.empty)((n * h * w, ld), dtype=dtype, device=src.device)
    assert out.dtype == dtype and out.stride(1) == 1
    check(lib.tt_nchw_to_tokens(_p(src), int(src.dtype == torch.float32), n, c, h * w, _p(out), out.stride(0), _code(dtype),
                                _stream()), "tt_nchw_to_tokens")
    _wrote(out)
    return out


def tokens_to_nchw(src, n, c, h, w, out_dtype):
    lib = _lib.load()
    src_f32 = src.dtype == torch.float32
    # `dtype` names the 16-bit side of the conversion; fp32 -> fp32 (eps of the UNet, TT_F32 tokens) has none and passes TT_F32
    tok_dtype = out_dtype if src_f32 and out_dtype != torch.float32 else (src.dtype if not src_f32 else torch.float32)
    dst_f32 = out_dtype == torch.float32
    if not dst_f32 and not src_f32:
        assert out_dtype == src.dtype
    dst = torch.empty((n, c, h, w), dtype=out_dtype, device=src.device)
    check(lib.tt_tokens_to_nchw(_p(src), int(src_f32), src.stride(0), n, c, h * w, _p(dst), int(dst_f32), _code(tok_dtype),
                                _stream()), "tt_tokens_to_nchw")
    return dst


def add_scaled(a, b, scale=1.0, out=None):
    lib = _lib.load()
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    if out is None:
        out = torch.empty_like(a)
    check(lib.tt_add_scaled(_p(a), _p(b), scale, _p(out), a.numel(), _code(a.dtype), _stream()), "tt_add_scaled")
    _wrote(out)
    return out
