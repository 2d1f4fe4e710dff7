"""Weight re-layout for the libttvdm kernels (done once at load; pure data movement)."""
from __future__ import annotations

import os

import torch


def pack_conv3x3(w: torch.Tensor) -> torch.Tensor:
    """[O, I, 3, 3] -> [O, 9*I], k index = (ky*3 + kx, ci): each K step of the implicit GEMM stays inside one tap."""
    o, i = w.shape[:2]
    return w.permute(0, 2, 3, 1).reshape(o, 9 * i).contiguous()


def pack_tconv3(w: torch.Tensor) -> torch.Tensor:
    """Conv3d (3,1,1) weight [O, I, 3, 1, 1] -> [O, 3*I], k index = (frame tap, ci)."""
    o, i = w.shape[:2]
    return w[:, :, :, 0, 0].permute(0, 2, 1).reshape(o, 3 * i).contiguous()


def pack_conv1x1(w: torch.Tensor) -> torch.Tensor:
    return w.reshape(w.shape[0], w.shape[1]).contiguous()


def pack_geglu(w: torch.Tensor, b: torch.Tensor):
    """GEGLU.proj weight [8C, C] (= [value 4C | gate 4C]) -> 16-row groups [8 value rows | 8 gate rows] so the
    GEMM epilogue finds value and gate of one output column in the same lane (include/ttvdm.h, tt_gemm)."""
    n2, k = w.shape
    half = n2 // 2
    assert half % 8 == 0
    wp = w.view(2, half // 8, 8, k).permute(1, 0, 2, 3).reshape(n2, k).contiguous()
    bp = b.view(2, half // 8, 8).permute(1, 0, 2).reshape(n2).contiguous()
    return wp, bp


def pad_rows(w: torch.Tensor, multiple: int) -> torch.Tensor:
    n = w.shape[0]
    np_ = (n + multiple - 1) // multiple * multiple
    if np_ == n:
        return w
    out = torch.zeros((np_,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    out[:n] = w
    return out


DEVICE_ROUNDING = os.environ.get("TT_ZSR_DEVICE", "1") != "0"       # zero_sum_round of device tensors in one HIP kernel (0: the tensor passes; A/B, tests)


def zero_sum_round(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Round the rows of ``w`` (fp32/fp64, each summing to ~0) to ``dtype`` such that every ROUNDED row still sums to zero
    (to ~1e-6 of an ulp-sized weight instead of ~sqrt(K) ulps): plain rounding leaves a residual
    r_n = sum_k round(w_nk) != 0 and the fused LayerNorm GEMM would then add  mean(x) * r_n * rstd  to its output -- an error
    that grows with the row mean of x.  Greedy mixed-radix correction, binade by binade from the coarsest spacing to the
    finest: within a binade every element has the same spacing u, so up to round(r / u) of them are moved by one ulp towards
    cancelling what is left of the residual r.  The first descent only moves elements whose own rounding error has the sign
    of the correction: for them the move lands on their OTHER rounding neighbour, so they stay within one ulp of the true
    value; what that cannot cancel (rare: a row needs more such elements than a binade holds) is handled by a second descent
    that may take any element (up to 1.5 ulp).  Both stop as soon as every row sums to exactly zero.
    fp32: returned as is (the residual is ~1e-7 of a weight)."""
    if dtype == torch.float32:
        return w.float().contiguous()
    mant, emin = {torch.bfloat16: (7, -126), torch.float16: (10, -14)}[dtype]
    if w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.shape[1] <= 16384 and DEVICE_ROUNDING:
        # the same algorithm in one kernel, one block per row (tt_zero_sum_round: bit-identical result, tests/test_ops_gpu.py): the tensor passes
        # below cost 2.4-2.9 s per process over the LayerNorm-folded matrices of both networks
        from . import _lib, ops
        wc = w.detach().contiguous()
        q0 = wc.to(dtype)
        nz = q0 != 0
        if not bool(nz.any()):
            return q0.contiguous()
        _, e = torch.frexp(q0.float().abs())
        e = torch.clamp(e - 1, min=emin)
        hi, lo = int(e[nz].max()), int(e[nz].min())
        out = torch.empty_like(q0)
        _lib.check(_lib.load().tt_zero_sum_round(wc.data_ptr(), wc.stride(0), wc.shape[0], wc.shape[1], hi, lo, out.data_ptr(), out.stride(0),
                                                 ops._code(dtype), ops._stream()), "tt_zero_sum_round")
        return out
    wd = w.detach().double()
    q = w.detach().to(dtype).double()
    up = q > wd                                               # rounded up: one ulp DOWN is its other rounding neighbour
    dn = q < wd
    _, e = torch.frexp(q.abs())                               # |q| = m * 2^e, m in [0.5, 1)
    e = torch.clamp(e - 1, min=emin)                          # binade exponent (subnormals share the lowest one)
    nz = q != 0
    r = q.sum(dim=1)                                          # residual of every row, fp64 (exact: all terms are dyadic)
    lo = int(e[nz].min()) if bool(nz.any()) else 0
    hi = int(e[nz].max()) if bool(nz.any()) else -1
    moved = torch.zeros_like(nz)
    for only_other_neighbour in (True, False):
        for lvl in range(hi, max(lo, hi - 48) - 1, -1):       # <= 49 levels of O(N K) vector work per descent, once per layer at load
            if not bool((r != 0).any()):
                break
            u = 2.0 ** (lvl - mant)
            want = torch.round(r / u).abs()                   # [N]: moves of one ulp of this binade that still fit into the residual
            if not bool((want != 0).any()):                   # (the binades coarser than every residual: no O(N K) work)
                continue
            mask = nz & (e == lvl) & ~moved
            if only_other_neighbour:
                mask = mask & torch.where((r > 0)[:, None], up, dn)
            cnt = mask.sum(dim=1)
            n = torch.minimum(want, cnt.double()) * torch.sign(r)
            sel = mask & (torch.cumsum(mask, dim=1, dtype=torch.int32) <= n.abs()[:, None])
            q = q - sel.double() * (torch.sign(n) * u)[:, None]
            moved = moved | sel
            r = r - n * u
    out = q.to(dtype)
    assert torch.equal(out.double(), q), "zero_sum_round: a corrected value is not representable"
    return out.contiguous()


def fold_layernorm(weight: torch.Tensor, bias, gamma: torch.Tensor, beta: torch.Tensor):
    """Fold the affine part of ``nn.LayerNorm`` (gamma, beta) into the ``nn.Linear`` that consumes it, for tt_gemm's
    ``ln_fold`` (include/ttvdm.h):   Linear(LN(x)) = rstd * ((x - mean) (W*gamma)^T) + (W beta + bias).
    Returns (W'', b') in fp32 with  W'' = W*gamma - rowmean_k(W*gamma):  rows that sum to zero make
    x W''^T == (x - mean) W''^T for ANY x (the row mean of x multiplies sum_k W''[n,k] = 0), so the kernel can run on the
    raw activations and only has to scale each output row by rstd, which it measures itself from the operand stream.
    Round W'' to a 16-bit storage type with ``zero_sum_round`` (keeps the zero row sums exact), not with ``.to(dtype)``.
    weight [N, K]; bias [N] or None; gamma, beta [K]."""
    w = weight.detach().float()
    wg = w * gamma.detach().float()[None, :]
    wc = wg - wg.mean(dim=1, keepdim=True)
    b = w @ beta.detach().float()
    if bias is not None:
        b = b + bias.detach().float()
    return wc.contiguous(), b.contiguous()


def permute_q_rows(t: torch.Tensor) -> torch.Tensor:
    """Rows (output features) of a query projection -- weight [heads*64, C] or bias [heads*64] -- reordered for the fused query
    projection of tt_attention (include/ttvdm.h, TtAttnArgs.qx): inside every group of 16 rows bits 2 and 3 of the row index are
    swapped, new[n] = old[d(n)] with d(n) = n with bits 2 <-> 3 exchanged.  The projection's MFMA accumulators then hold, in
    contraction slot s of lane half hi at key step ks, head dimension 16 ks + 8 hi + s -- what the K fragment of that step holds.
    (An involution: applying it twice restores the order.)"""
    n = t.shape[0]
    if n % 16:
        raise ValueError("permute_q_rows: row count must be a multiple of 16")
    idx = torch.arange(n, device=t.device)
    d = (idx & ~12) | ((idx & 4) << 1) | ((idx & 8) >> 1)
    return t[d].contiguous()


class PreSplitF32(torch.Tensor):
    """An fp32-typed [N, K] matrix whose bytes are the pre-split fp16 pairs of TtGemmArgs.presplit (see presplit_f32).  The subclass
    is the marker ops.gemm looks for; row slices / row permutations / contiguous() keep it (and stay valid: the pairing is per row,
    in aligned groups of 4 columns).  Arithmetic on it is meaningless -- pre-split LAST, after every fold that reads the values."""


def presplit_f32(w: torch.Tensor) -> torch.Tensor:
    """fp32 [N, K] (K % 4 == 0) -> the same shape and element size holding, per aligned group of 4 consecutive k, the eight fp16 values
    h0 h1 h2 h3 l0 l1 l2 l3 with h = fp16(x 2^-8) and l = fp16((x - 2^8 h) 2^3): what the split16 GEMM (tt_gemm_set_f32_split) forms
    from every fp32 operand on the fly, done once for operands that are constants of the request (include/ttvdm.h, TtGemmArgs.presplit).
    Values beyond 2^24 would overflow h, as in the kernel."""
    if isinstance(w, PreSplitF32):
        return w
    if w.dtype != torch.float32 or w.dim() != 2 or w.shape[1] % 4:
        raise ValueError(f"presplit_f32: fp32 [N, K] with K % 4 == 0 expected, got {tuple(w.shape)} {w.dtype}")
    n, k = w.shape
    xs = w.detach().contiguous() * 2.0 ** -8
    h = xs.to(torch.float16)
    l = ((xs - h.float()) * 2048.0).to(torch.float16)             # exact residual in fp32, one rounding (the kernel: one fused v_fma_mix)
    pairs = torch.stack([h.view(n, k // 4, 4), l.view(n, k // 4, 4)], 2)     # [N, K/4, 2, 4] halves = 16 bytes per group
    return pairs.reshape(n, 2 * k).contiguous().view(torch.float32).as_subclass(PreSplitF32)
