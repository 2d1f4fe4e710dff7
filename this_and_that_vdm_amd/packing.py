"""Weight re-layout for the libttvdm kernels (done once at load; pure data movement)."""
from __future__ import annotations

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


def fold_layernorm(weight: torch.Tensor, bias, gamma: torch.Tensor, beta: torch.Tensor):
    """Fold the affine part of ``nn.LayerNorm`` (gamma, beta) into the ``nn.Linear`` that consumes it, for tt_gemm's
    ``ln_fold`` (include/ttvdm.h):   Linear(LN(x)) = rstd * ((x - mean) (W*gamma)^T) + (W beta + bias).
    Returns (W'', b') in fp32 with  W'' = W*gamma - rowmean_k(W*gamma):  rows that sum to zero make
    x W''^T == (x - mean) W''^T for ANY x (the row mean of x multiplies sum_k W''[n,k] = 0), so the kernel can run on the
    raw activations and only has to scale each output row by rstd, which it measures itself from the operand stream.
    weight [N, K]; bias [N] or None; gamma, beta [K]."""
    w = weight.detach().float()
    wg = w * gamma.detach().float()[None, :]
    wc = wg - wg.mean(dim=1, keepdim=True)
    b = w @ beta.detach().float()
    if bias is not None:
        b = b + bias.detach().float()
    return wc.contiguous(), b.contiguous()
