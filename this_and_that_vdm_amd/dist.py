"""Multi-GPU plumbing for the request-parallel denoise path: one process per GPU, every rank holds
a full weight replica and serves its own requests; the only collective is the start-up weight broadcast
(RCCL over xGMI on the GPU box, gloo in the CPU tests).  Nothing here runs inside a denoise step."""
from __future__ import annotations

import time
from typing import Optional

import torch
import torch.distributed as dist


def flat_param_buffer(model: torch.nn.Module) -> torch.Tensor:
    """Re-home all parameters as views of ONE flat buffer (same dtype) so a model is broadcast by a single,
    large collective (3 GB for the UNet in bf16) instead of ~1400 small ones."""
    params = list(model.parameters())
    dt, dev = params[0].dtype, params[0].device
    if any(p.dtype != dt or p.device != dev for p in params):
        raise ValueError("flat_param_buffer needs all parameters on one device in one dtype")
    flat = torch.empty(sum(p.numel() for p in params), dtype=dt, device=dev)
    off = 0
    for p in params:
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view_as(p)
        off += n
    if hasattr(model, "invalidate_packs"):
        model.invalidate_packs()             # kernel-ready weight packs were built from the old storage
    return flat


def broadcast_model_(model: torch.nn.Module, src: int = 0, flat: Optional[torch.Tensor] = None) -> float:
    """In-place broadcast of every parameter from ``src``; returns seconds spent (0.0 when not distributed)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0.0
    flat = flat_param_buffer(model) if flat is None else flat
    if flat.is_cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.broadcast(flat, src=src)
    if flat.is_cuda:
        torch.cuda.synchronize()
    if hasattr(model, "invalidate_packs"):
        model.invalidate_packs()             # the broadcast wrote through .data: _version did not move
    return time.perf_counter() - t0


def max_over_ranks(seconds: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_requests(n_requests: int, rank: int, world: int):
    """request r -> rank r % world (independent units, no data-path collective)."""
    return [r for r in range(n_requests) if r % world == rank]
