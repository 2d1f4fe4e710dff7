"""Multi-GPU plumbing for the request-parallel denoise path: one process per GPU, every rank holds
a full weight replica and serves its own requests; the only collective is the start-up weight broadcast
(RCCL over xGMI on the GPU box, gloo in the CPU tests).  Nothing here runs inside a denoise step."""
from __future__ import annotations

import time
from typing import Optional

import torch
import torch.distributed as dist


class RendezvousError(RuntimeError):
    """The N > 1 start-up failed (bad rank / device mapping, process-group init, or the first collective)."""


def init_ranks(backend: str, rank: int, world: int, local_rank: int, device=None, timeout_s: float = 180.0) -> float:
    """Fail-loud start-up of the one-process-per-GPU job: checks the rank / LOCAL_RANK / device mapping BEFORE touching the
    process group, initialises it with a finite timeout, and proves the transport with a 1-element broadcast + all-reduce
    before any large collective (the 3 GB weight broadcast) is queued on it.  Raises RendezvousError with the cause in its
    text; returns the seconds spent.  ``backend`` "nccl" is RCCL on ROCm (one GPU per rank), "gloo" the CPU tests."""
    import datetime
    import os
    t0 = time.perf_counter()
    if world < 1 or not (0 <= rank < world):
        raise RendezvousError(f"rank {rank} outside world size {world} (RANK / WORLD_SIZE from the launcher)")
    if backend == "nccl":
        if not torch.cuda.is_available():
            raise RendezvousError("backend nccl (RCCL) needs GPUs: torch.cuda.is_available() is False")
        ndev = torch.cuda.device_count()
        if not (0 <= local_rank < ndev):
            raise RendezvousError(f"LOCAL_RANK {local_rank} has no GPU: {ndev} device(s) visible to rank {rank} "
                                  f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')!r}, "
                                  f"ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES')!r}); one rank per GPU is required")
        if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0") != "0":
            # a property of the host driver this was developed on (dmabuf IPC only), not of the library: warn, do not refuse --
            # if the setting really is wrong for the host, the probe collectives below fail with the transport's own message
            import warnings
            warnings.warn("HSA_ENABLE_IPC_MODE_LEGACY=" + repr(os.environ["HSA_ENABLE_IPC_MODE_LEGACY"]) + ": hosts whose driver only "
                          "supports dmabuf IPC need 0 for RCCL (hipIpcGetMemHandle fails otherwise)", RuntimeWarning, stacklevel=2)
    for var in ("MASTER_ADDR", "MASTER_PORT"):
        if world > 1 and not os.environ.get(var):
            raise RendezvousError(f"{var} is not set: launch with torch.distributed.run --master-addr 127.0.0.1 --master-port P")
    if world == 1:
        return 0.0
    kw = dict(rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    try:
        dist.init_process_group(backend, **kw)
    except Exception as e:                                   # noqa: BLE001 -- every cause is reported the same way
        raise RendezvousError(f"init_process_group({backend!r}, rank {rank}/{world}, {os.environ.get('MASTER_ADDR')}:"
                              f"{os.environ.get('MASTER_PORT')}, timeout {timeout_s:.0f} s) failed: {type(e).__name__}: {e}") from e
    try:
        dev = device if backend == "nccl" else "cpu"
        probe = torch.full((1,), float(rank + 1) if rank == 0 else -1.0, dtype=torch.float32, device=dev)
        dist.broadcast(probe, src=0)
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones)
        if backend == "nccl":
            torch.cuda.synchronize()
        if float(probe.item()) != 1.0 or float(ones.item()) != float(world):
            raise RendezvousError(f"first collective returned wrong data on rank {rank}: broadcast {float(probe.item())} "
                                  f"(expected 1.0), all-reduce {float(ones.item())} (expected {world})")
    except RendezvousError:
        _drop_group()
        raise
    except Exception as e:                                   # noqa: BLE001
        _drop_group()
        raise RendezvousError(f"first collective over {backend!r} failed on rank {rank}/{world}: {type(e).__name__}: {e}") from e
    return time.perf_counter() - t0


def _drop_group() -> None:
    """A failed probe must not leave a half-working process group behind: a caller that catches RendezvousError and retries (or
    the next test) would find dist.is_initialized() True."""
    try:
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:                                        # noqa: BLE001 -- the original failure is what gets reported
        pass


def gather_floats(values, device) -> list:
    """values (a list of floats of this rank) from every rank: [[rank 0's], [rank 1's], ...]; [values] when not distributed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [list(values)]
    mine = torch.tensor(list(values), dtype=torch.float64, device=device)
    allv = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allv, mine)
    return [[float(x) for x in v] for v in allv]


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
