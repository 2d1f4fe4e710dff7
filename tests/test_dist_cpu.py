"""CPU, world_size 2 over gloo: the N>1 plumbing of bench.py (single flat-buffer weight broadcast from rank 0,
request sharding, max-over-ranks timing).  The denoise step itself has no collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from this_and_that_vdm_amd.dist import broadcast_model_, flat_param_buffer, max_over_ranks, shard_requests


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from this_and_that_vdm_amd.svd.unet_spatio_temporal_condition import UNetSpatioTemporalConditionModel
    from this_and_that_vdm_amd.utils.synthetic import fill_parameters_
    torch.manual_seed(100 + rank)                       # different random init per rank
    m = UNetSpatioTemporalConditionModel(block_out_channels=(64, 64, 64, 64), num_attention_heads=(1, 1, 1, 1),
                                         cross_attention_dim=32, num_frames=2)
    flat = flat_param_buffer(m)
    assert all(p.data_ptr() >= flat.data_ptr() for p in m.parameters())
    if rank == 0:
        fill_parameters_(m, "unet.")
    secs = broadcast_model_(m, src=0, flat=flat)
    ref = UNetSpatioTemporalConditionModel(block_out_channels=(64, 64, 64, 64), num_attention_heads=(1, 1, 1, 1),
                                           cross_attention_dim=32, num_frames=2)
    fill_parameters_(ref, "unet.")
    same = all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), ref.state_dict().values()))
    mx = max_over_ranks(1.0 + rank, "cpu")
    q.put((rank, same, secs >= 0.0, mx, shard_requests(5, rank, world)))
    dist.destroy_process_group()


def test_weight_broadcast_sharding_and_max_timing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True], "every rank must hold rank 0's weights after the broadcast"
    assert all(r[3] == 2.0 for r in res)
    assert res[0][4] == [0, 2, 4] and res[1][4] == [1, 3]


def test_single_process_is_a_no_op():
    m = torch.nn.Linear(4, 4)
    assert broadcast_model_(m) == 0.0 and max_over_ranks(3.0, "cpu") == 3.0


def _init_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from this_and_that_vdm_amd.dist import gather_floats, init_ranks
    secs = init_ranks("gloo", rank, world, rank, timeout_s=60)
    got = gather_floats([float(rank), 10.0 + rank], "cpu")
    q.put((rank, secs > 0.0, got))
    dist.destroy_process_group()


def test_fail_loud_rendezvous_probe_and_gather():
    """init_ranks: process group with a finite timeout + 1-element broadcast / all-reduce probe; gather_floats: per-rank values."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_init_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert res[0][2] == res[1][2] == [[0.0, 10.0], [1.0, 11.0]]


def test_rendezvous_rejects_bad_mappings_before_touching_the_process_group(monkeypatch):
    import pytest
    from this_and_that_vdm_amd.dist import RendezvousError, gather_floats, init_ranks
    with pytest.raises(RendezvousError, match="outside world size"):
        init_ranks("gloo", 2, 2, 0)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    with pytest.raises(RendezvousError, match="MASTER_PORT"):
        init_ranks("gloo", 0, 2, 0)
    if not torch.cuda.is_available():                       # the RCCL branch refuses a box without GPUs by name
        with pytest.raises(RendezvousError, match="needs GPUs"):
            init_ranks("nccl", 0, 2, 0)
    assert init_ranks("gloo", 0, 1, 0) == 0.0 and not dist.is_initialized()
    assert gather_floats([1.5, 2.5], "cpu") == [[1.5, 2.5]]


def test_rendezvous_times_out_with_the_cause_in_the_message(monkeypatch):
    """A rank whose peer never arrives: init_process_group gives up after the timeout and the error names backend, rank and address."""
    import pytest
    from this_and_that_vdm_amd.dist import RendezvousError, init_ranks
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    with pytest.raises(RendezvousError, match=r"init_process_group\('gloo', rank 1/2"):
        init_ranks("gloo", 1, 2, 1, timeout_s=3)            # rank 1 without a rank 0: nobody hosts the store
    assert not dist.is_initialized()
