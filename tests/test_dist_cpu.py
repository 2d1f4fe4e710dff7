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
