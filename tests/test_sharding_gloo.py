"""CPU, world_size 2 over gloo: the N>1 path of the engine is a static partition + barrier + max-over-ranks time."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_samples, ens, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from dg_tta_amd.sharding import max_over_ranks, rank_world, units_for_rank
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert rank_world() == (rank, world)
    mine = units_for_rank(num_samples, ens, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    dist.barrier()
    t = max_over_ranks(1.0 + rank)                     # rank 1 is "slower"
    if rank == 0:
        out.put((gathered, t))
    dist.destroy_process_group()


def test_partition_is_disjoint_complete_and_time_is_max():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    flat = [u for part in gathered for u in part]
    assert sorted(flat) == [(s, e) for s in range(5) for e in range(3)]          # complete
    assert len(set(flat)) == len(flat)                                          # disjoint
    assert {s for s, _ in gathered[0]} == {0, 2, 4} and {s for s, _ in gathered[1]} == {1, 3}
    assert t == 2.0


def test_single_process_defaults():
    from dg_tta_amd.sharding import max_over_ranks, units_for_rank
    assert units_for_rank(2, 2, 0, 1) == [(0, 0), (0, 1), (1, 0), (1, 1)]
    assert max_over_ranks(0.5) == 0.5
