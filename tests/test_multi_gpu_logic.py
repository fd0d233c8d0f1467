"""N>1 path on CPU (gloo, world_size 2): the benchmark's only communication is the timing barrier + MAX/SUM of the
timing; reads are sharded by rank with disjoint seeds and no data-path collective exists."""
import os
import socket

import pytest


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import synthdata as S
    g = S.make_genome(1, [20000], repeats=())
    reads, _ = S.make_reads(bench.rank_read_seed(rank), g, 4, n50=1000, lo=200)
    nb = sum(len(r) for r in reads)
    dist.barrier()
    dt, aligned, bases = bench.aggregate(dist, 1.0 + rank, nb, nb)   # rank 1 is "slower"
    q.put((rank, dt, aligned, bases, nb, reads[0][:50]))
    dist.destroy_process_group()


def test_aggregate_two_ranks_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps: p.join(60)
    (r0, dt0, a0, b0, nb0, h0), (r1, dt1, a1, b1, nb1, h1) = res
    assert dt0 == dt1 == 2.0                       # MAX over ranks
    assert a0 == a1 == nb0 + nb1 and b0 == b1      # whole-job sums
    assert h0 != h1                                # ranks map different reads (weak scaling)


def test_single_rank_passthrough():
    import bench
    assert bench.aggregate(None, 0.5, 10, 20) == (0.5, 10.0, 20.0)
