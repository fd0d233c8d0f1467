"""N>1 path on CPU (gloo, world_size 2).  The path shards by reads with no exchange step: ONE read set is cut into
contiguous shards balanced by cumulative bases, every rank maps its own shard, and the only communication is the timing barrier +
MAX/SUM of the timing.  Two gloo ranks shard the same read set; the union of their results equals the single-rank result and the
shards are disjoint.  (The mapping engine of this CPU test is the oracle: the HIP path needs a GPU.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

SEED, N_TOTAL, READ_KW = 4, 96, dict(n50=1500, sigma=0.75, lo=300, hi=20000)


def _genome():
    import synthdata as S
    return S.make_genome(1, [60000], repeats=())


def _map_all(reads, g):
    from oracle import oracle as O
    orc = O.OracleAligner(codes=g, names=["chrT"], preset="map-ont", n_threads=2)
    return [[(h["target_start"], h["target_end"], h["strand"], h["cigar_str"], h["mapq"]) for h in orc.map(r, cs=True)] for r in reads]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import synthdata as S
    from mappy_rs import shard_by_bases
    g = _genome()
    wl = dict(seed=SEED, reads=READ_KW)
    reads = bench.shard_reads(wl, g, N_TOTAL // world, rank, world)          # the function bench.py shards with
    lens = S.read_set_lengths(SEED, N_TOTAL, **READ_KW)
    b = shard_by_bases(lens, world)
    res = _map_all(reads, g)
    nb = sum(len(r) for r in reads)
    dist.barrier()
    dt, aligned, bases = bench.aggregate(dist, 1.0 + rank, nb, nb)   # rank 1 is "slower"
    q.put((rank, dt, aligned, bases, nb, (b[rank], b[rank + 1]), res))
    dist.destroy_process_group()


def test_two_ranks_shard_one_read_set_gloo(built):
    import torch.multiprocessing as mp
    import synthdata as S
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in ps: p.join(60)
    (r0, dt0, a0, b0, nb0, span0, m0), (r1, dt1, a1, b1, nb1, span1, m1) = res
    assert dt0 == dt1 == 2.0                       # MAX over ranks
    assert a0 == a1 == nb0 + nb1 and b0 == b1      # whole-job sums
    # the shards are contiguous, disjoint and cover the read set
    assert span0[0] == 0 and span0[1] == span1[0] and span1[1] == N_TOTAL and span0[1] > 0 and span1[1] > span1[0]
    # balanced by cumulative bases, not by read count: the heavier side is within one read of half the bases
    g = _genome()
    whole = S.read_set_slice(SEED, 0, N_TOTAL, g, **READ_KW)
    assert abs(nb0 - nb1) <= max(len(r) for r in whole) * 2
    # union of the per-rank results == the single-rank result, in order
    assert m0 + m1 == _map_all(whole, g)
    assert sum(1 for m in m0 + m1 if m) > N_TOTAL // 2


def test_shard_by_bases_properties():
    from mappy_rs import shard_by_bases
    rng = np.random.default_rng(7)
    for n, w in ((0, 3), (1, 4), (5, 8), (1000, 8), (4096, 2), (333, 7)):
        lens = rng.integers(1, 50000, n)
        b = shard_by_bases(lens, w)
        assert len(b) == w + 1 and b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(w))
        if n >= 100 * w:
            tot = [int(lens[b[i]:b[i + 1]].sum()) for i in range(w)]
            assert max(tot) - min(tot) <= 2 * int(lens.max())
    assert shard_by_bases([10, 20, 30, 40, 50, 60], 3) == [0, 3, 5, 6]


def test_single_rank_passthrough():
    import bench
    assert bench.aggregate(None, 0.5, 10, 20) == (0.5, 10.0, 20.0)


def test_bench_gpus_flag_must_match_world_size():
    """`--gpus N` is honoured: under a launcher it has to agree with WORLD_SIZE (without one bench.py spawns the N ranks itself)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in p.stderr
