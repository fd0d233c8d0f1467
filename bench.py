#!/usr/bin/env python3
"""bench.py -- aligned Mbases/s of the MI355X mapping path (BASELINE.json metric: ONT reads vs GRCh38, map-ont).

A "step" is one pass of the hot path (sketch -> seed lookup -> chain -> extension -> hits) over one batch of synthetic ONT reads.
Default workload = BASELINE.json configs[2]: synthetic GRCh38-scale genome (3.09 Gbp, index built on the device), map-ont, reads
N50 ~10 kb.  `value` is the PCIe-inclusive rate (H2D of the reads and D2H of the results inside the timed region, index
upload outside); the rate with the reads already resident in HBM is reported beside it (`resident_mbases_per_s`).

Multi-GPU (--gpus N): ONE read set of N x --reads reads is cut into N contiguous shards balanced by cumulative bases;
rank r maps shard r against its own replica of the index; no data-path collective.  The driver launches the ranks with
torch.distributed.run; started by hand with WORLD_SIZE unset, `bench.py --gpus N` spawns its N ranks itself (before anything
touches the GPU).  The timed region is bracketed by a device synchronisation + barrier on both sides, the MAX over ranks is taken
and rank 0 prints ONE JSON line.

Workloads (--workload):
  human        configs[2] (default)     synthetic GRCh38-scale genome (seed 3), map-ont, read set seed 4
  human-hifi   configs[4]               same genome, map-hifi k19 w19, HiFi-like reads N(18 kb, 2.5 kb), 0.2 % error, read set seed 6
  ecoli        configs[1]               synthetic 4.64 Mbp genome (seed 1), map-ont, reads N50 ~8 kb, read set seed 2
  ecoli-hifi   configs[4] shape on the small genome
The JSON line also carries `roofline` (dominant kernel; algorithmic bytes of SURVEY 8d / live HIP-event time on the launch stream)
and `cpu_baseline` (the CPU oracle timed on the host cores on a bounded sample of the same reads, index built with host threads).
"""
import argparse
import os as _os
import sys as _sys

# ROCclr multiplexes all HIP streams of a process over GPU_MAX_HW_QUEUES hardware queues (default 4); with 8 contexts in flight the
# per-read front kernels of one context queue behind the extension grids of another.  8 queues measured best (16+ lets the
# extension rounds interleave again).  Must be set before the HIP runtime starts.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def _parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="human")
    ap.add_argument("--reads", type=int, default=0, help="reads per step per GPU (0 = workload default)")
    ap.add_argument("--streams", type=int, default=0, help="host threads per GPU, each driving its own context (HIP streams + buffers)")
    ap.add_argument("--depth", type=int, default=0, help="sub-batches each stream maps one after the other within a step")
    ap.add_argument("--scale", type=float, default=1.0, help="genome scale of the human workloads")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-resident", action="store_true", help="skip the second timed region (reads already resident in HBM)")
    return ap.parse_args()


def _spawn_ranks(args):
    """`bench.py --gpus N` without a launcher: start the N ranks as child processes (nothing in this process has touched the GPU),
    relay rank 0's JSON line, exit with the worst child status."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(_os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([_sys.executable, _os.path.abspath(__file__)] + _sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    _sys.exit(rc)


_ARGS = _parse_args() if __name__ == "__main__" else None
if _ARGS is not None:
    _ws = _os.environ.get("WORLD_SIZE")
    if _ws is None and _ARGS.gpus > 1:
        _spawn_ranks(_ARGS)
    if _ws is not None and int(_ws) != _ARGS.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s; launch with `python -m torch.distributed.run --nproc-per-node %d ... "
                         "bench.py --gpus %d`, or unset WORLD_SIZE and let bench.py spawn its ranks" % (_ARGS.gpus, _ws, _ARGS.gpus, _ARGS.gpus))


def _host_pool_threads():
    """size of the library's shared host pool (MM355_HOST_THREADS) for this rank: the CPUs this process may use (cgroup quota or
    affinity mask) divided by the ranks of the node, minus two for the context threads and the HIP runtime (the library's own default
    does the same for a single rank).  The path is host-bound: fewer threads starve the GPU, over-subscribing a CPU quota stalls every
    thread of the cgroup (16-CPU quota, one rank: 10 threads 790, 14: 860, 16: 845, 20: 747 Mbases/s)."""
    local_world = int(_os.environ.get("LOCAL_WORLD_SIZE", _os.environ.get("WORLD_SIZE", "1")))
    try:
        cpus = float(len(_os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        cpus = float(_os.cpu_count() or 16)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cpus = min(cpus, float(q) / float(per))
    except (OSError, ValueError):
        pass
    return max(4, min(32, int(cpus / max(1, local_world)) - 2))


_os.environ.setdefault("MM355_HOST_THREADS", str(_host_pool_threads()))
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mappy-rs_amd"))

import numpy as np

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured streaming copy)
HBM_COPY_GBS = 6290.0
METRIC = "aligned Mbases/sec, ONT reads vs GRCh38 map-ont, 1/2/4/8 MI355X"     # BASELINE.json "metric", verbatim


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def pmc_traffic(workload, reads_per_sub, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this workload shape (profiles/pmc_traffic.json:
    FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 corrections of MI355X_MICROARCH.md applied); None when that shape was not profiled"""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return None
    ent = tab.get("%s/%d" % (workload, reads_per_sub), {})
    return ent.get(kernel)


WORKLOADS = {
    # name: (genome, preset, read-set seed, read model, default reads/step/GPU, streams, depth, BASELINE config text)
    "human": dict(genome="human", preset="map-ont", seed=4, reads=dict(n50=10000, sigma=0.75, lo=500, hi=100000),
                  n_reads=73728, streams=6, depth=2, cfg="configs[2]", what="synthetic ONT reads N50~10kb 6% error (read set seed 4)"),
    "human-hifi": dict(genome="human", preset="map-hifi", seed=6, reads=dict(n50=18000, sigma=0.14, lo=5000, hi=60000, sub=0.0005, ins=0.00075, dele=0.00075),
                       n_reads=24576, streams=6, depth=2, cfg="configs[4]", what="synthetic HiFi reads ~N(18kb, 2.5kb) 0.2% error (read set seed 6)"),
    "ecoli": dict(genome="ecoli", preset="map-ont", seed=2, reads=dict(n50=8000, sigma=0.75, lo=500, hi=100000),
                  n_reads=131072, streams=8, depth=4, cfg="configs[1]", what="synthetic ONT reads N50~8kb 6% error (read set seed 2)"),
    "ecoli-hifi": dict(genome="ecoli", preset="map-hifi", seed=6, reads=dict(n50=18000, sigma=0.14, lo=5000, hi=60000, sub=0.0005, ins=0.00075, dele=0.00075),
                       n_reads=16384, streams=8, depth=2, cfg="configs[4] shape on the configs[1] genome", what="synthetic HiFi reads ~N(18kb, 2.5kb) 0.2% error (read set seed 6)"),
}


def make_genome(kind, scale):
    import synthdata as S
    t0 = time.time()
    if kind == "ecoli":
        g = S.make_genome(1, [4641652], gc=0.508, repeats=((5000, 7, 0.01), (1300, 20, 0.01)))
        log("[bench] synthetic E. coli-like genome in %.1fs" % (time.time() - t0))
        return g, ["chrE"], "synthetic 4.64 Mbp E. coli-like genome (seed 1)", False
    g, names = S.make_human_like(3, scale, log=log)
    tot = sum(len(c) for c in g)
    log("[bench] synthetic GRCh38-like genome (%.2f Gbp) in %.1fs" % (tot / 1e9, time.time() - t0))
    return g, names, "synthetic GRCh38-scale genome (24 contigs, %.3f Gbp, GC 41%%, SINE/LINE/satellite-like repeat families, seed 3, scale %g), " \
                     "index built on the device" % (tot / 1e9, scale), True


def shard_reads(wl, g, n_per_gpu, rank, world):
    """rank's shard of the ONE read set of world * n_per_gpu reads: contiguous, balanced by cumulative bases"""
    import synthdata as S
    from mappy_rs import shard_by_bases
    t0 = time.time()
    total = n_per_gpu * world
    lens = S.read_set_lengths(wl["seed"], total, **{k: v for k, v in wl["reads"].items() if k in ("n50", "sigma", "lo", "hi")})
    b = shard_by_bases(lens, world)
    reads = S.read_set_slice(wl["seed"], b[rank], b[rank + 1], g, **wl["reads"])
    log("[bench] read set: %d reads, rank %d maps reads [%d, %d) (%.1f Mbases) -- synthesised in %.1fs"
        % (total, rank, b[rank], b[rank + 1], sum(len(r) for r in reads) / 1e6, time.time() - t0))
    return reads


def aggregate(dist, dt, aligned, bases):
    """max-over-ranks time and whole-job sums; the only communication of the benchmark (never on the data path)"""
    if dist is None:
        return dt, float(aligned), float(bases)
    import torch
    dev = "cuda" if (torch.cuda.is_available() and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    a = torch.tensor([aligned, bases], dtype=torch.float64, device=dev)
    dist.all_reduce(a, op=dist.ReduceOp.SUM)
    return float(t.item()), float(a[0].item()), float(a[1].item())


SIG_FIELDS = ("target_name", "target_start", "target_end", "query_start", "query_end", "strand", "mapq", "is_primary", "NM", "match_len", "block_len", "cigar_str", "cs")


def cpu_baseline(g, names, preset, reads, budget_s, threads, gpu_sigs=None):
    """oracle (CPU restatement of the minimap2 2.26 path) on host threads, bounded sample of the same reads; the index is built from
    the same contigs with the oracle's threaded builder (not timed: the metric excludes index construction on both sides).
    gpu_sigs[i]: hash of the hit records the HIP path returned for read i (taken before the timed numbers were printed, outside the timed
    region): every read of the sample is also a full-size parity check of the run itself -- the oracle as the checker."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    tb = time.time()
    orc = O.OracleAligner(codes=g, names=names, preset=preset, n_threads=threads)
    t_build = time.time() - tb
    log("[bench] cpu_baseline: oracle index built on %d threads in %.1fs" % (threads, t_build))
    t0 = time.time()
    done = {"bases": 0, "aligned": 0, "n": 0}
    deadline = t0 + budget_s

    def work(rd):
        if time.time() > deadline:
            return None
        h = orc.map(rd, cs=True)
        return len(rd), (len(rd) if h else 0), hash(tuple(tuple(x[k] for k in SIG_FIELDS) for x in h)), len(h)

    par = {"reads": 0, "hits": 0, "mismatching_reads": 0}
    with ThreadPoolExecutor(threads) as ex:
        for i, r in enumerate(ex.map(work, reads)):
            if r is None:
                continue
            done["bases"] += r[0]; done["aligned"] += r[1]; done["n"] += 1
            if gpu_sigs is not None and i < len(gpu_sigs):
                par["reads"] += 1; par["hits"] += r[3]; par["mismatching_reads"] += int(gpu_sigs[i] != r[2])
    dt = time.time() - t0
    if gpu_sigs is not None:
        log("[bench] parity of the run itself: %d reads (%d hits) of the CPU sample compared with the HIP path's records, %d mismatching" %
            (par["reads"], par["hits"], par["mismatching_reads"]))
    return dict(parity=par if gpu_sigs is not None else None, value=round(done["aligned"] / dt / 1e6, 4), unit="aligned Mbases/s", cores=threads, kind="port",
                sample="the first %d reads (%.2f Mbases) of rank 0's shard of the same workload in %.1f s; CPU restatement of the minimap2 2.26 "
                       "path (oracle/), scalar ksw2, one read per thread task; oracle index of the same genome built on %d threads in %.0f s "
                       "(not timed)" % (done["n"], done["bases"] / 1e6, dt, threads, t_build))


def main():
    args = _ARGS
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_
        dist = dist_
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl" if torch.cuda.is_available() else "gloo")

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()
    from mappy_rs import _ffi
    import synthdata as S
    L = _ffi.lib()
    if L.mm355_device_count() <= local_rank:
        raise SystemExit("bench.py needs an MI355X: libmm355 has no CPU fallback (devices visible: %d)" % L.mm355_device_count())
    if args.workload not in WORKLOADS:
        raise SystemExit("unknown workload " + args.workload)
    wl = WORKLOADS[args.workload]
    n_per_gpu = args.reads or wl["n_reads"]
    n_thr = max(1, args.streams or wl["streams"])
    depth = max(1, args.depth or wl["depth"])

    g, names, gdesc, device_index = make_genome(wl["genome"], args.scale)
    reads = shard_reads(wl, g, n_per_gpu, rank, world)
    # index (replicated on every GPU; built on the device for the GRCh38-scale genome)
    t0 = time.time()
    io, mo = _ffi.IdxOpt(), _ffi.MapOpt()
    L.mm355_set_opt(None, C.byref(io), C.byref(mo))
    _ffi.check(L.mm355_set_opt(wl["preset"].encode(), C.byref(io), C.byref(mo)))
    mo.flag |= 4
    idx = C.c_void_p()
    nm = (C.c_char_p * len(g))(*[n.encode() for n in names])
    lens = (C.c_int64 * len(g))(*[len(c) for c in g])
    if device_index:
        ptrs = (C.c_char_p * len(g))(*[C.cast(c.ctypes.data, C.c_char_p) for c in g])   # raw codes 0..4, no copy
        _ffi.check(L.mm355_index_build_device(C.byref(io), len(g), ptrs, lens, nm, local_rank, C.byref(idx)))
    else:
        seqs = [S.codes_to_str(c).encode() for c in g]
        arr = (C.c_char_p * len(seqs))(*seqs)
        _ffi.check(L.mm355_index_build(C.byref(io), len(seqs), arr, lens, nm, min(16, os.cpu_count() or 1), C.byref(idx)))
    L.mm355_mapopt_update(C.byref(mo), idx)
    # one context (own HIP streams + working buffers) per host thread; every context holds `depth` sub-batches (mm355_batch_select)
    # and maps them one after the other within a step, so that the host tail of one sub-batch overlaps kernels of another.
    n_str = n_thr * depth                 # sub-batches per step
    ctxs, parts = [], []
    for ti in range(n_thr):
        ctx = C.c_void_p()
        _ffi.check(L.mm355_ctx_create(idx, local_rank, C.byref(ctx)))
        ctxs.append(ctx)
    for si in range(n_str):
        parts.append(reads[si::n_str])
    log("[bench] index built + uploaded in %.1fs (mid_occ=%d)" % (time.time() - t0, mo.mid_occ))

    packed = [_ffi.pack_reads(p) for p in parts]
    n_bases = sum(len(b) for pk in packed for b in pk[2])
    rlens_np = [np.asarray(pk[1], dtype=np.int64) for pk in packed]

    def step_one(si, resident):
        ctx, (rarr, rlens, keep) = ctxs[si % n_thr], packed[si]       # sub-batch si lives in context si % n_thr, slot si // n_thr
        _ffi.check(L.mm355_batch_select(ctx, si // n_thr))
        hp = C.POINTER(_ffi.Hits)()
        if resident:
            _ffi.check(L.mm355_map_resident(ctx, C.byref(mo), _ffi.OUT_CS, C.byref(hp)))
        else:   # the drop-in call: host buffers in (H2D), hit records out (D2H)
            _ffi.check(L.mm355_map_batch(ctx, C.byref(mo), len(keep), rarr, rlens, _ffi.OUT_CS, C.byref(hp)))
        h = hp.contents
        off = np.ctypeslib.as_array(h.hit_off, shape=(len(keep) + 1,))
        mapped = np.diff(off) > 0
        aligned = int(rlens_np[si][mapped].sum())
        n_hits = int(h.n_hits)
        L.mm355_free_hits(hp)
        st = _ffi.Stats()
        L.mm355_get_stats(ctx, C.byref(st))
        return aligned, n_hits, st

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(n_thr)

    def run_steps(resident, steps):
        """`steps` passes over the batch: every host thread maps its own sub-batches `steps` times, one after the other, without waiting
        for the other threads between passes (the passes of different threads overlap; the barriers bracket the whole region, not a pass)"""
        def thread(ti):
            return [step_one(si, resident) for _ in range(steps) for si in range(ti, n_str, n_thr)]
        res = [r for part in pool.map(thread, range(n_thr)) for r in part]
        agg_st = {}
        for _a, _h, st in res:
            for k, _t in _ffi.Stats._fields_:
                v = getattr(st, k)
                v = np.array(list(v), dtype=np.float64) if hasattr(v, "__len__") else v
                agg_st[k] = agg_st.get(k, 0) + v
        return sum(r[0] for r in res), sum(r[1] for r in res), agg_st

    def barrier():
        # both sides of the timed region: every stream of this rank's GPU drained (the library runs on its own HIP streams, which
        # torch.cuda.synchronize() would not see unless it synchronises the device -- hipDeviceSynchronize does), then the ranks meet
        _ffi.check(L.mm355_device_synchronize(local_rank))
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def timed(resident, steps):
        barrier()
        t0 = time.perf_counter()
        aligned_tot, n_hits, agg = run_steps(resident, steps)
        barrier()
        dt = time.perf_counter() - t0
        dt, aligned_all, bases_all = aggregate(dist, dt, aligned_tot, n_bases * steps)
        return dt, aligned_all, bases_all, agg

    if args.warmup:
        run_steps(False, args.warmup)
    K = args.steps
    dt, aligned_all, bases_all, agg = timed(False, K)           # `value`: H2D + map + D2H
    if args.no_resident:
        dt_res, aligned_res = None, None
    else:
        dt_res, aligned_res, _b, _agg = timed(True, K)          # reads already resident in HBM (uploaded by the steps above)

    if rank == 0:
        kern_ms = {"sketch": agg["ms_sketch"] / K, "seed_lookup": agg["ms_seed_lookup"] / K, "seed_expand": agg["ms_seed_expand"] / K,
                   "seed_select+mzflt": agg["ms_seed"] / K, "sort": agg["ms_sort"] / K, "chain": agg["ms_chain"] / K,
                   "backtrack": agg["ms_backtrack"] / K, "dp": agg["ms_dp"] / K, "host_glue": agg["ms_host"] / K}
        n_mz, n_hit, n_a, n_am = agg["n_mz"] / K, agg["n_hit"] / K, agg["n_a"] / K, agg["n_a_multi"] / K
        cells, pairs = agg["dp_cells"] / K, agg["chain_pairs"] / K
        # algorithmic bytes per launch
        lookup_bytes = 16 * n_mz + 16 * n_hit
        expand_bytes = 8 * n_am + 16 * n_a
        n_ldp = max(1.0, agg["n_launch_dp"] / K)          # extension launch groups per step (one per sub-batch and round)
        # the extension kernel that takes the most time, timed alone with HIP events on its own stream (group = 2 * size class + exact)
        # (the long-target classes are two launches, approx and exact alignments together: targets <= 4096 timed as group 8, longer ones as 10)
        gnames = ["k_ksw_reg<%d, %s>" % (np_, ex) for np_ in (1, 2, 4, 8) for ex in ("false", "true")] + \
                 ["k_ksw_extd2<512> (targets 1025..4096)", "-", "k_ksw_extd2<512> (targets > 4096)", "-", "-", "-", "k_ksw_row<2>", "k_ksw_row<4>", "k_ksw_row<8>", "k_ksw_rowl (targets 1025..8192)", "k_ksw_regw (exact, band <= 832, targets > 1024)"]
        cells_g = np.array(agg["dp_cells_group"], dtype=np.float64); cells_g[8] = cells_g[8:10].sum(); cells_g[10] = cells_g[10:14].sum(); cells_g[9] = 0; cells_g[11:14] = 0
        nl_g = np.array(agg["n_launch_group"], dtype=np.float64); nl_g[8] = nl_g[8:10].max(); nl_g[10] = nl_g[10:14].max(); nl_g[9] = 0; nl_g[11:14] = 0
        gi = int(np.argmax(agg["ms_dp_group"]))
        g_ms, g_cells, g_nl = agg["ms_dp_group"][gi] / K, cells_g[gi] / K, max(1.0, nl_g[gi] / K)
        n_lfront = float(n_str)                            # one launch of every front kernel per sub-batch
        cand = {   # name: (algorithmic bytes per step, summed kernel ms per step, launches per step, formula)
            gnames[gi]: (g_cells, g_ms, g_nl, "1 B/cell direction matrix written to HBM, %.4g cells per launch (HIP events on the kernel's own stream)" % (g_cells / g_nl)),
        }
        for i in range(19):   # every extension kernel that ran, each timed alone on its own stream (the dominant one is also `roofline`)
            if i != gi and nl_g[i] > 0 and agg["ms_dp_group"][i] > 0:
                cand[gnames[i]] = (cells_g[i] / K, agg["ms_dp_group"][i] / K, max(1.0, nl_g[i] / K),
                                   "1 B/cell direction matrix written to HBM, %.4g cells per launch (HIP events on the kernel's own stream)" % (cells_g[i] / max(1.0, nl_g[i])))
        cand.update({
            "extension launch group": (cells, kern_ms["dp"], n_ldp, "1 B/cell x %.4g cells per group = all k_ksw_reg<NP,exact> / k_ksw_extd2 size classes "
                                       "on their streams + k_ksw_backtrack; one HIP-event pair around the group" % (cells / n_ldp)),
            "k_seed_lookup": (lookup_bytes, kern_ms["seed_lookup"], n_lfront, "16*n_mz + 16*n_hit (minimizer read + one table slot)"),
            "k_seed_expand": (expand_bytes, kern_ms["seed_expand"], n_lfront, "8*n_a_multi + 16*n_a (pos[] entry read + anchor written)"),
            "anchor sort": (32 * n_a, kern_ms["sort"], n_lfront, "2*16*n_a (one read + one write of every anchor; the radix passes actually needed are not counted)"),
            "k_chain": (36 * n_a, kern_ms["chain"], n_lfront, "16*n_a read + 20*n_a written"),
        })
        # dominant kernel = the extension kernel with the largest summed duration (each is timed alone with HIP events on the stream it is launched
        # on; the rocprofv3 kernel statistics of the same command, profiles/, name the same kernel at the top).  The front stages are timed as
        # event spans on the context's main stream, where kernels of other contexts interleave: they stay in roofline_all.
        dom = gnames[gi]
        reads_per_sub = len(reads) // n_str
        roof = {}
        for k, (b, ms, nl, how) in cand.items():
            ach = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0     # bytes per launch / average launch duration (ratio of the per-step sums)
            roof[k] = dict(bound="hbm", achieved=round(ach, 3), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5),
                           frac_of_measured_copy=round(ach / HBM_COPY_GBS, 5), traffic=pmc_traffic(args.workload, reads_per_sub, k), kernel=k,
                           launches_per_step=round(nl, 2), ms_per_launch=round(ms / nl, 4), algorithmic_bytes=int(b / nl), formula=how)
        out = {
            "metric": METRIC,
            "value": round(aligned_all / dt / 1e6, 3), "unit": "Mbases/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32 (+f32 chaining gap cost)", "data": "synthetic",
            "config": dict(workload="%s: %s, %s %s, %s" % (wl["cfg"], gdesc, wl["preset"], "k%d w%d" % (io.k, io.w), wl["what"]),
                           reads_per_step_per_gpu=len(reads), read_set="%d reads, %d contiguous shard(s) balanced by cumulative bases" % (n_per_gpu * world, world),
                           streams_per_gpu=n_thr, sub_batches_per_stream=depth, mbases_per_step_per_gpu=round(n_bases / 1e6, 3),
                           preset=wl["preset"], parallelism="reads sharded over %d GPU(s), index replicated, no collective" % world),
            "value_basis": "PCIe-inclusive (SURVEY 8d): reads handed over as host buffers (H2D), hit records returned to the host (D2H), index resident",
            "input_mbases_per_s": round(bases_all / dt / 1e6, 3),
            "resident_mbases_per_s": None if dt_res is None else round(aligned_res / dt_res / 1e6, 3),
            "roofline": roof[dom], "roofline_all": roof, "kernel_ms_per_step": {k: round(v, 3) for k, v in kern_ms.items()},
            "dp_kernel_ms_per_step": {gnames[i]: round(float(agg["ms_dp_group"][i]) / K, 3) for i in range(19) if nl_g[i] > 0},
            "dp_cells_per_step": {gnames[i]: int(cells_g[i] / K) for i in range(19) if nl_g[i] > 0},
            "counters_per_step": dict(n_mz=int(n_mz), n_hit=int(n_hit), n_a=int(n_a), n_a_multi=int(n_am), chain_pairs=int(pairs), dp_cells=int(cells),
                                      n_dp_jobs=int(agg["n_dp_jobs"] / K)),
        }
    barrier()
    gpu_sigs = None
    if rank == 0 and not args.no_cpu and world == 1:   # (outside the timed region) the records of the reads the CPU sample will cover
        import mappy_rs
        gpu_sigs = []
        n_chk = len(reads) if args.cpu_seconds >= 60 else min(len(reads), 4 * 6144)   # a long CPU leg compares the whole read set
        for lo in range(0, n_chk, 6144):
            sub = reads[lo:min(n_chk, lo + 6144)]
            rarr, rlens, keep = _ffi.pack_reads(sub)
            hp = C.POINTER(_ffi.Hits)()
            _ffi.check(L.mm355_map_batch(ctxs[0], C.byref(mo), len(sub), rarr, rlens, _ffi.OUT_CS, C.byref(hp)))
            for ms in mappy_rs._batch_to_mappings(hp, len(sub), names):
                gpu_sigs.append(None if isinstance(ms, Exception) else hash(tuple(tuple(getattr(m, k) for k in SIG_FIELDS) for m in ms)))
            L.mm355_free_hits(hp)
    for ctx in ctxs:
        L.mm355_ctx_destroy(ctx)
    L.mm355_index_free(idx)
    if rank == 0:
        if not args.no_cpu and world == 1:   # after the GPU side has released its memory; rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(g, names, wl["preset"], reads, args.cpu_seconds, min(16, os.cpu_count() or 1), gpu_sigs)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
