#!/usr/bin/env python3
"""bench.py -- aligned Mbases/s of the MI355X mapping path (BASELINE.json metric).

A "step" is one pass of the hot path (mm355_map_resident: sketch -> seed lookup -> chain -> extension -> hits) over
one batch of synthetic ONT reads that is already resident in HBM when the timed region starts.  At --gpus N>1 the
driver launches one rank per GPU (torch.distributed.run); every rank maps its OWN batch against its own replica of the
index (weak scaling, reads are independent: no data-path collective), the timed region is bracketed by barriers, the
MAX over ranks is taken and rank 0 prints ONE JSON line.

Workloads (--workload):
  ecoli   BASELINE.json configs[1]: synthetic 4.64 Mbp genome (seed 1), map-ont, reads N50 ~8 kb, 6 % error
  human   BASELINE.json configs[2]: synthetic GRCh38-scale genome (seed 3), map-ont, reads N50 ~10 kb  [needs the
          device index builder; selected automatically when available]
The JSON line also carries `roofline` (dominant kernel; algorithmic bytes of SURVEY 8d / live HIP-event time on the
launch stream) and `cpu_baseline` (the CPU oracle timed on the host cores on a bounded sample of the same reads).
"""
import argparse
import os as _os
# ROCclr multiplexes all HIP streams of a process over GPU_MAX_HW_QUEUES hardware queues (default 4); with 8 contexts in flight the
# per-read front kernels of one context queue behind the extension grids of another.  8 queues measured best (16+ lets the
# extension rounds interleave again).  Must be set before the HIP runtime starts.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def _host_pool_threads():
    """size of the library's shared host pool (MM355_HOST_THREADS) for this rank: the CPUs this process may use (cgroup quota or
    affinity mask) divided by the ranks of the node, minus the threads that sit in the HIP runtime; 16 (the library default, one
    GPU's CPU share on the bench boxes) when the rank is alone.  Over-subscribing a CPU quota stalls every thread of the cgroup."""
    local_world = int(_os.environ.get("LOCAL_WORLD_SIZE", _os.environ.get("WORLD_SIZE", "1")))
    if local_world <= 1:
        return 16
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
    return max(6, min(16, int(cpus / local_world) - 4))


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

# HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 corrections of
# MI355X_MICROARCH.md applied); measured on the default workload's sub-batch, see profiles/README.md.  None = not measured.
PMC_TRAFFIC = {"k_ksw_reg<2, false>": 6.18e9}   # default workload, 4096-read sub-batch: WRITE_SIZE 6.020 GB + 2 x FETCH_SIZE 0.078 GB
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured streaming copy)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def make_workload(name, n_reads, rank, scale=1.0):
    import synthdata as S
    if name == "ecoli":
        t0 = time.time()
        g = S.make_genome(1, [4641652], gc=0.508, repeats=((5000, 7, 0.01), (1300, 20, 0.01)))
        reads, _ = S.make_reads(rank_read_seed(rank), g, n_reads, n50=8000, sigma=0.75, lo=500, hi=100000)
        log("[bench] synthetic E. coli-like genome + %d reads in %.1fs" % (n_reads, time.time() - t0))
        return g, ["chrE"], reads, dict(workload="configs[1]: synthetic 4.64 Mbp E. coli-like genome (seed 1), map-ont k15 w10, "
                                                 "synthetic ONT reads N50~8kb 6% error (seed 2)", preset="map-ont")
    if name == "ecoli-hifi":   # BASELINE configs[4] shape (HiFi reads, map-hifi k19 w19) on the E. coli-like genome
        t0 = time.time()
        g = S.make_genome(1, [4641652], gc=0.508, repeats=((5000, 7, 0.01), (1300, 20, 0.01)))
        reads, _ = S.make_reads(rank_read_seed(rank) + 4, g, n_reads, n50=18000, sigma=0.3, lo=5000, hi=60000, sub=0.001, ins=0.0005, dele=0.0005)
        log("[bench] synthetic E. coli-like genome + %d HiFi-like reads in %.1fs" % (n_reads, time.time() - t0))
        return g, ["chrE"], reads, dict(workload="configs[4] shape: synthetic 4.64 Mbp E. coli-like genome (seed 1), map-hifi k19 w19, "
                                                 "synthetic HiFi reads N50~18kb 0.2% error (seed 6)", preset="map-hifi")
    if name == "human":
        t0 = time.time()
        g, names = S.make_human_like(3, scale, log=log)
        reads, _ = S.make_reads_codes(4 + 1000 * rank, g, n_reads, n50=10000, sigma=0.75, lo=500, hi=100000)
        tot = sum(len(c) for c in g)
        log("[bench] synthetic GRCh38-like genome (%.2f Gbp) + %d reads in %.1fs" % (tot / 1e9, n_reads, time.time() - t0))
        return g, names, reads, dict(workload="configs[2]: synthetic GRCh38-scale genome (24 contigs, %.3f Gbp, GC 41%%, SINE/LINE/satellite-like "
                                              "repeat families, seed 3, scale %g), map-ont k15 w10, synthetic ONT reads N50~10kb 6%% error (seed 4); "
                                              "index built on the device" % (tot / 1e9, scale), preset="map-ont", device_index=True)
    raise SystemExit("unknown workload " + name)


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


def rank_read_seed(rank):
    """every rank maps its own reads (weak scaling): seed 2 for rank 0, disjoint streams for the others"""
    return 2 + 1000 * rank


def cpu_baseline(fa, preset, reads, budget_s, threads):
    """oracle (CPU restatement of the minimap2 2.26 path) on host threads, bounded sample"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    orc = O.OracleAligner(fa, preset=preset)
    t0 = time.time()
    done = {"bases": 0, "aligned": 0, "n": 0}
    deadline = t0 + budget_s

    def work(rd):
        if time.time() > deadline:
            return None
        h = orc.map(rd, cs=True)
        return len(rd), (len(rd) if h else 0)

    with ThreadPoolExecutor(threads) as ex:
        for r in ex.map(work, reads):
            if r is None:
                continue
            done["bases"] += r[0]; done["aligned"] += r[1]; done["n"] += 1
    dt = time.time() - t0
    return dict(value=round(done["aligned"] / dt / 1e6, 4), unit="aligned Mbases/s", cores=threads, kind="port",
                sample="%d reads (%.2f Mbases) of the same workload in %.1f s; CPU restatement of the minimap2 2.26 path "
                       "(oracle/), scalar ksw2, one read per thread task" % (done["n"], done["bases"] / 1e6, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="ecoli")
    ap.add_argument("--reads", type=int, default=131072, help="reads per step per GPU")
    ap.add_argument("--streams", type=int, default=8, help="host threads per GPU, each driving its own contexts (HIP streams + buffers)")
    ap.add_argument("--depth", type=int, default=4, help="sub-batches each stream maps one after the other within a step")
    ap.add_argument("--scale", type=float, default=1.0, help="genome scale of the human workload")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

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

    g, names, reads, wl = make_workload(args.workload, args.reads, rank, args.scale)
    # index (host builder for now; replicated on every GPU)
    t0 = time.time()
    io, mo = _ffi.IdxOpt(), _ffi.MapOpt()
    L.mm355_set_opt(None, C.byref(io), C.byref(mo))
    _ffi.check(L.mm355_set_opt(wl["preset"].encode(), C.byref(io), C.byref(mo)))
    mo.flag |= 4
    idx = C.c_void_p()
    nm = (C.c_char_p * len(g))(*[n.encode() for n in names])
    lens = (C.c_int64 * len(g))(*[len(c) for c in g])
    if wl.get("device_index"):
        ptrs = (C.c_char_p * len(g))(*[C.cast(c.ctypes.data, C.c_char_p) for c in g])   # raw codes 0..4, no copy
        _ffi.check(L.mm355_index_build_device(C.byref(io), len(g), ptrs, lens, nm, local_rank, C.byref(idx)))
    else:
        seqs = [S.codes_to_str(c).encode() for c in g]
        arr = (C.c_char_p * len(seqs))(*seqs)
        _ffi.check(L.mm355_index_build(C.byref(io), len(seqs), arr, lens, nm, min(16, os.cpu_count() or 1), C.byref(idx)))
    L.mm355_mapopt_update(C.byref(mo), idx)
    # `--streams S`: S contexts (own HIP stream + buffers) on this GPU, each with 1/S of the step's reads resident in HBM;
    # a step maps all of them concurrently from S host threads, so the host tail of one sub-batch overlaps kernels of another.
    n_thr = max(1, args.streams)
    depth = max(1, args.depth)
    n_str = n_thr * depth                 # sub-batches per step
    # one context (own HIP streams + working buffers) per host thread; every context holds `depth` sub-batches resident in HBM
    # (mm355_batch_select) and maps them one after the other within a step.
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
    t0 = time.time()
    for si, (rarr, rlens, keep) in enumerate(packed):      # sub-batch si lives in context si % n_thr, slot si // n_thr
        _ffi.check(L.mm355_batch_select(ctxs[si % n_thr], si // n_thr))
        _ffi.check(L.mm355_batch_upload(ctxs[si % n_thr], len(keep), rarr, rlens))
    t_upload = time.time() - t0

    rlens_np = [np.asarray(pk[1], dtype=np.int64) for pk in packed]

    def step_one(si):
        ctx, (rarr, rlens, keep) = ctxs[si % n_thr], packed[si]
        tt0 = time.perf_counter()
        _ffi.check(L.mm355_batch_select(ctx, si // n_thr))
        hp = C.POINTER(_ffi.Hits)()
        _ffi.check(L.mm355_map_resident(ctx, C.byref(mo), _ffi.OUT_CS, C.byref(hp)))
        tt1 = time.perf_counter()
        h = hp.contents
        off = np.ctypeslib.as_array(h.hit_off, shape=(len(keep) + 1,))
        mapped = np.diff(off) > 0
        aligned = int(rlens_np[si][mapped].sum())
        n_hits = int(h.n_hits)
        tt2 = time.perf_counter()
        L.mm355_free_hits(hp)
        tt3 = time.perf_counter()
        st = _ffi.Stats()
        L.mm355_get_stats(ctx, C.byref(st))
        if os.environ.get("BENCH_PY_TIMES"):
            log("[py] map %.1f ms, numpy %.1f ms, free %.1f ms, stats %.1f ms" % ((tt1 - tt0) * 1e3, (tt2 - tt1) * 1e3, (tt3 - tt2) * 1e3, (time.perf_counter() - tt3) * 1e3))
        return aligned, n_hits, st

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(n_thr)

    def step_thread(ti):   # sub-batches ti, ti + n_thr, ... one after the other: their phases interleave with the other threads'
        return [step_one(si) for si in range(ti, n_str, n_thr)]

    def step():
        res = [r for part in pool.map(step_thread, range(n_thr)) for r in part]
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

    for _ in range(args.warmup):
        step()
    barrier()
    agg = {}
    t0 = time.perf_counter()
    aligned_tot = 0
    for _ in range(args.steps):
        aligned, n_hits, st = step()
        aligned_tot += aligned
        for k, _t in _ffi.Stats._fields_:
            agg[k] = agg.get(k, 0) + st[k]
    barrier()
    dt = time.perf_counter() - t0
    dt, aligned_all, bases_all = aggregate(dist, dt, aligned_tot, n_bases * args.steps)

    # PCIe-inclusive variant (never `value`): upload + map of the same reads
    def pcie_one(si):
        ctx, (rarr, rlens, keep) = ctxs[si % n_thr], packed[si]
        _ffi.check(L.mm355_batch_select(ctx, si // n_thr))
        hp = C.POINTER(_ffi.Hits)()
        _ffi.check(L.mm355_map_batch(ctx, C.byref(mo), len(keep), rarr, rlens, _ffi.OUT_CS, C.byref(hp)))
        L.mm355_free_hits(hp)
    t0 = time.perf_counter()
    list(pool.map(lambda ti: [pcie_one(si) for si in range(ti, n_str, n_thr)], range(n_thr)))
    dt_pcie = time.perf_counter() - t0

    if rank == 0:
        K = args.steps
        kern_ms = {"sketch": agg["ms_sketch"] / K, "seed_lookup": agg["ms_seed_lookup"] / K, "seed_expand": agg["ms_seed_expand"] / K,
                   "seed_select+mzflt": agg["ms_seed"] / K, "sort": agg["ms_sort"] / K, "chain": agg["ms_chain"] / K,
                   "backtrack": agg["ms_backtrack"] / K, "dp": agg["ms_dp"] / K, "host_glue": agg["ms_host"] / K}
        n_mz, n_hit, n_a, n_am = agg["n_mz"] / K, agg["n_hit"] / K, agg["n_a"] / K, agg["n_a_multi"] / K
        cells, pairs = agg["dp_cells"] / K, agg["chain_pairs"] / K
        # algorithmic bytes per launch
        seed_bytes = 16 * n_mz + 16 * n_hit + 8 * n_am + 16 * n_a
        seed_ms = kern_ms["seed_lookup"] + kern_ms["seed_expand"]
        dp_bytes = cells                                                   # 1 B/cell direction matrix written to HBM
        chain_bytes = 36 * n_a
        n_ldp = max(1.0, agg["n_launch_dp"] / K)          # extension launch groups per step (one per sub-batch and round)
        # the extension kernel that takes the most time, timed alone with HIP events on its own stream (group = 2 * size class + exact)
        gnames = ["k_ksw_reg<%d, %s>" % (np_, ex) for np_ in (1, 2, 4, 8) for ex in ("false", "true")] + \
                 ["k_ksw_extd2<512> (lds 4096, %s)" % m for m in ("approx", "exact")] + ["k_ksw_extd2<512> (lds 12288, %s)" % m for m in ("approx", "exact")] + \
                 ["k_ksw_extd2<512> (hbm state, %s)" % m for m in ("approx", "exact")] + ["-", "-"]
        gi = int(np.argmax(agg["ms_dp_group"]))
        g_ms, g_cells, g_nl = agg["ms_dp_group"][gi] / K, agg["dp_cells_group"][gi] / K, max(1.0, agg["n_launch_group"][gi] / K)
        n_lfront = float(n_str)                            # one launch of every front kernel per sub-batch
        cand = {   # name: (algorithmic bytes per step, summed kernel ms per step, launches per step, formula)
            gnames[gi]: (g_cells, g_ms, g_nl, "1 B/cell direction matrix written to HBM, %.4g cells per launch (HIP events on the kernel's own stream)" % (g_cells / g_nl)),
            "extension launch group": (dp_bytes, kern_ms["dp"], n_ldp, "1 B/cell x %.4g cells per group = all k_ksw_reg<NP,exact> / k_ksw_extd2 size classes "
                                       "on their streams + k_ksw_backtrack; one HIP-event pair around the group" % (cells / n_ldp)),
            "k_seed_lookup+k_seed_expand": (seed_bytes, seed_ms, n_lfront, "16*n_mz+16*n_hit+8*n_a_multi+16*n_a"),
            "k_chain": (chain_bytes, kern_ms["chain"], n_lfront, "16*n_a read + 20*n_a written"),
        }
        dom = gnames[gi]
        roof = {}
        for k, (b, ms, nl, how) in cand.items():
            ach = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0     # bytes per launch / average launch duration (ratio of the per-step sums)
            roof[k] = dict(bound="hbm", achieved=round(ach, 3), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5),
                           traffic=(PMC_TRAFFIC.get(k) if args.workload == "ecoli" and len(reads) // n_str == 4096 else None), kernel=k, launches_per_step=round(nl, 2), ms_per_launch=round(ms / nl, 4),
                           algorithmic_bytes=int(b / nl), formula=how)
        out = {
            "metric": "aligned Mbases/sec, synthetic ONT reads, map-ont, MI355X",
            "value": round(aligned_all / dt / 1e6, 3), "unit": "Mbases/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32 (+f32 chaining gap cost)", "data": "synthetic",
            "config": dict(workload=wl["workload"], reads_per_step_per_gpu=len(reads), streams_per_gpu=n_thr, sub_batches_per_stream=max(1, args.depth), mbases_per_step_per_gpu=round(n_bases / 1e6, 3),
                           preset=wl["preset"], parallelism="reads sharded over %d GPU(s), index replicated, no collective" % world),
            "input_mbases_per_s": round(bases_all / dt / 1e6, 3),
            "pcie_inclusive_mbases_per_s": round(n_bases / dt_pcie / 1e6, 3),
            "roofline": roof[dom], "roofline_all": roof, "kernel_ms_per_step": {k: round(v, 3) for k, v in kern_ms.items()},
            "dp_kernel_ms_per_step": {gnames[i]: round(float(agg["ms_dp_group"][i]) / K, 3) for i in range(14) if agg["n_launch_group"][i] > 0},
            "counters_per_step": dict(n_mz=int(n_mz), n_hit=int(n_hit), n_a=int(n_a), n_a_multi=int(n_am), chain_pairs=int(pairs), dp_cells=int(cells),
                                      n_dp_jobs=int(agg["n_dp_jobs"] / K)),
        }
        if wl.get("device_index") and not args.no_cpu:
            out["cpu_baseline"] = dict(value=None, unit="aligned Mbases/s", cores=0, kind="port",
                                       sample="not run: the single-threaded oracle index build of a GRCh38-scale genome does not fit the bench time budget; "
                                              "see configs[1] (default workload) for the CPU baseline")
        elif not args.no_cpu:
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                fa = os.path.join(td, "ref.fa")
                S.write_fasta(fa, g, names)
                out["cpu_baseline"] = cpu_baseline(fa, wl["preset"], reads, args.cpu_seconds, min(16, os.cpu_count() or 1))
        print(json.dumps(out), flush=True)
    barrier()
    for ctx in ctxs:
        L.mm355_ctx_destroy(ctx)
    L.mm355_index_free(idx)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
