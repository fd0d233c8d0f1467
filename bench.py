#!/usr/bin/env python3
"""bench.py -- aligned Mbases/s of the MI355X mapping path (BASELINE.json metric: ONT reads vs GRCh38, map-ont).

A "step" is one pass of the hot path (sketch -> seed lookup -> chain -> extension -> hits) over one block of synthetic ONT reads.
Default workload = BASELINE.json configs[2]: synthetic GRCh38-scale genome (3.09 Gbp, index built on the device), map-ont, 1 M reads
N50 ~10 kb: the read set is cut into steps + warmup DISTINCT blocks of 73 728 reads (14 + 1 by default: 1 032 192 reads in the timed
region, no read mapped twice in it; the warm-up block is a block of its own).  `value` is the PCIe-inclusive rate of the drop-in call
over ALL timed blocks (H2D of the reads and D2H of the hit records inside the timed region): every sub-batch is handed to
mm355_map_batch as host buffers, in input order (contiguous chunks, the way mappy_rs.map_batch cuts them).  The rate with the reads
already resident in HBM (mm355_map_resident) is timed afterwards on a few of the same blocks: `resident_mbases_per_s`.

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
    ap.add_argument("--steps", type=int, default=14)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="human")
    ap.add_argument("--reads", type=int, default=0, help="reads per step per GPU (0 = workload default)")
    ap.add_argument("--streams", type=int, default=0, help="host threads per GPU, each driving its own context (HIP streams + buffers)")
    ap.add_argument("--depth", type=int, default=0, help="sub-batches each stream maps one after the other within a step")
    ap.add_argument("--scale", type=float, default=1.0, help="genome scale of the human workloads")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pcie", "--no-resident", dest="no_resident", action="store_true", help="skip the second timed region (reads resident in HBM: mm355_map_resident)")
    ap.add_argument("--resident-steps", "--pcie-steps", dest="resident_steps", type=int, default=3, help="blocks of the second (HBM-resident) timed region")
    ap.add_argument("--resident-value", action="store_true", help="experiment: time the whole region with the reads resident in HBM (then `value` is NOT the metric of SURVEY 8d; flagged in value_basis)")
    ap.add_argument("--synth-procs", type=int, default=0, help="worker processes that synthesise the reads (0 = automatic; 1 under a profiler)")
    ap.add_argument("--oversubscribe", action="store_true", help="test hook: more ranks than GPUs on the node -- rank r uses GPU r mod (GPUs visible); the JSON line is flagged `oversubscribed`")
    ap.add_argument("--bin", action="store_true", help="experiment: sort the reads of a step by length before cutting the sub-batches (untimed preprocessing the drop-in path does not do)")
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
                  n_reads=73728, streams=8, depth=1, cfg="configs[2]", what="synthetic ONT reads N50~10kb 6% error (read set seed 4)"),
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


_SYNTH = {}


def _synth_block(b):
    import synthdata as S
    rd, _ = S.make_read_block(_SYNTH["seed"], b, _SYNTH["g"], **_SYNTH["kw"])
    return rd


def shard_reads(wl, g, n_per_gpu, rank, world, procs=1):
    """rank's shard of the ONE read set of world * n_per_gpu reads: contiguous, balanced by cumulative bases.  The blocks of
    the read set (synthdata.READ_BLOCK reads each, one PRNG stream per block) are synthesised by forked worker processes -- this runs
    before anything in the process has touched the GPU."""
    import multiprocessing as mp
    import synthdata as S
    from mappy_rs import shard_by_bases
    t0 = time.time()
    total = n_per_gpu * world
    lens = S.read_set_lengths(wl["seed"], total, **{k: v for k, v in wl["reads"].items() if k in ("n50", "sigma", "lo", "hi")})
    b = shard_by_bases(lens, world)
    lo, hi = b[rank], b[rank + 1]
    blocks = list(range(lo // S.READ_BLOCK, (hi + S.READ_BLOCK - 1) // S.READ_BLOCK))
    _SYNTH.update(seed=wl["seed"], g=g, kw=wl["reads"])
    if procs > 1 and len(blocks) > 1:
        with mp.get_context("fork").Pool(min(procs, len(blocks))) as pool:
            parts = pool.map(_synth_block, blocks, chunksize=1)
    else:
        parts = [_synth_block(bk) for bk in blocks]
    reads = []
    for bk, rd in zip(blocks, parts):
        b0 = bk * S.READ_BLOCK
        reads.extend(rd[max(lo, b0) - b0:min(hi, b0 + S.READ_BLOCK) - b0])
    log("[bench] read set: %d reads, rank %d maps reads [%d, %d) (%.1f Mbases) -- synthesised in %.1fs on %d processes"
        % (total, rank, lo, hi, sum(len(r) for r in reads) / 1e6, time.time() - t0, min(procs, max(1, len(blocks)))))
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


def reference_on_box(g, names, preset, reads, gpu_recs, threads, budget_s=600.0):
    """SURVEY 8d / BASELINE.md: if the real thing is on the GPU box -- the `minimap2` binary or the `mappy` module -- time it on the CPU sample and
    diff its records with the HIP path's (/root/reference/tests/benchmark.py:53-90 times mappy beside mappy_rs the same way).  None when
    neither exists (the case on every box this was run on: no network, nothing to install); the check itself is always made."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("minimap2")
    try:
        import mappy as _mappy          # noqa: F401  (the reference's own baseline module)
    except Exception:
        _mappy = None
    if exe is None and _mappy is None:
        return None
    import synthdata as S
    out = {"minimap2": exe, "mappy": getattr(_mappy, "__version__", None) if _mappy else None}
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "ref.fa")
        S.write_fasta(fa, g, names)
        n = len(reads)
        try:
            if exe is not None:
                fq = os.path.join(td, "reads.fa")
                with open(fq, "w") as f:
                    for i, r in enumerate(reads):
                        f.write(">r%d\n%s\n" % (i, r if isinstance(r, str) else S.codes_to_str(np.frombuffer(r, np.uint8))))
                ver = subprocess.run([exe, "--version"], capture_output=True, text=True, timeout=30).stdout.strip()
                t0 = time.time()
                p = subprocess.run([exe, "-x", preset, "-c", "--cs", "-t", str(threads), fa, fq], capture_output=True, text=True, timeout=budget_s)
                dt = time.time() - t0
                got = {}
                for line in p.stdout.splitlines():
                    c = line.split("\t")
                    tags = {t[:2]: t[5:] for t in c[12:]}
                    got.setdefault(int(c[0][1:]), []).append((c[5], int(c[7]), int(c[8]), int(c[2]), int(c[3]), c[4], int(c[11]), tags.get("cg"), tags.get("cs")))
                out.update(version=ver, seconds=round(dt, 1), note="index construction included in the time (minimap2 CLI)")
            else:
                al = _mappy.Aligner(fa, preset=preset, n_threads=threads)
                t0 = time.time()
                got = {}
                for i, r in enumerate(reads):
                    s_ = r if isinstance(r, str) else S.codes_to_str(np.frombuffer(r, np.uint8))
                    got[i] = [(h.ctg, h.r_st, h.r_en, h.q_st, h.q_en, "+" if h.strand > 0 else "-", h.mapq, h.cigar_str, h.cs) for h in al.map(s_, cs=True)]
                dt = time.time() - t0
                out.update(version=out["mappy"], seconds=round(dt, 1), note="mappy.Aligner.map loop, one thread")
            aligned = sum(len(reads[i]) for i in range(n) if got.get(i))
            out["value"] = round(aligned / dt / 1e6, 4); out["unit"] = "aligned Mbases/s"
            if gpu_recs is not None:
                bad = sum(1 for i in range(min(n, len(gpu_recs))) if sorted(got.get(i, [])) != sorted(gpu_recs[i]))
                out["parity"] = dict(reads=min(n, len(gpu_recs)), mismatching_reads=bad)
        except Exception as e:      # a broken install must not take the bench line down
            out["error"] = repr(e)[:300]
    return out


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


# VALU issue peak of the chip: 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz lane-operations per second
VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9
# Arithmetic a unit of work needs at least (DESIGN.md section 4 spells the counts out):
#   one cell of the two-piece affine recurrence with its direction byte: 34 integer operations (4 adds for the four gap candidates, 4 max
#   + 8 compare/select for z and its source, 2 subtractions for u' / v', 4 x 4 for the four gap states' open-or-extend updates and flag
#   bits), two int16 cells per lane-operation (v_pk_*) -> 17 lane-operations per cell;
#   one predecessor evaluation of the chaining recurrence (comput_sc + window test + running max): 30 operations, no packing.
DP_LANE_OPS_PER_CELL = 17.0
CHAIN_LANE_OPS_PER_PAIR = 30.0


def main():
    args = _ARGS
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import __graft_entry__ as ge
    import synthdata as S
    if args.workload not in WORKLOADS:
        raise SystemExit("unknown workload " + args.workload)
    wl = WORKLOADS[args.workload]
    n_per_step = args.reads or wl["n_reads"]          # reads per step per GPU
    n_thr = max(1, args.streams or wl["streams"])
    depth = max(1, args.depth or wl["depth"])
    K, W = max(1, args.steps), max(0, args.warmup)
    # every step maps a block of its own; the library keeps at most 64 resident batches per context (depth of them per block)
    max_blocks = 63 // depth if args.resident_value else 1 << 30   # (resident batches: 64 slots per context, slot 0 is the drop-in call's)
    n_blocks = min(K + W, max_blocks)
    n_timed_blocks = min(K, n_blocks - min(W, 1)) if n_blocks > 1 else 1

    # ---- host-side synthesis first: nothing below this block may have touched the GPU (worker processes are forked)
    g, names, gdesc, device_index = make_genome(wl["genome"], args.scale)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    procs = max(1, min(16, int((os.cpu_count() or 2) // max(1, local_world)) - 1))
    # never fork under a profiler: its preloaded tool library may have initialised the GPU before this program started (rocprofv3 --pmc does),
    # and a forked copy of such a process hangs on the device
    if args.synth_procs > 0:
        procs = args.synth_procs
    elif any(k.startswith(("ROCP", "ROCPROF")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        procs = 1
    reads = shard_reads(wl, g, n_per_step * n_blocks, rank, world, procs)

    dist = None
    if os.environ.get("WORLD_SIZE") is not None:   # launched by torch.distributed.run (or by _spawn_ranks): also with a single rank
        import torch
        import torch.distributed as dist_
        dist = dist_
        n_dev = torch.cuda.device_count()
        if args.oversubscribe and n_dev > 0:
            local_rank = local_rank % n_dev
        torch.cuda.set_device(local_rank)
        # (MM355_BENCH_DIST_BACKEND=gloo: the oversubscribed test's fallback when RCCL refuses two ranks on one device)
        dist.init_process_group(backend=os.environ.get("MM355_BENCH_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo"))
    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()
    from mappy_rs import _ffi
    L = _ffi.lib()
    if L.mm355_device_count() <= local_rank:
        raise SystemExit("bench.py needs an MI355X: libmm355 has no CPU fallback (devices visible: %d)" % L.mm355_device_count())

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
    log("[bench] index built + uploaded in %.1fs (mid_occ=%d)" % (time.time() - t0, mo.mid_occ))

    # one context (own HIP streams + working buffers) per host thread.  Block b of the read set (one step) is cut into n_str = streams x
    # depth sub-batches; sub-batch si of block b lives in context si % n_thr as resident batch b * depth + si // n_thr.  A host thread
    # maps its sub-batches of a step one after the other, so that the host tail of one overlaps kernels of another.
    n_str = n_thr * depth
    ctxs = []
    for ti in range(n_thr):
        ctx = C.c_void_p()
        _ffi.check(L.mm355_ctx_create(idx, local_rank, C.byref(ctx)))
        ctxs.append(ctx)
    per_block = len(reads) // n_blocks
    t0 = time.time()
    packed, rlens_np, block_bases = {}, {}, []
    resident_value = bool(args.resident_value)
    n_res_blocks = n_timed_blocks if resident_value else (0 if args.no_resident else min(args.resident_steps, n_timed_blocks))
    from mappy_rs import shard_by_bases, order_by_length
    for b in range(n_blocks):
        blk = reads[b * per_block:(b + 1) * per_block]
        block_bases.append(sum(len(r) for r in blk))
        # sub-batches of a step: contiguous chunks of the block IN INPUT ORDER, the way mappy_rs.map_batch cuts an iterable (SUB_BATCH_READS
        # reads each).  --bin: reads of similar length together (experiment; a preprocessing step the drop-in path does not perform)
        if args.bin:
            blk = [blk[i] for i in order_by_length([len(r) for r in blk])]
            cut = shard_by_bases([len(r) for r in blk], n_str)
        else:
            cut = [(len(blk) * si) // n_str for si in range(n_str + 1)]
        for si in range(n_str):
            pk = _ffi.pack_reads(blk[cut[si]:cut[si + 1]])
            rlens_np[(b, si)] = np.asarray(pk[1], dtype=np.int64)
            packed[(b, si)] = pk              # host buffers: what the drop-in call is handed
            if b < n_res_blocks or (resident_value and b >= n_timed_blocks):   # blocks of the HBM-resident region: uploaded before it (slot 0 is the drop-in call's)
                _ffi.check(L.mm355_batch_select(ctxs[si % n_thr], 1 + (b % (63 // depth)) * depth + si // n_thr))
                _ffi.check(L.mm355_batch_upload(ctxs[si % n_thr], len(pk[2]), pk[0], pk[1]))
    cpu_sample = reads[:min(len(reads), per_block)]   # the CPU leg samples the first block
    n_reads_rank = len(reads)
    del reads
    log("[bench] %d blocks of %d reads packed (%d sub-batches of ~%d reads; %d blocks also resident in HBM for the second region) in %.1fs" %
        (n_blocks, per_block, n_blocks * n_str, per_block // n_str, n_res_blocks, time.time() - t0))
    reads_per_sub_nominal = per_block // n_str

    def step_one(b, si, resident):
        ctx = ctxs[si % n_thr]
        hp = C.POINTER(_ffi.Hits)()
        if resident:   # inputs already in HBM
            _ffi.check(L.mm355_batch_select(ctx, 1 + (b % (63 // depth)) * depth + si // n_thr))
            _ffi.check(L.mm355_map_resident(ctx, C.byref(mo), _ffi.OUT_CS, C.byref(hp)))
        else:          # the drop-in call, the timed call of `value`: host buffers in (H2D), hit records out (D2H)
            _ffi.check(L.mm355_batch_select(ctx, 0))
            rarr, rl, keep = packed[(b, si)]
            _ffi.check(L.mm355_map_batch(ctx, C.byref(mo), len(keep), rarr, rl, _ffi.OUT_CS, C.byref(hp)))
        h = hp.contents
        nr = len(rlens_np[(b, si)])
        off = np.ctypeslib.as_array(h.hit_off, shape=(nr + 1,))
        mapped = np.diff(off) > 0
        aligned = int(rlens_np[(b, si)][mapped].sum())
        n_hits = int(h.n_hits)
        L.mm355_free_hits(hp)
        st = _ffi.Stats()
        L.mm355_get_stats(ctx, C.byref(st))
        return aligned, n_hits, st, nr, time.perf_counter()

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(n_thr)

    def run_blocks(resident, blocks):
        """one pass over every listed block: every host thread maps its own sub-batches of block after block without waiting for the other
        threads between blocks (the barriers bracket the whole region, not a step)"""
        def thread(ti):
            return [(k,) + step_one(b, si, resident) for k, b in enumerate(blocks) for si in range(ti, n_str, n_thr)]
        res = [r for part in pool.map(thread, range(n_thr)) for r in part]
        done = [0.0] * len(blocks)                 # when the last sub-batch of every step finished
        for r in res:
            done[r[0]] = max(done[r[0]], r[5])
        res = [r[1:5] for r in res]
        run_blocks.step_done = done
        agg_st = {}
        for _a, _h, st, _n in res:
            for k, _t in _ffi.Stats._fields_:
                v = getattr(st, k)
                v = np.array(list(v), dtype=np.float64) if hasattr(v, "__len__") else v
                agg_st[k] = agg_st.get(k, 0) + v
        return sum(r[0] for r in res), sum(r[1] for r in res), agg_st, sum(r[3] for r in res)

    def barrier():
        # both sides of the timed region: every stream of this rank's GPU drained (the library runs on its own HIP streams, which
        # torch.cuda.synchronize() would not see unless it synchronises the device -- hipDeviceSynchronize does), then the ranks meet
        _ffi.check(L.mm355_device_synchronize(local_rank))
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def timed(resident, blocks):
        barrier()
        t0 = time.perf_counter()
        aligned_tot, n_hits, agg, n_mapped = run_blocks(resident, blocks)
        barrier()
        dt = time.perf_counter() - t0
        timed.step_ms = [round((b - a) * 1e3, 1) for a, b in zip([t0] + run_blocks.step_done[:-1], run_blocks.step_done)]
        dt, aligned_all, bases_all = aggregate(dist, dt, aligned_tot, sum(block_bases[b] for b in blocks))
        return dt, aligned_all, bases_all, agg, n_mapped

    # blocks: the timed region maps blocks 0..; the warm-up maps the blocks behind them (its own reads), wrapping only when the 64-slot
    # limit of a context leaves no block for it
    timed_blocks = [k % n_timed_blocks for k in range(K)]
    warm_blocks = [(n_timed_blocks + k) % n_blocks for k in range(W)]
    if warm_blocks:
        run_blocks(resident_value, warm_blocks)
    dt, aligned_all, bases_all, agg, n_mapped = timed(resident_value, timed_blocks)   # `value`: the drop-in call, H2D / D2H inside
    step_ms = timed.step_ms
    if n_res_blocks and not resident_value:
        dt_r, aligned_r, _b, _agg, _n = timed(True, list(range(n_res_blocks)))      # the same reads again, resident in HBM
    else:
        dt_r, aligned_r = None, None

    if rank == 0:
        kern_ms = {"sketch": agg["ms_sketch"] / K, "seed_lookup": agg["ms_seed_lookup"] / K, "seed_expand": agg["ms_seed_expand"] / K,
                   "seed_select+mzflt": agg["ms_seed"] / K, "sort": agg["ms_sort"] / K, "chain": agg["ms_chain"] / K,
                   "backtrack": agg["ms_backtrack"] / K, "rmq": agg["ms_rmq"] / K, "dp": agg["ms_dp"] / K, "host_glue": agg["ms_host"] / K}
        n_mz, n_hit, n_a, n_am = agg["n_mz"] / K, agg["n_hit"] / K, agg["n_a"] / K, agg["n_a_multi"] / K
        cells, pairs = agg["dp_cells"] / K, agg["chain_pairs"] / K
        # algorithmic bytes per launch
        lookup_bytes = 16 * n_mz + 16 * n_hit
        expand_bytes = 8 * n_am + 16 * n_a
        n_ldp = max(1.0, agg["n_launch_dp"] / K)          # extension launch groups per step (one per sub-batch and round)
        # the extension kernels, each timed alone with HIP events on its own stream (group = 2 * size class + exact)
        # (the long-target classes are two launches, approx and exact alignments together: targets <= 4096 timed as group 8, longer ones as 10)
        gnames = ["k_ksw_reg<%d, %s>" % (np_, ex) for np_ in (1, 2, 4, 8) for ex in ("false", "true")] + \
                 ["k_ksw_extd2<512> (targets 1025..4096)", "-", "k_ksw_extd2<512> (targets > 4096)", "-", "-", "-", "k_ksw_row<2>", "k_ksw_row<4>", "k_ksw_row<8>", "k_ksw_rowl (targets 1025..8192)", "k_ksw_regw8 (exact, band <= 832, targets > 1024)",
                  "k_ksw_band<1> (128 diagonals)", "k_ksw_band<2> (256 diagonals)", "k_ksw_band<4> (512 diagonals)", "k_ksw_band2 (64 diagonals, two problems per wave)",
                  "second run of band problems (k_ksw_row / k_ksw_rowl)"]
        cells_g = np.array(agg["dp_cells_group"], dtype=np.float64); cells_g[8] = cells_g[8:10].sum(); cells_g[10] = cells_g[10:14].sum(); cells_g[9] = 0; cells_g[11:14] = 0
        nl_g = np.array(agg["n_launch_group"], dtype=np.float64); nl_g[8] = nl_g[8:10].max(); nl_g[10] = nl_g[10:14].max(); nl_g[9] = 0; nl_g[11:14] = 0
        n_lfront = float(n_str)                            # one launch of every front kernel per sub-batch
        reads_per_sub = per_block // n_str
        roof = {}

        def add_hbm(name, b, ms, nl, how):
            ach = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0     # bytes per launch / average launch duration (ratio of the per-step sums)
            roof[name] = dict(bound="hbm", achieved=round(ach, 3), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5),
                              frac_of_measured_copy=round(ach / HBM_COPY_GBS, 5), traffic=pmc_traffic(args.workload, reads_per_sub, name), kernel=name,
                              launches_per_step=round(nl, 2), ms_per_launch=round(ms / nl, 4), algorithmic_bytes=int(b / nl), formula=how)

        def add_valu(name, units, ms, nl, ops_per_unit, unit, how, bytes_per_unit):
            ach = units / (ms * 1e-3) / 1e9 if ms > 0 else 0.0   # G units per second
            peak = VALU_LANE_OPS / ops_per_unit / 1e9
            roof[name] = dict(bound="valu", achieved=round(ach, 3), peak=round(peak, 1), unit=unit, frac=round(ach / peak, 5),
                              traffic=pmc_traffic(args.workload, reads_per_sub, name), kernel=name, launches_per_step=round(nl, 2),
                              ms_per_launch=round(ms / nl, 4), units_per_launch=int(units / nl), lane_ops_per_unit=ops_per_unit,
                              hbm_gbs=round(units * bytes_per_unit / (ms * 1e-3) / 1e9, 3) if ms > 0 else 0.0,
                              hbm_frac=round(units * bytes_per_unit / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms > 0 else 0.0, formula=how)

        for i in range(24):   # every extension kernel that ran, each timed alone on its own stream
            if nl_g[i] > 0 and agg["ms_dp_group"][i] > 0:
                add_valu(gnames[i], cells_g[i] / K, agg["ms_dp_group"][i] / K, max(1.0, nl_g[i] / K), DP_LANE_OPS_PER_CELL, "Gcells/s",
                         "cell updates of the two-piece affine recurrence per second against the VALU issue peak (%.3g lane-ops/s / %g lane-ops per cell, "
                         "packed int16); hbm_*: the 1 B/cell direction matrix against 8 TB/s; HIP events on the kernel's own stream" % (VALU_LANE_OPS, DP_LANE_OPS_PER_CELL), 1.0)
        add_valu("extension launch group", cells, kern_ms["dp"], n_ldp, DP_LANE_OPS_PER_CELL, "Gcells/s",
                 "all extension kernels of a round on their streams + k_ksw_backtrack; one HIP-event pair around the group", 1.0)
        add_valu("k_chain", pairs, kern_ms["chain"], n_lfront, CHAIN_LANE_OPS_PER_PAIR, "Gpairs/s",
                 "predecessor evaluations of mg_lchain_dp per second against the VALU issue peak (%g lane-ops per pair); hbm_*: 16*n_a read + 20*n_a written" % CHAIN_LANE_OPS_PER_PAIR,
                 36.0 * n_a / max(1.0, pairs))
        # every kernel outside the extension rounds, each timed alone with a HIP-event pair on its stream (mm355_stats_t::ms_kernel)
        mk = np.array(agg["ms_kernel"], dtype=np.float64) / K
        n_keep = agg["n_a_kept"] / K if agg["n_a_kept"] > 0 else n_a
        n_lit, n_vr, pairs_big = agg["n_a_literal"] / K, agg["n_v_rmq"] / K, agg["chain_pairs_big"] / K
        bases_step = agg["n_bases"] / K
        front = [   # (slot, kernel, algorithmic bytes per step or None, formula)
            (0, "k_sketch", bases_step + 16 * n_mz, "L + 16*n_mz (bases read, minimizers written)"),
            (1, "k_mzflt", 32 * n_mz, "2*16*n_mz"),
            (2, "k_seed_lookup", lookup_bytes, "16*n_mz + 16*n_hit (minimizer read + one table slot; k_lookup_tiles + k_seed_lookup)"),
            (3, "k_seed_select", 16 * n_hit + 8 * n_mz, "16*n_hit + 8*n_mz (hit records read, kept seeds written)"),
            (4, "k_seed_expand", expand_bytes, "8*n_a_multi + 16*n_a (pos[] entry read + anchor written)"),
            (5, "k_cull", 16 * n_a + 8 * n_a + 8 * n_keep, "16*n_a read + 8*n_a position words + 8*n_kept survivors written"),
            (6, "k_asort", 16 * n_keep + 32 * n_keep, "8*n_kept words in + out, 16*n_kept anchors gathered + written"),
            (7, "k_sort_level_mw<1024> (radix_sort_128x emulation, buckets > 16384)", 32 * n_lit * 2, "2 levels of 2*16 B over the anchors of the reads with equal keys; ms_per_launch: all level launches of a sub-batch together (7 for GRCh38), traffic: bytes of ONE of them"),
            (20, "k_sort_level_mw<256> (radix_sort_128x emulation, buckets > 2048)", 32 * n_lit, "2*16 B over the anchors of the reads with equal keys; ms_per_launch: all level launches of a sub-batch together, traffic: bytes of ONE of them"),
            (21, "k_sort_tasks (radix_sort_128x emulation, one wave per bucket)", 32 * n_lit, "2*16 B over the anchors of the reads with equal keys"),
            (22, "k_tie_copy + k_asort + k_tie_tcnt (plain sort of the reads with equal keys)", 56 * n_lit, "(16 + 16 + 8 + 8 + 8) B per anchor of those reads"),
            (8, "k_chain_segments", 16 * n_keep, "16*n_kept"),
            (11, "k_backtrack", 28 * n_keep, "(16 + 12)*n_kept: anchors, f / p / v"),
            (12, "k_rmq_sort", 32 * n_vr, "2*16*n_v of the re-chained reads"),
            (13, "k_rmq_dp", 40 * n_vr, "16*n_v read + 24*n_v written"),
            (14, "k_rmq_backtrack", 28 * n_vr, "(16 + 12)*n_v"),
            (15, "k_dp_gather", None, None),
            (16, "k_ksw_backtrack", None, None),
            (17, "k_extra", None, None),
            (18, "k_read_codes", 3 * bases_step, "L read, 2*L written"),
            (19, "k_pack_chains", None, None),
        ]
        for slot, name, b_, how in front:
            if mk[slot] <= 0:
                continue
            if b_ is None:
                roof[name] = dict(bound="hbm", achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None, traffic=pmc_traffic(args.workload, reads_per_sub, name), kernel=name,
                                  launches_per_step=n_lfront, ms_per_launch=round(mk[slot] / n_lfront, 4), formula="latency-bound walk: no algorithmic byte count defined")
            else:
                add_hbm(name, b_, mk[slot], n_lfront, how)
        if mk[9] > 0:
            add_valu("k_chain_big", pairs_big, mk[9], n_lfront, CHAIN_LANE_OPS_PER_PAIR, "Gpairs/s", "predecessor evaluations of the long segments (a wave each) against the VALU issue peak", 36.0 * n_keep / max(1.0, pairs))
        if mk[10] > 0:
            add_valu("k_chain_small", pairs - pairs_big, mk[10], n_lfront, CHAIN_LANE_OPS_PER_PAIR, "Gpairs/s", "predecessor evaluations of the short segments (a lane each) against the VALU issue peak", 36.0 * n_keep / max(1.0, pairs))
        add_hbm("anchor sort stage", 32 * n_a, kern_ms["sort"], n_lfront, "2*16*n_a over the event span of the whole stage (cull + sort + literal emulation + its host round trips)")
        # RULE: `roofline` = the kernel with the largest summed duration among ALL kernels of the path, each timed alone with HIP events on the
        # stream it is launched on: the extension kernels (ms_dp_group) and every other kernel (ms_kernel).  No exclusion list; the rocprofv3
        # kernel statistics of the same command (profiles/) rank the same kernels by the same quantity.
        ms_all = {gnames[i]: float(agg["ms_dp_group"][i]) / K for i in range(24) if nl_g[i] > 0 and agg["ms_dp_group"][i] > 0}
        ms_all.update({name: float(mk[slot]) for slot, name, _b, _h in front if mk[slot] > 0})
        ms_all.update({n_: float(mk[sl]) for n_, sl in (("k_chain_big", 9), ("k_chain_small", 10)) if mk[sl] > 0})
        dom = max(ms_all, key=ms_all.get)
        LATENCY_GROUPS = (8, 10, 17, 18)
        latency_chains = {gnames[i]: dict(ms_per_launch=round(agg["ms_dp_group"][i] / max(1.0, nl_g[i]), 3), ms_per_step=round(agg["ms_dp_group"][i] / K, 3),
                                          cells_per_launch=int(cells_g[i] / max(1.0, nl_g[i])))
                          for i in LATENCY_GROUPS if nl_g[i] > 0 and agg["ms_dp_group"][i] > 0}
        # measured issue roof of the row sweep's instruction mix (profiles/r03_valubench.txt: 4.4 cycles per wave-instruction per SIMD at full
        # occupancy = 0.558e12 wave-instructions/s on 1024 SIMDs) beside the spec roof: x cells per wave-instruction of the kernel
        MEASURED_WAVE_INSTR_PER_S = 256 * 4 * 2.4e9 / 4.4
        for nm_, cpi in (("k_ksw_row<2>", 128 / 74.0), ("k_ksw_row<4>", 128 / 74.0), ("k_ksw_row<8>", 128 / 74.0)):
            if nm_ in roof:
                roof[nm_]["peak_measured_issue"] = round(MEASURED_WAVE_INSTR_PER_S * cpi / 1e9, 1)
                roof[nm_]["frac_of_measured_issue"] = round(roof[nm_]["achieved"] / roof[nm_]["peak_measured_issue"], 4)
        # SURVEY 8(d)'s B_seed has four terms: lookup (16*n_mz + 16*n_hit) and expansion (8*n_a_multi + 16*n_a) -- over both kernels' time
        seed4 = dict(bound="hbm", achieved=round((lookup_bytes + expand_bytes) / ((mk[2] + mk[4]) * 1e-3) / 1e9, 3) if mk[2] + mk[4] > 0 else 0.0, peak=HBM_PEAK_GBS, unit="GB/s",
                     kernel="k_seed_lookup + k_seed_expand", algorithmic_bytes=int((lookup_bytes + expand_bytes) / n_lfront), ms_per_launch=round((mk[2] + mk[4]) / n_lfront, 4),
                     formula="B_seed = 16*n_mz + 16*n_hit + 8*n_a_multi + 16*n_a over the two kernels' HIP-event time")
        seed4["frac"] = round(seed4["achieved"] / HBM_PEAK_GBS, 5)
        # the kernel the north star names: seed lookup against the HBM roof, and against what the memory system delivers for uniformly random
        # 128-byte lines of a table this size (profiles/r03_random_line_roof.json, measured by tools/linebench on the same chip)
        rl = dict(roof.get("k_seed_lookup", {}))
        rl["four_term_B_seed"] = seed4
        try:
            rr = json.load(open(os.path.join(ROOT, "profiles", "r03_random_line_roof.json")))
            rl["random_line_roof_gbs"] = rr["random_128B_lines_GBs"]
            if rl.get("traffic"):
                rl["traffic_gbs"] = round(rl["traffic"] / (rl["ms_per_launch"] * 1e-3) / 1e9, 1)
                rl["traffic_frac_of_random_line_roof"] = round(rl["traffic_gbs"] / rr["random_128B_lines_GBs"], 4)
        except (OSError, ValueError, KeyError):
            pass
        pool_threads = int(os.environ.get("MM355_HOST_THREADS", "14"))
        out = {
            "metric": METRIC, "oversubscribed": bool(args.oversubscribe),
            "value": round(aligned_all / dt / 1e6, 3), "unit": "Mbases/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32 (+f32 chaining gap cost, f64 RMQ priority)", "data": "synthetic",
            "config": dict(workload="%s: %s, %s %s, %s" % (wl["cfg"], gdesc, wl["preset"], "k%d w%d" % (io.k, io.w), wl["what"]),
                           reads_total=int(per_block * n_timed_blocks * world), reads_mapped_in_timed_region=int(n_mapped * world) if dist is None else None,
                           distinct_blocks_in_timed_region=n_timed_blocks, reads_per_step_per_gpu=per_block,
                           read_set="%d reads, %d contiguous shard(s) balanced by cumulative bases, %d distinct blocks per shard (%d timed + %d warm-up)"
                                    % (n_per_step * n_blocks * world, world, n_blocks, n_timed_blocks, n_blocks - n_timed_blocks),
                           streams_per_gpu=n_thr, sub_batches_per_stream=depth, mbases_per_step_per_gpu=round(sum(block_bases[b] for b in timed_blocks) / K / 1e6, 3),
                           preset=wl["preset"], parallelism="reads sharded over %d GPU(s), index replicated, no collective" % world),
            "value_basis": ("EXPERIMENT (--resident-value): reads resident in HBM when the timed region starts; not the metric of SURVEY 8d" if resident_value else
                            "PCIe-inclusive (SURVEY 8d): every timed sub-batch is handed to the drop-in call mm355_map_batch as host buffers (H2D of the reads, "
                            "D2H of the hit records inside the timed region), sub-batches cut in input order; resident_mbases_per_s: the same call sequence "
                            "with the reads already in HBM, a few of the same blocks") + ("; EXPERIMENT (--bin): blocks sorted by length beforehand" if args.bin else ""),
            "input_mbases_per_s": round(bases_all / dt / 1e6, 3),
            "step_ms": dict(median=float(np.median(step_ms)), min=min(step_ms), max=max(step_ms), all=step_ms,
                            note="completion of the last sub-batch of every step, rank 0 (the threads do not wait for one another between steps)"),
            "resident_mbases_per_s": None if dt_r is None else round(aligned_r / dt_r / 1e6, 3),
            "roofline": dict(roof[dom], ms_per_step=round(ms_all[dom], 3),
                             rule="the kernel with the largest summed duration among ALL kernels of the path (extension kernels and every other kernel, each timed "
                                  "alone with HIP events on the stream it is launched on; kernel_ms_per_step_all) -- no exclusion list"),
            "kernel_ms_per_step_all": {k: round(v, 3) for k, v in sorted(ms_all.items(), key=lambda kv: -kv[1])},
            "roofline_seed_lookup": rl, "roofline_all": roof, "latency_chains": latency_chains, "kernel_ms_per_step": {k: round(v, 3) for k, v in kern_ms.items()},
            "dp_kernel_ms_per_step": {gnames[i]: round(float(agg["ms_dp_group"][i]) / K, 3) for i in range(24) if nl_g[i] > 0},
            "dp_cells_per_step": {gnames[i]: int(cells_g[i] / K) for i in range(24) if nl_g[i] > 0},
            "counters_per_step": dict(n_mz=int(n_mz), n_hit=int(n_hit), n_a=int(n_a), n_a_kept_by_cull=int(n_keep), n_a_sorted_literally=int(n_lit), n_a_multi=int(n_am), chain_pairs=int(pairs), dp_cells=int(cells),
                                      n_dp_jobs=int(agg["n_dp_jobs"] / K), n_dp_band=int(agg["n_dp_band"] / K), n_dp_band_redo=int(agg["n_dp_band_redo"] / K), n_rounds_split=int(agg["n_rounds_split"]), n_sort_tie_reads=int(agg["n_sort_tie_reads"] / K),
                                      n_rmq_reads=int(agg["n_rmq_reads"] / K), n_rmq_host_fallback=int(agg["n_rmq_host"] / K),
                                      rmq_window_elements=int(agg["rmq_scanned"] / K)),
            "host": dict(cpu_us_per_read=round(agg["host_cpu_ms"] * 1e3 / max(1, n_mapped), 2), pool_threads=pool_threads, context_threads=n_thr,
                         busy_frac_of_pool=round(agg["host_cpu_ms"] * 1e-3 / (dt * (pool_threads + n_thr)), 4),
                         note="CPU time (thread clocks) of the context threads and the shared pool inside mm355_map_resident, rank 0; "
                              "busy_frac = that / (wall x (pool + context threads))"),
        }
    barrier()
    gpu_sigs = None
    gpu_recs = None
    if rank == 0 and not args.no_cpu and world == 1:   # (outside the timed region) the records of the reads the CPU sample will cover
        import mappy_rs
        gpu_sigs = []
        import shutil as _sh
        import importlib.util as _iu
        ref_wanted = _sh.which("minimap2") is not None or _iu.find_spec("mappy") is not None
        gpu_recs = []
        n_chk = len(cpu_sample) if args.cpu_seconds >= 60 else min(len(cpu_sample), 4 * 6144)   # a long CPU leg compares the whole block
        _ffi.check(L.mm355_batch_select(ctxs[0], 63))   # a slot of its own: the resident blocks stay as they are
        for lo in range(0, n_chk, 6144):
            sub = cpu_sample[lo:min(n_chk, lo + 6144)]
            rarr, rl_, keep = _ffi.pack_reads(sub)
            hp = C.POINTER(_ffi.Hits)()
            _ffi.check(L.mm355_map_batch(ctxs[0], C.byref(mo), len(sub), rarr, rl_, _ffi.OUT_CS, C.byref(hp)))
            for ms in mappy_rs._batch_to_mappings(hp, len(sub), names):
                gpu_sigs.append(None if isinstance(ms, Exception) else hash(tuple(tuple(getattr(m, k) for k in SIG_FIELDS) for m in ms)))
                if ref_wanted:
                    gpu_recs.append([] if isinstance(ms, Exception) else [(m.target_name, m.target_start, m.target_end, m.query_start, m.query_end, "+" if m.strand > 0 else "-",
                                                                             m.mapq, m.cigar_str, m.cs) for m in ms])
            L.mm355_free_hits(hp)
            if lo == 0:   # this sub-batch had the GPU to itself: the seed-lookup kernel without other contexts' kernels in front of it on its queue
                st1 = _ffi.Stats()
                L.mm355_get_stats(ctxs[0], C.byref(st1))
                if st1.ms_seed_lookup > 0:
                    b1 = 16.0 * st1.n_mz + 16.0 * st1.n_hit
                    a1 = b1 / (st1.ms_seed_lookup * 1e-3) / 1e9
                    rl = out["roofline_seed_lookup"]
                    rl["alone"] = dict(reads=len(sub), ms_per_launch=round(st1.ms_seed_lookup, 4), algorithmic_bytes=int(b1), achieved=round(a1, 1), frac=round(a1 / HBM_PEAK_GBS, 5),
                                       lookups_per_s=round(st1.n_mz / (st1.ms_seed_lookup * 1e-3) / 1e9, 2), unit="GB/s; G lookups/s",
                                       note="one sub-batch alone on the GPU (the parity sample), same kernel, HIP events on its stream")
                    if rl.get("traffic"):
                        rl["alone"]["traffic_gbs"] = round(rl["traffic"] * (st1.n_mz / max(1.0, out["counters_per_step"]["n_mz"] / n_str)) / (st1.ms_seed_lookup * 1e-3) / 1e9, 1)
    for ctx in ctxs:
        L.mm355_ctx_destroy(ctx)
    L.mm355_index_free(idx)
    if rank == 0:
        if not args.no_cpu and world == 1:   # after the GPU side has released its memory; rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(g, names, wl["preset"], cpu_sample, args.cpu_seconds, min(16, os.cpu_count() or 1), gpu_sigs)
            # the real minimap2 / mappy, if the box has one (it never did): timed on the same sample, records diffed -- null otherwise
            n_ref = len(gpu_recs) if gpu_recs else min(len(cpu_sample), 6144)
            out["reference_on_box"] = reference_on_box(g, names, wl["preset"], cpu_sample[:n_ref], gpu_recs or None, min(16, os.cpu_count() or 1))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
