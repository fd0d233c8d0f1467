"""How much of the human workload's latency is the anchor-rich (satellite) reads?  Anchor-count distribution of the first reads of the
bench read set, and the mapping rate of the same reads with the heaviest ones left out / concentrated in their own sub-batches.
python tools/heavyprobe.py [n_reads]"""
import ctypes as C, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mappy-rs_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import synthdata as S
from mappy_rs import _ffi
from concurrent.futures import ThreadPoolExecutor
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
L = _ffi.lib()
g, names = S.make_human_like(3, 1.0)
reads = S.read_set_slice(4, 0, n, g, n50=10000, sigma=0.75, lo=500, hi=100000)
io, mo = _ffi.IdxOpt(), _ffi.MapOpt()
L.mm355_set_opt(None, C.byref(io), C.byref(mo)); L.mm355_set_opt(b"map-ont", C.byref(io), C.byref(mo)); mo.flag |= 4
ptrs = (C.c_char_p * len(g))(*[C.cast(c.ctypes.data, C.c_char_p) for c in g])
lens = (C.c_int64 * len(g))(*[len(c) for c in g]); nm = (C.c_char_p * len(g))(*[x.encode() for x in names])
idx = C.c_void_p(); _ffi.check(L.mm355_index_build_device(C.byref(io), len(g), ptrs, lens, nm, 0, C.byref(idx)))
L.mm355_mapopt_update(C.byref(mo), idx)
ctx = C.c_void_p(); _ffi.check(L.mm355_ctx_create(idx, 0, C.byref(ctx)))
na = []
cap = 150_000_000
a = np.zeros((cap, 2), np.uint64)
for lo in range(0, n, 2048):
    sub = reads[lo:lo + 2048]
    arr, ln, keep = _ffi.pack_reads(sub)
    off = np.zeros(len(sub) + 1, np.int64)
    rep = np.zeros(len(sub), np.int32); nmp = np.zeros(len(sub), np.int32)
    _ffi.check(L.mm355_stage_anchors(ctx, C.byref(mo), len(sub), arr, ln, 0, off.ctypes.data, a.ctypes.data, cap, rep.ctypes.data, nmp.ctypes.data))
    na += list(np.diff(off))
del a
na = np.array(na); rl = np.array([len(r) for r in reads])
print("reads %d, anchors total %.1fM, mean %.0f, median %.0f, max %d" % (n, na.sum() / 1e6, na.mean(), np.median(na), na.max()), flush=True)
for thr in (4096, 16384, 32768, 65536, 131072, 262144):
    m = na > thr
    print("  n_a > %6d: %5.2f%% of reads, %5.1f%% of anchors, %5.1f%% of bases" % (thr, 100 * m.mean(), 100 * na[m].sum() / na.sum(), 100 * rl[m].sum() / rl.sum()), flush=True)
L.mm355_ctx_destroy(ctx)


def run_parts(parts, label, n_thr=6, dynamic=True):
    packed = [_ffi.pack_reads(p) for p in parts]
    ctxs = []
    for _ in range(n_thr):
        c = C.c_void_p(); _ffi.check(L.mm355_ctx_create(idx, 0, C.byref(c))); ctxs.append(c)
    nxt = [0]; lk = threading.Lock()

    def worker(ti):
        while True:
            with lk:
                si = nxt[0]; nxt[0] += 1
            if si >= len(packed):
                return
            arr, ln, keep = packed[si]
            hp = C.POINTER(_ffi.Hits)(); _ffi.check(L.mm355_map_batch(ctxs[ti], C.byref(mo), len(keep), arr, ln, 1, C.byref(hp))); L.mm355_free_hits(hp)
    pool = ThreadPoolExecutor(n_thr)

    def step():
        nxt[0] = 0
        list(pool.map(worker, range(n_thr)))
    step()
    t0 = time.perf_counter(); step(); step(); dt = (time.perf_counter() - t0) / 2
    nb = sum(len(r) for p in parts for r in p)
    print("%-34s %6d reads %3d sub-batches (first sizes %s) %7.1f Mbases  %.3f s/step  %.1f Mbases/s" %
          (label, sum(len(p) for p in parts), len(parts), [len(p) for p in parts[:3]], nb / 1e6, dt, nb / dt / 1e6), flush=True)
    for c in ctxs:
        L.mm355_ctx_destroy(c)


def strided(sel, n_str=24):
    sub = [reads[i] for i in sel]
    return [sub[s::n_str] for s in range(n_str)]


run_parts(strided(range(n)), "all reads, strided")
run_parts(strided([i for i in range(n) if na[i] <= 65536]), "n_a <= 65536 only, strided")
run_parts(strided([i for i in range(n) if na[i] <= 16384]), "n_a <= 16384 only, strided")
run_parts(strided([i for i in range(n) if na[i] > 16384], 6), "n_a > 16384 only, 6 sub-batches")
# heavy reads concentrated: contiguous sub-batches of the n_a-sorted order, cut by weight (anchors, floor 2000 per read)
order = np.argsort(-na, kind="stable")
w = np.maximum(na[order], 2000).astype(np.float64); cum = np.cumsum(w); tot = cum[-1]
for n_str in (24, 48):
    cuts = sorted(set([0] + [min(len(order), int(np.searchsorted(cum, tot * (s + 1) / n_str)) + 1) for s in range(n_str - 1)] + [len(order)]))
    parts = [[reads[i] for i in order[a_:b_]] for a_, b_ in zip(cuts[:-1], cuts[1:]) if b_ > a_]
    run_parts(parts, "contiguous by n_a, %d weighted" % n_str)
L.mm355_index_free(idx)
