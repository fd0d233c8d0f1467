"""How many anchors of a GRCh38-scale read can never chain?  
An anchor only ever interacts with anchors of its own x-component (maximal run of the x-sorted array with the same strand | rid and
consecutive gaps <= max_dist_x); a component with fewer than ceil(min_chain_score / k) anchors puts nothing into z[].
Prints, for a sample of configs[2]: the fraction of anchors in such components (exact components, 8192-base bins, hashed bins),
the survivors per read, and the reads with equal keys before / after."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mappy-rs_amd"))
import numpy as np
import synthdata as S
from mappy_rs import _ffi
L = _ffi.lib()
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
g, names = S.make_human_like(3, scale)
reads, _ = S.make_reads_codes(4, g, n_reads, n50=10000)
io, mo = _ffi.IdxOpt(), _ffi.MapOpt()
L.mm355_set_opt(None, C.byref(io), C.byref(mo)); mo.flag |= 4
ptrs = (C.c_char_p * len(g))(*[C.cast(c.ctypes.data, C.c_char_p) for c in g])
lens = (C.c_int64 * len(g))(*[len(c) for c in g]); nm = (C.c_char_p * len(g))(*[n.encode() for n in names])
idx = C.c_void_p(); _ffi.check(L.mm355_index_build_device(C.byref(io), len(g), ptrs, lens, nm, 0, C.byref(idx)))
L.mm355_mapopt_update(C.byref(mo), idx)
sr = _ffi.StageRunner(idx, mo, 0)
a, _, _ = sr.anchors(reads, sorted_=False, cap=60_000_000)
D, T = 5000, 3
tot = cul_exact = cul_bin = cul_hash = 0
surv = []; ties0 = ties1 = 0; comp_sizes = []
for x in a:
    x = np.ascontiguousarray(x[:, 0]) if len(x) else np.zeros(0, np.uint64)
    n = len(x); tot += n
    if n == 0: surv.append(0); continue
    xs = np.sort(x)
    brk = np.concatenate(([True], (xs[1:] - xs[:-1]) > np.uint64(D)))     # (rid / strand changes are gaps >= 2^32 - 2^31)
    cid = np.cumsum(brk) - 1
    cs = np.bincount(cid)
    keep = cs[cid] >= T
    cul_exact += int((~keep).sum())
    comp_sizes.append(cs[cs >= T])
    # bins of 8192 bases: coarse component = run of consecutive non-empty bins
    b = xs >> np.uint64(13)
    ub, cnt = np.unique(b, return_counts=True)
    bb = np.concatenate(([True], (ub[1:] - ub[:-1]) > np.uint64(1)))
    rid = np.cumsum(bb) - 1
    rs = np.bincount(rid, weights=cnt)
    keepb = rs[rid] >= T
    kept_b = int(cnt[keepb].sum())
    cul_bin += n - kept_b
    # hashed bins: table of 2^ceil(log2(2n)) saturating counters, run = consecutive bins whose hashed slot is non-empty
    tb = 1 << max(6, int(np.ceil(np.log2(2 * n))))
    h = ((ub * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)).astype(np.int64) & (tb - 1)
    table = np.zeros(tb, np.int64); np.add.at(table, h, cnt)
    def slot(v): return table[(((v * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)).astype(np.int64)) & (tb - 1)]
    tot_run = slot(ub).copy()
    for sgn in (1, -1):          # extend the run to both sides while the hashed slot is non-empty (up to T steps are enough to reach T)
        alive = np.ones(len(ub), bool); v = ub.copy()
        for _ in range(T):
            v = v + np.uint64(1) if sgn > 0 else v - np.uint64(1)
            s = slot(v); alive &= s > 0
            tot_run += np.where(alive, s, 0)
    keeph = tot_run >= T
    cul_hash += n - int(cnt[keeph].sum())
    xk = xs[keep]
    surv.append(len(xk))
    ties0 += bool(n > 64 and (xs[1:] == xs[:-1]).any()); ties1 += bool(len(xk) > 64 and (xk[1:] == xk[:-1]).any())
surv = np.array(surv); na = np.array([len(x) for x in a])
cz = np.concatenate(comp_sizes) if comp_sizes else np.zeros(1)
print("reads %d anchors %d (%.0f per read, max %d)" % (len(a), tot, tot / len(a), na.max()))
print("culled: exact components %.4f | 8192-base bins %.4f | hashed bins %.4f" % (cul_exact / tot, cul_bin / tot, cul_hash / tot))
print("survivors per read: mean %.0f median %.0f p90 %.0f p99 %.0f max %d; reads > 4096: %d, > 8192: %d, > 16384: %d" % (surv.mean(), np.median(surv), np.percentile(surv, 90), np.percentile(surv, 99), surv.max(), (surv > 4096).sum(), (surv > 8192).sum(), (surv > 16384).sum()))
print("surviving components: %d, sizes mean %.1f median %.0f p90 %.0f p99 %.0f max %d; anchors in components > 64: %.3f" % (len(cz), cz.mean(), np.median(cz), np.percentile(cz, 90), np.percentile(cz, 99), cz.max(), cz[cz > 64].sum() / max(1, cz.sum())))
print("reads with equal keys: all anchors %d, survivors %d" % (ties0, ties1))
