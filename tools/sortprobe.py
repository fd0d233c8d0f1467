import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mappy-rs_amd"))
import numpy as np
import synthdata as S
from mappy_rs import _ffi
L = _ffi.lib()
g, names = S.make_human_like(3, 0.1)
reads, _ = S.make_reads_codes(4, g, 4096, n50=10000)
io, mo = _ffi.IdxOpt(), _ffi.MapOpt()
L.mm355_set_opt(None, C.byref(io), C.byref(mo)); mo.flag |= 4
ptrs = (C.c_char_p * len(g))(*[C.cast(c.ctypes.data, C.c_char_p) for c in g])
lens = (C.c_int64 * len(g))(*[len(c) for c in g]); nm = (C.c_char_p * len(g))(*[n.encode() for n in names])
idx = C.c_void_p(); _ffi.check(L.mm355_index_build_device(C.byref(io), len(g), ptrs, lens, nm, 0, C.byref(idx)))
L.mm355_mapopt_update(C.byref(mo), idx)
sr = _ffi.StageRunner(idx, mo, 0)
a, _, _ = sr.anchors(reads, sorted_=False, cap=120_000_000)
na = np.array([len(x) for x in a]); order = np.argsort(-na)
print("n_a: total %d max %d p99 %d median %d" % (na.sum(), na.max(), np.percentile(na, 99), np.median(na)))
for k in (0, 1, 5, 20, 100, 400, 2000):
    r = reads[order[k]]
    for stage in ("anchors_sorted", "chains"):
        t0 = time.time()
        if stage == "anchors_sorted": sr.anchors([r], sorted_=True, cap=4_000_000)
        else: sr.chains([r], cap=4_000_000)
        st = sr.stats()
        print("rank %4d n_a %7d len %6d  %s: sort %.2f ms chain %.2f ms backtrack %.2f ms" % (k, na[order[k]], len(r), stage, st.ms_sort, st.ms_chain, st.ms_backtrack))
for rep in range(2):
    sr.anchors(reads, sorted_=True, cap=120_000_000); st = sr.stats()
    print("whole batch: sort %.2f ms (n_heavy>16384: %d, >65536: %d)" % (st.ms_sort, (na > 16384).sum(), (na > 65536).sum()))
