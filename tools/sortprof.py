"""rocprofv3 target: sort stage of the heaviest read alone, then of the whole batch (10 %-scale human-like genome)"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mappy-rs_amd"))
import numpy as np
import synthdata as S
from mappy_rs import _ffi
L = _ffi.lib()
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
nr = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
g, names = S.make_human_like(3, scale)
reads, _ = S.make_reads_codes(4, g, nr, n50=10000)
io, mo = _ffi.IdxOpt(), _ffi.MapOpt()
L.mm355_set_opt(None, C.byref(io), C.byref(mo)); mo.flag |= 4
ptrs = (C.c_char_p * len(g))(*[C.cast(c.ctypes.data, C.c_char_p) for c in g])
lens = (C.c_int64 * len(g))(*[len(c) for c in g]); nm = (C.c_char_p * len(g))(*[n.encode() for n in names])
idx = C.c_void_p(); _ffi.check(L.mm355_index_build_device(C.byref(io), len(g), ptrs, lens, nm, 0, C.byref(idx)))
L.mm355_mapopt_update(C.byref(mo), idx)
sr = _ffi.StageRunner(idx, mo, 0)
a, _, _ = sr.anchors(reads, sorted_=False, cap=120_000_000)
na = np.array([len(x) for x in a]); order = np.argsort(-na)
mode = sys.argv[1] if len(sys.argv) > 1 else "one"
if mode == "one":
    sr.chains([reads[order[0]]], cap=4_000_000)
else:
    sr.chains(reads, cap=120_000_000)
st = sr.stats()
print("n_a: total %d max %d p99 %d median %d heavy %d" % (na.sum(), na.max(), np.percentile(na, 99), np.median(na), (na > 16384).sum()))
print("mode %s: sort %.2f chain %.2f backtrack %.2f" % (mode, st.ms_sort, st.ms_chain, st.ms_backtrack))
