"""how many reads have equal-x anchors (ties) at GRCh38-like scale: python tools/tieprobe.py [scale=1.0] [reads=2048]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mappy-rs_amd"))
import numpy as np
import synthdata as S
from mappy_rs import _ffi
L = _ffi.lib()
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
g, names = S.make_human_like(3, scale)
reads, _ = S.make_reads_codes(4, g, nr, n50=10000)
io, mo = _ffi.IdxOpt(), _ffi.MapOpt()
L.mm355_set_opt(None, C.byref(io), C.byref(mo)); mo.flag |= 4
ptrs = (C.c_char_p * len(g))(*[C.cast(c.ctypes.data, C.c_char_p) for c in g])
lens = (C.c_int64 * len(g))(*[len(c) for c in g]); nm = (C.c_char_p * len(g))(*[n.encode() for n in names])
idx = C.c_void_p(); _ffi.check(L.mm355_index_build_device(C.byref(io), len(g), ptrs, lens, nm, 0, C.byref(idx)))
L.mm355_mapopt_update(C.byref(mo), idx)
sr = _ffi.StageRunner(idx, mo, 0)
a, _, _ = sr.anchors(reads, sorted_=True, cap=200_000_000)
na = np.array([len(x) for x in a])
ties = np.array([int((x[1:, 0] == x[:-1, 0]).sum()) if len(x) > 1 else 0 for x in a])
print("reads %d, anchors total %d, mean %.0f, max %d" % (nr, na.sum(), na.mean(), na.max()))
for lo, hi in ((0, 2048), (2048, 16384), (16384, 65536), (65536, 10**9)):
    m = (na > lo) & (na <= hi)
    if m.sum():
        print("n_a in (%d, %d]: %4d reads, %9d anchors, with ties: %4d reads (%.0f%%), tie pairs %d" % (lo, hi, m.sum(), na[m].sum(), (ties[m] > 0).sum(), 100.0 * (ties[m] > 0).sum() / m.sum(), ties[m].sum()))
