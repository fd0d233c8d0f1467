"""latency of the single-read surface (Aligner.map): python tools/maplat.py"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mappy-rs_amd"))
import synthdata as S, mappy_rs
g = S.ecoli_like(1); S.write_fasta("/tmp/ml.fa", g)
reads, _ = S.make_reads(2, g, 300, n50=8000)
al = mappy_rs.Aligner("/tmp/ml.fa", preset="map-ont")
for r in reads[:20]: al.map(r)
t0 = time.time(); n = 0
for r in reads[20:]:
    n += len(al.map(r, cs=True))
dt = time.time() - t0
print("Aligner.map: %.2f ms per read (%d reads, %.1f Mbases/s)" % (dt / 280 * 1e3, 280, sum(len(r) for r in reads[20:]) / dt / 1e6))
# device time of the stages for the last reads (HIP events around each stage's launches)
import ctypes as C
from mappy_rs import _ffi
acc = {}
for r in reads[20:120]:
    al.map(r)
    st = _ffi.Stats(); al._L.mm355_get_stats(al._ctx, C.byref(st))
    for f, _ in st._fields_:
        if f.startswith("ms_") and not f.endswith("group"): acc[f] = acc.get(f, 0.0) + getattr(st, f)
    acc["n_launch_dp"] = acc.get("n_launch_dp", 0) + st.n_launch_dp
print("per read: " + "  ".join("%s %.3f" % (k, v / 100) for k, v in acc.items()))
