"""small-batch latency of the drop-in surface: Aligner.map (one read) and one mm355_map_batch call of 1 / 16 / 256 / 4096 reads
(E. coli-like genome, ONT-like reads N50 ~8 kb).  python tools/latency_table.py"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mappy-rs_amd"))
import numpy as np
import synthdata as S, mappy_rs
from mappy_rs import _ffi
g = S.ecoli_like(1); S.write_fasta("/tmp/ml.fa", g)
reads, _ = S.make_reads(2, g, 4096 + 64, n50=8000)
al = mappy_rs.Aligner("/tmp/ml.fa", preset="map-ont")
for r in reads[:8]: al.map(r, cs=True)
r8 = [r for r in reads if 7000 <= len(r) <= 9000][:40]
t0 = time.perf_counter()
for r in r8: al.map(r, cs=True)
dt = (time.perf_counter() - t0) / len(r8)
print("Aligner.map, one ~8 kb read: %.2f ms" % (dt * 1e3))
for n in (1, 16, 256, 4096):
    sub = reads[64:64 + n]
    al._map_many(sub, 1)
    reps = 5 if n <= 256 else 2
    t0 = time.perf_counter()
    for _ in range(reps): al._map_many(sub, 1)
    dt = (time.perf_counter() - t0) / reps
    nb = sum(len(r) for r in sub)
    print("batch of %4d reads (%6.2f Mbases): %8.2f ms per call, %7.3f ms per read, %7.1f Mbases/s" % (n, nb / 1e6, dt * 1e3, dt * 1e3 / n, nb / dt / 1e6))
