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
