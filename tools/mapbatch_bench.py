"""throughput of the drop-in surface itself (Aligner.map_batch, Python Mapping objects included): python tools/mapbatch_bench.py [reads=32768] [threads=8] [result channel capacity]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mappy-rs_amd"))
import synthdata as S
import mappy_rs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if len(sys.argv) > 3: mappy_rs.RESULT_CHANNEL_CAP = int(sys.argv[3])      # (experiments: the reference's bounded(20000) result channel)
g = S.ecoli_like(1); S.write_fasta("/tmp/mb.fa", g)
reads, _ = S.make_reads(2, g, n, n50=8000)
al = mappy_rs.Aligner("/tmp/mb.fa", preset="map-ont")
al.enable_threading(thr)
bases = sum(len(r) for r in reads)
for rep in range(3):
    t0 = time.time()
    it = al.map_batch([{"seq": r, "id": i} for i, r in enumerate(reads)])
    t1 = time.time()
    first = next(it); t2 = time.time()
    nres = 1 + sum(1 for _ in it)
    t3 = time.time()          # streaming: the call returns when the iterable is consumed, the results follow in completion order
    print("map_batch(%d reads, %.1f Mbases, %d threads): call %.2f s, first result after %.2f s, all %d results after %.2f s -> %.1f Mbases/s" %
          (n, bases / 1e6, thr, t1 - t0, t2 - t0, nres, t3 - t0, bases / 1e6 / (t3 - t0)), flush=True)
