"""DP kernel micro-benchmark through mm355_stage_dp: n jobs of ~L x L (5 % divergence), approx or exact.
usage: python tools/dpbench.py [L=210] [n=100000] [flag=8] [w=500]   (w=30001: the full band of a gap fill)"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mappy-rs_amd"))
import numpy as np
import synthdata as S
from mappy_rs import _ffi
import mappy_rs
L = _ffi.lib()
Lt = int(sys.argv[1]) if len(sys.argv) > 1 else 210
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
flag = int(sys.argv[3]) if len(sys.argv) > 3 else 8
band = int(sys.argv[4]) if len(sys.argv) > 4 else 500
rng = np.random.default_rng(5)
g = S.make_genome(1, [200000])
S.write_fasta("/tmp/dpb.fa", g)
al = mappy_rs.Aligner("/tmp/dpb.fa", preset="map-ont")
base = S.random_codes(rng, Lt * 64)
qs, ts = [], []
for i in range(64):
    t = base[i * Lt:(i + 1) * Lt]
    qs.append(S.mutate(t, rng, 0.03, 0.015, 0.015).astype(np.uint8)); ts.append(t.astype(np.uint8))
qcat = np.concatenate(qs); tcat = np.concatenate(ts)
qo = np.concatenate([[0], np.cumsum([len(x) for x in qs])]); to = np.concatenate([[0], np.cumsum([len(x) for x in ts])])
ja = (_ffi.DpJob * n)()
for i in range(n):
    k = i % 64
    ja[i].qlen, ja[i].tlen, ja[i].qoff, ja[i].toff, ja[i].w, ja[i].zdrop, ja[i].end_bonus, ja[i].flag = len(qs[k]), Lt, int(qo[k]), int(to[k]), band, 400, -1, flag
res = (_ffi.DpRes * n)()
cap = int(n * (2 * Lt + 40))
cig = np.zeros(cap, np.uint32)
sr = al._stage_runner()
for rep in range(3):
    t0 = time.time()
    _ffi.check(L.mm355_stage_dp(sr.ctx, C.byref(al._mo), n, ja, qcat.ctypes.data, qcat.size, tcat.ctypes.data, tcat.size, res, cig.ctypes.data, cap))
    st = sr.stats()
    print("L=%d n=%d flag=%#x: dp stage %.2f ms (wall %.1f ms), %.1f Gcells/s" % (Lt, n, flag, st.ms_dp, (time.time() - t0) * 1e3, st.dp_cells / st.ms_dp / 1e6))
