"""shape statistics of the extension problems dumped with MM355_DP_DUMP=<file>: python tools/dpshape.py <file>"""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int32).reshape(-1, 4)
q, t, w, fl = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2], a[:, 3]
ok = (q > 0) & (t > 0)
q, t, w, fl = q[ok], t[ok], w[ok], fl[ok]
approx = (fl & 8) != 0
band = np.where(w < 0, np.maximum(q, t), w)
full = band >= q + t
diag_len = np.minimum(np.minimum(q, t), band + 1)
cells = np.where(full, q * t, np.minimum(q * t, (band + 1) * (q + t)))          # rough for banded
blocks_per_diag = np.ceil((diag_len + 15) / 128.0)
bd = (q + t - 1) * blocks_per_diag                                                 # block-diagonals (lower bound)
print("jobs %d  approx %.1f%%  full-band %.1f%%" % (len(q), 100 * approx.mean(), 100 * full.mean()))
print("cells %.3g  block-diagonals %.3g  -> cells per block-diagonal %.1f of 128" % (cells.sum(), bd.sum(), cells.sum() / bd.sum()))
for name, sel in (("T<=128", t <= 128), ("128<T<=256", (t > 128) & (t <= 256)), ("256<T<=512", (t > 256) & (t <= 512)), ("512<T<=1024", (t > 512) & (t <= 1024)), ("T>1024", t > 1024)):
    for ex, sel2 in (("approx", sel & approx), ("exact", sel & ~approx)):
        if sel2.sum() == 0: continue
        qq, tt = q[sel2], t[sel2]
        print("  %-12s %-6s jobs %8d  cells %.3g (%.1f%%)  blockdiag %.3g (%.1f%%)  cells/blockdiag %.1f  median q %d t %d  q<64: %.1f%% of jobs, %.1f%% of blockdiags"
              % (name, ex, sel2.sum(), cells[sel2].sum(), 100 * cells[sel2].sum() / cells.sum(), bd[sel2].sum(), 100 * bd[sel2].sum() / bd.sum(),
                 cells[sel2].sum() / bd[sel2].sum(), np.median(qq), np.median(tt), 100 * (qq < 64).mean(), 100 * bd[sel2][qq < 64].sum() / bd[sel2].sum()))
# distribution of min(q,t) weighted by block-diagonals
m = np.minimum(q, t)
for lo, hi in ((0, 16), (16, 32), (32, 64), (64, 128), (128, 256), (256, 100000)):
    s = (m >= lo) & (m < hi)
    print("  min(q,t) in [%d,%d): %.1f%% of jobs, %.1f%% of block-diagonals, %.1f%% of cells" % (lo, hi, 100 * s.mean(), 100 * bd[s].sum() / bd.sum(), 100 * cells[s].sum() / cells.sum()))
