"""summary of a rocprofv3 rocpd database grouped by (kernel, lds_size, workgroup): python tools/ksum.py <db>"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count from kernels order by start").fetchall()
agg = collections.OrderedDict()
for r in rows:
    k = (r[0].split('(')[0][:50], r[5], r[4], r[6])
    a = agg.setdefault(k, [0, 0.0, 0, 0.0])
    a[0] += 1; a[1] += (r[2] - r[1]) / 1e6; a[2] += r[3] // max(1, r[4]); a[3] = max(a[3], (r[2] - r[1]) / 1e6)
tot = sum(a[1] for a in agg.values())
print("total kernel time %.1f ms over %d launches; span %.1f ms" % (tot, len(rows), (rows[-1][2] - rows[0][1]) / 1e6))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-50s lds %6d wg %4d vgpr %3d  n %5d  total %9.2f ms (%4.1f%%)  avg %8.3f max %8.3f  blocks/launch %8d" % (k[0], k[1], k[2], k[3], a[0], a[1], 100 * a[1] / tot, a[1] / a[0], a[3], a[2] // a[0]))
