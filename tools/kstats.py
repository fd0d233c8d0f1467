"""per-kernel rows of a rocprofv3 rocpd database: python tools/kstats.py <db> [substr ...]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pats = sys.argv[2:]
rows = db.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count from kernels order by start").fetchall()
for r in rows:
    nm = r[0].split('(')[0][:56]
    if not pats or any(p in nm for p in pats):
        print("%-56s %9.3f ms  blocks %7d x %4d lds %6d vgpr %3d start %9.3f" % (nm, (r[2] - r[1]) / 1e6, r[3] // max(1, r[4]), r[4], r[5], r[6], (r[1] - rows[0][1]) / 1e6))
