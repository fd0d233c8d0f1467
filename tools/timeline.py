"""GPU concurrency timeline of a rocprofv3 kernel trace (rocpd): python tools/timeline.py <db> [t_from_ms t_to_ms]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
t0 = rows[0][1]
if len(sys.argv) > 3:
    a, b = t0 + float(sys.argv[2]) * 1e6, t0 + float(sys.argv[3]) * 1e6
    rows = [r for r in rows if r[2] > a and r[1] < b]
    t0 = rows[0][1]
t1 = max(r[2] for r in rows)
ev = []
for i, r in enumerate(rows): ev.append((r[1], 1, i)); ev.append((r[2], -1, i))
ev.sort()
h = collections.Counter(); alone = collections.Counter(); act = set(); last = ev[0][0]
for t, d, i in ev:
    h[len(act)] += t - last
    if len(act) == 1: alone[rows[next(iter(act))][0].split('(')[0][:40]] += t - last
    last = t
    if d == 1: act.add(i)
    else: act.discard(i)
span = (t1 - t0) / 1e6
print("span %.1f ms, idle %.1f ms (%.0f%%), sum of kernel durations %.1f ms" % (span, h[0] / 1e6, 100 * h[0] / 1e6 / span, sum(r[2] - r[1] for r in rows) / 1e6))
for k in sorted(h): print("  %2d kernels running: %8.1f ms" % (k, h[k] / 1e6))
for n, v in alone.most_common(8): print("  alone: %-40s %8.1f ms" % (n, v / 1e6))
