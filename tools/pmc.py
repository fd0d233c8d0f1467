"""sum PMC counters per kernel from a rocprofv3 rocpd database: python tools/pmc.py <db> [kernel substr]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: [0.0, 0])
for kn, cn, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
    if pat in kn:
        a = agg[(kn.split('(')[0][:40], cn)]; a[0] += v; a[1] += 1
for (kn, cn), (v, n) in sorted(agg.items()):
    print("%-40s %-24s %16.0f  (%d dispatches, %.0f each)" % (kn, cn, v, n, v / n))
