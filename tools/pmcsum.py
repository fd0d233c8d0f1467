"""per-kernel sums of a rocprofv3 --pmc counter_collection CSV: python tools/pmcsum.py <csv> [<csv> ...]
prints kernel, counter, dispatches, total, per dispatch (kernel names cut at the argument list)"""
import csv, sys, collections, re
agg = collections.defaultdict(lambda: [0.0, set()])
for fn in sys.argv[1:]:
    for r in csv.DictReader(open(fn)):
        kn = re.sub(r"\(.*", "", r["Kernel_Name"])[:80]
        a = agg[(kn, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1].add((fn, r["Dispatch_Id"]))
for (kn, cn), (v, ds) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-80s %-12s dispatches %5d total %16.0f per dispatch %14.0f" % (kn, cn, len(ds), v, v / max(1, len(ds))))
