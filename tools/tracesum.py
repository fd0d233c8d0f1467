"""per-phase summary of an MM355_TRACE file: python tools/tracesum.py <file>  (mean / total duration of every host-side phase)"""
import sys, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    c, ph, s, e = l.rstrip("\n").split("\t")
    d[ph].append(float(e) - float(s))
tot = sum(sum(v) for k, v in d.items() if k in ("front", "pack", "pre", "align", "dp", "finish", "asm"))
for ph, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%-12s n %5d  mean %8.2f ms  total %9.1f ms" % (ph, len(v), sum(v) / len(v), sum(v)))
print("sum of the top-level phases (front pack pre align dp finish asm): %.1f ms" % tot)
