"""text timeline of an MM355_TRACE file: python tools/hosttrace.py <file> [ms_per_char=2] [t_from t_to]"""
import sys, collections
rows = [l.rstrip("\n").split("\t") for l in open(sys.argv[1])]
res = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
ev = [(r[0], r[1], float(r[2]), float(r[3])) for r in rows]
t0 = min(e[2] for e in ev); t1 = max(e[3] for e in ev)
if len(sys.argv) > 4: a, b = t0 + float(sys.argv[3]), t0 + float(sys.argv[4])
else: a, b = t0, t1
sym = {"dpk": "#", "front": "F", "pack": "p", "pre": "r", "align": "a", "dp": "D", "finish": "f", "asm": "s"}
ctxs = sorted(set(e[0] for e in ev))
n = int((b - a) / res) + 1
for c in ctxs:
    line = [" "] * n
    for cc, ph, s, e in ev:
        if cc != c or e < a or s > b: continue
        for k in range(max(0, int((s - a) / res)), min(n, int((e - a) / res) + 1)): line[k] = sym.get(ph, "?")
    print(c[-6:], "".join(line))
# utilisation of the extension turn (GPU saturated) inside the window
iv = sorted((max(s, a), min(e, b)) for cc, ph, s, e in ev if ph == "dpk" and e > a and s < b)
busy = 0.0; cs = ce = None
for s, e in iv:
    if cs is None: cs, ce = s, e
    elif s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
if cs is not None: busy += ce - cs
print("extension rounds hold the GPU %.0f%% of the window (%d rounds)" % (100 * busy / (b - a), len(iv)))
print("legend: # extension kernels running, F front(GPU) p pack r pre_align a align_step D dp round f finish s assemble; %.1f ms per char, window %.0f..%.0f ms" % (res, a - t0, b - t0))
