"""profiles/pmc_traffic.json from the two per-kernel PMC summaries of tools/profile_round.sh (tools/pmcsum.py output of a FETCH_SIZE pass and
of a WRITE_SIZE pass of ONE sub-batch): python tools/pmc2json.py <workload>/<reads per sub-batch> <fetch.txt> <write.txt> [<out.json>]
HBM bytes per launch = FETCH_SIZE x 1024 x c + WRITE_SIZE x 1024 (MI355X_MICROARCH.md: both counters are in KB; on gfx950 FETCH_SIZE tallies a
128-B request as 64 B, i.e. reports half of the bytes of 16-B-per-lane coalesced loads -> c = 2 for the kernels whose fetches are such loads
(k_seed_lookup: eight lanes x dwordx4 = one 128-B table line); c = 1 for every other kernel, whose access widths are uncalibrated: their
figure is a lower bound of the fetched bytes and exact for the written bytes of 16-B stores)."""
import json, os, re, sys
key, ff, wf = sys.argv[1:4]
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
WIDE_LOADS = ("k_seed_lookup",)
# bench.py roofline names <- kernel name prefixes
ALIAS = {"k_ksw_extd2<512> (targets 1025..4096)": "k_ksw_extd2<512>", "k_ksw_extd2<512> (targets > 4096)": "k_ksw_extd2<512>",
         "k_ksw_rowl (targets 1025..8192)": "k_ksw_rowl", "k_ksw_regw8 (exact, band <= 832, targets > 1024)": "k_ksw_regw8",
         "k_ksw_band2 (64 diagonals, two problems per wave)": "k_ksw_band2", "k_ksw_band<1> (128 diagonals)": "k_ksw_band<1>",
         "k_ksw_band<2> (256 diagonals)": "k_ksw_band<2>", "k_ksw_band<4> (512 diagonals)": "k_ksw_band<4>",
         "k_sort_level_mw<1024> (radix_sort_128x emulation, buckets > 16384)": "k_sort_level_mw<mm128, mm_key_x, 1024, 122880>",
         "k_sort_level_mw<256> (radix_sort_128x emulation, buckets > 2048)": "k_sort_level_mw<mm128, mm_key_x, 256, 16448>",
         "k_sort_tasks (radix_sort_128x emulation, one wave per bucket)": "k_sort_tasks<mm128, mm_key_x>"}


def load(fn):
    d = {}
    for ln in open(fn):
        m = re.match(r"(?:void )?(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches\s+(\d+) total\s+(\d+)", ln)
        if m:
            d[m.group(1).strip()] = (int(m.group(3)), float(m.group(4)))
    return d


f, w = load(ff), load(wf)
ent = {}
for kn in sorted(set(f) | set(w)):
    nf, vf = f.get(kn, (0, 0.0)); nw, vw = w.get(kn, (0, 0.0))
    n = max(nf, nw, 1)
    c = 2.0 if any(kn.startswith(p) for p in WIDE_LOADS) else 1.0
    ent[kn] = int((vf * c + vw) * 1024 / n)
for a, k in ALIAS.items():
    if k in ent:
        ent[a] = ent[k]
try:
    tab = json.load(open(out))
except (OSError, ValueError):
    tab = {}
tab[key] = ent
tab["_how"] = __doc__.split("\n", 2)[2].strip()
json.dump(tab, open(out, "w"), indent=1, sort_keys=True)
print("wrote %s [%s]: %d kernels" % (out, key, len(ent)))
