"""Generates tests/golden/oracle_ont_small.json and oracle_presets_small.json (hits of the other presets / extra_flags): per-stage dumps of the CPU oracle on a small seeded synthetic case
(G3 of SURVEY 8c).  These vectors pin GPU == oracle and oracle == its own past self; they are NOT reference output
(the reference's arithmetic, minimap2 2.26, is an un-vendored dependency that cannot be built or imported here).
Run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O   # noqa: E402
import synthdata as S            # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def build():
    g = S.make_genome(101, [60000, 40000], repeats=((1500, 5, 0.01), (400, 20, 0.03)), n_runs=1)
    reads, truth = S.make_reads(102, g, 24, n50=2500, lo=300)
    orc = O.OracleAligner(seqs=[S.codes_to_str(c) for c in g], names=["gA", "gB"], preset="map-ont")
    out = dict(genome_sha=[sha(c) for c in g], mid_occ=orc.mo.mid_occ, reads=[])
    for rd in reads:
        mz = orc.sketch(rd)
        a, rep, mp, _ = orc.anchors(rd)
        f, p, v, _ = orc.chain_fill(a, len(rd))
        u, b = orc.chains(a, len(rd))
        hits = orc.map(rd, cs=True, MD=True)
        out["reads"].append(dict(
            seq=rd, n_mz=len(mz), mz_sha=sha(mz), n_a=len(a), a_sha=sha(a), rep_len=rep, n_mini=len(mp),
            f_sha=sha(f), p_sha=sha(p), u=[int(x) for x in u], chained_sha=sha(b),
            hits=[{k: h[k] for k in ("target_name", "target_start", "target_end", "query_start", "query_end", "strand", "mapq",
                                     "is_primary", "NM", "cigar_str", "cs", "MD", "match_len", "block_len")} for h in hits]))
    return out


HIT_KEYS = ("target_name", "target_start", "target_end", "query_start", "query_end", "strand", "mapq", "is_primary", "NM", "cigar_str",
            "cs", "MD", "match_len", "block_len")
# (label, Aligner kwargs): the other presets and the extra_flags the path honours
PRESET_CASES = (("map-hifi", dict(preset="map-hifi")), ("asm20", dict(preset="asm20")), ("asm5", dict(preset="asm5")),
                ("ava-ont", dict(preset="ava-ont")), ("map-ont+EQX", dict(preset="map-ont", extra_flags=0x4000000)),
                ("map-ont+REV_ONLY+NO_LJOIN", dict(preset="map-ont", extra_flags=0x200000 | 0x400)),
                ("map-ont k13 w7 scoring", dict(preset="map-ont", k=13, w=7, best_n=8, scoring=(3, 6, 5, 2, 20, 1, 2))))


def presets_inputs():
    g = S.make_genome(111, [80000, 30000], repeats=((2500, 4, 0.01), (600, 12, 0.03)), n_runs=1)
    reads, _ = S.make_reads(112, g, 10, n50=4000, lo=400, sub=0.01, ins=0.004, dele=0.004)
    rng = np.random.default_rng(113)
    comp = lambda c: np.where(c < 4, 3 - c, 4).astype(np.uint8)[::-1]
    c = np.concatenate([g[0][10000:13000], comp(g[1][5000:7500]), g[0][15500:18000]])   # chimera with an inverted piece and a 2.5 kb deletion
    reads.append(S.codes_to_str(S.mutate(c, rng, 0.01, 0.004, 0.004)))
    return g, reads


def build_presets():
    g, reads = presets_inputs()
    out = dict(genome_sha=[sha(c) for c in g], reads=reads, cases={})
    for label, kw in PRESET_CASES:
        orc = O.OracleAligner(seqs=[S.codes_to_str(c) for c in g], names=["pA", "pB"], **kw)
        out["cases"][label] = [[{k: h[k] for k in HIT_KEYS} for h in orc.map(rd, cs=True, MD=True)] for rd in reads]
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    dp = build_presets()
    with open(os.path.join(here, "oracle_presets_small.json"), "w") as fh:
        json.dump(dp, fh, indent=0)
    print("presets: %d cases x %d reads, %d hits" % (len(dp["cases"]), len(dp["reads"]), sum(len(h) for c in dp["cases"].values() for h in c)))
    d = build()
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_ont_small.json"), "w") as fh:
        json.dump(d, fh, indent=0)
    print("wrote %d reads" % len(d["reads"]))
