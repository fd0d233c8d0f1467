"""Generates tests/golden/oracle_ont_small.json: per-stage dumps of the CPU oracle on a small seeded synthetic case
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


if __name__ == "__main__":
    d = build()
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_ont_small.json"), "w") as fh:
        json.dump(d, fh, indent=0)
    print("wrote %d reads" % len(d["reads"]))
