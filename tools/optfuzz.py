"""option fuzzing of the whole path against the oracle: random presets / k / w / chaining and DP thresholds / scoring tuples / extra_flags,
24 reads per configuration (ordinary + chimeric).  python tools/optfuzz.py [seed=1] [n_configs=30] [noisy]   (needs the GPU)"""
import os
import sys

import numpy as np

FIELDS = ("target_name", "target_start", "target_end", "query_start", "query_end", "strand", "target_len", "match_len", "block_len", "mapq",
          "is_primary", "NM", "cs", "MD")
# NO_LJOIN, ALL_CHAINS, NO_END_FLT, HARD_MLEVEL, EQX, NO_INV, FOR_ONLY, REV_ONLY
FLAGS = (0x400, 0x800000, 0x10000000, 0x20000000, 0x4000000, 0x200000000, 0x100000, 0x200000)


def random_config(rng):
    kw = dict(preset=str(rng.choice(["map-ont", "map-hifi", "asm20", "ava-ont"])))
    if rng.random() < 0.5: kw["k"] = int(rng.integers(11, 25))
    if rng.random() < 0.5: kw["w"] = int(rng.integers(3, 30))
    if rng.random() < 0.4: kw["min_cnt"] = int(rng.integers(1, 6))
    if rng.random() < 0.4: kw["min_chain_score"] = int(rng.integers(10, 120))
    if rng.random() < 0.4: kw["min_dp_score"] = int(rng.integers(20, 300))
    if rng.random() < 0.5: kw["bw"] = int(rng.choice([50, 100, 250, 500, 1000, 3000]))
    if rng.random() < 0.4: kw["best_n"] = int(rng.integers(1, 12))
    if rng.random() < 0.3: kw["max_frag_len"] = int(rng.choice([800, 5000, 20000]))
    if rng.random() < 0.6:
        a, b, q, e = int(rng.integers(1, 6)), int(rng.integers(1, 12)), int(rng.integers(1, 14)), int(rng.integers(1, 5))
        sc = [a, b, q, e]
        if rng.random() < 0.7: sc += [int(rng.integers(q, 40)), int(rng.integers(1, e + 1))]
        if len(sc) == 6 and rng.random() < 0.5: sc += [int(rng.integers(0, 4))]
        q2, e2 = (sc[4], sc[5]) if len(sc) >= 6 else ((26, 1) if ("hifi" in kw["preset"] or "asm" in kw["preset"]) else (24, 1))
        lim = 2 * min(q + e, q2 + e2)       # ksw2's domain: beyond it upstream has no defined result (the product refuses it)
        if max(b, sc[6] if len(sc) == 7 else 1) > lim: sc[1] = lim
        kw["scoring"] = tuple(sc)
    fl = 0
    for f in FLAGS:
        if rng.random() < 0.2: fl |= f
    if fl: kw["extra_flags"] = fl
    return kw


def run(seed, n_configs, fa_path, verbose=False, noisy=False):
    """returns (configurations run, hits compared, mismatching reads)"""
    import mappy_rs
    import synthdata as S
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    g = S.make_genome(100 + seed, [300000, 120000], repeats=((4000, 5, 0.01), (1000, 20, 0.02), (250, 60, 0.05)), n_runs=3)
    S.write_fasta(fa_path, g, ["a", "b"])
    comp = lambda c: np.where(c < 4, 3 - c, 4).astype(np.uint8)[::-1]
    tot_hits = tot_bad = 0
    for ci in range(n_configs):
        kw = random_config(rng)
        if noisy: reads, _ = S.make_reads(int(rng.integers(1, 1 << 30)), g, 20, n50=9000, lo=300, sub=0.06, ins=0.04, dele=0.05)   # 15 % error
        else: reads, _ = S.make_reads(int(rng.integers(1, 1 << 30)), g, 20, n50=5000, lo=300)
        for _ in range(4):
            a0, b0 = int(rng.integers(0, 250000)), int(rng.integers(0, 100000))
            c = np.concatenate([g[0][a0:a0 + 2500], comp(g[1][b0:b0 + 2000]), g[0][a0 + 4000:a0 + 6000]])
            reads.append(S.codes_to_str(S.mutate(c, rng, 0.03, 0.01, 0.01)))
        orc = O.OracleAligner(fa_path, **kw)
        al = mappy_rs.Aligner(fa_path, **kw)
        al.enable_threading(2)
        got = al._map_many(reads, 3)
        # cs only: mm_update_extra's walk and the cs string then run on the device (k_extra; MM355_EXTRA_MIN_READS=1 set below) -- unless the
        # configuration asks for '=' / 'X' CIGARs, which keep the host walk like an MD request
        got_dev = al._map_many(reads, 1)
        FD = FIELDS[:-1]
        bad = nh = 0
        for i, rd in enumerate(reads):
            exp = orc.map(rd, cs=True, MD=True)
            nh += len(exp)
            ok = len(exp) == len(got[i]) and all(tuple(getattr(m, k) for k in FIELDS) + (m.cigar_str,) == tuple(e[k] for k in FIELDS) + (e["cigar_str"],)
                                                 for m, e in zip(got[i], exp))
            ok = ok and len(exp) == len(got_dev[i]) and all(tuple(getattr(m, k) for k in FD) + (m.cigar_str,) == tuple(e[k] for k in FD) + (e["cigar_str"],)
                                                            for m, e in zip(got_dev[i], exp))
            bad += not ok
        tot_hits += nh; tot_bad += bad
        if verbose or bad: print("cfg %2d hits %4d mismatching reads %d %s" % (ci, nh, bad, kw), flush=True)
    return n_configs, tot_hits, tot_bad


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "mappy-rs_amd"), os.path.join(root, "tests")): sys.path.insert(0, p)
    os.environ.setdefault("MM355_EXTRA_MIN_READS", "1")
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    print("configs %d, hits %d, mismatching reads %d" % run(seed, n, "/tmp/optfuzz.fa", verbose=True, noisy=len(sys.argv) > 3 and sys.argv[3] == "noisy"))
