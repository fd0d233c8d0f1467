"""End-to-end parity on the GPU: full Mapping records from the HIP path (through mm355_map_batch) must equal the CPU
oracle's on the same seeded inputs -- coordinates, strand, CIGAR, NM, cs, MD, MAPQ, primary flag (bit-exact).
Also the banded-extension kernel alone vs the oracle's ksw_extd2 restatement on random problems."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
import synthdata as S

FIELDS = ("target_name", "target_start", "target_end", "query_start", "query_end", "strand", "target_len", "match_len",
          "block_len", "mapq", "is_primary", "NM", "cs", "MD")


def rec(m):
    return (m.target_name, m.target_start, m.target_end, m.query_start, m.query_end, m.strand, m.target_len, m.match_len,
            m.block_len, m.mapq, m.is_primary, m.NM, m.cs, m.MD, m.cigar_str)


def orec(o):
    return tuple(o[k] for k in FIELDS) + (o["cigar_str"],)


def check_reads(al, orc, reads, cs=True, MD=True):
    al.enable_threading(2)
    n_hits = n_sec = 0
    batch = al._map_many(reads, (1 if cs else 0) | (2 if MD else 0))
    # an MD request keeps mm_update_extra's walk on the host; the cs-only request (what map_batch makes) leaves it to the device (k_extra):
    # both forms are held against the same oracle records
    batch_dev = al._map_many(reads, 1) if (cs and MD) else None
    for i, rd in enumerate(reads):
        exp = orc.map(rd, cs=cs, MD=MD)
        got = batch[i]
        assert len(got) == len(exp), (i, len(got), len(exp))
        for g, e in zip(got, exp):
            assert rec(g) == orec(e), (i, rec(g), orec(e))
            n_hits += 1
            n_sec += not e["is_primary"]
        if batch_dev is not None:
            assert len(batch_dev[i]) == len(exp), (i, len(batch_dev[i]), len(exp))
            for g, e in zip(batch_dev[i], exp):
                rg, re_ = rec(g), orec(e)
                assert rg[:13] + rg[14:] == re_[:13] + re_[14:] and g.MD is None, (i, rg, re_)
    return n_hits, n_sec


@pytest.fixture(scope="module")
def ont(built, tmp_path_factory):
    import mappy_rs
    td = tmp_path_factory.mktemp("gm")
    g = S.make_genome(51, [500000, 300000], repeats=((5000, 6, 0.01), (1200, 30, 0.02), (300, 80, 0.05)), n_runs=4)
    fa = str(td / "ref.fa")
    S.write_fasta(fa, g, ["chr1", "chr2"])
    return dict(al=mappy_rs.Aligner(fa, preset="map-ont"), orc=O.OracleAligner(fa, preset="map-ont"), g=g, fa=fa)


def test_map_parity_ont(ont):
    reads, _ = S.make_reads(52, ont["g"], 200, n50=6000, lo=200)
    n_hits, n_sec = check_reads(ont["al"], ont["orc"], reads)
    assert n_hits >= 195


def test_map_parity_chimeric_and_sv(ont):
    """reads that force z-drop splits, long-join re-chaining, inversions and secondary hits"""
    g = ont["g"]
    rng = np.random.default_rng(3)
    comp = lambda c: np.where(c < 4, 3 - c, 4).astype(np.uint8)[::-1]
    reads = []
    for _ in range(12):   # chimera of two loci
        a0 = int(rng.integers(0, 400000)); b0 = int(rng.integers(0, 250000))
        c = np.concatenate([g[0][a0:a0 + 3000], comp(g[1][b0:b0 + 2500])])
        reads.append(S.codes_to_str(S.mutate(c, rng, 0.02, 0.01, 0.01)))
    for _ in range(12):   # large deletion / insertion relative to the reference
        a0 = int(rng.integers(0, 400000))
        c = np.concatenate([g[0][a0:a0 + 2500], g[0][a0 + 2500 + 1500:a0 + 6500]])
        reads.append(S.codes_to_str(S.mutate(c, rng, 0.02, 0.01, 0.01)))
        c = np.concatenate([g[0][a0:a0 + 2000], S.random_codes(rng, 800), g[0][a0 + 2000:a0 + 4500]])
        reads.append(S.codes_to_str(S.mutate(c, rng, 0.02, 0.01, 0.01)))
    for _ in range(10):   # inversion in the middle
        a0 = int(rng.integers(0, 400000))
        c = np.concatenate([g[0][a0:a0 + 2500], comp(g[0][a0 + 2500:a0 + 3300]), g[0][a0 + 3300:a0 + 6000]])
        reads.append(S.codes_to_str(S.mutate(c, rng, 0.015, 0.01, 0.01)))
    n_hits, n_sec = check_reads(ont["al"], ont["orc"], reads)
    assert n_hits > len(reads)   # splits produce more than one hit per read


@pytest.mark.parametrize("preset", ["map-ont", "asm20"])
def test_map_parity_tandem_arrays(built, tmp_path, preset):
    """end to end over reads whose re-chain takes every route of row a9: device (order-checked and sorted inner walks), host fallback on equal
    range-minimum priorities, and -- asm20 -- the primary RMQ chainer followed by the long-join pass"""
    import mappy_rs
    from test_gpu_stages import _tandem_world
    fa, reads = _tandem_world(tmp_path)
    n_hits, _ = check_reads(mappy_rs.Aligner(fa, preset=preset), O.OracleAligner(fa, preset=preset), reads[:80])
    assert n_hits >= 80


def test_map_parity_edge_reads(ont):
    g = ont["g"]
    reads = ["ACGT", "A" * 500, "N" * 300, S.codes_to_str(g[0][1000:1400]), S.codes_to_str(g[1][50000:50100]),
             S.codes_to_str(g[0][0:3000]), S.codes_to_str(g[0][-3000:]), S.codes_to_str(g[0][20000:21000]) + "N" * 40 + S.codes_to_str(g[0][21040:22000])]
    check_reads(ont["al"], ont["orc"], reads)
    with pytest.raises(RuntimeError, match="Sequence is empty"):
        ont["al"].map("")


def test_map_parity_hifi(built, tmp_path):
    import mappy_rs
    g = S.make_genome(61, [400000], repeats=((4000, 5, 0.005), (800, 20, 0.01)))
    fa = str(tmp_path / "h.fa")
    S.write_fasta(fa, g, ["chrH"])
    reads, _ = S.make_reads(62, g, 40, n50=12000, lo=3000, sub=0.0005, ins=0.00075, dele=0.00075)
    al = mappy_rs.Aligner(fa, preset="map-hifi")
    orc = O.OracleAligner(fa, preset="map-hifi")
    n_hits, _ = check_reads(al, orc, reads)
    assert n_hits >= 40


@pytest.mark.parametrize("preset", ["asm20", "asm10", "asm5", "ava-ont"])
def test_map_parity_other_presets(built, tmp_path, preset):
    """asm*: MM_F_RMQ makes mg_lchain_rmq the primary chainer (host-resident, from the device-sorted anchors), bw 1000 / bw_long 100000,
    heavier gap costs; ava-ont: all chains kept, no long-join, bw 2000 (qname is NULL through the reference, so NO_DIAG/NO_DUAL are inert)"""
    import mappy_rs
    g = S.make_genome(81, [350000, 150000], repeats=((4000, 4, 0.01), (900, 12, 0.02)), n_runs=2)
    fa = str(tmp_path / "p.fa")
    S.write_fasta(fa, g, ["ctgA", "ctgB"])
    div = {"asm5": 0.002, "asm10": 0.008, "asm20": 0.02, "ava-ont": 0.03}[preset]
    reads, _ = S.make_reads(82, g, 48, n50=15000, lo=1500, sub=div, ins=div / 4, dele=div / 4)
    rng = np.random.default_rng(83)
    for _ in range(6):   # contigs with a large indel: several chains that the long-join / RMQ pass has to bridge
        a0 = int(rng.integers(0, 300000))
        c = np.concatenate([g[0][a0:a0 + 9000], g[0][a0 + 9000 + 2500:a0 + 20000]])
        reads.append(S.codes_to_str(S.mutate(c, rng, div, div / 4, div / 4)))
    al = mappy_rs.Aligner(fa, preset=preset)
    orc = O.OracleAligner(fa, preset=preset)
    n_hits, _ = check_reads(al, orc, reads)
    assert n_hits >= 50


def test_map_parity_ultra_long_reads(built, tmp_path):
    """150 kb - 1 Mb reads (ultra-long ONT) and a 430 kb read with an inverted 130 kb block: the block-level sort, the wave-per-segment
    chainer over very long segments, z-drop splits of one huge chain, and the large-target extension classes"""
    import mappy_rs
    g = S.make_genome(91, [1500000], repeats=((6000, 5, 0.01), (1500, 15, 0.02)), n_runs=2)
    fa = str(tmp_path / "ul.fa")
    S.write_fasta(fa, g, ["chrU"])
    rng = np.random.default_rng(92)
    reads = []
    for L in (150000, 300000, 650000, 1000000):
        a0 = int(rng.integers(0, 1500000 - L))
        reads.append(S.codes_to_str(S.mutate(g[0][a0:a0 + L], rng, 0.024, 0.016, 0.02)))
    comp = lambda c: np.where(c < 4, 3 - c, 4).astype(np.uint8)[::-1]
    c = np.concatenate([g[0][100000:260000], comp(g[0][700000:830000]), g[0][262000:400000]])
    reads.append(S.codes_to_str(S.mutate(c, rng, 0.03, 0.02, 0.02)))
    al = mappy_rs.Aligner(fa, preset="map-ont")
    orc = O.OracleAligner(fa, preset="map-ont")
    n_hits, _ = check_reads(al, orc, reads)
    assert n_hits >= 7


def test_map_parity_adversarial_inputs(built, tmp_path):
    """low-complexity reference and reads (homopolymer, di-/heptanucleotide repeats: masses of equal sort keys and over-represented
    minimizers), IUPAC / lower-case / N-rich reads, thousands of small contigs, duplicate and tiny reads, tandem-duplicated loci"""
    import mappy_rs
    rng = np.random.default_rng(7)
    low = [S.random_codes(rng, 20000), np.zeros(30000, np.uint8), np.tile(np.array([0, 1], np.uint8), 10000),
           np.tile(np.array([0, 1, 2, 3, 3, 1, 0], np.uint8), 4000), S.random_codes(rng, 20000)]
    g = [np.concatenate(low), S.random_codes(rng, 200000)]
    fa = str(tmp_path / "adv1.fa")
    S.write_fasta(fa, g, ["lc", "rnd"])
    lc = [S.codes_to_str(S.mutate(g[0][a:a + L], rng, 0.02, 0.01, 0.01))
          for a, L in ((15000, 12000), (19000, 25000), (45000, 15000), (55000, 20000), (60000, 30000), (70000, 40000), (0, 118000))]
    lc += ["A" * 50000, "AC" * 12000, "ACGTTCA" * 3000]
    base = S.codes_to_str(g[1][1000:9000])
    odd = [base.lower(), base[:3000] + "RYKMSWN" * 5 + base[3035:], "".join(c if rng.random() > 0.02 else "N" for c in base),
           base[:200] + "N" * 3000 + base[3200:], "NNNN" + base[4:60], base[:4000].lower() + base[4000:]]
    dup = [base[:5000]] * 64 + [base[:n] for n in (1, 14, 15, 16, 24, 25, 26, 39, 40, 41, 64, 100)]
    loc = g[1][50000:56000]
    tandem = [S.codes_to_str(S.mutate(np.concatenate([loc, loc, loc]), rng, 0.02, 0.01, 0.01)),
              S.codes_to_str(S.mutate(np.concatenate([g[1][20000:26000], g[1][23000:30000]]), rng, 0.02, 0.01, 0.01))]
    al, orc = mappy_rs.Aligner(fa, preset="map-ont"), O.OracleAligner(fa, preset="map-ont")
    n_hits, _ = check_reads(al, orc, lc + odd + dup + tandem)
    assert n_hits >= 70
    kw = dict(preset="map-ont", k=13, w=5, min_chain_score=20, best_n=10)
    check_reads(mappy_rs.Aligner(fa, **kw), O.OracleAligner(fa, **kw), lc[:7])
    check_reads(mappy_rs.Aligner(fa, preset="map-hifi"), O.OracleAligner(fa, preset="map-hifi"), tandem + odd[:3])
    # thousands of small contigs (rid-rich index), reads within one contig and reads joining two
    gs = [S.random_codes(rng, int(L)) for L in rng.integers(300, 900, 6000)]
    fa3 = str(tmp_path / "adv3.fa")
    S.write_fasta(fa3, gs, ["c%d" % i for i in range(len(gs))])
    r3 = [S.codes_to_str(S.mutate(gs[int(rng.integers(0, len(gs)))], rng, 0.02, 0.01, 0.01)) for _ in range(150)]
    for _ in range(30):
        i, j = (int(x) for x in rng.integers(0, len(gs), 2))
        r3.append(S.codes_to_str(S.mutate(np.concatenate([gs[i], gs[j]]), rng, 0.02, 0.01, 0.01)))
    n_hits, _ = check_reads(mappy_rs.Aligner(fa3, preset="map-ont"), O.OracleAligner(fa3, preset="map-ont"), r3)
    assert n_hits >= 170


@pytest.mark.parametrize("flag", [0x100000, 0x200000, 0x300000])
def test_map_parity_strand_restricted(ont, flag):
    """MM_F_FOR_ONLY / MM_F_REV_ONLY (U:map.c::skip_seed): seed hits of the excluded strand produce no anchors; both = nothing maps"""
    import mappy_rs
    al = mappy_rs.Aligner(ont["fa"], preset="map-ont", extra_flags=flag)
    orc = O.OracleAligner(ont["fa"], preset="map-ont", extra_flags=flag)
    reads, _ = S.make_reads(74, ont["g"], 60, n50=4000, lo=500)
    n_hits, _ = check_reads(al, orc, reads)
    strands = {m.strand for r in reads[:30] for m in al.map(r)}
    assert strands == ({1} if flag == 0x100000 else {-1} if flag == 0x200000 else set()) and (n_hits > 15) == (flag != 0x300000)


@pytest.mark.parametrize("preset", ["map-ont", "asm20"])
def test_map_parity_large_gaps(built, tmp_path, preset):
    """6 - 19 kb of unrelated sequence, or a deletion of that size, in the middle of a 16 kb read: long-join re-chaining across the gap,
    gap fills with one very long side (the eight-wave / HBM-state extension classes), the max_sw_mat rule, z-drop splits"""
    import mappy_rs
    rng = np.random.default_rng(5)
    g = [S.random_codes(rng, 400000)]
    fa = str(tmp_path / "sw.fa")
    S.write_fasta(fa, g, ["c"])
    reads = []
    for gap in (6000, 9000, 10500, 11000, 12000, 15000, 19000):
        a0 = 50000 + gap * 3
        c = np.concatenate([g[0][a0:a0 + 8000], S.random_codes(rng, gap), g[0][a0 + 8000 + gap:a0 + 16000 + gap]])
        reads.append(S.codes_to_str(S.mutate(c, rng, 0.01, 0.005, 0.005)))
        c = np.concatenate([g[0][a0:a0 + 8000], g[0][a0 + 8000 + gap:a0 + 16000 + gap]])
        reads.append(S.codes_to_str(S.mutate(c, rng, 0.01, 0.005, 0.005)))
    n_hits, _ = check_reads(mappy_rs.Aligner(fa, preset=preset), O.OracleAligner(fa, preset=preset), reads)
    assert n_hits >= 21


def test_map_parity_eqx(ont):
    """extra_flags=MM_F_EQX: '='/'X' CIGAR (U:align.c::mm_update_cigar_eqx)"""
    import mappy_rs
    al = mappy_rs.Aligner(ont["fa"], preset="map-ont", extra_flags=0x4000000)
    orc = O.OracleAligner(ont["fa"], preset="map-ont", extra_flags=0x4000000)
    reads, _ = S.make_reads(73, ont["g"], 40, n50=4000, lo=500)
    reads.append("N" * 30 + S.codes_to_str(ont["g"][0][7000:9000]) + "N" * 10)
    n_hits, _ = check_reads(al, orc, reads)
    got = al.map(reads[0])
    assert n_hits >= 40 and "=" in got[0].cigar_str and "M" not in got[0].cigar_str and any(op == 8 for _, op in got[0].cigar)


def test_unsupported_flags_fail_loudly(ont):
    """options outside the long-read path are refused, never mapped with different semantics"""
    import mappy_rs
    for fl in (0x80, 0x100, 0x200, 0x1000, 0x400000, 0x100000000):   # SPLICE, SPLICE_FOR/REV, SR, HEAP_SORT, QSTRAND
        al = mappy_rs.Aligner(ont["fa"], preset="map-ont", extra_flags=fl)
        with pytest.raises(RuntimeError, match="outside the long-read hot path"):
            al.map(S.codes_to_str(ont["g"][0][1000:3000]))


def test_option_fuzz_parity(built, tmp_path):
    """random option sets (presets, k/w, thresholds, scoring tuples, extra_flags) x 24 reads each, bit-exact against the oracle"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("optfuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "optfuzz.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    n_cfg, n_hits, n_bad = fz.run(5, 14, str(tmp_path / "fz.fa"))
    assert n_bad == 0 and n_hits > 300


def test_scoring_outside_ksw2_domain_is_refused(ont):
    """mismatch (or ambiguity) penalty > 2*(q+e): ksw_extd2_sse returns without aligning and mm_align1 dereferences a NULL r->p upstream
    -- there is no defined result to be identical to, so the option set is refused instead of crashing"""
    import mappy_rs
    al = mappy_rs.Aligner(ont["fa"], preset="map-ont", scoring=(5, 11, 3, 1, 7, 1))
    with pytest.raises(RuntimeError):
        al.map(S.codes_to_str(ont["g"][0][1000:3000]))


def test_map_parity_option_overrides(ont):
    import mappy_rs
    kw = dict(preset="map-ont", best_n=2, min_chain_score=60, bw=300, scoring=(2, 5, 5, 3, 20, 1))
    al = mappy_rs.Aligner(ont["fa"], **kw)
    orc = O.OracleAligner(ont["fa"], **kw)
    reads, _ = S.make_reads(72, ont["g"], 40, n50=4000, lo=500)
    check_reads(al, orc, reads)


def test_dp_kernel_parity(ont):
    """k_ksw_extd2 vs the oracle's restatement: extension (left/right) and global fills, approx and exact, band-limited"""
    from mappy_rs import _ffi
    L = _ffi.lib()
    OL = O.lib()
    rng = np.random.default_rng(9)
    al = ont["al"]
    jobs, qs, ts = [], [], []
    EXTZ, RIGHT, REV, APPROX = 0x40, 0x02, 0x80, 0x08
    APPROX_DROP = 0x10
    kinds = [(EXTZ, 751, 400), (EXTZ | RIGHT | REV, 751, 200), (APPROX, 30001, 400), (0, None, 400), (APPROX | RIGHT, 30001, 400),
             (EXTZ | APPROX | APPROX_DROP, 751, 100), (RIGHT, None, 400), (APPROX, None, 400), (EXTZ, None, 60)]
    spans = [(1, 40), (100, 140), (240, 270), (500, 530), (1000, 1040), (1, 700), (900, 2600)]   # around the size-class borders
    for i in range(540):
        lo, hi = spans[i % len(spans)]
        tl = int(rng.integers(lo, hi))
        t = S.random_codes(rng, tl)
        q = S.mutate(t, rng, 0.05, 0.03, 0.03)
        if i % 11 == 0 and len(q) > 20:
            q[len(q) // 2:len(q) // 2 + 3] = 4
        if i % 17 == 0 and tl > 30:
            t[tl // 3:tl // 3 + 2] = 4
        if i % 13 == 0:
            q = np.concatenate([q[:len(q) // 2], S.random_codes(rng, 300)])   # forces z-drop
        if i % 19 == 0:
            q = q[:max(1, len(q) // 3)]                                       # query much shorter than the target
        if len(q) == 0:
            q = S.random_codes(rng, 5)
        flag, w, zd = kinds[i % len(kinds)]
        if w is None:
            w = int(rng.integers(5, 200))
        jobs.append((len(q), tl, w, zd, -1 if i % 5 else 10, flag)); qs.append(q.astype(np.uint8)); ts.append(t.astype(np.uint8))
    # gap fills whose band never binds (KSW_EZ_APPROX_MAX, w >= qlen + tlen, targets <= 1024): the row-sweep kernel k_ksw_row -- every
    # register-set border (128 / 256 / ... / 1024), query much longer / shorter than the target, ambiguous bases, left- and right-aligned
    # gaps, reversed CIGARs, long indels (the second gap piece), empty-ish problems
    for i in range(420):
        tl = int(rng.choice([1, 2, 3, 17, 127, 128, 129, 200, 255, 256, 257, 300, 383, 384, 385, 511, 512, 513, 640, 767, 768, 769, 1023, 1024])) if i % 3 == 0 else int(rng.integers(1, 1025))
        t = S.random_codes(rng, tl)
        q = S.mutate(t, rng, 0.06, 0.03, 0.03)
        if i % 7 == 0 and len(q) > 40:
            cut = int(rng.integers(5, len(q) - 30)); q = np.concatenate([q[:cut], q[cut + int(rng.integers(1, 30)):]])        # deletion
        if i % 7 == 3:
            cut = int(rng.integers(0, len(q) + 1)); q = np.concatenate([q[:cut], S.random_codes(rng, int(rng.integers(1, 120))), q[cut:]])   # insertion
        if i % 10 == 0:
            q = S.random_codes(rng, int(rng.integers(1, 900)))                 # unrelated query, any length ratio
        if i % 11 == 0 and len(q) > 6:
            q[len(q) // 2:len(q) // 2 + 3] = 4
        if i % 13 == 0 and tl > 6:
            t[tl // 3:tl // 3 + 2] = 4
        if len(q) == 0:
            q = S.random_codes(rng, 1)
        flag = APPROX | (RIGHT if i % 2 else 0) | (REV if i % 4 == 1 else 0)
        jobs.append((len(q), tl, len(q) + tl + int(rng.integers(0, 50)), 400, -1, flag)); qs.append(q.astype(np.uint8)); ts.append(t.astype(np.uint8))
    # the same with targets of 1025..4096 bases: the eight-wave row sweep k_ksw_rowl (one 512-column panel per wave, 64-row batches):
    # every panel border, one to eight panels, queries shorter than / equal to / just over one batch and up to the LDS limit of 5120 rows
    n_short = len(jobs)
    for i in range(52):   # (the last eight: targets beyond 4096 -- a wave takes a second panel)
        tl = int([1025, 1536, 1537, 2047, 2048, 2049, 2560, 2561, 3000, 3583, 3584, 3585, 4000, 4095, 4096][i % 15]) if i < 30 else int(rng.integers(1025, 4097)) if i < 44 else \
             int([4097, 4608, 4609, 5000, 5120, 6000, 7000, 8192][i - 44])
        t = S.random_codes(rng, tl)
        q = S.mutate(t, rng, 0.06, 0.03, 0.03)
        if i % 5 == 0:
            cut = int(rng.integers(5, len(q) - 600)); q = np.concatenate([q[:cut], q[cut + int(rng.integers(1, 500)):]])        # deletion
        if i % 5 == 3:
            cut = int(rng.integers(0, len(q) + 1)); q = np.concatenate([q[:cut], S.random_codes(rng, int(rng.integers(1, 700))), q[cut:]])   # insertion
        if i % 8 == 1:
            q = q[:int([1, 63, 64, 65, 128, 700][(i // 8) % 6])]               # a few rows only
        if i % 8 == 6:
            q = S.random_codes(rng, int(rng.integers(1, 5121)))                # unrelated query, any length ratio
        if len(q) > 5120:
            q = q[:5120]                                                       # the kernel's row limit (its LDS column buffers)
        if i % 6 == 0:
            q[len(q) // 2:len(q) // 2 + 3] = 4
        if i % 7 == 0:
            t[tl // 3:tl // 3 + 2] = 4
        flag = APPROX | (RIGHT if i % 2 else 0) | (REV if i % 4 == 1 else 0)
        jobs.append((len(q), tl, len(q) + tl + int(rng.integers(0, 50)), 400, -1, flag)); qs.append(q.astype(np.uint8)); ts.append(t.astype(np.uint8))
    # exact sweeps with a narrow band over long targets: k_ksw_regw, the register kernel whose 1024-position window follows the band
    # (extensions of read ends: w = 751; the widest band it takes, 832; narrow ones; band-limited global fills; z-drop half way; queries
    # much shorter / longer than the target, so that the band leaves the matrix early; ambiguous bases across a window move)
    n_rowl = len(jobs)
    for i in range(40):
        tl = int([1025, 1100, 1151, 1152, 1153, 2000, 2047, 2048, 3000, 5000, 7000, 9000][i % 12]) if i < 24 else int(rng.integers(1025, 6000))
        t = S.random_codes(rng, tl)
        q = S.mutate(t, rng, 0.05, 0.03, 0.03)
        if i % 6 == 1:
            cut = int(rng.integers(400, len(q) - 400)); q = np.concatenate([q[:cut], q[cut + int(rng.integers(20, 300)):]])        # deletion: the path moves off the main diagonal
        if i % 6 == 4:
            cut = int(rng.integers(400, len(q) - 400)); q = np.concatenate([q[:cut], S.random_codes(rng, int(rng.integers(20, 300))), q[cut:]])
        if i % 7 == 2:
            q = np.concatenate([q[:len(q) * 2 // 3], S.random_codes(rng, 900)])      # z-drop after two thirds
        if i % 9 == 3:
            q = q[:int(rng.integers(1, 900))]                                      # short query: band limited by the query
        if i % 9 == 5:
            q = np.concatenate([q, S.random_codes(rng, 1500)])                     # query runs past the target
        if i % 5 == 0:
            q[len(q) // 2:len(q) // 2 + 3] = 4; t[tl // 2 + 100:tl // 2 + 102] = 4
        flag, w, zd = [(EXTZ, 751, 400), (EXTZ | RIGHT | REV, 751, 200), (0, 832, 400), (RIGHT, 300, 400), (EXTZ, 16, 100), (0, 751, 10000)][i % 6]
        jobs.append((len(q), tl, w, zd, -1 if i % 4 else 10, flag)); qs.append(q.astype(np.uint8)); ts.append(t.astype(np.uint8))
    # ... and two whose query + target need more than 64 KB of LDS (the eight-wave kernel stages both sequences there: dynamic LDS beyond the
    # default limit, opted into per launch); qlen * tlen stays below max_sw_mat (beyond it the stage answers "z-dropped" without aligning)
    for tl, fl in ((65000, EXTZ), (64800, EXTZ | RIGHT | REV)):
        t = S.random_codes(rng, tl)
        q = S.mutate(t[:1450], rng, 0.05, 0.03, 0.03)[:1500]
        jobs.append((len(q), tl, 751, 100000, -1, fl)); qs.append(q.astype(np.uint8)); ts.append(t.astype(np.uint8))
    # paths along the matrix border and along the band edge while the band spans five or more 128-cell blocks (the catch-all instance
    # of the register kernels): a global alignment that opens with a 560-base deletion runs through the top-row cells t = r >= 512, whose
    # y / u are boundary values; one that opens with a (w - 1)-base insertion rides the lower band edge, where x[st - 1] / v[st - 1] are
    # defaults whenever st did not move
    for i in range(24):
        tl = int([1020, 1000, 900, 3000, 2500, 5000][i % 6])
        t = S.random_codes(rng, tl)
        m = S.mutate(t, rng, 0.03, 0.01, 0.01)
        w = 751
        if i % 4 == 0: q = m[int([560, 600, 700, 520][(i // 4) % 4]):]
        elif i % 4 == 1: q = np.concatenate([S.random_codes(rng, w - 1 - (i // 4) % 3), m])
        elif i % 4 == 2: q = np.concatenate([m[:len(m) // 2], S.random_codes(rng, 700), m[len(m) // 2:]])   # the same in the middle of the matrix
        else: q = np.concatenate([m[:len(m) // 3], m[len(m) // 3 + 650:]])
        flag = [0, RIGHT, EXTZ, EXTZ | RIGHT | REV][(i // 2) % 4]
        jobs.append((len(q), tl, w, 100000, -1, flag)); qs.append(q.astype(np.uint8)); ts.append(t.astype(np.uint8))
    qcat = np.concatenate(qs); tcat = np.concatenate(ts)
    ja = (_ffi.DpJob * len(jobs))()
    qo = to = 0
    for i, (ql, tl, w, zd, eb, fl) in enumerate(jobs):
        ja[i].qlen, ja[i].tlen, ja[i].qoff, ja[i].toff, ja[i].w, ja[i].zdrop, ja[i].end_bonus, ja[i].flag = ql, tl, qo, to, w, zd, eb, fl
        qo += ql; to += tl
    res = (_ffi.DpRes * len(jobs))()
    cap = int(qcat.size + tcat.size + 4 * len(jobs))
    cig = np.zeros(cap, np.uint32)
    sr = al._stage_runner()
    _ffi.check(L.mm355_stage_dp(sr.ctx, C.byref(al._mo), len(jobs), ja, qcat.ctypes.data, qcat.size, tcat.ctypes.data, tcat.size, res, cig.ctypes.data, cap))
    st = sr.stats()
    if not os.environ.get("MM355_DP_BAND_FORCE"):
        assert st.n_launch_group[17] == 1 and st.dp_cells_group[17] + st.dp_cells_group[21] + st.dp_cells_group[20] >= sum(j[0] * j[1] for j in jobs[n_short:n_rowl]), "long full-band fills must run on k_ksw_rowl (or a band)"
    assert st.n_launch_group[18] == 1 and st.dp_cells_group[18] > 0, "long narrow-band exact sweeps must run on k_ksw_regw"
    mat = np.zeros(25, np.int8)
    mo = al._mo
    OL.mmo_ksw_gen_simple_mat(5, mat.ctypes.data, mo.a, mo.b, mo.sc_ambi)
    n_zd = 0
    for i, (ql, tl, w, zd, eb, fl) in enumerate(jobs):
        ez = O.Extz()
        OL.mmo_ksw_extd2(ql, qs[i].ctypes.data, tl, ts[i].ctypes.data, 5, mat.ctypes.data, mo.q, mo.e, mo.q2, mo.e2, w, zd, eb, fl, C.byref(ez))
        emax, ezd = ez.max_zd & 0x7fffffff, ez.max_zd >> 31
        r = res[i]
        assert (r.max, r.zdropped, r.max_q, r.max_t, r.mqe, r.mqe_t, r.mte, r.mte_q, r.score, r.reach_end, r.n_cigar) == \
               (emax, ezd, ez.max_q, ez.max_t, ez.mqe, ez.mqe_t, ez.mte, ez.mte_q, ez.score, ez.reach_end, ez.n_cigar), (i, jobs[i])
        exp = [ez.cigar[k] for k in range(ez.n_cigar)]
        assert list(cig[r.cigar_off:r.cigar_off + r.n_cigar]) == exp, i
        n_zd += ezd
        if ez.n_cigar: OL.free(ez.cigar)
    assert n_zd > 0
    groups = list(sr.stats().n_launch_group)
    st = sr.stats()
    forced = os.environ.get("MM355_DP_BAND_FORCE") or os.environ.get("MM355_DP_BAND") == "0"
    if not forced:
        assert groups[14] > 0 and groups[15] > 0 and groups[16] > 0, groups      # k_ksw_row<2>, <4> and <8> ran ...
        assert groups[19] + groups[22] > 0 and st.n_dp_band > 50, (groups, st.n_dp_band)          # ... and the band kernel with its sufficiency proof (mm355_dpband.h)
        assert 0 < st.n_dp_band_redo < st.n_dp_band and groups[23] == 1, (st.n_dp_band_redo, st.n_dp_band, groups)   # some proofs fail (long indels): run again on the full matrix
    elif os.environ.get("MM355_DP_BAND") == "0":
        assert groups[19] + groups[20] + groups[21] + groups[22] == 0 and st.n_dp_band == 0
    else:
        k = {"64": 22, "1": 19, "2": 20, "4": 21}[os.environ["MM355_DP_BAND_FORCE"]]
        assert groups[k] > 0 and st.n_dp_band > 300 and st.n_dp_band_redo > 20, (groups, st.n_dp_band, st.n_dp_band_redo)   # forced: every problem whose end cell fits the band tries it
    sr.close()


@pytest.mark.parametrize("env", [{"MM355_DP_BAND_FORCE": "64"}, {"MM355_DP_BAND_FORCE": "1"}, {"MM355_DP_BAND_FORCE": "2"}, {"MM355_DP_BAND_FORCE": "4"}, {"MM355_DP_BAND": "0"}])
def test_dp_band_kernels_forced(built, env):
    """the 1000 problems of test_dp_kernel_parity once more (a child process: the switches are read once) with every full-band fill pushed onto a
    band of 128 / 256 / 512 diagonals whenever its end cell fits -- unrelated sequences, 700-base insertions and all: the proof must fail for
    those and the second run on the full matrix must deliver the oracle's result -- and with the band kernels switched off"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_map.py", "-x", "-q", "-k", "test_dp_kernel_parity or test_dp_row_kernel_with_reordered_gap_costs"],
                       cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_dp_row_kernel_with_reordered_gap_costs(ont):
    """k_ksw_row under options ksw2 re-orders (q + e > q2 + e2: the pieces are swapped, the absolute score keeps the given q + e) and under
    a single-piece cost (q == q2, e == e2); an irregular cost (e == e2, q != q2) must stay on the literal kernels"""
    from mappy_rs import _ffi
    import copy
    L = _ffi.lib(); OL = O.lib()
    rng = np.random.default_rng(31)
    al = ont["al"]
    for (a, b, amb, q, e, q2, e2), expect_row in (((4, 6, 1, 10, 1, 3, 4), True), ((2, 5, 1, 5, 2, 5, 2), True), ((4, 10, 1, 3, 3, 12, 3), False)):
        mo = copy.copy(al._mo)
        mo.a, mo.b, mo.sc_ambi, mo.q, mo.e, mo.q2, mo.e2 = a, b, amb, q, e, q2, e2
        qs, ts, jobs = [], [], []
        for i in range(120):
            tl = int(rng.integers(1, 513)); t = S.random_codes(rng, tl); x = S.mutate(t, rng, 0.06, 0.03, 0.03)
            if i % 6 == 0 and len(x) > 40:
                cut = int(rng.integers(5, len(x) - 30)); x = np.concatenate([x[:cut], x[cut + int(rng.integers(1, 40)):]])
            if i % 9 == 0 and len(x) > 6: x[len(x) // 2:len(x) // 2 + 2] = 4
            if len(x) == 0: x = S.random_codes(rng, 1)
            qs.append(x.astype(np.uint8)); ts.append(t.astype(np.uint8)); jobs.append((len(x), tl, len(x) + tl + 1, 8 | (2 if i % 2 else 0)))
        for i, tl in enumerate((1300, 2100, 4090)):   # long targets: k_ksw_rowl where its int16 range allows (checked per problem on the host)
            t = S.random_codes(rng, tl); x = S.mutate(t, rng, 0.06, 0.03, 0.03)
            qs.append(x.astype(np.uint8)); ts.append(t.astype(np.uint8)); jobs.append((len(x), tl, len(x) + tl + 1, 8 | (2 if i % 2 else 0)))
        qcat = np.concatenate(qs); tcat = np.concatenate(ts)
        ja = (_ffi.DpJob * len(jobs))()
        qo = to = 0
        for i, (ql, tl, w, fl) in enumerate(jobs):
            ja[i].qlen, ja[i].tlen, ja[i].qoff, ja[i].toff, ja[i].w, ja[i].zdrop, ja[i].end_bonus, ja[i].flag = ql, tl, qo, to, w, 400, -1, fl
            qo += ql; to += tl
        res = (_ffi.DpRes * len(jobs))(); cap = int(qcat.size + tcat.size + 4 * len(jobs)); cig = np.zeros(cap, np.uint32)
        sr = al._stage_runner()
        _ffi.check(L.mm355_stage_dp(sr.ctx, C.byref(mo), len(jobs), ja, qcat.ctypes.data, qcat.size, tcat.ctypes.data, tcat.size, res, cig.ctypes.data, cap))
        groups = list(sr.stats().n_launch_group)
        assert (groups[14] + groups[15] + groups[16] + groups[19] + groups[20] + groups[21] + groups[22] > 0) == expect_row, (groups, (q, e, q2, e2))
        # k_ksw_rowl takes a long problem only where its int16 range allows (mm355_dp.hip::rowl_range_ok, restated here): with
        # (4, 6, 1, 10, 1, 3, 4) -- match 4, e = 4 after ksw2's ordering -- 4096 columns do not fit, nor with the single piece (5, 2)
        def range_ok(ql, tl):
            qq, ee, qq2, ee2 = (q, e, q2, e2) if q + e <= q2 + e2 else (q2, e2, q, e)
            tlr = (tl + 127) & ~127
            cost = lambda k: min(qq + k * ee, qq2 + k * ee2)
            emax, hi = max(ee, ee2), a * min(ql, tlr)
            lo = cost(ql + 1) + cost(tlr + 1) + max(qq + ee, qq2 + ee2, b, amb) + emax + 64
            return hi + emax * (tlr + 1) <= 32000 and lo <= 16384 - 64 and hi + lo <= 32000
        cells17 = sr.stats().dp_cells_group[17]
        want17 = sum(j[0] * j[1] for j in jobs[-3:] if range_ok(j[0], j[1])) if expect_row else 0
        if os.environ.get("MM355_DP_BAND") == "0":        # (a band kernel takes a long problem whose proof looks within reach; without them: exactly these)
            assert cells17 == want17 and (not expect_row or 0 < want17 < sum(j[0] * j[1] for j in jobs[-3:])), (cells17, want17, (q, e, q2, e2))
        else:
            assert cells17 <= want17, (cells17, want17, (q, e, q2, e2))
        mat = np.zeros(25, np.int8); OL.mmo_ksw_gen_simple_mat(5, mat.ctypes.data, a, b, amb)
        for i, (ql, tl, w, fl) in enumerate(jobs):
            ez = O.Extz()
            OL.mmo_ksw_extd2(ql, qs[i].ctypes.data, tl, ts[i].ctypes.data, 5, mat.ctypes.data, q, e, q2, e2, w, 400, -1, fl, C.byref(ez))
            assert res[i].score == ez.score and list(cig[res[i].cigar_off:res[i].cigar_off + res[i].n_cigar]) == [ez.cigar[k] for k in range(ez.n_cigar)], (i, jobs[i], (q, e, q2, e2))
            if ez.n_cigar: OL.free(ez.cigar)
        sr.close()


def test_stats_counters(ont):
    reads, _ = S.make_reads(82, ont["g"], 30, n50=3000, lo=500)
    al = ont["al"]
    al._map_many(reads, 1)
    from mappy_rs import _ffi
    st = _ffi.Stats()
    _ffi.check(al._L.mm355_get_stats(al._ctx, C.byref(st)))
    tot = dict(n_mz=0, n_hit=0, n_a=0, n_a_multi=0)
    for rd in reads:
        ont["orc"].map(rd)
        s = ont["orc"].stats()
        for k in tot: tot[k] += getattr(s, k)
    assert (st.n_mz, st.n_hit, st.n_a, st.n_a_multi) == (tot["n_mz"], tot["n_hit"], tot["n_a"], tot["n_a_multi"])
    assert st.dp_cells > 0 and st.n_dp_jobs > 0 and st.ms_seed_lookup > 0


def test_round_cut_by_hbm_budget_is_counted_and_exact(ont, monkeypatch):
    """an extension round whose direction matrices do not fit the HBM budget is cut into several launches: slower, never different -- and
    counted (mm355_stats_t::n_rounds_split), so that a workload that outgrows the budget shows up in the numbers instead of halving the rate
    silently"""
    from mappy_rs import _ffi
    reads, _ = S.make_reads(84, ont["g"], 40, n50=3000, lo=500)
    al = ont["al"]
    ref = [[rec(m) for m in ms] for ms in al._map_many(reads, 1)]
    st = _ffi.Stats(); _ffi.check(al._L.mm355_get_stats(al._ctx, C.byref(st)))
    assert st.n_rounds_split == 0 and st.n_ext_rounds >= 1
    monkeypatch.setenv("MM355_DP_BUDGET_MB", "1")
    cut = [[rec(m) for m in ms] for ms in al._map_many(reads, 1)]
    _ffi.check(al._L.mm355_get_stats(al._ctx, C.byref(st)))
    assert st.n_rounds_split >= 1, st.n_rounds_split
    assert cut == ref


def test_map_batch_pipeline_workers(ont, monkeypatch):
    """map_batch over several worker contexts (sub-batches of 7 reads, 4 host threads) returns what the sequential path returns
    (results stream in completion order, as in the reference: compared by id), and equals the oracle for a sample"""
    import mappy_rs
    reads, _ = S.make_reads(93, ont["g"], 45, n50=2500, lo=300)
    reads[5] = ""                                  # empty read: no result for it (worker error in the reference)
    al = ont["al"]
    items = [{"seq": r, "id": i} for i, r in enumerate(reads)]
    al.enable_threading(1)
    seq = sorted((i["id"], [rec(m) for m in ms]) for ms, i in al.map_batch(items))
    monkeypatch.setattr(mappy_rs, "SUB_BATCH_READS", 7)
    al.enable_threading(4)
    par = sorted((i["id"], [rec(m) for m in ms]) for ms, i in al.map_batch(iter(items)))
    assert par == seq and len(par) == len(reads) - 1 and all(k != 5 for k, _ in par)
    assert sum(len(v) for v in al._wctx.values()) == 4          # every worker context went back to the pool
    for k in (0, 9, 44):
        assert [r for r in dict(par)[k]] == [orec(o) for o in ont["orc"].map(reads[k], cs=True)]


def test_resident_batch_slots(ont):
    """mm355_batch_select: two batches resident in ONE context, mapped alternately, give the hits of a one-shot mm355_map_batch"""
    from mappy_rs import _ffi
    al = ont["al"]; L = al._L
    ra, _ = S.make_reads(91, ont["g"], 24, n50=3000, lo=500)
    rb, _ = S.make_reads(92, ont["g"], 17, n50=5000, lo=300)
    ctx = C.c_void_p()
    _ffi.check(L.mm355_ctx_create(al._idx, 0, C.byref(ctx)))
    try:
        packs = [_ffi.pack_reads(ra), _ffi.pack_reads(rb)]
        for slot, (rarr, rlens, keep) in enumerate(packs):
            _ffi.check(L.mm355_batch_select(ctx, slot))
            _ffi.check(L.mm355_batch_upload(ctx, len(keep), rarr, rlens))

        def hits_of(hp, n):
            h = hp.contents
            off = np.ctypeslib.as_array(h.hit_off, shape=(n + 1,)).copy()
            rec = [(h.hits[k].rid, h.hits[k].target_start, h.hits[k].target_end, h.hits[k].query_start, h.hits[k].query_end, h.hits[k].strand,
                    h.hits[k].mapq, h.hits[k].n_cigar, h.hits[k].NM) for k in range(int(h.n_hits))]
            L.mm355_free_hits(hp)
            return list(off), rec

        expect = []
        ctx2 = C.c_void_p()
        _ffi.check(L.mm355_ctx_create(al._idx, 0, C.byref(ctx2)))
        for rarr, rlens, keep in packs:   # reference: a fresh context, one-shot upload + map
            hp = C.POINTER(_ffi.Hits)()
            _ffi.check(L.mm355_map_batch(ctx2, C.byref(al._mo), len(keep), rarr, rlens, 1, C.byref(hp)))
            expect.append(hits_of(hp, len(keep)))
        L.mm355_ctx_destroy(ctx2)
        for order in ((0, 1), (1, 0, 1)):
            for slot in order:
                _ffi.check(L.mm355_batch_select(ctx, slot))
                hp = C.POINTER(_ffi.Hits)()
                _ffi.check(L.mm355_map_resident(ctx, C.byref(al._mo), 1, C.byref(hp)))
                assert hits_of(hp, len(packs[slot][2])) == expect[slot], slot
        assert sum(len(e[1]) for e in expect) > 20
    finally:
        L.mm355_ctx_destroy(ctx)


def test_map_parity_reads_running_into_n_runs(built, tmp_path):
    """the reference lives in HBM as a 2-bit image plus a sorted table of its N runs: reads that end inside a run, start inside one, span a
    short one, and single ambiguous bases scattered in the target (runs of length 1, several per extension window) against the oracle,
    which reads the 4-bit image"""
    import mappy_rs
    rng = np.random.default_rng(77)
    g = S.make_genome(131, [260000, 140000, 30], repeats=((3000, 4, 0.01), (800, 10, 0.02)), n_runs=0)
    runs = [(0, 40000, 700), (0, 90000, 37), (0, 150000, 1), (1, 20000, 1500), (1, 139000, 1000), (0, 0, 120)]   # (contig, start, length); the last: a contig that starts with N
    for ci, st, ln in runs:
        g[ci][st:st + ln] = 4
    for p_ in range(200000, 203000, 61):         # lone ambiguous bases, ~50 in a 3 kb stretch
        g[0][p_] = 4
    g[2][:] = 4                                    # a contig of nothing but N
    fa = str(tmp_path / "n.fa")
    S.write_fasta(fa, g, ["c0", "c1", "c2"])
    al = mappy_rs.Aligner(fa, preset="map-ont")
    orc = O.OracleAligner(fa, preset="map-ont")
    reads = []
    def take(ci, a, b, rc=False):
        c = g[ci][max(0, a):b].copy()
        n = c > 3
        c[n] = rng.integers(0, 4, int(n.sum()))     # the read has real bases where the assembly has none
        c = S.mutate(c, rng, 0.04, 0.02, 0.02)
        if rc: c = S._COMP[c[::-1]]
        reads.append(S.codes_to_str(c))
    for ci, st, ln in runs:
        take(ci, st - 4000, st + ln // 2)            # ends inside the run
        take(ci, st + ln // 2, st + ln + 4000, True)  # starts inside it
        take(ci, st - 3000, st + ln + 3000)           # spans it
        take(ci, st - 2500, st + 10, True)            # touches its first bases
    take(0, 198000, 205000); take(0, 199500, 204000, True); take(0, 200100, 201900)
    n_hits, _ = check_reads(al, orc, reads)
    assert n_hits >= 20
    from mappy_rs import _ffi
    assert al.seq("c0", 39990, 40010) == "".join("ACGT"[x] for x in g[0][39990:40000]) + "N" * 10   # (host image, U:index.c::mm_idx_getseq)


def test_committed_golden_vectors_on_gpu(built, golden_dir):
    """the HIP path reproduces the committed golden hits (tests/golden/oracle_ont_small.json) without running the oracle"""
    import json
    import mappy_rs
    want = json.load(open(os.path.join(golden_dir, "oracle_ont_small.json")))
    g = S.make_genome(101, [60000, 40000], repeats=((1500, 5, 0.01), (400, 20, 0.03)), n_runs=1)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "g.fa")
        S.write_fasta(fa, g, ["gA", "gB"])
        al = mappy_rs.Aligner(fa, preset="map-ont")
        res = al._map_many([r["seq"] for r in want["reads"]], 3)
        for got, w in zip(res, want["reads"]):
            assert len(got) == len(w["hits"])
            for m, h in zip(got, w["hits"]):
                assert (m.target_name, m.target_start, m.target_end, m.query_start, m.query_end, m.strand, m.mapq, m.is_primary, m.NM,
                        m.cigar_str, m.cs, m.MD, m.match_len, m.block_len) == tuple(h[k] for k in (
                            "target_name", "target_start", "target_end", "query_start", "query_end", "strand", "mapq", "is_primary", "NM",
                            "cigar_str", "cs", "MD", "match_len", "block_len"))


def test_committed_preset_vectors_on_gpu(built, golden_dir, tmp_path):
    """the HIP path reproduces the committed hits of the other presets / extra_flags (oracle_presets_small.json) without running the oracle"""
    import importlib.util
    import json
    import mappy_rs
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(golden_dir, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = json.load(open(os.path.join(golden_dir, "oracle_presets_small.json")))
    g, reads = mg.presets_inputs()
    assert reads == want["reads"]
    fa = str(tmp_path / "p.fa")
    S.write_fasta(fa, g, ["pA", "pB"])
    for label, kw in mg.PRESET_CASES:
        al = mappy_rs.Aligner(fa, **kw)
        res = al._map_many(reads, 3)
        for got, w in zip(res, want["cases"][label]):
            assert [tuple(getattr(m, k) for k in mg.HIT_KEYS) for m in got] == [tuple(h[k] for k in mg.HIT_KEYS) for h in w], label


def test_full_size_properties(built, tmp_path):
    """BASELINE configs[1] shape at bench size (E. coli-like genome, 8192 ONT-like reads): properties that need no oracle --
    (1) the hits of a read do not depend on which other reads share its batch, (2) a second run is identical, (3) every CIGAR spans
    exactly its query and target intervals, (4) cs agrees with the CIGAR and NM, (5) reads map back to where they were drawn."""
    import re
    from mappy_rs import _ffi
    import mappy_rs
    g = S.ecoli_like(1)
    fa = str(tmp_path / "ecoli.fa")
    S.write_fasta(fa, g)
    reads, truth = S.make_reads(2, g, 8192, n50=8000)
    al = mappy_rs.Aligner(fa, preset="map-ont")
    L = al._L
    ctx = C.c_void_p()
    _ffi.check(L.mm355_ctx_create(al._idx, 0, C.byref(ctx)))

    def run(sub):
        rarr, rlens, keep = _ffi.pack_reads(sub)
        hp = C.POINTER(_ffi.Hits)()
        _ffi.check(L.mm355_map_batch(ctx, C.byref(al._mo), len(sub), rarr, rlens, 1, C.byref(hp)))
        h = hp.contents
        off = np.ctypeslib.as_array(h.hit_off, shape=(len(sub) + 1,)).copy()
        hits = np.frombuffer(C.string_at(h.hits, int(h.n_hits) * C.sizeof(_ffi.Hit)), dtype=mappy_rs._HIT_DTYPE).copy()
        cig = np.ctypeslib.as_array(h.cigar, shape=(max(int(h.n_cigar), 1),)).copy()
        sbuf = C.string_at(h.str, int(h.n_str)) if h.n_str else b""
        L.mm355_free_hits(hp)
        return off, hits, cig, sbuf

    def per_read(off, hits, cig, sbuf):
        out = []
        for i in range(len(off) - 1):
            rs = []
            for k in range(off[i], off[i + 1]):
                x = hits[k]
                rs.append((int(x["rid"]), int(x["target_start"]), int(x["target_end"]), int(x["query_start"]), int(x["query_end"]), int(x["strand"]),
                           int(x["mapq"]), int(x["NM"]), cig[x["cigar_off"]:x["cigar_off"] + x["n_cigar"]].tobytes(),
                           sbuf[x["cs_off"]:x["cs_off"] + x["cs_len"]]))
            out.append(rs)
        return out

    try:
        whole = per_read(*run(reads))
        assert per_read(*run(reads)) == whole                                  # (2)
        cuts = [0, 1000, 1003, 5000, len(reads)]
        parts = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            parts += per_read(*run(reads[a:b]))
        assert parts == whole                                                  # (1)
        n_hits = n_right = 0
        for i, rs in enumerate(whole):
            for (rid, ts, te, qs, qe, strand, mapq, nm, cg, cs) in rs:
                ops = np.frombuffer(cg, dtype=np.uint32)
                ln, op = ops >> 4, ops & 0xf
                assert int(ln[(op == 0) | (op == 1)].sum()) == qe - qs and int(ln[(op == 0) | (op == 2)].sum()) == te - ts   # (3)
                cs = cs.decode()
                n_match = sum(int(v) for v in re.findall(r":(\d+)", cs))
                n_sub = cs.count("*")
                n_ins = sum(len(v) for v in re.findall(r"\+([acgtn]+)", cs)); n_del = sum(len(v) for v in re.findall(r"-([acgtn]+)", cs))
                assert n_match + n_sub == int(ln[op == 0].sum()) and n_ins == int(ln[op == 1].sum()) and n_del == int(ln[op == 2].sum())   # (4)
                assert nm == n_sub + n_ins + n_del
                n_hits += 1
            if rs:
                ci, st, en, sd = truth[i]
                rid, ts, te, qs, qe, strand, mapq, *_ = rs[0]
                n_right += (min(te, en) - max(ts, st)) > 0.5 * (en - st) and (strand > 0) == (sd > 0)
        assert n_hits >= 8000 and n_right >= 0.97 * len(reads)                 # (5)
    finally:
        L.mm355_ctx_destroy(ctx)


def test_repeat_rich_genome_device_index_parity(built, tmp_path):
    """GRCh38-like miniature (24 contigs, SINE/LINE/satellite families, N runs; 12 Mbp): index built ON THE DEVICE from memory vs the
    oracle's index from the FASTA; full records of repeat-rich reads -- thousands of anchors per read, equal-key ties, the block-level
    and the segmented-sort paths -- must equal the oracle's"""
    from mappy_rs import _ffi
    import mappy_rs
    L = _ffi.lib()
    g, names = S.make_human_like(3, 0.004)
    fa = str(tmp_path / "mini.fa")
    S.write_fasta(fa, g, names)
    reads, _ = S.make_reads(4, g, 160, n50=6000, lo=500)
    io, mo = _ffi.IdxOpt(), _ffi.MapOpt()
    L.mm355_set_opt(None, C.byref(io), C.byref(mo))                      # defaults first, then the preset (as mm_set_opt is used upstream)
    _ffi.check(L.mm355_set_opt(b"map-ont", C.byref(io), C.byref(mo))); mo.flag |= 4
    ptrs = (C.c_char_p * len(g))(*[C.cast(c.ctypes.data, C.c_char_p) for c in g])
    lens = (C.c_int64 * len(g))(*[len(c) for c in g]); nm = (C.c_char_p * len(g))(*[n.encode() for n in names])
    idx = C.c_void_p()
    _ffi.check(L.mm355_index_build_device(C.byref(io), len(g), ptrs, lens, nm, 0, C.byref(idx)))
    L.mm355_mapopt_update(C.byref(mo), idx)
    orc = O.OracleAligner(fa, preset="map-ont")
    ctx = C.c_void_p()
    _ffi.check(L.mm355_ctx_create(idx, 0, C.byref(ctx)))
    try:
        rarr, rlens, keep = _ffi.pack_reads(reads)
        hp = C.POINTER(_ffi.Hits)()
        _ffi.check(L.mm355_map_batch(ctx, C.byref(mo), len(reads), rarr, rlens, 1, C.byref(hp)))
        got = mappy_rs._batch_to_mappings(hp, len(reads), names)
        L.mm355_free_hits(hp)
        st = _ffi.Stats(); L.mm355_get_stats(ctx, C.byref(st))
        assert st.n_a / len(reads) > 800, "the miniature must be repeat-rich (anchors per read: %.0f)" % (st.n_a / len(reads))
        n_hits = 0
        for i in range(0, len(reads), 2):
            exp = orc.map(reads[i], cs=True)
            assert len(got[i]) == len(exp), (i, len(got[i]), len(exp))
            for m, e in zip(got[i], exp):
                assert (m.target_name, m.target_start, m.target_end, m.query_start, m.query_end, m.strand, m.mapq, m.is_primary, m.NM, m.cigar_str, m.cs) == \
                       (e["target_name"], e["target_start"], e["target_end"], e["query_start"], e["query_end"], e["strand"], e["mapq"], e["is_primary"],
                        e["NM"], e["cigar_str"], e["cs"]), i
                n_hits += 1
        assert n_hits > 60
    finally:
        L.mm355_ctx_destroy(ctx)
        L.mm355_index_free(idx)


def test_many_zdrop_splits_need_more_than_64_rounds(built, tmp_path):
    """ADVICE r1 / VERDICT r1: the extension loop used to stop after 64 rounds and emit unaligned regions silently.  A read that is
    co-linear with the reference but carries ~90 diverged blocks forces a z-drop split per block (each costs a round or two); the result
    must still equal the oracle's, and the stats must show that more than 64 rounds ran."""
    from mappy_rs import _ffi
    import mappy_rs
    rng = np.random.default_rng(123)
    g = S.make_genome(61, [400000], repeats=())
    c = g[0]
    seg = c[20000:20000 + 140000].copy()
    for b in range(90):                                   # 700 random bases every 1500: no anchors inside, a z-drop in every gap fill
        p0 = 800 + b * 1500
        seg[p0:p0 + 700] = S.random_codes(rng, 700, 0.5)
    rd = S.codes_to_str(S.mutate(seg, rng, 0.01, 0.003, 0.003))
    fa = str(tmp_path / "z.fa")
    S.write_fasta(fa, g, ["chrZ"])
    al = mappy_rs.Aligner(fa, preset="map-ont")
    orc = O.OracleAligner(fa, preset="map-ont")
    ctx = C.c_void_p()
    _ffi.check(al._L.mm355_ctx_create(al._idx, 0, C.byref(ctx)))
    try:
        rarr, rlens, keep = _ffi.pack_reads([rd])
        hp = C.POINTER(_ffi.Hits)()
        _ffi.check(al._L.mm355_map_batch(ctx, C.byref(al._mo), 1, rarr, rlens, 3, C.byref(hp)))
        got = mappy_rs._batch_to_mappings(hp, 1, al._names())[0]
        al._L.mm355_free_hits(hp)
        st = _ffi.Stats(); al._L.mm355_get_stats(ctx, C.byref(st))
        exp = orc.map(rd, cs=True, MD=True)
        assert len(got) == len(exp) and len(exp) > 40, (len(got), len(exp))
        for m, e in zip(got, exp):
            assert rec(m) == orec(e)
        assert st.n_ext_rounds > 64, st.n_ext_rounds
    finally:
        al._L.mm355_ctx_destroy(ctx)


def test_no_seq_index_is_refused(built, golden_dir, tmp_path):
    """MM_I_NO_SEQ index with MM_F_CIGAR (always set): no target sequence to extend against -> error, never garbage"""
    import mappy_rs
    from test_oracle_golden import ENTERO
    mmi = bytearray(open(os.path.join(golden_dir, "test.mmi"), "rb").read())
    mmi[20:24] = (2).to_bytes(4, "little")
    f = tmp_path / "noseq.mmi"; f.write_bytes(bytes(mmi[:-((4 * 400 + 7) // 8 * 4)]))
    al = mappy_rs.Aligner(str(f))
    with pytest.raises(RuntimeError):
        al.map(ENTERO)


def test_map_batch_streams_results(ont):
    """SURVEY 8 f3 / lib.rs:793-839: results leave the iterator while later sub-batches are still being mapped; every input yields exactly
    one tuple carrying a COPY of its dict; completion order may differ from input order"""
    import time
    al = ont["al"]
    reads, _ = S.make_reads(57, ont["g"], 6000, n50=2500, lo=200)
    items = [{"seq": r, "id": i} for i, r in enumerate(reads)]
    al.enable_threading(2)
    it = al.map_batch(iter(items))                    # unknown length: ramped sub-batches
    first = next(it)
    t_first = time.perf_counter()
    rest = list(it)
    assert it.n_sub_batches >= 4
    assert t_first < max(it.t_sub_done), "the first result must be available before the last sub-batch has been mapped"
    out = [first] + rest
    assert sorted(d["id"] for _, d in out) == list(range(len(items)))
    assert all(d is not items[d["id"]] and d == items[d["id"]] for _, d in out)
    # same records as the single-call path
    by_id = {d["id"]: m for m, d in out}
    ref = al._map_many(reads[:64], 1)
    for i in range(64):
        assert [rec(m) for m in by_id[i]] == [rec(m) for m in ref[i]]
    with pytest.raises(StopIteration):
        next(it)
    # a worker-side failure surfaces from the iterator; a validation failure raises from map_batch itself and leaves the aligner usable
    with pytest.raises(ValueError, match="`seq` must be a string"):
        al.map_batch(items[:3000] + [{"seq": 5}])
    assert len(list(al.map_batch(items[:10]))) == 10


def test_multi_device_api_on_one_gpu(ont):
    """Aligner(devices=[...]) / mm355_upload: replicas are per device and shared by the contexts of that device (exercised with the one GPU
    of this box; the N-GPU dealing logic itself is covered on CPU by tests/test_multi_gpu_logic.py)"""
    import mappy_rs
    from mappy_rs import _ffi
    al = mappy_rs.Aligner(ont["fa"], preset="map-ont", devices=[0])
    L = al._L
    arr = (C.c_int32 * 1)(0)
    assert L.mm355_upload(al._idx, arr, 1) == 0 and L.mm355_upload(al._idx, arr, 1) == 0      # idempotent
    bad = (C.c_int32 * 1)(L.mm355_device_count())
    assert L.mm355_upload(al._idx, bad, 1) == _ffi.MM355_ENODEV
    reads, _ = S.make_reads(58, ont["g"], 300, n50=3000, lo=200)
    al.enable_threading(3)
    got = {d["i"]: m for m, d in al.map_batch([{"seq": r, "i": i} for i, r in enumerate(reads)])}
    ref = ont["al"]._map_many(reads, 1)
    assert all([rec(m) for m in got[i]] == [rec(m) for m in ref[i]] for i in range(len(reads)))


def _log2f_approx(x):
    """U:mmpriv.h::mg_log2 in float32 arithmetic (the gap cost of mm_update_extra)"""
    z = np.array([x], np.float32).view(np.uint32)
    log_2 = np.float32(int((z[0] >> 23) & 255) - 128)
    z[0] &= np.uint32(~(255 << 23) & 0xffffffff); z[0] += np.uint32(127 << 23)
    f = z.view(np.float32)[0]
    t = np.float32(-0.34484843) * f
    t = np.float32(t + np.float32(2.02466578)); t = np.float32(t * f); t = np.float32(t - np.float32(0.67487759))
    return np.float32(log_2 + t)


def _update_extra_ref(q, t, cigar, a, b, amb, go, ge):
    """plain restatement of U:align.c::mm_update_extra (the walk after mm_fix_cigar) and U:format.c::write_cs_core (short form)"""
    s = mx = 0.0
    mlen = blen = n_ambi_tot = 0
    qo = to = 0
    cs = []
    nt = "acgtn"
    for op, ln in cigar:
        if op == 0:
            n_ambi = n_diff = run = 0
            for l in range(ln):
                cq, ct = int(q[qo + l]), int(t[to + l])
                if ct > 3 or cq > 3: n_ambi += 1; sc = -amb
                elif ct != cq: n_diff += 1; sc = -b
                else: sc = a
                s += sc
                if s < 0: s = 0.0
                else: mx = max(mx, s)
                if cq == ct: run += 1
                else:
                    if run: cs.append(":%d" % run); run = 0
                    cs.append("*" + nt[ct] + nt[cq])
            if run: cs.append(":%d" % run)
            blen += ln - n_ambi; mlen += ln - (n_ambi + n_diff); n_ambi_tot += n_ambi; qo += ln; to += ln
        elif op == 1:
            n_ambi = int((q[qo:qo + ln] > 3).sum()); blen += ln - n_ambi; n_ambi_tot += n_ambi
            cs.append("+" + "".join(nt[int(x)] for x in q[qo:qo + ln]))
            s -= go + float(ge) * float(_log2f_approx(np.float32(1.0 + ln)))
            if s < 0: s = 0.0
            qo += ln
        elif op == 2:
            n_ambi = int((t[to:to + ln] > 3).sum()); blen += ln - n_ambi; n_ambi_tot += n_ambi
            cs.append("-" + "".join(nt[int(x)] for x in t[to:to + ln]))
            s -= go + float(ge) * float(_log2f_approx(np.float32(1.0 + ln)))
            if s < 0: s = 0.0
            to += ln
    return mlen, blen, n_ambi_tot, int(mx + .499), "".join(cs)


@pytest.mark.gpu
def test_update_extra_and_cs_on_device(ont):
    """k_extra (row f2) through its stage entry: mlen / blen / n_ambi / dp_max and the cs string of regions with given CIGARs -- one to
    several hundred operations (1 .. 6 segments of 64), long gaps (log cost), runs of mismatches (the score clamps at 0), ambiguous bases,
    regions that begin or end with a gap, a region without operations, match operations of up to 12 000 columns (cut between lanes: runs of
    matches, the cs number and the MD count cross the cuts) -- against the oracle's own mm_update_extra walk, write_cs_core and write_MD_core"""
    from mappy_rs import _ffi
    L = _ffi.lib()
    rng = np.random.default_rng(41)
    al = ont["al"]
    t_all = np.asarray(ont["g"][0], np.uint8)     # codes 0..4 of chr1 (with N runs)
    jobs, qs, cigs, refs = [], [], [], []
    for i in range(72):
        n_ops = int([0, 1, 2, 63, 64, 65, 127, 128, 129, 300][i % 10]) if i < 30 else int(rng.integers(1, 400)) if i < 60 else int(rng.integers(1, 12))
        t_st = int(rng.integers(0, len(t_all) - (200000 if i >= 60 else 60000)))
        ops, q, to = [], [], 0
        for k in range(n_ops):
            last = ops[-1][0] if ops else -1
            op = 0 if (k % 2 == 0 and last != 0) else int(rng.choice([1, 2]))
            if op == last: op = 0
            if i % 7 == 3 and k == 0: op = int(rng.choice([1, 2]))        # region that begins with a gap
            ln = int(rng.integers(1, 40)) if op == 0 else int(rng.choice([1, 1, 2, 3, 10, 60, 700, 3000]))
            if i >= 60 and op == 0: ln = int(rng.choice([500, 1792, 2048, 2049, 2304, 4096, 7000, 12000]))   # HiFi-like: match operations longer than a segment (cut between lanes)
            if op == 0:
                seg = t_all[t_st + to:t_st + to + ln].copy()
                mm = rng.random(ln) < (0.6 if (i % 5 == 1 and k % 8 < 4) else 0.08 if i < 60 else (0.0 if i % 3 == 0 else 0.002))     # stretches of mismatches: s falls back to 0; long operations: runs of matches across the cuts
                seg[mm] = (seg[mm] + rng.integers(1, 4, int(mm.sum()))) % 4
                if i % 6 == 2: seg[rng.random(ln) < 0.05] = 4
                q.append(seg); to += ln
            elif op == 1: q.append(S.random_codes(rng, ln).astype(np.uint8))
            else: to += ln
            ops.append((op, ln))
        qa = np.concatenate(q).astype(np.uint8) if q else np.zeros(0, np.uint8)
        jobs.append((t_st, ops)); qs.append(qa)
    qcat = np.concatenate(qs + [np.zeros(8, np.uint8)])
    cig = np.array([ln << 4 | op for _, ops in jobs for op, ln in ops] + [0], np.uint32)
    ja = (_ffi.ExtraJob * len(jobs))()
    qo = co = 0
    for i, (t_st, ops) in enumerate(jobs):
        ja[i].q_off, ja[i].cigar_off, ja[i].rid, ja[i].t_st, ja[i].n_cigar = qo, co, 0, t_st, len(ops)
        qo += len(qs[i]); co += len(ops)
    res = (_ffi.ExtraRes * len(jobs))()
    cap = int(5 * (qcat.size + sum(ln for _, ops in jobs for op, ln in ops)) + 64 * len(cig))
    cs = np.zeros(cap, np.uint8)
    mo = al._mo
    sr = al._stage_runner()
    _ffi.check(L.mm355_stage_extra(sr.ctx, C.byref(mo), len(jobs), ja, qcat.ctypes.data, qcat.size, cig.ctypes.data, len(cig) - 1, 3, res, cs.ctypes.data, cap))   # cs and MD
    for i, (t_st, ops) in enumerate(jobs):
        t_len = sum(ln for op, ln in ops if op != 1)
        exp = O.update_extra_stage(qs[i], t_all[t_st:t_st + t_len + 8], ops, mo.a, mo.b, mo.sc_ambi, mo.q, mo.e)   # the oracle's own walk + write_cs_core
        got = (res[i].mlen, res[i].blen, res[i].n_ambi, res[i].dp_max, bytes(cs[res[i].cs_off:res[i].cs_off + res[i].cs_len]).decode(),
               bytes(cs[res[i].md_off:res[i].md_off + res[i].md_len]).decode())
        assert got == exp, (i, len(ops), got[:4], exp[:4], got[4][:80], exp[4][:80], got[5][:80], exp[5][:80])
        assert exp[:5] == _update_extra_ref(qs[i], t_all[t_st:], ops, mo.a, mo.b, mo.sc_ambi, mo.q, mo.e), i       # (and the independent restatement in this file)
    sr.close()


def test_update_extra_host_walk(ont, monkeypatch):
    """the host form of mm_update_extra / cs (what batches below 1024 reads, MD and EQX requests use) against the oracle and against the
    device form on the same reads"""
    reads, _ = S.make_reads(77, ont["g"], 60, n50=5000, lo=300)
    al = ont["al"]
    dev = [[(m.r_st, m.r_en, m.mlen, m.blen, m.NM, m.mapq, m.cs) for m in al.map(r, cs=True)] for r in reads]
    monkeypatch.setenv("MM355_EXTRA_HOST", "1")
    n_hits, _ = check_reads(al, ont["orc"], reads)
    host = [[(m.r_st, m.r_en, m.mlen, m.blen, m.NM, m.mapq, m.cs) for m in al.map(r, cs=True)] for r in reads]
    assert n_hits > 50 and host == dev


def test_more_contexts_than_the_stream_pool_holds(ont):
    """mm355_ctx_create hands the main and the sort stream of the first eight contexts of a device out of a pool (two streams of one context
    on one hardware queue, mm355_pipeline.hip); the ninth and later contexts get streams of their own, and a destroyed context's pool slot
    goes to the next one created.  Eleven contexts map the same reads at the same time, three are destroyed and created again: every result
    equals the first."""
    import threading
    from mappy_rs import _ffi
    al = ont["al"]
    L = al._L
    reads, _ = S.make_reads(97, ont["g"], 40, n50=4000, lo=300)
    rarr, rlens, keep = _ffi.pack_reads(reads)

    def run(ctx, out, k):
        hp = C.POINTER(_ffi.Hits)()
        _ffi.check(L.mm355_map_batch(ctx, C.byref(al._mo), len(keep), rarr, rlens, 1, C.byref(hp)))
        h = hp.contents
        off = list(np.ctypeslib.as_array(h.hit_off, shape=(len(keep) + 1,)))
        rec = [(h.hits[i].rid, h.hits[i].target_start, h.hits[i].target_end, h.hits[i].query_start, h.hits[i].query_end, h.hits[i].strand, h.hits[i].mapq,
                h.hits[i].n_cigar, h.hits[i].NM) for i in range(int(h.n_hits))]
        cig = list(np.ctypeslib.as_array(h.cigar, shape=(max(1, int(h.n_cigar)),))[:int(h.n_cigar)])
        L.mm355_free_hits(hp)
        out[k] = (off, rec, cig)

    ctxs = []
    try:
        for _ in range(11):
            c = C.c_void_p()
            _ffi.check(L.mm355_ctx_create(al._idx, 0, C.byref(c)))
            ctxs.append(c)
        for rnd in range(2):
            out = [None] * len(ctxs)
            th = [threading.Thread(target=run, args=(c, out, k)) for k, c in enumerate(ctxs)]
            for t in th: t.start()
            for t in th: t.join()
            assert all(o is not None for o in out) and len(out[0][1]) > 20
            for k in range(1, len(out)):
                assert out[k] == out[0], (rnd, k)
            if rnd == 0:   # slots 1, 4 and 6 of the pool become free and are taken again
                for k in (1, 4, 6):
                    L.mm355_ctx_destroy(ctxs[k])
                    ctxs[k] = None
                for k in (1, 4, 6):
                    c = C.c_void_p()
                    _ffi.check(L.mm355_ctx_create(al._idx, 0, C.byref(c)))
                    ctxs[k] = c
    finally:
        for c in ctxs:
            if c is not None:
                L.mm355_ctx_destroy(c)
