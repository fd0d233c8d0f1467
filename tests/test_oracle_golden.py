"""Oracle vs the reference's own fixtures: CPU-only.

G1  test.fa <-> test.mmi : mm_sketch + index key/value encoding + MMI\\2 format, bit-exact
G2  map_one (R:src/lib.rs:1094-1106, R:tests/python_test.py:124-137) + the derived full record
plus the reference's property tests (k, w, n_seq, seq_names, seq).
"""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from oracle import oracle as O

ENTERO = ("AGAGCAGGTAGGATCGTTGAAAAAAGAGTACTCAGGATTCCATTCAACTTTTACTGATTTGAAGCGTAC"
          "TGTTTATGGCCAAGAATATTTACGTCTTTACAACCAATACGCAAAAAAAGGTTCATTGAGTTTGGTTGT"
          "GATTTGATGAAAATTACTGAGAATAACAGGATTATTAAGCTGATTGATGAACTAAATCAGCTTAATAAA"
          "TATTCTTTGCAGATAGGAATATTTGGGGAAAATGATTCTTTTATGGCGATGTTGGCCCAAGTTCATGAA"
          "TTTGGGGTGACTATTCGTCCCAAAGGTCGTTTTCTTGTTATACCACTTATGAAAAAGTATAGAGGTAAA"
          "AGTCCACGTCAATTTGATTTGTTTTTTATGCAAACTAAAGAAAATCACAAGTTTT")
BACILLUS = ("AGAGTGAAGCCAATATTCCGATAACGATTGCTTTCATGATATCCCTCATTCTGGCATTATTTTTTTATA"
            "CTATACTATTCGATATCGCACAGATCAATGGAGTCGTGAGAAAATAAACATGTTTTGCGAACCGCTATG"
            "TGTGGAAGACAAAAAATGGAGGTGAAATTGATGGAAGCAAAGACACAGGCGTACTTTTTTCAGGATGAT"
            "GGCAGGATTCCGAATCACCCTGATTTTCCGCTCGTTGTGTATCAAAACGCACTCAAGGACACCGGTCAG"
            "GCAGAGCGGATCGTCAACCGGCATGGCTGGTCAAACAGCTGGTCGGGGAGTGTTTTTCCATACCATCAT"
            "TATCACAGCAATACGCATGAAGTCCTGATTGCAGTTCGGGGAGAGGCTGTGATTC")


def parse_mmi(fn):
    d = open(fn, "rb").read()
    assert d[:4] == b"MMI\x02"
    w, k, b, n, flag = struct.unpack("<5I", d[4:24])
    o = 24
    seqs = []
    for _ in range(n):
        l = d[o]; o += 1
        nm = d[o:o + l]; o += l
        ln, = struct.unpack("<I", d[o:o + 4]); o += 4
        seqs.append((nm.decode(), ln))
    ent = {}
    for i in range(1 << b):
        nn, = struct.unpack("<i", d[o:o + 4]); o += 4
        p = struct.unpack("<%dQ" % nn, d[o:o + 8 * nn]); o += 8 * nn
        sz, = struct.unpack("<I", d[o:o + 4]); o += 4
        for _ in range(sz):
            kk, vv = struct.unpack("<2Q", d[o:o + 16]); o += 16
            minier = (kk >> 1) << b | i
            ent[minier] = (vv,) if kk & 1 else tuple(p[(vv >> 32):(vv >> 32) + (vv & 0xffffffff)])
    S = d[o:]
    return dict(w=w, k=k, b=b, flag=flag, seqs=seqs, ent=ent, S=S)


def read_fasta(fn):
    out, name, buf = [], None, []
    for line in open(fn):
        if line.startswith(">"):
            if name is not None: out.append((name, "".join(buf)))
            name, buf = line[1:].split()[0], []
        else:
            buf.append(line.strip())
    out.append((name, "".join(buf)))
    return out


@pytest.fixture(scope="module")
def mmi(golden_dir):
    return parse_mmi(os.path.join(golden_dir, "test.mmi"))


@pytest.fixture(scope="module")
def al(golden_dir):
    return O.OracleAligner(os.path.join(golden_dir, "test.mmi"))


def test_fixture_header(mmi):
    assert (mmi["w"], mmi["k"], mmi["b"], mmi["flag"]) == (10, 15, 14, 0)
    assert mmi["seqs"] == [("Bacillus_subtilis", 400), ("Enterococcus_faecalis", 400), ("Escherichia_coli_1", 400), ("Escherichia_coli_2", 400)]
    assert len(mmi["ent"]) == 280 and all(len(v) == 1 for v in mmi["ent"].values())


def test_sketch_reproduces_fixture(mmi, al, golden_dir):
    """mm_sketch over test.fa with rid=0..3 yields exactly the 280 (minimizer -> y) entries of test.mmi"""
    got = {}
    per = []
    for rid, (name, seq) in enumerate(read_fasta(os.path.join(golden_dir, "test.fa"))):
        mz = al.sketch(seq, rid)
        per.append(len(mz))
        for x, y in mz:
            assert int(x) & 0xff == 15
            got.setdefault(int(x) >> 8, []).append(int(y))
    assert per == [66, 75, 73, 66]
    assert {k: tuple(sorted(v)) for k, v in got.items()} == mmi["ent"]


def test_index_rebuild_is_identical(mmi, golden_dir, tmp_path):
    b = O.OracleAligner(os.path.join(golden_dir, "test.fa"))
    out = str(tmp_path / "rebuilt.mmi")
    assert O.lib().mmo_idx_dump(b.idx, out.encode()) == 0
    r = parse_mmi(out)
    assert r["ent"] == mmi["ent"] and r["S"] == mmi["S"] and r["seqs"] == mmi["seqs"]
    assert b.mo.mid_occ == 10


def test_properties(al):
    assert al.k == 15 and al.w == 10 and al.n_seq == 4
    assert sorted(al.seq_names) == ["Bacillus_subtilis", "Enterococcus_faecalis", "Escherichia_coli_1", "Escherichia_coli_2"]
    assert al.seq("Bacillus_subtilis") == BACILLUS
    assert al.seq("Bacillus_subtilis", 10, 20) == BACILLUS[10:20]
    assert al.seq("nope") is None and al.seq("Bacillus_subtilis", 400) is None
    assert al.mo.mid_occ == 10


def test_map_one(al):
    hits = al.map(ENTERO, cs=True)
    assert len(hits) == 1                                   # asserted by the reference
    h = hits[0]
    assert h["target_start"] == 0 and h["target_end"] == 400    # asserted by the reference
    # derived full record
    assert (h["query_start"], h["query_end"], h["strand"], h["target_name"], h["target_len"]) == (0, 400, 1, "Enterococcus_faecalis", 400)
    assert (h["match_len"], h["block_len"], h["NM"], h["mapq"], h["is_primary"]) == (400, 400, 0, 60, True)
    assert h["cigar"] == [(400, 0)] and h["cs"] == ":400" and h["MD"] is None
    assert h["score0"] == 393 and h["cnt"] == 75 and h["dp_max"] == 800


def test_map_all_contigs_and_revcomp(al, golden_dir):
    comp = str.maketrans("ACGT", "TGCA")
    for name, seq in read_fasta(os.path.join(golden_dir, "test.fa")):
        h = al.map(seq, cs=True, MD=True)
        assert len(h) == 1 and h[0]["target_name"] == name and h[0]["cigar_str"] == "400M" and h[0]["strand"] == 1 and h[0]["MD"] == "400"
        h = al.map(seq.translate(comp)[::-1])
        assert len(h) == 1 and h[0]["strand"] == -1 and (h[0]["target_start"], h[0]["target_end"]) == (0, 400)


def test_eqx_cigar(golden_dir):
    """MM_F_EQX (extra_flags=0x4000000, U:align.c::mm_update_cigar_eqx): M is split into '=' / 'X' runs that agree with the
    sequences; merged back they give the default CIGAR, and every other field is unchanged"""
    import re
    fn = os.path.join(golden_dir, "test.mmi")
    plain, eqx = O.OracleAligner(fn), O.OracleAligner(fn, extra_flags=0x4000000)
    rng = np.random.default_rng(5)
    q = list(ENTERO)
    for pos in rng.choice(len(q) - 40, 9, replace=False) + 20:          # substitutions
        q[pos] = "ACGT"[("ACGT".index(q[pos]) + 1 + int(rng.integers(0, 3))) % 4]
    q = "".join(q[:150]) + "TT" + "".join(q[150:260]) + "".join(q[263:])   # an insertion and a deletion
    a, b = plain.map(q, cs=True, MD=True), eqx.map(q, cs=True, MD=True)
    assert len(a) == len(b) == 1
    a, b = a[0], b[0]
    assert "X" in b["cigar_str"] and "=" in b["cigar_str"] and "M" not in b["cigar_str"]
    merged, run = [], 0
    for n, op in re.findall(r"(\d+)([MIDN=X])", b["cigar_str"]):
        if op in "=X": run += int(n)
        else:
            if run: merged.append("%dM" % run); run = 0
            merged.append(n + op)
    if run: merged.append("%dM" % run)
    assert "".join(merged) == a["cigar_str"]
    for k in a:
        if k not in ("cigar", "cigar_str"): assert a[k] == b[k], k
    t = plain.seq(b["target_name"], b["target_start"], b["target_end"])
    qs = q[b["query_start"]:b["query_end"]]
    ti = qi = 0
    for n, op in b["cigar"]:
        if op == 7: assert t[ti:ti + n] == qs[qi:qi + n]
        if op == 8: assert all(x != y for x, y in zip(t[ti:ti + n], qs[qi:qi + n]))
        if op in (0, 7, 8): ti += n; qi += n
        elif op == 1: qi += n
        elif op == 2: ti += n
    assert ti == len(t) and qi == len(qs)


def test_errors(al):
    with pytest.raises(RuntimeError, match="Sequence is empty"):
        al.map("")
    assert al.map("ACGT") == []


def test_region_hash_salt():
    """salt for qlen=400, seed=11, qname=NULL is 0x734db24f"""
    def wang(k):
        k &= 0xffffffff
        k = (k + (~(k << 15) & 0xffffffff)) & 0xffffffff
        k ^= k >> 10
        k = (k + (k << 3)) & 0xffffffff
        k ^= k >> 6
        k = (k + (~(k << 11) & 0xffffffff)) & 0xffffffff
        k ^= k >> 16
        return k
    assert wang((wang(400) + wang(11)) & 0xffffffff) == 0x734db24f


def test_committed_stage_vectors(golden_dir):
    """G3: the oracle reproduces its committed per-stage dumps (regression pin; generator: tests/golden/make_golden.py)"""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(golden_dir, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = json.load(open(os.path.join(golden_dir, "oracle_ont_small.json")))
    got = mg.build()
    assert got["genome_sha"] == want["genome_sha"] and got["mid_occ"] == want["mid_occ"]
    assert len(got["reads"]) == len(want["reads"])
    for g, w in zip(got["reads"], want["reads"]):
        assert g == w


def test_committed_preset_vectors(golden_dir):
    """the oracle reproduces its committed hits for the other presets / extra_flags (tests/golden/oracle_presets_small.json)"""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(golden_dir, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = json.load(open(os.path.join(golden_dir, "oracle_presets_small.json")))
    got = mg.build_presets()
    assert got["genome_sha"] == want["genome_sha"] and got["reads"] == want["reads"]
    assert sorted(got["cases"]) == sorted(want["cases"]) and len(want["cases"]) == len(mg.PRESET_CASES)
    for label in want["cases"]:
        assert got["cases"][label] == want["cases"][label], label
    assert any("X" in h["cigar_str"] for hits in want["cases"]["map-ont+EQX"] for h in hits)
    assert {h["strand"] for hits in want["cases"]["map-ont+REV_ONLY+NO_LJOIN"] for h in hits} == {-1}


def test_hpc_sketch_is_the_plain_sketch_of_the_run_compressed_sequence():
    """MM_I_HPC (map-pb / ava-pb).  The reference holds no fixture for it (test.mmi is a plain index): the HPC branch of the oracle's
    mm_sketch is pinned to its plain branch -- which test.mmi pins -- by what homopolymer compression MEANS: the minimizers of a
    sequence are those of its run-compressed image, at the position of each run's last base, with the run lengths of the k-mer as span."""
    L = O.lib()
    rng = np.random.default_rng(5)
    for k, w in ((19, 10), (15, 5), (14, 8)):
        codes = rng.integers(0, 4, 4000)
        codes = codes[np.r_[True, codes[1:] != codes[:-1]]]                      # the compressed image: no two equal neighbours
        rl = np.ones(len(codes), np.int64)
        pos = rng.integers(0, len(codes), len(codes) // 5)
        rl[pos] = rng.integers(2, 9, len(pos))                                   # (spans stay below 256)
        full = "".join("ACGT"[c] * int(n) for c, n in zip(codes, rl)).encode()
        comp = "".join("ACGT"[c] for c in codes).encode()
        end = np.cumsum(rl) - 1                                                  # last base of every run
        csum = np.r_[0, np.cumsum(rl)]
        def sk(b, hpc):
            v = O.MM128V()
            L.mmo_sketch(b, len(b), w, k, 7, hpc, C.byref(v))
            out = np.ctypeslib.as_array(C.cast(v.a, C.POINTER(C.c_uint64)), shape=(v.n, 2)).copy()
            L.free(v.a)
            return out
        plain, hpc = sk(comp, 0), sk(full, 1)
        assert len(plain) == len(hpc) > 300
        p = (plain[:, 1] & 0xffffffff) >> 1                                      # position in the compressed image
        assert np.all((plain[:, 0] & 0xff) == k)
        assert np.array_equal(hpc[:, 0] >> 8, plain[:, 0] >> 8)                  # same k-mers chosen
        assert np.array_equal((hpc[:, 1] & 0xffffffff) >> 1, end[p].astype(np.uint64))
        assert np.array_equal(hpc[:, 1] & 1, plain[:, 1] & 1) and np.all(hpc[:, 1] >> 32 == 7)
        assert np.array_equal(hpc[:, 0] & 0xff, (csum[p + 1] - csum[p + 1 - k]).astype(np.uint64))
