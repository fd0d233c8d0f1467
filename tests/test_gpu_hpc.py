"""MM_I_HPC on the device (the map-pb / ava-pb presets, /root/reference/src/lib.rs:333-336 passes any preset to mm_set_opt): homopolymer-
compressed minimizers in the chunked sketch kernels (reads: k_sketch_hpc / k_sketch_sparse_hpc, contigs: k_sketch_contig_hpc), the index
built from them, and U:align.c::mm_adjust_minier's HPC branch in the extension driver -- all against the CPU oracle, bit for bit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
import synthdata as S
from test_host import _hp_genome
from test_gpu_map import check_reads


def _hp_reads(seed, g, n, **kw):
    reads, truth = S.make_reads(seed, g, n, **kw)
    return reads, truth


@pytest.mark.parametrize("sparse_max", ["0", "1000000000"])   # lane-per-chunk grid / one chunk per block (small batches)
def test_hpc_chunked_sketch_parity(built, tmp_path, monkeypatch, sparse_max):
    """the chunk machine's warm-up proof on compressed sequence: chunks that start inside a run, runs across chunk borders, runs longer
    than a chunk, spans >= 256 (no record), N inside and between runs, even k (symmetric k-mers), reads around the chunk size"""
    import mappy_rs
    monkeypatch.setenv("MM355_SKETCH_SPARSE_MAX", sparse_max)
    rng = np.random.default_rng(13)
    g = _hp_genome(73, [50000], repeats=())
    fa = str(tmp_path / "s.fa")
    S.write_fasta(fa, g, ["c"])
    base = S.codes_to_str(g[0][:24000])
    plain = S.codes_to_str(S.random_codes(rng, 4000))
    long_runs = plain[:700] + "A" * 380 + plain[700:1100] + "C" * 800 + plain[1100:1500] + "G" * 255 + plain[1500:1900] + "T" * 256 + plain[1900:]
    with_n = "".join(c if i % 11 else "N" for i, c in enumerate(base[:5000]))
    n_in_run = plain[:900] + "AAAAANAAAAA" + plain[900:1300] + "N" * 400 + "CCCC" + plain[1300:2500]
    pal = "ACGT" * 300 + "AT" * 500 + "GATC" * 200 + "AACCGGTT" * 150
    reads = [base, plain, long_runs, with_n, n_in_run, pal, "A" * 3000, "AC" * 1500, "N" * 700 + base[:900], base[:383], base[:384], base[:385],
             base[:769], "A" * 384 + plain[:500], plain[:380] + "T" * 10 + plain[380:900], S.codes_to_str(S.random_codes(rng, 5000, gc=0.08)), "ACGT", "A"]
    for k, w in ((19, 10), (19, 5), (14, 8), (16, 5), (21, 11), (15, 19)):
        al = mappy_rs.Aligner(fa, preset="map-pb", k=k, w=w)
        orc = O.OracleAligner(fa, preset="map-pb", k=k, w=w)
        sr = al._stage_runner()
        got = sr.sketch(reads)
        n = 0
        for i, rd in enumerate(reads):
            exp = orc.sketch(rd)
            assert got[i].shape == exp.shape and np.array_equal(got[i], exp), (k, w, i)
            n += len(exp)
        assert n > 3000
        sr.close()


def test_hpc_device_index_builder_equals_host_builder_and_oracle(built, tmp_path):
    """the index sketched on the GPU (k_sketch_contig_hpc) == the host builder's == the oracle's: statistics, mid_occ, anchors"""
    import mappy_rs
    from mappy_rs import _ffi
    L = _ffi.lib()
    g = _hp_genome(93, [200000, 900, 120000, 41], repeats=((2500, 6, 0.0), (600, 40, 0.01), (200, 200, 0.02)), n_runs=3)
    names = ["c0", "c1", "c2", "c3"]
    fa = str(tmp_path / "d.fa")
    S.write_fasta(fa, g, names)
    al = mappy_rs.Aligner(fa, preset="map-pb")            # host builder
    orc = O.OracleAligner(fa, preset="map-pb")
    io, mo = _ffi.IdxOpt(), _ffi.MapOpt()
    L.mm355_set_opt(None, C.byref(io), C.byref(mo))
    assert L.mm355_set_opt(b"map-pb", C.byref(io), C.byref(mo)) == 0
    mo.flag |= 4
    seqs = [bytes(bytearray(c.tolist())) for c in g]
    arr = (C.c_char_p * len(seqs))(*seqs)
    lens = (C.c_int64 * len(seqs))(*[len(s) for s in seqs])
    nm = (C.c_char_p * len(seqs))(*[n.encode() for n in names])
    h = C.c_void_p()
    _ffi.check(L.mm355_index_build_device(C.byref(io), len(seqs), arr, lens, nm, 0, C.byref(h)))
    L.mm355_mapopt_update(C.byref(mo), h)
    assert mo.mid_occ == al._mo.mid_occ == orc.mo.mid_occ
    nmz, nd, nmz2, nd2 = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    L.mm355_index_stat(h, C.byref(nmz), C.byref(nd), None, None)
    L.mm355_index_stat(al._idx, C.byref(nmz2), C.byref(nd2), None, None)
    ond = C.c_int64()
    assert (nmz.value, nd.value) == (nmz2.value, nd2.value) == (O.lib().mmo_idx_n_minimizers(orc.idx, C.byref(ond)), ond.value)
    reads, _ = S.make_reads(94, g, 50, n50=3000, lo=200)
    sr_h = al._stage_runner()
    sr_d = _ffi.StageRunner(h, mo, 0)
    a_h, rep_h, _ = sr_h.anchors(reads, sorted_=True)
    a_d, rep_d, _ = sr_d.anchors(reads, sorted_=True)
    tot = 0
    for i, rd in enumerate(reads):
        exp, erep, _, _ = orc.anchors(rd, sorted_=True)
        assert np.array_equal(a_h[i], exp) and np.array_equal(a_d[i], exp) and rep_h[i] == rep_d[i] == erep, i
        tot += len(exp)
    assert tot > 2000
    sr_h.close(); sr_d.close()
    L.mm355_index_free(h)


@pytest.mark.parametrize("preset", ["map-pb", "ava-pb"])
def test_hpc_map_parity(built, tmp_path, preset):
    """whole records (coordinates, CIGAR, NM, cs, MD, MAPQ) with an HPC index: the alignment is cut at the first base of a seed's last
    homopolymer run (mm_adjust_minier) instead of the middle of the k-mer; reads with PacBio-like indel errors inside the runs"""
    import mappy_rs
    g = _hp_genome(53, [300000, 150000], repeats=((4000, 5, 0.01), (1000, 25, 0.02)), n_runs=3)
    fa = str(tmp_path / "ref.fa")
    S.write_fasta(fa, g, ["chr1", "chr2"])
    al = mappy_rs.Aligner(fa, preset=preset)
    orc = O.OracleAligner(fa, preset=preset)
    assert al.k == 19 and al.w == (10 if preset == "map-pb" else 5)
    reads, _ = S.make_reads(54, g, 90 if preset == "map-pb" else 30, n50=5000, lo=300, sub=0.015, ins=0.05, dele=0.03)
    reads += [reads[0][:60], "A" * 500, reads[1][:700] + "N" * 12 + reads[1][700:1600]]
    n_hits, n_sec = check_reads(al, orc, reads)
    assert n_hits >= (80 if preset == "map-pb" else 24)


def _fast_path_chains(al, orc, reads, expect_cull):
    sr = al._stage_runner()
    try:
        got = sr.chains(reads)
        st = sr.stats()
        assert st.n_sort_fast_reads == len(reads), st.n_sort_fast_reads
        assert (0 < st.n_a_kept < st.n_a) if expect_cull else st.n_a_kept == st.n_a, (st.n_a_kept, st.n_a)
        n_short = 0
        for i, rd in enumerate(reads):
            ea, _, _, _ = orc.anchors(rd, sorted_=True)
            eu, eb = orc.chains(ea, len(rd))
            assert np.array_equal(got[i][0], eu) and np.array_equal(got[i][1], eb), i
            n_short += int(((eu & 0xffffffff) < 6).sum())
        return n_short
    finally:
        sr.close()


def test_hpc_index_is_never_culled(built, tmp_path):
    """row a6 on an HPC index: a seed's span is a sum of run lengths (up to 255), so ONE seed can reach min_chain_score and sit in z[]
    while mg_chain_backtrack's unstable sort orders equal scores -- no component can be dropped before the sort (T = ceil(min_chain_score
    / 255) = 1).  With the rule as it read before this test (T = max(min_cnt, ceil(min_chain_score / k))) 3 of these 40 reads got a chain
    with one anchor more or less.  Anchor-rich reads (w = 5, 15-kb reads: the LDS-sort path is taken), chains == the oracle's."""
    import mappy_rs
    g = _hp_genome(57, [400000], repeats=((700, 60, 0.01), (300, 100, 0.02), (2000, 8, 0.005)), n_runs=1)
    fa = str(tmp_path / "c.fa")
    S.write_fasta(fa, g, ["c"])
    reads, _ = S.make_reads(58, g, 40, n50=18000, lo=13000, sub=0.01, ins=0.02, dele=0.02)
    n_short = _fast_path_chains(mappy_rs.Aligner(fa, preset="ava-pb"), O.OracleAligner(fa, preset="ava-pb"), reads, expect_cull=False)
    assert n_short >= 100, n_short
