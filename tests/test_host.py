"""CPU-only checks of the product library: it loads, exports every symbol include/mm355.h declares,
and its host side (index load/build, flat table, options, accessors) agrees with the oracle and the
reference's fixtures.  No compute call is made here (no GPU in this container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import oracle as O
from test_oracle_golden import BACILLUS, parse_mmi, read_fasta


@pytest.fixture(scope="module")
def ffi(built):
    from mappy_rs import _ffi
    return _ffi


def test_library_exports_header_symbols(ffi):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "mm355.h")).read()
    declared = set(re.findall(r"\b(mm355_[a-z0-9_]+)\s*\(", hdr))
    assert declared and declared == set(ffi.EXPORTS)
    L = ffi.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.mm355_version()


def test_aligner_properties_and_seq(ffi, golden_dir):
    import mappy_rs
    al = mappy_rs.Aligner(os.path.join(golden_dir, "test.mmi"))
    assert al and al.k == 15 and al.w == 10 and al.n_seq == 4
    assert sorted(al.seq_names) == ["Bacillus_subtilis", "Enterococcus_faecalis", "Escherichia_coli_1", "Escherichia_coli_2"]
    assert al.seq("Bacillus_subtilis") == BACILLUS
    assert al.seq("Bacillus_subtilis", 5, 25) == BACILLUS[5:25]
    assert al.seq("Bacillus_subtilis", 390, 1000) == BACILLUS[390:]
    assert al.seq("missing") is None and al.seq("Bacillus_subtilis", 400) is None and al.seq("Bacillus_subtilis", 10, 10) is None
    assert al._mo.mid_occ == 10 and al._mo.flag & 4


def test_constructor_errors(ffi, golden_dir):
    import mappy_rs
    with pytest.raises(RuntimeError, match="Did not create or open an index"):
        mappy_rs.Aligner()
    with pytest.raises(RuntimeError, match="Did not create or open an index"):
        mappy_rs.Aligner("/nonexistent/file.mmi")
    with pytest.raises(NotImplementedError, match="Not Implemented"):
        mappy_rs.Aligner(os.path.join(golden_dir, "test.mmi"), seq="ACGT")
    with pytest.raises(NotImplementedError, match="Not Implemented"):
        mappy_rs.Aligner(os.path.join(golden_dir, "test.mmi"), fn_idx_out="x.mmi")
    al = mappy_rs.Aligner(os.path.join(golden_dir, "test.mmi"))
    with pytest.raises(NotImplementedError, match="Using `seq2` is not implemented"):
        al.map("ACGT", seq2="ACGT")
    with pytest.raises(RuntimeError, match="Multi threading not enabled"):
        al.map_batch([{"seq": "ACGT"}])


def test_option_presets_match_oracle(ffi):
    L = ffi.lib()
    for preset in (None, "map-ont", "map-hifi", "asm20", "asm5", "ava-ont", "map-pb", "map10k", "ava-pb"):
        io, mo = ffi.IdxOpt(), ffi.MapOpt()
        L.mm355_set_opt(None, C.byref(io), C.byref(mo))
        oio, omo = O.IdxOpt(), O.MapOpt()
        O.lib().mmo_set_opt(None, C.byref(oio), C.byref(omo))
        if preset:
            assert L.mm355_set_opt(preset.encode(), C.byref(io), C.byref(mo)) == 0
            assert O.lib().mmo_set_opt(preset.encode(), C.byref(oio), C.byref(omo)) == 0
        assert (io.k, io.w, io.flag, io.bucket_bits) == (oio.k, oio.w, oio.flag, oio.bucket_bits)
        for name, _ in ffi.MapOpt._fields_:
            assert getattr(mo, name) == getattr(omo, name), (preset, name)
    before = bytes(mo), bytes(io)
    for known in (b"sr", b"splice", b"splice:hq", b"short", b"cdna"):
        assert L.mm355_set_opt(known, C.byref(io), C.byref(mo)) == ffi.MM355_EUNSUP
    for unknown in (b"map-ontt", b"asm7", b""):
        assert L.mm355_set_opt(unknown, C.byref(io), C.byref(mo)) == ffi.MM355_EINVAL    # U:options.c::mm_set_opt returns -1
    assert (bytes(mo), bytes(io)) == before                                               # ... and leaves the options untouched


def test_unsupported_presets_raise(ffi, golden_dir):
    """a preset minimap2 knows but this path does not implement must not be mapped with other parameters"""
    import mappy_rs
    mmi = os.path.join(golden_dir, "test.mmi")
    for preset in ("sr", "splice", "splice:hq"):
        with pytest.raises(NotImplementedError, match="not implemented by the MI355X mapping path"):
            mappy_rs.Aligner(mmi, preset=preset)
    al = mappy_rs.Aligner(mmi, preset="no-such-preset")      # the reference ignores mm_set_opt's -1 (lib.rs:336): defaults stay
    assert al._mo.bw == 500 and al._mo.flag & 4 and al.k == 15
    assert mappy_rs.Aligner(mmi, preset="map-hifi")._mo.max_gap == 10000


def test_map_no_op_record(ffi, golden_dir):
    """lib.rs:675-693"""
    import mappy_rs
    al = mappy_rs.Aligner(os.path.join(golden_dir, "test.mmi"))
    (m,) = al.map_no_op("ACGT")
    assert (m.query_start, m.query_end, m.strand, m.target_name, m.target_len, m.target_start, m.target_end, m.match_len, m.block_len,
            m.mapq, m.is_primary, m.cigar, m.NM, m.MD, m.cs) == (0, 1000, 1, "Hello", 101010, 10, 1010, 1000, 1000, 60, True, [], 0, None, "Cigar string")
    with pytest.raises(NotImplementedError, match="Using `seq2` is not implemented"):
        al.map_no_op("ACGT", seq2="A")


def test_gzip_fasta_and_corrupt_inputs(ffi, golden_dir, tmp_path):
    """.fa.gz is read through zlib like the reference does; a truncated gzip stream or a corrupt .mmi header gives no index"""
    import gzip
    import mappy_rs
    fa = open(os.path.join(golden_dir, "test.fa"), "rb").read()
    gz = tmp_path / "test.fa.gz"
    with gzip.open(gz, "wb") as fh:
        fh.write(fa)
    al = mappy_rs.Aligner(str(gz))
    ref = mappy_rs.Aligner(os.path.join(golden_dir, "test.fa"))
    assert al.n_seq == 4 and al.seq_names == ref.seq_names and al.seq("Bacillus_subtilis") == BACILLUS
    nm, nd = C.c_int64(), C.c_int64()
    ffi.lib().mm355_index_stat(al._idx, C.byref(nm), C.byref(nd), None, None)
    assert (nm.value, nd.value) == (280, 280)
    bad = tmp_path / "trunc.fa.gz"
    bad.write_bytes(gz.read_bytes()[:len(gz.read_bytes()) // 2])
    with pytest.raises(RuntimeError, match="Did not create or open an index"):
        mappy_rs.Aligner(str(bad))
    mmi = bytearray(open(os.path.join(golden_dir, "test.mmi"), "rb").read())
    for off, val in ((12, 40), (8, 0), (4, 0), (8, 200)):      # b = 40, k = 0, w = 0, k = 200
        c = bytearray(mmi); c[off:off + 4] = int(val).to_bytes(4, "little")
        f = tmp_path / ("bad%d_%d.mmi" % (off, val)); f.write_bytes(bytes(c))
        with pytest.raises(RuntimeError, match="Did not create or open an index"):
            mappy_rs.Aligner(str(f))


def test_no_seq_index_has_no_sequence(ffi, golden_dir, tmp_path):
    """minimap2 --idx-no-seq (MM_I_NO_SEQ): `seq()` reports nothing (lib.rs:710-714); mapping is refused on the GPU (test_gpu_map)"""
    import mappy_rs
    mmi = bytearray(open(os.path.join(golden_dir, "test.mmi"), "rb").read())
    flag = int.from_bytes(mmi[20:24], "little")
    assert flag == 0
    mmi[20:24] = (flag | 2).to_bytes(4, "little")
    n_S = (4 * 400 + 7) // 8 * 4
    f = tmp_path / "noseq.mmi"; f.write_bytes(bytes(mmi[:-n_S]))
    al = mappy_rs.Aligner(str(f))
    assert al.n_seq == 4 and al.seq("Bacillus_subtilis") is None


def _all_minimizers(al, seqs):
    out = set()
    for rid, s in enumerate(seqs):
        for x, _ in al.sketch(s, rid):
            out.add(int(x) >> 8)
    return out


@pytest.mark.parametrize("src", ["test.mmi", "test.fa"])
def test_flat_table_equals_fixture(ffi, golden_dir, src):
    """every minimizer of test.mmi is found in the flat 128-B-line table with the same positions; absent keys miss"""
    L = ffi.lib()
    io, mo = ffi.IdxOpt(), ffi.MapOpt()
    L.mm355_set_opt(None, C.byref(io), C.byref(mo))
    h = C.c_void_p()
    assert L.mm355_index_load(os.path.join(golden_dir, src).encode(), C.byref(io), 2, C.byref(h)) == 0
    ref = parse_mmi(os.path.join(golden_dir, "test.mmi"))["ent"]
    buf = np.zeros(16, np.uint64)
    for minier, vals in ref.items():
        n = L.mm355_index_get(h, minier, buf.ctypes.data, 16)
        assert n == len(vals) and tuple(int(v) for v in buf[:n]) == vals
    rng = np.random.default_rng(5)
    for minier in rng.integers(0, 1 << 30, 200):
        if int(minier) not in ref:
            assert L.mm355_index_get(h, int(minier), buf.ctypes.data, 16) == 0
    nm, nd = C.c_int64(), C.c_int64()
    L.mm355_index_stat(h, C.byref(nm), C.byref(nd), None, None)
    assert (nm.value, nd.value) == (280, 280)
    L.mm355_mapopt_update(C.byref(mo), h)
    assert mo.mid_occ == 10
    L.mm355_index_free(h)


def test_index_build_with_repeats_matches_oracle(ffi, tmp_path):
    """synthetic genome with repeat families: multi-occurrence runs, mid_occ, 4-bit sequence all agree with the oracle"""
    import synthdata as S
    g = S.make_genome(21, [150000, 90000], repeats=((3000, 6, 0.0), (700, 30, 0.01)), n_runs=2)
    fa = str(tmp_path / "g.fa")
    S.write_fasta(fa, g, ["c0", "c1"])
    orc = O.OracleAligner(fa, preset="map-ont")
    L = ffi.lib()
    io, mo = ffi.IdxOpt(), ffi.MapOpt()
    L.mm355_set_opt(None, C.byref(io), C.byref(mo))
    h = C.c_void_p()
    assert L.mm355_index_load(fa.encode(), C.byref(io), 2, C.byref(h)) == 0
    L.mm355_mapopt_update(C.byref(mo), h)
    assert mo.mid_occ == orc.mo.mid_occ
    seqs = [S.codes_to_str(c) for c in g]
    keys = _all_minimizers(orc, seqs)
    nm, nd = C.c_int64(), C.c_int64()
    L.mm355_index_stat(h, C.byref(nm), C.byref(nd), None, None)
    ond = C.c_int64()
    assert nm.value == O.lib().mmo_idx_n_minimizers(orc.idx, C.byref(ond)) and nd.value == ond.value == len(keys)
    buf = np.zeros(4096, np.uint64)
    n_multi = 0
    for minier in keys:
        n = L.mm355_index_get(h, minier, buf.ctypes.data, 4096)
        on = C.c_int()
        p = O.lib().mmo_idx_get(orc.idx, minier, C.byref(on))
        assert n == on.value
        assert [int(v) for v in buf[:n]] == [p[i] for i in range(n)]
        n_multi += n > 1
    assert n_multi > 50
    out = (C.c_uint8 * 1000)()
    assert L.mm355_index_getseq(h, 1, 500, 1500, out) == 1000
    assert bytes(out) == bytes(int(c) for c in g[1][500:1500])
    L.mm355_index_free(h)


# ---- map_batch plumbing with the GPU call mocked out (host logic only: channel, back-pressure, worker lifetime)
def _mocked_aligner(golden_dir, monkeypatch, delay=0.0):
    import time
    import mappy_rs
    al = mappy_rs.Aligner(os.path.join(golden_dir, "test.mmi"))
    taken, given = [], []
    monkeypatch.setattr(al, "_ctx_acquire", lambda slot: (taken.append(slot), (0, slot))[1])
    monkeypatch.setattr(al, "_ctx_release", lambda dc: given.append(dc[1]))

    def fake_map_many(seqs, flags, ctx=None):
        if delay:
            time.sleep(delay)
        return [[] for _ in seqs]
    monkeypatch.setattr(al, "_map_many", fake_map_many)
    al.enable_threading(4)
    return al, taken, given


def test_abandoned_map_batch_iterator_releases_workers_and_contexts(ffi, golden_dir, monkeypatch):
    """more results than the channel holds, the iterator dropped after one item: workers blocked on the full channel must give up,
    hand their contexts back and exit (they used to poll for the life of the process)"""
    import gc
    import time
    import mappy_rs
    monkeypatch.setattr(mappy_rs, "RESULT_CHANNEL_CAP", 500)
    al, taken, given = _mocked_aligner(golden_dir, monkeypatch)
    it = al.map_batch([{"seq": "ACGT", "i": i} for i in range(9000)])
    st = it._st
    next(it)
    threads = list(st.threads)
    assert len(threads) >= 3 and any(t.is_alive() for t in threads)      # blocked: 9000 results do not fit 500 slots
    del it
    gc.collect()
    deadline = time.time() + 10
    while any(t.is_alive() for t in threads) and time.time() < deadline:
        time.sleep(0.05)
    assert not any(t.is_alive() for t in threads)
    assert sorted(given) == sorted(taken) and len(taken) >= 2               # every context went back to the pool


def test_map_batch_back_off_bounds_pending_work(ffi, golden_dir, monkeypatch):
    """with back_off the producer waits once 50 000 reads are pending (lib.rs:867-888); all results still arrive, in any order"""
    import mappy_rs
    monkeypatch.setattr(mappy_rs, "WORK_QUEUE_CAP", 3000)
    al, taken, given = _mocked_aligner(golden_dir, monkeypatch, delay=0.002)
    ids = set()
    for m, d in al.map_batch(({"seq": "ACGT", "i": i} for i in range(20000))):
        ids.add(d["i"])
    assert len(ids) == 20000
    with pytest.raises(RuntimeError, match="without backoff"):
        monkeypatch.setattr(mappy_rs, "WORK_QUEUE_CAP", 50000)
        al.map_batch([{"seq": "ACGT"}] * 50001, back_off=False)


def test_mapping_views_read_the_hit_rows(ffi):
    """_batch_to_mappings: Mapping records are views of the C-ABI arrays; every field, cs / MD and the CIGAR come out on access"""
    import mappy_rs
    from mappy_rs import _ffi
    hits = (_ffi.Hit * 2)()
    hits[0].query_start, hits[0].query_end, hits[0].strand, hits[0].rid, hits[0].target_len = 3, 40, -1, 1, 999
    hits[0].target_start, hits[0].target_end, hits[0].match_len, hits[0].block_len, hits[0].mapq, hits[0].is_primary, hits[0].NM = 10, 47, 30, 37, 60, 1, 7
    hits[0].n_cigar, hits[0].cigar_off, hits[0].cs_off, hits[0].cs_len, hits[0].md_off, hits[0].md_len = 2, 1, 0, 3, 4, 2
    hits[1].rid, hits[1].strand, hits[1].n_cigar, hits[1].cigar_off, hits[1].cs_len, hits[1].md_len = 0, 1, 1, 0, -1, -1
    cig = (C.c_uint32 * 3)(5 << 4, (30 << 4) | 0, (7 << 4) | 2)
    sbuf = C.create_string_buffer(b":30\x0012\x00")
    off = (C.c_int64 * 4)(0, 1, 1, 2)
    status = (C.c_int32 * 3)(0, _ffi.MM355_EEMPTY, 0)
    H = _ffi.Hits()
    H.n_reads, H.hit_off, H.status, H.hits = 3, off, status, hits
    H.cigar, H.str, H.n_hits, H.n_cigar, H.n_str = cig, C.cast(sbuf, type(H.str)), 2, 3, 7
    out = mappy_rs._batch_to_mappings(C.pointer(H), 3, ["chrA", "chrB"])
    assert isinstance(out[1], RuntimeError) and str(out[1]) == "Sequence is empty"
    m = out[0][0]
    assert (m.q_st, m.q_en, m.strand, m.ctg, m.ctg_len, m.r_st, m.r_en, m.mlen, m.blen, m.mapq, m.is_primary, m.NM) == (3, 40, -1, "chrB", 999, 10, 47, 30, 37, 60, True, 7)
    assert m.cigar == [(30, 0), (7, 2)] and m.cigar_str == "30M7D" and m.cs == ":30" and m.MD == "12"
    m2 = out[2][0]
    assert m2.ctg == "chrA" and m2.cs is None and m2.MD is None and m2.cigar == [(5, 0)] and not m2.is_primary
    assert m == m and m != m2 and "cg:Z:30M7D" in str(m)


def test_pack_reads_passes_ascii_str_by_address_and_copies_the_rest(ffi):
    """plain ASCII str reads go to the library by the address of their own buffer (no encode); bytes, mixed lists and non-ASCII text take the
    copying path; both give the same pointers' contents and lengths"""
    import ctypes as C
    reads = ["ACGTNACGT", "", "G" * 5000]
    arr, lens, keep = ffi.pack_reads(reads)
    assert len(keep) == 3 and list(lens) == [9, 0, 5000] and [C.string_at(C.cast(arr, C.POINTER(C.c_void_p))[i], lens[i]) for i in range(3)] == [r.encode() for r in reads]
    assert list(arr) == [r.encode() for r in reads]
    arr2, lens2, keep2 = ffi.pack_reads(["ACGT", b"GG"])
    assert list(arr2) == [b"ACGT", b"GG"] and list(lens2) == [4, 2] and len(keep2) == 2
    arr3, lens3, _ = ffi.pack_reads(["ACéT"])
    assert list(lens3) == [5] and list(arr3) == ["ACéT".encode()]


def test_map_batch_list_fast_path_and_elementwise_errors(ffi, golden_dir, monkeypatch):
    """lists of plain dicts are cut a sub-batch at a time; a bad element anywhere still raises what the element-wise loop raises"""
    import mappy_rs
    monkeypatch.setattr(mappy_rs, "SUB_BATCH_READS", 64)
    al, taken, given = _mocked_aligner(golden_dir, monkeypatch)
    items = [{"seq": "ACGT" * 10, "i": i} for i in range(3000)]
    out = sorted(d["i"] for _m, d in al.map_batch(items))
    assert out == list(range(3000)) and all(it == {"seq": "ACGT" * 10, "i": i} for i, it in enumerate(items))
    for bad, exc in ((("seq", "ACGT"), TypeError), ({"id": 1}, KeyError), ({"seq": b"ACGT"}, ValueError)):
        broken = list(items); broken[2500] = bad
        with pytest.raises(exc):
            al.map_batch(broken)
    assert sorted(taken) == sorted(given)          # every worker context went back


def _hp_genome(seed, lens, **kw):
    """contigs with homopolymer runs of 2..40 bases at one position in ~12 (and one of 300): what MM_I_HPC compresses"""
    import synthdata as S
    rng = np.random.default_rng(seed)
    out = []
    for c in S.make_genome(seed, lens, **kw):
        rep = np.ones(len(c), np.int64)
        pos = rng.integers(0, len(c), len(c) // 12)
        rep[pos] = rng.integers(2, 41, len(pos))
        rep[int(rng.integers(0, len(c)))] = 300                      # a run whose k-mers have no record (span >= 256)
        rep[c > 3] = 1
        out.append(np.repeat(c, rep).astype(np.uint8))
    return out


def test_hpc_index_build_matches_oracle(ffi, tmp_path):
    """MM_I_HPC (map-pb): the host builder's homopolymer-compressed index == the oracle's (U:sketch.c::mm_sketch with is_hpc)"""
    import synthdata as S
    g = _hp_genome(23, [60000, 25000], repeats=((1500, 5, 0.0), (400, 20, 0.01)), n_runs=2)
    fa = str(tmp_path / "hp.fa")
    S.write_fasta(fa, g, ["c0", "c1"])
    orc = O.OracleAligner(fa, preset="map-pb")
    assert orc.k == 19 and orc.w == 10
    L = ffi.lib()
    io, mo = ffi.IdxOpt(), ffi.MapOpt()
    L.mm355_set_opt(None, C.byref(io), C.byref(mo))
    assert L.mm355_set_opt(b"map-pb", C.byref(io), C.byref(mo)) == 0 and io.flag & 1 and io.k == 19
    h = C.c_void_p()
    assert L.mm355_index_load(fa.encode(), C.byref(io), 2, C.byref(h)) == 0
    L.mm355_mapopt_update(C.byref(mo), h)
    assert mo.mid_occ == orc.mo.mid_occ
    seqs = [S.codes_to_str(c) for c in g]
    keys = _all_minimizers(orc, seqs)
    spans = {int(x) & 0xff for rid, s in enumerate(seqs) for x, _ in orc.sketch(s, rid)}
    assert max(spans) > 60 and min(spans) >= 19                      # spans are sums of run lengths
    nm, nd = C.c_int64(), C.c_int64()
    L.mm355_index_stat(h, C.byref(nm), C.byref(nd), None, None)
    ond = C.c_int64()
    assert nm.value == O.lib().mmo_idx_n_minimizers(orc.idx, C.byref(ond)) and nd.value == ond.value == len(keys)
    buf = np.zeros(4096, np.uint64)
    for minier in keys:
        n = L.mm355_index_get(h, minier, buf.ctypes.data, 4096)
        on = C.c_int()
        p = O.lib().mmo_idx_get(orc.idx, minier, C.byref(on))
        assert n == on.value and [int(v) for v in buf[:n]] == [p[i] for i in range(n)]
    L.mm355_index_free(h)


def test_hpc_mmi_file_is_loaded_with_its_flag(ffi, tmp_path):
    """an .mmi written from an HPC index (minimap2 -H / map-pb; here: the oracle's writer) carries MM_I_HPC in its header: the loader keeps the
    flag (it was refused until map-pb went in), the table equals the FASTA build's, and the preset given at load time does not override it"""
    import synthdata as S
    import mappy_rs
    g = _hp_genome(29, [40000, 9000], repeats=((800, 6, 0.01),), n_runs=1)
    fa = str(tmp_path / "hp.fa")
    S.write_fasta(fa, g, ["c0", "c1"])
    orc = O.OracleAligner(fa, preset="map-pb")
    mmi = str(tmp_path / "hp.mmi")
    assert O.lib().mmo_idx_dump(orc.idx, mmi.encode()) == 0
    L = ffi.lib()
    io, mo = ffi.IdxOpt(), ffi.MapOpt()
    L.mm355_set_opt(None, C.byref(io), C.byref(mo))              # plain defaults: the FILE decides
    h, h2 = C.c_void_p(), C.c_void_p()
    assert L.mm355_index_load(mmi.encode(), C.byref(io), 2, C.byref(h)) == 0
    assert L.mm355_set_opt(b"map-pb", C.byref(io), C.byref(mo)) == 0
    assert L.mm355_index_load(fa.encode(), C.byref(io), 2, C.byref(h2)) == 0
    a = mappy_rs.Aligner(mmi)
    assert a.k == 19 and a.w == 10 and a.n_seq == 2
    st = [C.c_int64() for _ in range(4)]
    L.mm355_index_stat(h, C.byref(st[0]), C.byref(st[1]), None, None)
    L.mm355_index_stat(h2, C.byref(st[2]), C.byref(st[3]), None, None)
    assert (st[0].value, st[1].value) == (st[2].value, st[3].value) and st[0].value > 1000
    buf, buf2 = np.zeros(4096, np.uint64), np.zeros(4096, np.uint64)
    seqs = [S.codes_to_str(c) for c in g]
    for minier in list(_all_minimizers(orc, seqs))[:3000]:
        n, n2 = L.mm355_index_get(h, minier, buf.ctypes.data, 4096), L.mm355_index_get(h2, minier, buf2.ctypes.data, 4096)
        assert n == n2 > 0 and np.array_equal(buf[:n], buf2[:n])
    L.mm355_index_free(h); L.mm355_index_free(h2)
