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
    for preset in (None, "map-ont", "map-hifi", "asm20"):
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
    for known in (b"sr", b"splice", b"splice:hq", b"map-pb", b"ava-pb", b"map10k", b"short", b"cdna"):
        assert L.mm355_set_opt(known, C.byref(io), C.byref(mo)) == ffi.MM355_EUNSUP
    for unknown in (b"map-ontt", b"asm7", b""):
        assert L.mm355_set_opt(unknown, C.byref(io), C.byref(mo)) == ffi.MM355_EINVAL    # U:options.c::mm_set_opt returns -1
    assert (bytes(mo), bytes(io)) == before                                               # ... and leaves the options untouched


def test_unsupported_presets_raise(ffi, golden_dir):
    """a preset minimap2 knows but this path does not implement must not be mapped with other parameters"""
    import mappy_rs
    mmi = os.path.join(golden_dir, "test.mmi")
    for preset in ("sr", "splice", "map-pb", "ava-pb", "splice:hq"):
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
