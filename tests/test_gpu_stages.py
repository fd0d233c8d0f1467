"""Kernel-by-kernel parity: HIP path (through the C-ABI of libmm355.so) vs the CPU oracle on the same
seeded inputs.  Bit-exact (integer/index work): minimizers, anchor lists in generation order and after the
literal radix_sort_128x emulation, chaining f/p/v, chains after backtrack+compact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
import synthdata as S


@pytest.fixture(scope="module")
def world(built, tmp_path_factory):
    import mappy_rs
    td = tmp_path_factory.mktemp("gs")
    g = S.make_genome(31, [400000, 250000], repeats=((4000, 6, 0.01), (900, 40, 0.02), (300, 120, 0.05)), n_runs=3)
    fa = str(td / "ref.fa")
    S.write_fasta(fa, g, ["chrA", "chrB"])
    reads, truth = S.make_reads(32, g, 160, n50=5000, lo=200)
    # edge cases: tiny reads, a read with Ns, a read made of a tandem repeat, an unrelated random read, a 1-base read
    rng = np.random.default_rng(7)
    unit = S.codes_to_str(g[0][1000:1037])
    extra = ["ACGT", "A", reads[0][:14], reads[1][:600] + "NNNNNNNNNN" + reads[1][600:1500], unit * 60,
             S.codes_to_str(S.random_codes(rng, 3000)), S.codes_to_str(g[1][5000:5400]), "N" * 50]
    reads = reads + extra
    al = mappy_rs.Aligner(fa, preset="map-ont")
    orc = O.OracleAligner(fa, preset="map-ont")
    sr = al._stage_runner()
    yield dict(al=al, orc=orc, sr=sr, reads=reads, genome=g)
    sr.close()


@pytest.mark.parametrize("sparse_max", ["0", "1000000000"])
def test_sketch_parity(world, monkeypatch, sparse_max):
    monkeypatch.setenv("MM355_SKETCH_SPARSE_MAX", sparse_max)
    got = world["sr"].sketch(world["reads"])
    for rd, g in zip(world["reads"], got):
        exp = world["orc"].sketch(rd)
        assert g.shape == exp.shape and np.array_equal(g, exp)


def test_anchor_parity_generation_order(world):
    got, rep, nmp = world["sr"].anchors(world["reads"], sorted_=False)
    tot = 0
    for i, rd in enumerate(world["reads"]):
        exp, erep, emp, _ = world["orc"].anchors(rd, sorted_=False)
        assert got[i].shape == exp.shape, (i, got[i].shape, exp.shape)
        assert np.array_equal(got[i], exp), i
        assert rep[i] == erep and nmp[i] == len(emp)
        tot += len(exp)
    assert tot > 10000


def test_anchor_parity_sorted(world):
    """same permutation as the unstable in-place radix sort, including equal-x ties"""
    got, _, _ = world["sr"].anchors(world["reads"], sorted_=True)
    n_ties = 0
    for i, rd in enumerate(world["reads"]):
        exp, _, _, _ = world["orc"].anchors(rd, sorted_=True)
        assert np.array_equal(got[i], exp), i
        if len(exp) > 1:
            n_ties += int((exp[1:, 0] == exp[:-1, 0]).sum())
    assert n_ties > 0, "test data must exercise equal-key ties"


def test_chain_fill_parity(world):
    got = world["sr"].chain(world["reads"])
    for i, rd in enumerate(world["reads"]):
        a, f, p, v = got[i]
        ea, _, _, _ = world["orc"].anchors(rd, sorted_=True)
        ef, ep, ev, _ = world["orc"].chain_fill(ea, len(rd))
        assert np.array_equal(a, ea)
        assert np.array_equal(f, ef), i
        assert np.array_equal(p.astype(np.int64), ep), i
        assert np.array_equal(v, ev), i


def test_chains_parity(world):
    got = world["sr"].chains(world["reads"])
    n_multi = 0
    for i, rd in enumerate(world["reads"]):
        u, a = got[i]
        ea, _, _, _ = world["orc"].anchors(rd, sorted_=True)
        eu, eb = world["orc"].chains(ea, len(rd))
        assert np.array_equal(u, eu), i
        assert np.array_equal(a, eb), i
        n_multi += len(eu) > 1
    assert n_multi > 0


def test_chains_parity_long_probes(world):
    """mg_chain_backtrack on chains of more than 4096 anchors: the probe outruns the LDS list of visited nodes (the chase-again path of
    k_backtrack), its register window is reloaded many times, compact_a copies tens of thousands of anchors of one chain"""
    g = world["genome"]
    rng = np.random.default_rng(17)
    comp = lambda c: np.where(c < 4, 3 - c, 4).astype(np.uint8)[::-1]
    reads = [S.codes_to_str(S.mutate(g[0][1000:121000], rng, 0.01, 0.005, 0.005)),
             S.codes_to_str(S.mutate(comp(g[0][150000:215000]), rng, 0.02, 0.01, 0.01)),
             S.codes_to_str(S.mutate(g[1][20000:70000], rng, 0.03, 0.015, 0.015))]
    got = world["sr"].chains(reads)
    longest = 0
    for i, rd in enumerate(reads):
        u, a = got[i]
        ea, _, _, _ = world["orc"].anchors(rd, sorted_=True)
        eu, eb = world["orc"].chains(ea, len(rd))
        assert np.array_equal(u, eu), i
        assert np.array_equal(a, eb), i
        longest = max(longest, max(int(x) & 0xffffffff for x in eu))
    assert longest > 4096, longest


def _sv_reads(g, rng, n_each=10):
    """reads whose chains mg_lchain_dp (bw 500) cannot join: a 1.5 - 6 kb deletion, an inserted block, a chimera of two loci"""
    comp = lambda c: np.where(c < 4, 3 - c, 4).astype(np.uint8)[::-1]
    out = []
    for _ in range(n_each):
        a0 = int(rng.integers(0, 330000)); gap = int(rng.integers(1500, 6000))
        out.append(S.codes_to_str(S.mutate(np.concatenate([g[0][a0:a0 + 3000], g[0][a0 + 3000 + gap:a0 + 8000 + gap]]), rng, 0.02, 0.01, 0.01)))
        out.append(S.codes_to_str(S.mutate(np.concatenate([g[0][a0:a0 + 2500], S.random_codes(rng, int(rng.integers(700, 3000))), g[0][a0 + 2500:a0 + 6000]]), rng, 0.02, 0.01, 0.01)))
        b0 = int(rng.integers(0, 200000))
        out.append(S.codes_to_str(S.mutate(np.concatenate([g[0][a0:a0 + 3000], comp(g[1][b0:b0 + 2500])]), rng, 0.02, 0.01, 0.01)))
    return out


def _check_rmq(sr, orc, reads, must_rechain):
    got = sr.rmq(reads)
    n_dev = n_host = n_keep = 0
    for i, rd in enumerate(reads):
        u, a, state = got[i]
        ea, _, _, _ = orc.anchors(rd, sorted_=True)
        eu, eb, did = orc.chains_final(ea, len(rd))      # did: bit 0 primary RMQ chainer, bit 1 long-join re-chain
        if state >= 2:      # handed to the literal host code: the device leaves the anchors sorted by x (checked end to end by the mapping tests)
            n_host += 1
            assert did != 0, i
            assert len(u) == 0 and np.all(a[1:, 0] >= a[:-1, 0]), i
            continue
        assert (state == 1) == bool(did & 2), (i, state, did)
        assert np.array_equal(u, eu), (i, state)
        assert np.array_equal(a, eb), (i, state)
        n_dev += did != 0; n_keep += state == 0
    assert n_dev >= must_rechain, (n_dev, n_host, n_keep)
    assert n_host <= max(2, len(reads) // 10), (n_dev, n_host)      # the fallback is the exception
    return n_dev, n_host, n_keep


def test_rmq_rechain_parity(world):
    """row a9 on the device: long-join re-chain (rescue test, radix_sort_128x of the chained anchors, mg_lchain_rmq with bw_long,
    backtrack, compact_a) against the oracle's mg_lchain_rmq, read by read"""
    reads = world["reads"] + _sv_reads(world["genome"], np.random.default_rng(91))
    n_dev, n_host, n_keep = _check_rmq(world["sr"], world["orc"], reads, must_rechain=25)
    assert n_keep > 0


def test_rmq_stage_off_leaves_the_host_in_charge(world, monkeypatch):
    monkeypatch.setenv("MM355_RMQ_ON_HOST", "1")
    got = world["sr"].rmq(world["reads"][:8])
    assert all(st == 3 for _, _, st in got)


@pytest.mark.parametrize("preset", ["asm5", "asm20"])
def test_rmq_primary_chainer_parity(built, tmp_path, preset):
    """MM_F_RMQ presets: mg_lchain_rmq over all sorted anchors of a read on the device (bw 1000, inner distance 1000, cap 100000)"""
    import mappy_rs
    g = S.make_genome(81, [350000, 150000], repeats=((4000, 4, 0.01), (900, 12, 0.02)), n_runs=2)
    fa = str(tmp_path / "p.fa")
    S.write_fasta(fa, g, ["ctgA", "ctgB"])
    div = {"asm5": 0.002, "asm20": 0.02}[preset]
    reads, _ = S.make_reads(82, g, 40, n50=15000, lo=1500, sub=div, ins=div / 4, dele=div / 4)
    rng = np.random.default_rng(83)
    for _ in range(6):
        a0 = int(rng.integers(0, 300000))
        reads.append(S.codes_to_str(S.mutate(np.concatenate([g[0][a0:a0 + 9000], g[0][a0 + 9000 + 2500:a0 + 20000]]), rng, div, div / 4, div / 4)))
    reads += ["ACGT", S.codes_to_str(g[1][5000:5400])]
    al = mappy_rs.Aligner(fa, preset=preset)
    orc = O.OracleAligner(fa, preset=preset)
    sr = al._stage_runner()
    _check_rmq(sr, orc, reads, must_rechain=40)
    sr.close()


def _tandem_world(tmp_path):
    """a genome full of tandem arrays (exact and 1 % diverged copies of 171 / 350 / 700 bp units) and reads across them: chains of
    neighbouring diagonals interleave (the inner walk of mg_lchain_rmq has to order its candidates) and symmetric anchor pairs share a
    range-minimum priority (the device hands those reads to the literal host code)"""
    rng = np.random.default_rng(5)
    bg = S.random_codes(rng, 300000)
    pieces, pos = [], 0
    for k in range(12):
        seg = bg[pos:pos + 20000]; pos += 20000
        unit = S.random_codes(rng, int(rng.choice([171, 350, 700])))
        ncopy = int(rng.integers(3, 12))
        pieces += [seg, np.concatenate([S.mutate(unit, rng, 0.0 if k % 2 == 0 else 0.01, 0, 0) for _ in range(ncopy)])]
    g = [np.concatenate(pieces).astype(np.uint8)]
    fa = str(tmp_path / "tandem.fa")
    S.write_fasta(fa, g, ["chrT"])
    reads = []
    for i in range(200):
        st = int(rng.integers(0, len(g[0]) - 9000)); ln = int(rng.integers(2000, 9000))
        err = 0.0 if i % 2 == 0 else 0.01
        reads.append(S.codes_to_str(S.mutate(g[0][st:st + ln], rng, err, err / 2, err / 2)))
    return fa, reads


@pytest.mark.parametrize("preset", ["map-ont", "asm20", "map-hifi"])
def test_rmq_tandem_arrays_order_and_fallback(built, tmp_path, preset):
    import mappy_rs
    fa, reads = _tandem_world(tmp_path)
    al = mappy_rs.Aligner(fa, preset=preset)
    orc = O.OracleAligner(fa, preset=preset)
    sr = al._stage_runner()
    got = sr.rmq(reads)
    n_dev = n_host = 0
    for i, rd in enumerate(reads):
        u, a, state = got[i]
        if state >= 2:
            n_host += 1
            continue
        ea, _, _, _ = orc.anchors(rd, sorted_=True)
        eu, eb, did = orc.chains_final(ea, len(rd))
        assert np.array_equal(u, eu) and np.array_equal(a, eb), (i, state, did)
        n_dev += did != 0
    sr.close()
    assert n_host > 0 and n_dev > 0, (n_dev, n_host)          # both routes taken: equal priorities exist here, and they are the minority
    assert n_host < len(reads) // 2, (n_dev, n_host)


def test_empty_and_ragged_batches(world):
    sr = world["sr"]
    assert sr.sketch([]) == []
    got = sr.sketch(["", "ACGT", world["reads"][0]])
    assert len(got[0]) == 0 and len(got[1]) == 0 and len(got[2]) > 0
    ch = sr.chains(["", "ACGTACGTAC"])
    assert len(ch[0][0]) == 0 and len(ch[1][0]) == 0


def test_hifi_preset_parity(built, tmp_path):
    import mappy_rs
    g = S.make_genome(41, [300000], repeats=((2500, 5, 0.005),))
    fa = str(tmp_path / "h.fa")
    S.write_fasta(fa, g, ["chrH"])
    reads, _ = S.make_reads(42, g, 24, n50=9000, lo=3000, sub=0.0005, ins=0.00075, dele=0.00075)
    al = mappy_rs.Aligner(fa, preset="map-hifi")
    orc = O.OracleAligner(fa, preset="map-hifi")
    assert al.k == 19 and al.w == 19
    sr = al._stage_runner()
    got = sr.chains(reads)
    for i, rd in enumerate(reads):
        ea, _, _, _ = orc.anchors(rd, sorted_=True)
        eu, eb = orc.chains(ea, len(rd))
        assert np.array_equal(got[i][0], eu) and np.array_equal(got[i][1], eb)
    sr.close()


@pytest.mark.parametrize("sparse_max", ["0", "1000000000"])   # lane-per-chunk grid / one chunk per wave (small batches)
def test_chunked_sketch_adversarial(built, tmp_path, monkeypatch, sparse_max):
    """the chunked sketch kernel must equal the sequential machine on inputs that stress its warm-up proof:
    even k (symmetric k-mers are skipped without advancing the ring), N every few bases, long N runs, homopolymers,
    palindromic repeats, and reads around the chunk size"""
    import mappy_rs
    monkeypatch.setenv("MM355_SKETCH_SPARSE_MAX", sparse_max)
    rng = np.random.default_rng(11)
    g = S.make_genome(71, [60000], repeats=())
    fa = str(tmp_path / "s.fa")
    S.write_fasta(fa, g, ["c"])
    base = S.codes_to_str(g[0][:20000])
    pal = "ACGT" * 300 + "AT" * 500 + "GATC" * 200
    sparse_n = "".join(c if i % 9 else "N" for i, c in enumerate(base[:6000]))
    runs = base[:1000] + "N" * 700 + base[1000:1400] + "N" * 33 + base[1400:3000]
    reads = [base, pal, sparse_n, runs, "A" * 3000, "N" * 2000 + base[:900], base[:383], base[:384], base[:385], base[:769],
             pal + base[:2000] + pal, S.codes_to_str(S.random_codes(rng, 5000, gc=0.1))]
    for k, w in ((15, 10), (14, 8), (19, 19), (16, 5), (21, 11)):
        al = mappy_rs.Aligner(fa, k=k, w=w)
        orc = O.OracleAligner(fa, k=k, w=w)
        sr = al._stage_runner()
        got = sr.sketch(reads)
        for i, rd in enumerate(reads):
            exp = orc.sketch(rd)
            assert got[i].shape == exp.shape and np.array_equal(got[i], exp), (k, w, i)
        sr.close()


def test_device_index_builder_equals_host_builder(built, tmp_path):
    """the index built on the GPU (sketch + radix sort + CAS table fill) gives the same mid_occ, the same
    anchors and the same final hits as the host-built one"""
    import ctypes as C
    import mappy_rs
    from mappy_rs import _ffi
    L = _ffi.lib()
    g = S.make_genome(91, [300000, 1000, 170000, 37], repeats=((3000, 6, 0.0), (700, 40, 0.01), (200, 300, 0.02)), n_runs=3)
    names = ["c0", "c1", "c2", "c3"]
    fa = str(tmp_path / "d.fa")
    S.write_fasta(fa, g, names)
    al = mappy_rs.Aligner(fa, preset="map-ont")           # host builder
    io, mo = _ffi.IdxOpt(), _ffi.MapOpt()
    L.mm355_set_opt(None, C.byref(io), C.byref(mo))
    mo.flag |= 4
    seqs = [bytes(bytearray(c.tolist())) for c in g]      # raw codes 0..4 are accepted like ASCII
    arr = (C.c_char_p * len(seqs))(*seqs)
    lens = (C.c_int64 * len(seqs))(*[len(s) for s in seqs])
    nm = (C.c_char_p * len(seqs))(*[n.encode() for n in names])
    h = C.c_void_p()
    _ffi.check(L.mm355_index_build_device(C.byref(io), len(seqs), arr, lens, nm, 0, C.byref(h)))
    L.mm355_mapopt_update(C.byref(mo), h)
    assert mo.mid_occ == al._mo.mid_occ
    nmz, nd, nmz2, nd2 = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    L.mm355_index_stat(h, C.byref(nmz), C.byref(nd), None, None)
    L.mm355_index_stat(al._idx, C.byref(nmz2), C.byref(nd2), None, None)
    assert (nmz.value, nd.value) == (nmz2.value, nd2.value)
    out = (C.c_uint8 * 500)()
    assert L.mm355_index_getseq(h, 2, 100, 600, out) == 500 and bytes(out) == bytes(int(c) for c in g[2][100:600])
    reads, _ = S.make_reads(92, g, 60, n50=3000, lo=200)
    sr_h = al._stage_runner()
    sr_d = _ffi.StageRunner(h, mo, 0)
    a_h, rep_h, _ = sr_h.anchors(reads, sorted_=True)
    a_d, rep_d, _ = sr_d.anchors(reads, sorted_=True)
    n_multi = 0
    for x, y in zip(a_h, a_d):
        assert np.array_equal(x, y)
    assert np.array_equal(rep_h, rep_d)
    sr_h.close(); sr_d.close()
    L.mm355_index_free(h)


def _repeat_world(td):
    """genome with a 150-copy 1-kb family; reads made of several unit copies collect > 100k anchors each"""
    rng = np.random.default_rng(77)
    g = S.random_codes(rng, 700000)
    unit = S.random_codes(rng, 1000)
    for _ in range(150):
        pos = int(rng.integers(0, len(g) - 1000))
        g[pos:pos + 1000] = S.mutate(unit, rng, 0.01, 0.0, 0.0)[:1000]
    fa = os.path.join(str(td), "rep.fa")
    S.write_fasta(fa, [g], ["chrR"])
    reads = [S.codes_to_str(S.mutate(np.tile(unit, m), rng, 0.02, 0.01, 0.01)) for m in (2, 5, 9)]
    reads += [S.codes_to_str(g[5000:9000]), S.codes_to_str(S.mutate(np.tile(unit, 3)[::-1].copy(), rng, 0.02, 0.0, 0.0))]
    return fa, reads


def _check_sorted_and_chains(fa, reads, min_heavy):
    import mappy_rs
    al = mappy_rs.Aligner(fa, preset="map-ont")
    orc = O.OracleAligner(fa, preset="map-ont")
    sr = al._stage_runner()
    try:
        got, _, _ = sr.anchors(reads, sorted_=True)
        ch = sr.chains(reads)
        big = 0
        for i, rd in enumerate(reads):
            exp, _, _, _ = orc.anchors(rd, sorted_=True)
            assert np.array_equal(got[i], exp), i
            eu, eb = orc.chains(exp, len(rd))
            assert np.array_equal(ch[i][0], eu), i
            assert np.array_equal(ch[i][1], eb), i
            big += len(exp) > min_heavy
        return big
    finally:
        sr.close()


def test_heavy_read_block_level_sort(built, tmp_path):
    """reads above the heavy threshold (16384 anchors) take the 1024-thread level kernel + task lists; the permutation
    must still be the literal radix_sort_128x one"""
    fa, reads = _repeat_world(tmp_path)
    assert _check_sorted_and_chains(fa, reads, 16384) >= 2


def test_heavy_sort_path_on_ordinary_reads(built, tmp_path):
    """MM355_SORT_HEAVY_MIN=100 sends every read with > 100 anchors through the block-level path (one child process:
    the threshold is read once per process)"""
    import subprocess, sys
    code = ("import sys; sys.path[:0] = %r; import tests.test_gpu_stages as T, synthdata as S, numpy as np, os\n"
            "td = sys.argv[1]\n"
            "g = S.make_genome(31, [400000, 250000], repeats=((4000, 6, 0.01), (900, 40, 0.02), (300, 120, 0.05)), n_runs=3)\n"
            "fa = os.path.join(td, 'ref.fa'); S.write_fasta(fa, g, ['chrA', 'chrB'])\n"
            "reads, _ = S.make_reads(32, g, 120, n50=5000, lo=200)\n"
            "n = T._check_sorted_and_chains(fa, reads, 100)\n"
            "fa2, r2 = T._repeat_world(td)\n"
            "n += T._check_sorted_and_chains(fa2, r2, 100)\n"
            "assert n > 50, n\nprint('heavy-ok', n)\n") % ([os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                          os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mappy-rs_amd")],)
    env = dict(os.environ, MM355_SORT_HEAVY_MIN="100")
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "heavy-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_fast_sort_path_keeps_the_reference_tie_order(built, tmp_path):
    """MM355_FAST_SORT=1 sends every read through the segmented radix sort; reads with equal keys must still come out in the
    literal radix_sort_128x order (they are re-sorted by the emulation), tie-free reads are identical by construction"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = %r; import tests.test_gpu_stages as T, synthdata as S, numpy as np, os\n"
            "td = sys.argv[1]\n"
            "g = S.make_genome(31, [400000, 250000], repeats=((4000, 6, 0.01), (900, 40, 0.02), (300, 120, 0.05)), n_runs=3)\n"
            "fa = os.path.join(td, 'ref.fa'); S.write_fasta(fa, g, ['chrA', 'chrB'])\n"
            "reads, _ = S.make_reads(32, g, 120, n50=5000, lo=200)\n"
            "unit = S.codes_to_str(g[0][1000:1037])\n"
            "reads += ['ACGT', 'A', unit * 60, 'N' * 50]\n"
            "n = T._check_sorted_and_chains(fa, reads, 100)\n"
            "fa2, r2 = T._repeat_world(td)\n"
            "n += T._check_sorted_and_chains(fa2, r2, 100)\n"
            "print('fast-ok', n)\n") % ([root, os.path.join(root, "mappy-rs_amd")],)
    env = dict(os.environ, MM355_FAST_SORT="1")
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fast-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
