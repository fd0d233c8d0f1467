"""BASELINE.json configs[2] (GRCh38-scale, map-ont) and configs[4] (GRCh38-scale, map-hifi) on the GPU.
 * full size (3.09 Gbp synthetic genome, index built on the device): the size-independent properties of tests/_props.py;
 * mid scale (155 Mbp: mid_occ in the hundreds, thousands of anchors per read): bit-exact parity with the oracle, and the code paths
   that only anchor-rich reads reach must actually have run (segmented sort + tie emulation, the eight-wave extension classes incl. the
   HBM-state one) -- asserted from mm355_get_stats."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
import synthdata as S
from _props import check_properties

HIFI = dict(n50=18000, sigma=0.14, lo=5000, hi=60000, sub=0.0005, ins=0.00075, dele=0.00075)
ONT = dict(n50=10000, sigma=0.75, lo=500, hi=100000)


def build_device_index(L, _ffi, g, names, preset, extra=None):
    io, mo = _ffi.IdxOpt(), _ffi.MapOpt()
    L.mm355_set_opt(None, C.byref(io), C.byref(mo))
    _ffi.check(L.mm355_set_opt(preset.encode(), C.byref(io), C.byref(mo))); mo.flag |= 4
    for k, v in (extra or {}).items():          # the kwargs of mappy_rs.Aligner (lib.rs:354-362)
        if k == "extra_flags":
            mo.flag |= v
        else:
            setattr(mo, k, v)
    ptrs = (C.c_char_p * len(g))(*[C.cast(c.ctypes.data, C.c_char_p) for c in g])
    lens = (C.c_int64 * len(g))(*[len(c) for c in g]); nm = (C.c_char_p * len(g))(*[n.encode() for n in names])
    idx = C.c_void_p()
    _ffi.check(L.mm355_index_build_device(C.byref(io), len(g), ptrs, lens, nm, 0, C.byref(idx)))
    L.mm355_mapopt_update(C.byref(mo), idx)
    return idx, mo


@pytest.fixture(scope="module")
def human_full(built):
    return S.make_human_like(3, 1.0)


@pytest.fixture(scope="module")
def human_mid(built):
    return S.make_human_like(3, 0.05)


@pytest.mark.parametrize("preset,seed,kw,n", [("map-ont", 4, ONT, 3072), ("map-hifi", 6, HIFI, 1024)])
def test_full_size_human_properties(human_full, preset, seed, kw, n):
    from mappy_rs import _ffi
    L = _ffi.lib()
    g, names = human_full
    reads, truth = S.make_read_block(seed, 0, g, **kw)
    reads, truth = reads[:n], truth[:n]
    idx, mo = build_device_index(L, _ffi, g, names, preset)
    ctx = C.c_void_p()
    _ffi.check(L.mm355_ctx_create(idx, 0, C.byref(ctx)))
    try:
        assert mo.mid_occ > 100, mo.mid_occ                      # a GRCh38-scale occurrence threshold (383 for map-ont on this genome)
        n_hits, n_right = check_properties(L, ctx, mo, reads, truth, [0, 500, 503, n // 2, n])
        assert n_hits >= 0.95 * n and n_right >= 0.93 * n, (n_hits, n_right)
        st = _ffi.Stats(); L.mm355_get_stats(ctx, C.byref(st))
        if preset == "map-ont":
            assert st.n_a / st.n_reads > 2048                    # anchor-rich: the segmented-sort path
    finally:
        L.mm355_ctx_destroy(ctx)
        L.mm355_index_free(idx)


def _sv_reads(g, rng, big_deletion=False):
    """reads that reach the long-target extension classes: (a) a long diverged stretch between two clean flanks (gap fills and z-drops
    inside it), (b) 12 kb of unrelated sequence at one end -- the end extension then runs max_gap query bases against max_gap reference
    bases (U:align.c::mm_align1 caps the window at max_gap: 5 000 for map-ont, 10 000 for map-hifi: the LDS classes of k_ksw_extd2<512>).
    No default preset produces a target beyond 12288 (windows and chain gaps are capped at max_gap <= 10 000); the HBM-state class is
    reached through the Aligner kwargs `bw`, `max_frag_len` and `extra_flags` (lib.rs:354-362): with bw = 30 000, max_frag_len = 100 000
    and MM_F_NO_END_FLT (mm_fix_bad_ends would cut the chain at the jump) the chain crosses a 20 kb deletion, and the gap fill is a
    ~200 x 20 000 problem (c: big_deletion)."""
    out = []
    for ci in (0, 3, 7):
        c = g[ci]
        st = int(rng.integers(len(c) // 8, len(c) // 4))
        seg = c[st:st + 44000].copy()
        if (seg == 4).any():
            continue
        if big_deletion:
            rd = np.concatenate([S.mutate(seg[:11000], rng, 0.005, 0.002, 0.002), S.mutate(seg[31000:42000], rng, 0.005, 0.002, 0.002)])
            out.append(rd.tobytes())
            out.append(np.where(rd < 4, 3 - rd, 4).astype(np.uint8)[::-1].tobytes())
            continue
        mid = S.mutate(seg[9000:24000], rng, 0.30, 0.02, 0.02)
        rd = np.concatenate([S.mutate(seg[:9000], rng, 0.01, 0.005, 0.005), mid, S.mutate(seg[24000:34000], rng, 0.01, 0.005, 0.005)])
        out.append(rd.tobytes())
        out.append(np.where(rd < 4, 3 - rd, 4).astype(np.uint8)[::-1].tobytes())
        tail = np.concatenate([S.mutate(seg[:14000], rng, 0.002, 0.001, 0.001), S.random_codes(rng, 12000, 0.41)])
        out.append(tail.tobytes())
        out.append(np.where(tail < 4, 3 - tail, 4).astype(np.uint8)[::-1].tobytes())
    return out


@pytest.mark.parametrize("preset,seed,kw,extra", [("map-ont", 4, ONT, None), ("map-hifi", 6, HIFI, None), ("map-ont", 4, ONT, dict(bw=30000, max_frag_len=100000, extra_flags=0x10000000))])
def test_mid_scale_human_parity(human_mid, preset, seed, kw, extra):
    from mappy_rs import _ffi
    import mappy_rs
    L = _ffi.lib()
    g, names = human_mid
    reads, _ = S.make_read_block(seed, 1, g, **kw)
    reads = reads[:96 if extra is None else 24] + _sv_reads(g, np.random.default_rng(77), big_deletion=extra is not None)
    idx, mo = build_device_index(L, _ffi, g, names, preset, extra)
    orc = O.OracleAligner(codes=g, names=names, preset=preset, n_threads=16, **(extra or {}))
    assert orc.mo.mid_occ == mo.mid_occ and mo.mid_occ >= 100, (orc.mo.mid_occ, mo.mid_occ)
    ctx = C.c_void_p()
    _ffi.check(L.mm355_ctx_create(idx, 0, C.byref(ctx)))
    try:
        rarr, rlens, keep = _ffi.pack_reads(reads)
        hp = C.POINTER(_ffi.Hits)()
        _ffi.check(L.mm355_map_batch(ctx, C.byref(mo), len(reads), rarr, rlens, 1, C.byref(hp)))
        got = mappy_rs._batch_to_mappings(hp, len(reads), names)
        L.mm355_free_hits(hp)
        st = _ffi.Stats(); L.mm355_get_stats(ctx, C.byref(st))
        n_hits = 0
        for i, rd in enumerate(reads):
            exp = orc.map(rd, cs=True)
            assert len(got[i]) == len(exp), (i, len(got[i]), len(exp))
            for m, e in zip(got[i], exp):
                assert (m.target_name, m.target_start, m.target_end, m.query_start, m.query_end, m.strand, m.mapq, m.is_primary, m.NM, m.cigar_str, m.cs) == \
                       (e["target_name"], e["target_start"], e["target_end"], e["query_start"], e["query_end"], e["strand"], e["mapq"], e["is_primary"],
                        e["NM"], e["cigar_str"], e["cs"]), i
                n_hits += 1
        assert n_hits >= (90 if extra is None else 25)
        groups = list(st.n_launch_group)
        assert sum(groups[8:12]) > 0, groups                       # eight-wave classes (targets 1k..12k) ran
        if extra is not None:
            assert groups[12] + groups[13] > 0, groups             # ... and the HBM-state class (targets > 12288; see _sv_reads)
        if preset == "map-ont":
            assert st.n_a / st.n_reads >= 2048, st.n_a / st.n_reads   # anchor-rich batch: segmented radix sort for all reads ...
            assert st.n_sort_fast_reads == len(reads) and st.n_sort_tie_reads > 0, (st.n_sort_fast_reads, st.n_sort_tie_reads)   # ... + literal tie emulation
    finally:
        L.mm355_ctx_destroy(ctx)
        L.mm355_index_free(idx)


def _check_culled(full, got, max_dist_x, t_min):
    """`got` (stage sorted = 2) must be `full` (the reference's sorted array) minus whole x-components, in the same order, and must still
    hold every anchor of every x-component (same strand | contig, consecutive gaps <= max_dist_x) of at least t_min anchors"""
    pos = {(int(x), int(y)): i for i, (x, y) in enumerate(full)}
    assert len(pos) == len(full)
    idx = np.array([pos[(int(x), int(y))] for x, y in got], np.int64)         # KeyError: an anchor that is not in the reference's array
    assert np.all(idx[1:] > idx[:-1]), "order differs from the reference's sorted array"
    if len(full) == 0:
        return 0
    x = full[:, 0]
    brk = np.concatenate(([True], (x[1:] >> np.uint64(32) != x[:-1] >> np.uint64(32)) | ((x[1:] - x[:-1]) > np.uint64(max_dist_x))))
    cid = np.cumsum(brk) - 1
    need = np.bincount(cid)[cid] >= t_min
    have = np.zeros(len(full), bool); have[idx] = True
    assert np.all(have[need]), "an x-component that can chain lost anchors"
    return int(len(full) - len(got))


def test_anchor_cull_and_sort_mid_scale(human_mid):
    """row a6 on an anchor-rich batch: the hand-written path (mm355_cullsort.hip).  sorted = 1: the reference's whole sorted array, ties in
    the order of the unstable radix_sort_128x (the cull switched off: every anchor goes through the LDS sort / the literal emulation);
    sorted = 2: what the chainer is given -- whole x-components dropped, nothing else touched; chains: identical to the oracle's chains
    of the WHOLE array (the cull is invisible downstream)."""
    from mappy_rs import _ffi
    L = _ffi.lib()
    g, names = human_mid
    reads, _ = S.make_read_block(4, 2, g, **ONT)
    reads = reads[:64]
    idx, mo = build_device_index(L, _ffi, g, names, "map-ont")
    orc = O.OracleAligner(codes=g, names=names, preset="map-ont", n_threads=16)
    sr = _ffi.StageRunner(idx, mo, 0)
    try:
        full_g, _, _ = sr.anchors(reads, sorted_=1, cap=40_000_000)
        st = sr.stats()
        assert st.n_sort_fast_reads == len(reads) and st.n_a_kept == st.n_a, (st.n_sort_fast_reads, st.n_a_kept, st.n_a)
        cul_g, _, _ = sr.anchors(reads, sorted_=2, cap=40_000_000)
        st = sr.stats()
        assert st.n_sort_fast_reads == len(reads) and 0 < st.n_a_kept < 0.9 * st.n_a, (st.n_a_kept, st.n_a)   # (155 Mbp: a third is dropped; 3.1 Gbp: nine tenths)
        n_ties = n_tie_reads = n_culled = 0
        exp_all = []
        for i, rd in enumerate(reads):
            exp, _, _, _ = orc.anchors(rd, sorted_=True)
            exp_all.append(exp)
            assert np.array_equal(full_g[i], exp), i
            t = int((exp[1:, 0] == exp[:-1, 0]).sum()) if len(exp) > 1 else 0
            n_ties += t; n_tie_reads += t > 0
            n_culled += _check_culled(exp, cul_g[i], mo.max_gap, 3)
        assert n_tie_reads >= 2 and n_culled == st.n_a - st.n_a_kept, (n_tie_reads, n_culled, st.n_a - st.n_a_kept)
        assert st.n_sort_tie_reads >= 1
        ch = sr.chains(reads, cap=40_000_000)
        for i, rd in enumerate(reads):
            eu, eb = orc.chains(exp_all[i], len(rd))
            assert np.array_equal(ch[i][0], eu), i
            assert np.array_equal(ch[i][1], eb), i
    finally:
        sr.close()
        L.mm355_index_free(idx)


def test_anchor_cull_rule_does_not_lean_on_min_cnt(human_mid):
    """min_chain_score = 25 with k = 15: two seeds reach it, so a component of two anchors puts elements into z[] although it leaves no chain
    (min_cnt = 3) -- and z[] is sorted by score with the unstable radix sort, whose order of equal scores decides which chain end claims a
    shared anchor first.  The cull's T is ceil(min_chain_score / k) = 2 (only lone anchors go), not max(min_cnt, ..) = 3; chains == the
    oracle's chains of the WHOLE sorted array.  (Found on an HPC index, tests/test_gpu_hpc.py::test_hpc_index_is_never_culled.)"""
    from mappy_rs import _ffi
    L = _ffi.lib()
    g, names = human_mid
    reads, _ = S.make_read_block(4, 3, g, **ONT)
    reads = reads[:48]
    idx, mo = build_device_index(L, _ffi, g, names, "map-ont", extra={"min_chain_score": 25})
    orc = O.OracleAligner(codes=g, names=names, preset="map-ont", min_chain_score=25, n_threads=16)
    sr = _ffi.StageRunner(idx, mo, 0)
    try:
        ch = sr.chains(reads, cap=40_000_000)
        st = sr.stats()
        assert st.n_sort_fast_reads == len(reads) and 0 < st.n_a_kept < st.n_a, (st.n_sort_fast_reads, st.n_a_kept, st.n_a)
        n_u = 0
        for i, rd in enumerate(reads):
            exp, _, _, _ = orc.anchors(rd, sorted_=True)
            eu, eb = orc.chains(exp, len(rd))
            assert np.array_equal(ch[i][0], eu) and np.array_equal(ch[i][1], eb), i
            n_u += len(eu)
        assert n_u > 500
    finally:
        sr.close()
        L.mm355_index_free(idx)


def test_anchor_cull_in_many_passes(built):
    """k_cull with 8192-bin tables (1 KB per bitmap level instead of 48 KB) and 256 threads: the mid-scale genome's 38 k bins then take five
    passes over a read's anchors -- the pass boundaries and their guard bins, which only the 3.1-Gbp genome reaches with the default table --
    and everything test_anchor_cull_and_sort_mid_scale checks must still hold (a child process: the switches are read once)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MM355_CULL_WPL="256", MM355_CULL_MAX_PASS="16", MM355_CULL_NT="256")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_human.py", "-x", "-q", "-k", "test_anchor_cull_and_sort_mid_scale"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
