"""The claim k_cull (mappy-rs_amd/csrc/mm355_cullsort.hip) rests on, checked on the CPU with the oracle's literal mg_lchain_dp: anchors whose
run of non-empty position bins holds fewer than T = ceil(min_chain_score / k) anchors can be deleted from the SORTED array
before chaining -- u[] and the compacted anchors do not change.  The rule is restated here in numpy exactly as the kernel applies it (bins of
2^sh >= max_dist_x bases of the concatenated, strand-doubled reference; a run = consecutive non-empty bins; counts as they are, no
saturation -- the kernel only ever keeps MORE), on a genome whose repeat families scatter lone hits all over it; also with a larger min_cnt (T unchanged: min_cnt is not part of the rule) /
min_chain_score (larger T) and with bins that are too NARROW (sh below log2 max_dist_x), where the deletion must be seen to change results --
the bin width is what the argument needs.  GPU side: tests/test_gpu_human.py::test_anchor_cull_and_sort_mid_scale."""
import numpy as np
import pytest

from oracle import oracle as O
import synthdata as S


def cull(a, seq_off, tot_len, sh, T):
    """keep[i] for the sorted anchor array a[n, 2]: the rule of k_cull"""
    x = a[:, 0]
    strand = (x >> np.uint64(63)).astype(np.int64); rid = ((x >> np.uint64(32)) & np.uint64(0x7fffffff)).astype(np.int64); rpos = (x & np.uint64(0xffffffff)).astype(np.int64)
    pos = strand * tot_len + seq_off[rid] + rpos
    b = pos >> sh
    ub, inv, cnt = np.unique(b, return_inverse=True, return_counts=True)
    run = np.cumsum(np.concatenate(([True], ub[1:] - ub[:-1] > 1))) - 1
    tot = np.bincount(run, weights=cnt)
    return tot[run][inv] >= T


@pytest.fixture(scope="module")
def world(built, tmp_path_factory):
    td = tmp_path_factory.mktemp("cull")
    g = S.make_genome(41, [14000000, 9000000], repeats=((300, 9000, 0.10), (1500, 700, 0.06), (171, 2500, 0.03)), n_runs=2)
    fa = str(td / "ref.fa")
    S.write_fasta(fa, g, ["chrA", "chrB"])
    reads, _ = S.make_reads(42, g, 40, n50=6000, lo=500)
    return dict(fa=fa, g=g, reads=reads)


@pytest.mark.parametrize("kw", [{}, {"min_cnt": 5}, {"min_chain_score": 100}])
def test_culled_anchors_never_chain(world, kw):
    orc = O.OracleAligner(world["fa"], preset="map-ont", **kw)
    g = world["g"]
    seq_off = np.concatenate(([0], np.cumsum([len(c) for c in g])[:-1])).astype(np.int64)
    tot_len = int(sum(len(c) for c in g))
    mo = orc.mo
    D = max(mo.max_gap_ref if mo.max_gap_ref > 0 else mo.max_gap, mo.bw)
    sh = int(np.ceil(np.log2(D)))
    T = -(-mo.min_chain_score // orc.k)
    n_all = n_kept = n_chains = 0
    for rd in world["reads"]:
        a, _, _, _ = orc.anchors(rd, sorted_=True)
        if len(a) == 0:
            continue
        keep = cull(a, seq_off, tot_len, sh, T)
        u0, b0 = orc.chains(a, len(rd))
        u1, b1 = orc.chains(a[keep], len(rd))
        assert np.array_equal(u0, u1) and np.array_equal(b0, b1)
        n_all += len(a); n_kept += int(keep.sum()); n_chains += len(u0)
    assert n_chains > 30 and n_kept < 0.8 * n_all, (n_all, n_kept, n_chains)      # the test data must have something to drop


def test_bins_narrower_than_max_dist_x_are_not_enough(world):
    """with 64-base bins anchors of one chain fall into runs of their own: deleting by that rule loses chains"""
    orc = O.OracleAligner(world["fa"], preset="map-ont")
    g = world["g"]
    seq_off = np.concatenate(([0], np.cumsum([len(c) for c in g])[:-1])).astype(np.int64)
    tot_len = int(sum(len(c) for c in g))
    differs = 0
    for rd in world["reads"][:30]:
        a, _, _, _ = orc.anchors(rd, sorted_=True)
        if len(a) == 0:
            continue
        keep = cull(a, seq_off, tot_len, 6, 3)
        u0, b0 = orc.chains(a, len(rd))
        u1, b1 = orc.chains(a[keep], len(rd))
        differs += not (np.array_equal(u0, u1) and np.array_equal(b0, b1))
    assert differs > 0


def test_equal_positions_share_the_decision(world):
    """What the literal path of mm355_cull_sort relies on when it only descends into buckets that hold equal positions among KEPT anchors
    (k_tie_tcnt, kept_only): anchors with the same x fall into the same bin, so the rule keeps all of them or none -- a pair of culled
    anchors is dropped whatever order the unstable sort would have left it in."""
    orc = O.OracleAligner(world["fa"], preset="map-ont")
    g = world["g"]
    seq_off = np.concatenate(([0], np.cumsum([len(c) for c in g])[:-1])).astype(np.int64)
    tot_len = int(sum(len(c) for c in g))
    mo = orc.mo
    D = max(mo.max_gap_ref if mo.max_gap_ref > 0 else mo.max_gap, mo.bw)
    sh = int(np.ceil(np.log2(D)))
    T = -(-mo.min_chain_score // orc.k)
    n_pairs = n_kept_pairs = 0
    for rd in world["reads"]:
        a, _, _, _ = orc.anchors(rd, sorted_=True)
        if len(a) < 2:
            continue
        keep = cull(a, seq_off, tot_len, sh, T)
        same = a[1:, 0] == a[:-1, 0]
        assert np.array_equal(keep[1:][same], keep[:-1][same])
        n_pairs += int(same.sum()); n_kept_pairs += int((same & keep[1:]).sum())
    assert n_pairs > 0 and n_kept_pairs < n_pairs, (n_pairs, n_kept_pairs)   # the data has such pairs, and the rule drops some of them


def test_the_rule_counts_spans_not_min_cnt(built, tmp_path):
    """Why T is ceil(min_chain_score / LARGEST SPAN) and nothing else.  On a homopolymer-compressed index (ava-pb: k = 19, min_chain_score 100,
    min_cnt 3) a seed's span is a sum of run lengths up to 255: one or two seeds reach min_chain_score, sit in z[] while mg_chain_backtrack
    sorts it with the unstable radix sort, and the order of equal scores decides which chain end claims a shared anchor first.  Dropping
    components of fewer than ceil(100 / 19) = 6 anchors changes the chains of half the reads, fewer than min_cnt = 3 still of some (the GPU
    path did exactly that for one session: tests/test_gpu_hpc.py::test_hpc_index_is_never_culled); T = ceil(100 / 255) = 1 drops nothing."""
    from test_host import _hp_genome
    g = _hp_genome(57, [400000], repeats=((700, 60, 0.01), (300, 100, 0.02), (2000, 8, 0.005)), n_runs=1)
    fa = str(tmp_path / "c.fa")
    S.write_fasta(fa, g, ["c"])
    orc = O.OracleAligner(fa, preset="ava-pb")
    mo = orc.mo
    reads, _ = S.make_reads(58, g, 40, n50=18000, lo=13000, sub=0.01, ins=0.02, dele=0.02)
    sh = int(np.ceil(np.log2(max(mo.max_gap, mo.bw))))
    differ = {}
    for T in (-(-mo.min_chain_score // 255), mo.min_cnt, -(-mo.min_chain_score // orc.k)):
        d = 0
        for rd in reads:
            a, _, _, _ = orc.anchors(rd, sorted_=True)
            keep = cull(a, np.array([0], np.int64), len(g[0]), sh, T)
            u0, b0 = orc.chains(a, len(rd))
            u1, b1 = orc.chains(a[keep], len(rd))
            d += not (np.array_equal(u0, u1) and np.array_equal(b0, b1))
        differ[T] = d
    assert differ[1] == 0 and differ[3] > 0 and differ[6] > differ[3], differ
