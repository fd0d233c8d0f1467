"""The claim k_chain_segments / k_chain_small / k_chain_big (mappy-rs_amd/csrc/mm355_kernels.hip) rest on, checked on the CPU with the oracle's
literal mg_lchain_dp fill: the sorted anchor array of a read can be cut wherever strand / contig change or two consecutive reference positions
lie more than max_dist_x apart, and every piece chained ON ITS OWN gives the f / p / v / t of the whole-array run (p and t shifted by the
piece's offset) -- no window, no `t[]` mark, no `max_ii` rescue and no `max_iter` / `max_skip` cut-off reaches across such a boundary.
Options that move the heuristics (short max_chain_iter, small max_chain_skip, a band wider than max_gap, max_gap_ref) included."""
import numpy as np
import pytest

from oracle import oracle as O
import synthdata as S


@pytest.fixture(scope="module")
def world(built, tmp_path_factory):
    td = tmp_path_factory.mktemp("seg")
    g = S.make_genome(43, [2500000, 1500000], repeats=((300, 1500, 0.08), (1500, 150, 0.04), (171, 600, 0.03), (5000, 6, 0.01)), n_runs=2)
    fa = str(td / "ref.fa")
    S.write_fasta(fa, g, ["chrA", "chrB"])
    reads, _ = S.make_reads(44, g, 36, n50=7000, lo=400)
    return dict(fa=fa, reads=reads)


@pytest.mark.parametrize("kw", [{}, {"max_chain_iter": 40}, {"max_chain_skip": 3}, {"bw": 8000}, {"max_gap_ref": 1500}, {"max_gap": 800}])
def test_segments_chain_independently(world, kw):
    fields = {k: kw.pop(k) for k in list(kw) if k in ("max_chain_iter", "max_chain_skip", "max_gap_ref", "max_gap")}
    orc = O.OracleAligner(world["fa"], preset="map-ont", **kw)
    for k, v in fields.items():
        setattr(orc.mo, k, v)
    mo = orc.mo
    n_seg = n_multi = n_anchor = 0
    for rd in world["reads"]:
        a, _, _, _ = orc.anchors(rd, sorted_=True)
        n = len(a)
        if n == 0:
            continue
        qlen = len(rd)
        mdx = mo.max_gap_ref if mo.max_gap_ref > 0 else (max(mo.max_frag_len - qlen, mo.max_gap) if mo.max_frag_len > 0 else mo.max_gap)
        mdx = max(mdx, mo.bw)
        x = a[:, 0]
        start = np.ones(n, bool)
        start[1:] = ((x[1:] >> np.uint64(32)) != (x[:-1] >> np.uint64(32))) | (x[1:] > x[:-1] + np.uint64(mdx))      # the rule of k_chain_segments
        f, p, v, t = orc.chain_fill(a, qlen)
        bounds = np.append(np.nonzero(start)[0], n)
        for b, e in zip(bounds[:-1], bounds[1:]):
            fs, ps, vs, ts = orc.chain_fill(a[b:e], qlen)
            assert np.array_equal(fs, f[b:e]) and np.array_equal(vs, v[b:e])
            assert np.array_equal(np.where(ps >= 0, ps + b, ps), p[b:e])
            n_seg += 1; n_multi += e - b > 1
        n_anchor += n
    assert n_seg > 500 and n_multi > 50 and n_anchor > 5000, (n_seg, n_multi, n_anchor)
