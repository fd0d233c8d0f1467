"""The claim the band kernels (mappy-rs_amd/csrc/mm355_dpband.h: k_ksw_band2, k_ksw_band<1|2|4>) rest on, checked on the CPU against the
oracle's literal restatement of U:ksw2_extd2_sse.c.  For a full-band approximate gap fill, run the two-piece affine recurrence on the diagonals
d = t - q in [dlo, dlo + W) only -- everything outside at -infinity -- and let L be the end cell's score.  With
    U(d) = a (qlen + tlen - |d| - |D0 - d|) / 2 - cost(|d|) - cost(|D0 - d|),   D0 = tlen - qlen,  cost(g) = min(q + g e, q2 + g e2)
(no path through diagonal d outside [min(0, D0), max(0, D0)] scores more), U(dlo - 1) < L and U(dlo + W) < L imply that score and CIGAR are
those of the FULL matrix.  This numpy model is written in row coordinates with a mask per row -- not the kernel's diagonal registers -- so
it is a second derivation of the same restricted recurrence; the HIP kernels are compared with the oracle on the GPU
(tests/test_gpu_map.py::test_dp_kernel_parity, ::test_dp_band_kernels_forced).  Checked here: whenever the proof holds the result equals the
oracle's; the proof holds for nearly all plain fills and fails for the problems a band cannot hold (unrelated sequences, long indels)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
import synthdata as S
from test_row_sweep_model import consts, hb, backtrack

NEG = -(1 << 28)


def band_sweep(query, target, a, b, amb, q, e, q2, e2, right, dlo, W):
    """the row sweep of test_row_sweep_model.py with every cell outside the band at -infinity (the border column t = -1 and the border row
    q = -1 are cells of the extended matrix: inside the band they carry the boundary values, outside they are -infinity too)"""
    qe_given = q + e
    q, e, q2, e2, lt, ld = consts(q, e, q2, e2)
    Q, T = len(query), len(target)
    sc_n = -e2 if amb == 0 else -abs(amb)
    cols = np.arange(T)
    inb = lambda row, t: (t - row >= dlo) & (t - row < dlo + W)
    Hp = np.where(inb(-1, cols), np.array([hb(t, q, e, q2, e2, lt, ld) for t in range(T)], dtype=np.int64), NEG)
    Fp = np.full(T, NEG, dtype=np.int64); F2p = np.full(T, NEG, dtype=np.int64)
    P = np.full((Q, T), 255, dtype=np.uint8)
    hl_prev = 0 if inb(-1, -1) else NEG                      # H(-1, -1)
    border = lambda row: hb(row, q, e, q2, e2, lt, ld) if inb(row, -1) else NEG
    hl = border(0)
    ke, ke2 = cols * e, cols * e2
    for j in range(Q):
        m = inb(j, cols)
        Hd = np.concatenate([[hl_prev], Hp[:-1]])
        s = np.where(target == query[j], a, -b); s = np.where((target == 4) | (query[j] == 4), sc_n, s)
        M = np.where(Hd > NEG // 2, Hd + s, NEG)
        F = np.maximum(np.where(Hp > NEG // 2, Hp - q - e, NEG), np.where(Fp > NEG // 2, Fp - e, NEG))
        F2 = np.maximum(np.where(Hp > NEG // 2, Hp - q2 - e2, NEG), np.where(F2p > NEG // 2, F2p - e2, NEG))
        G = np.where(m, np.maximum(np.maximum(M, F), F2), NEG)
        g1 = np.where(G > NEG // 2, G + ke, NEG); g2 = np.where(G > NEG // 2, G + ke2, NEG)
        pre = np.maximum.accumulate(np.concatenate([[hl - e if hl > NEG // 2 else NEG], g1]))[:-1]
        pre2 = np.maximum.accumulate(np.concatenate([[hl - e2 if hl > NEG // 2 else NEG], g2]))[:-1]
        E = np.where(pre > NEG // 2, pre - q - ke, NEG); E2 = np.where(pre2 > NEG // 2, pre2 - q2 - ke2, NEG)
        H = np.where(m, np.maximum(np.maximum(G, E), E2), NEG)
        cands = [M, E, F, E2, F2]
        d = np.zeros(T, dtype=np.uint8)
        for k in (range(4, -1, -1) if not right else range(5)):
            d = np.where(cands[k] == H, k, d)
        gt = (lambda x: x >= 0) if right else (lambda x: x > 0)
        d = d | np.where(gt(E - H + q), 8, 0) | np.where(gt(F - H + q), 16, 0) | np.where(gt(E2 - H + q2), 32, 0) | np.where(gt(F2 - H + q2), 64, 0)
        P[j] = np.where(m, d, 255)
        Hp, Fp, F2p = H, np.where(m, F, NEG), np.where(m, F2, NEG)
        hl_prev, hl = hl, border(j + 1)
    return int(Hp[T - 1]), int(Hp[T - 1]) + (q + e) - qe_given, P


def ubound(qlen, tlen, d, a, q, e, q2, e2):
    D0 = tlen - qlen
    g1, g2 = abs(d), abs(D0 - d)
    cost = lambda g: 0 if g <= 0 else min(q + g * e, q2 + g * e2)
    return a * max(0, (qlen + tlen - g1 - g2) // 2) - cost(g1) - cost(g2)


def plan(qlen, tlen, W, a, q, e, q2, e2):
    D0 = tlen - qlen
    lo, hi = min(0, D0), max(0, D0)
    span = hi - lo + 1
    if W < span + 16:
        return None
    dlo = lo - (W - span) // 2
    return dlo, max(ubound(qlen, tlen, dlo - 1, a, q, e, q2, e2), ubound(qlen, tlen, dlo + W, a, q, e, q2, e2)) + 1


CFG = [(2, 4, 1, 4, 2, 24, 1), (1, 4, 1, 6, 2, 26, 1), (1, 9, 1, 16, 2, 41, 1), (4, 6, 1, 10, 1, 3, 4), (2, 5, 1, 5, 2, 5, 2)]


@pytest.mark.parametrize("cfg", CFG)
def test_a_proven_band_gives_the_full_matrix_result(built, cfg):
    a, b, amb, q, e, q2, e2 = cfg
    qq, ee, qq2, ee2, _lt, _ld = consts(q, e, q2, e2)
    OL = O.lib()
    mat = np.zeros(25, np.int8); OL.mmo_ksw_gen_simple_mat(5, mat.ctypes.data, a, b, amb)
    rng = np.random.default_rng(sum(cfg) + 7)
    n_pass = n_fail = n_hard_fail = 0
    for it in range(60):
        tl = int(rng.integers(20, 330)); t = S.random_codes(rng, tl); x = S.mutate(t, rng, 0.04, 0.02, 0.02)
        hard = False
        if it % 6 == 0 and len(x) > 60:
            cut = int(rng.integers(5, len(x) - 50)); x = np.concatenate([x[:cut], x[cut + int(rng.integers(20, 45)):]]); hard = True     # a long deletion: the path leaves a narrow band
        if it % 10 == 3: x = S.random_codes(rng, int(rng.integers(max(1, tl - 30), tl + 30))); hard = True                         # unrelated sequences
        if it % 9 == 0 and len(x) > 6: x[len(x) // 2:len(x) // 2 + 2] = 4
        if len(x) == 0: x = S.random_codes(rng, 1)
        x, t = x.astype(np.uint8), t.astype(np.uint8)
        right = it % 2
        ez = O.Extz()
        OL.mmo_ksw_extd2(len(x), x.ctypes.data, tl, t.ctypes.data, 5, mat.ctypes.data, q, e, q2, e2, len(x) + tl + 5, 400, -1, 8 | (2 if right else 0), C.byref(ez))
        exp = [ez.cigar[k] for k in range(ez.n_cigar)]
        for W in (64, 128):
            pl = plan(len(x), tl, W, a, qq, ee, qq2, ee2)
            if pl is None:
                continue
            dlo, lmin = pl
            h_end, sc, P = band_sweep(x, t, a, b, amb, q, e, q2, e2, right, dlo, W)
            if h_end >= lmin:                                # the proof holds: identical score and CIGAR, and the walk never leaves the band
                assert sc == ez.score and backtrack(P, len(x), tl) == exp, (cfg, it, W, len(x), tl)
                n_pass += 1
            else:
                n_fail += 1; n_hard_fail += hard
        if ez.n_cigar: OL.free(ez.cigar)
    assert n_pass >= 30 and n_fail >= 3 and n_hard_fail >= 3, (n_pass, n_fail, n_hard_fail)      # plain fills prove their band (fewer under harsh costs); the hard ones go to the full matrix


def test_the_bound_is_needed(built):
    """without the proof a band result can differ: a 40-base deletion on a band of 64 diagonals centred on the main diagonal"""
    a, b, amb, q, e, q2, e2 = 2, 4, 1, 4, 2, 24, 1
    OL = O.lib()
    mat = np.zeros(25, np.int8); OL.mmo_ksw_gen_simple_mat(5, mat.ctypes.data, a, b, amb)
    rng = np.random.default_rng(11)
    t = S.random_codes(rng, 200).astype(np.uint8)
    x = np.concatenate([t[:80], t[80:200]]).astype(np.uint8)
    x = np.concatenate([x[:60], S.random_codes(rng, 40).astype(np.uint8), x[60:]])      # an insertion of 40 unrelated bases in the query
    x = np.concatenate([x, x[-45:]])[:len(x) + 45]                                     # ... and 45 more at its end: D0 = -85 does not fit a band that holds 0 +- 20
    ez = O.Extz()
    OL.mmo_ksw_extd2(len(x), x.ctypes.data, 200, t.ctypes.data, 5, mat.ctypes.data, q, e, q2, e2, len(x) + 205, 400, -1, 8, C.byref(ez))
    assert plan(len(x), 200, 64, a, q, e, q2, e2) is None          # the host does not even try: the end cell's diagonal is outside any 64-band around 0
    pl = plan(len(x), 200, 128, a, q, e, q2, e2)
    assert pl is not None
    h_end, sc, P = band_sweep(x, t, a, b, amb, q, e, q2, e2, 0, pl[0], 128)
    assert (h_end >= pl[1]) == (sc == ez.score)                   # the proof never passes a wrong score
    if ez.n_cigar: OL.free(ez.cigar)
