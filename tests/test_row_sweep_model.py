"""The claim k_ksw_row (mappy-rs_amd/csrc/mm355_dprow.h) rests on, checked on the CPU against the oracle's literal restatement of
U:ksw2_extd2_sse.c: for a gap fill whose band never binds (KSW_EZ_APPROX_MAX, w >= qlen + tlen) and a regular two-piece cost (after
ksw2's ordering e > e2, or two identical pieces) the SSE kernel's score and CIGAR are those of the plain two-piece affine recurrence
evaluated in ANY order, with the direction byte of a cell a function of the true H / E / F / E2 / F2 --
    d & 7 = first (KSW_EZ_RIGHT: last) maximum among (H(t-1,q-1)+s, E, F, E2, F2);  0x08: E - H + q > 0 (RIGHT >= 0); 0x10: F; 0x20/0x40: E2, F2 with q2
and E along a row an exclusive prefix maximum, E(t) = max_{k<t}(G(k) + k e) - q - t e with G = max(M, F, F2).
This numpy model is the row sweep; the HIP kernel is compared with the oracle on the GPU (tests/test_gpu_map.py::test_dp_kernel_parity).
For an IRREGULAR cost (e == e2 with q != q2, or e < e2) the SSE kernel's boundary follows the dearer piece and its clamp z <= sc_mch
becomes active: there the plain recurrence differs (second test), which is why the product keeps those options on the literal kernels."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
import synthdata as S


def consts(q, e, q2, e2):
    if q2 + e2 < q + e: q, q2, e, e2 = q2, q, e2, e
    lt = (q2 - q) // (e - e2) - 1 if e != e2 else 0
    if q2 + e2 + lt * e2 > q + e + lt * e: lt += 1
    return q, e, q2, e2, lt, lt * (e - e2) - (q2 - q) - e2


def hb(t, q, e, q2, e2, lt, ld):
    """H(t, -1) = H(-1, t): the sum of the first t + 1 boundary differences of the SSE kernel"""
    if t < 0: return 0
    n1 = max(min(lt - 1, t), 0)
    has = 1 if 1 <= lt <= t else 0
    return -(q + e) - n1 * e + has * ld - (t - n1 - has) * e2


def row_sweep(query, target, a, b, amb, q, e, q2, e2, right):
    qe_given = q + e
    q, e, q2, e2, lt, ld = consts(q, e, q2, e2)
    Q, T = len(query), len(target)
    sc_n = -e2 if amb == 0 else -abs(amb)
    Hp = np.array([hb(t, q, e, q2, e2, lt, ld) for t in range(T)], dtype=np.int64)
    Fp = np.full(T, -16384); F2p = np.full(T, -16384)
    P = np.zeros((Q, T), dtype=np.uint8)
    hl_prev, hl = 0, hb(0, q, e, q2, e2, lt, ld)
    ke, ke2 = np.arange(T) * e, np.arange(T) * e2
    for j in range(Q):
        Hd = np.concatenate([[hl_prev], Hp[:-1]])
        s = np.where(target == query[j], a, -b); s = np.where((target == 4) | (query[j] == 4), sc_n, s)
        M = Hd + s
        F = np.maximum(Hp - q - e, Fp - e); F2 = np.maximum(Hp - q2 - e2, F2p - e2)
        G = np.maximum(np.maximum(M, F), F2)
        pre = np.maximum.accumulate(np.concatenate([[hl - e], G + ke]))[:-1]       # exclusive prefix maximum, the left border first
        pre2 = np.maximum.accumulate(np.concatenate([[hl - e2], G + ke2]))[:-1]
        E, E2 = pre - q - ke, pre2 - q2 - ke2
        H = np.maximum(np.maximum(G, E), E2)
        cands = [M, E, F, E2, F2]
        d = np.zeros(T, dtype=np.uint8)
        for k in (range(4, -1, -1) if not right else range(5)):
            d = np.where(cands[k] == H, k, d)
        gt = (lambda x: x >= 0) if right else (lambda x: x > 0)
        d = d | np.where(gt(E - H + q), 8, 0) | np.where(gt(F - H + q), 16, 0) | np.where(gt(E2 - H + q2), 32, 0) | np.where(gt(F2 - H + q2), 64, 0)
        P[j] = d
        Hp, Fp, F2p = H, F, F2
        hl_prev, hl = hl, hb(j + 1, q, e, q2, e2, lt, ld)
    # (the SSE kernel anchors the absolute score with the (q + e) it was GIVEN, before ordering the two pieces: a constant when it swaps them)
    return int(Hp[T - 1]) + (q + e) - qe_given, P


def backtrack(P, Q, T):
    """U:ksw2.h::ksw_backtrack on a row-major direction matrix (every visited cell is inside the matrix: no forced states)"""
    i, j, state, cig = T - 1, Q - 1, 0, []

    def push(op, n):
        if cig and cig[-1][1] == op: cig[-1][0] += n
        else: cig.append([n, op])
    while i >= 0 and j >= 0:
        tmp = int(P[j, i])
        if state == 0: state = tmp & 7
        elif not (tmp >> (state + 2)) & 1: state = 0
        if state == 0: state = tmp & 7
        if state == 0: push(0, 1); i -= 1; j -= 1
        elif state in (1, 3): push(2, 1); i -= 1
        else: push(1, 1); j -= 1
    if i >= 0: push(2, i + 1)
    if j >= 0: push(1, j + 1)
    return [(n << 4) | op for n, op in cig[::-1]]


def run_config(rng, a, b, amb, q, e, q2, e2, n_jobs):
    OL = O.lib()
    mat = np.zeros(25, np.int8); OL.mmo_ksw_gen_simple_mat(5, mat.ctypes.data, a, b, amb)
    bad = 0
    for it in range(n_jobs):
        tl = int(rng.integers(1, 260)); t = S.random_codes(rng, tl); x = S.mutate(t, rng, 0.06, 0.03, 0.03)
        if it % 5 == 0 and len(x) > 40:
            cut = int(rng.integers(5, len(x) - 30)); x = np.concatenate([x[:cut], x[cut + int(rng.integers(1, 60)):]])
        if it % 7 == 0: x = S.random_codes(rng, int(rng.integers(1, 400)))
        if it % 9 == 0 and len(x) > 6: x[len(x) // 2:len(x) // 2 + 2] = 4
        if len(x) == 0: x = S.random_codes(rng, 1)
        x, t = x.astype(np.uint8), t.astype(np.uint8)
        right = it % 2
        ez = O.Extz()
        OL.mmo_ksw_extd2(len(x), x.ctypes.data, tl, t.ctypes.data, 5, mat.ctypes.data, q, e, q2, e2, len(x) + tl + 5, 400, -1, 8 | (2 if right else 0), C.byref(ez))
        exp = [ez.cigar[k] for k in range(ez.n_cigar)]
        sc, P = row_sweep(x, t, a, b, amb, q, e, q2, e2, right)
        bad += sc != ez.score or backtrack(P, len(x), tl) != exp
        if ez.n_cigar: OL.free(ez.cigar)
    return bad


REGULAR = [(2, 4, 1, 4, 2, 24, 1), (1, 4, 1, 6, 2, 26, 1), (1, 19, 1, 39, 3, 81, 1), (1, 9, 1, 16, 2, 41, 1), (5, 6, 1, 5, 4, 28, 2), (1, 10, 0, 1, 4, 11, 2),
           (4, 10, 1, 3, 2, 3, 2), (3, 8, 2, 5, 4, 13, 1), (2, 3, 1, 1, 4, 23, 1), (4, 1, 0, 8, 4, 28, 1), (4, 6, 1, 10, 1, 3, 4)]   # the last one is re-ordered by ksw2 (q + e > q2 + e2)


@pytest.mark.parametrize("cfg", REGULAR)
def test_row_sweep_equals_the_sse_kernel_for_regular_costs(built, cfg):
    a, b, amb, q, e, q2, e2 = cfg
    qq, ee, qq2, ee2, _lt, _ld = consts(q, e, q2, e2)
    assert ee > ee2 or (ee == ee2 and qq == qq2)
    assert run_config(np.random.default_rng(sum(cfg)), a, b, amb, q, e, q2, e2, 40) == 0


def test_irregular_costs_are_not_the_plain_recurrence(built):
    """e == e2 with q != q2: the SSE kernel's boundary follows the dearer piece, its clamp is active, the plain recurrence scores higher"""
    assert run_config(np.random.default_rng(3), 4, 10, 1, 3, 3, 12, 3, 60) > 0
