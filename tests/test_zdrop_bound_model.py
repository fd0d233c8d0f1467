"""The shortcut of the extension driver (mappy-rs_amd/csrc/mm355_glue.cpp::task_run, "quiet"): U:align.c::mm_test_zdrop walks the path of a
gap fill and returns 0 unless some score drop along it exceeds zdrop (or zdrop_inv); the driver skips the walk -- and the fetch of the
reference bases it needs -- when a bound on every drop is within both thresholds:
    drop <= a M - G2 - score + G1       (M match columns, G1 / G2 the one- / two-piece costs of the gaps, score the fill's global score)
(a drop is minus a sum of increments of the walk's score: gap costs, charged one-piece by the walk, and substitution scores, each at most a;
a M - sum of substitution scores = a M - G2 - score, because a full-band fill's `score` IS the two-piece score of the path in its CIGAR).
Here on the CPU: gap fills by the oracle's ksw_extd2 on sequence pairs from clean to heavily diverged, the bound restated in numpy, and the
oracle's literal mm_test_zdrop -- whenever the bound says quiet, the walk returns 0; and the bound is not vacuous (most clean fills are quiet)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
import synthdata as S


@pytest.mark.parametrize("scoring", [(2, 4, 4, 2, 24, 1), (1, 4, 6, 2, 26, 1), (1, 9, 16, 2, 41, 1), (2, 6, 5, 2, 30, 1)])
def test_quiet_bound_implies_no_zdrop(built, scoring):
    L = O.lib()
    L.mmo_test_zdrop.argtypes = [C.POINTER(O.MapOpt), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    a, b, q, e, q2, e2 = scoring
    io, mo = O.IdxOpt(), O.MapOpt()
    L.mmo_set_opt(None, C.byref(io), C.byref(mo))
    mo.a, mo.b, mo.q, mo.e, mo.q2, mo.e2 = a, b, q, e, q2, e2
    mat = np.zeros(25, np.int8)
    L.mmo_ksw_gen_simple_mat(5, mat.ctypes.data, a, b, mo.sc_ambi)
    rng = np.random.default_rng(17)
    n_quiet = n_loud_bound = n_zdrop = n = 0
    for it in range(400):
        ln = int(rng.integers(30, 900))
        t = S.random_codes(rng, ln)
        div = float(rng.choice([0.0, 0.02, 0.06, 0.15, 0.35]))
        qy = S.mutate(t, rng, div * 0.5, div * 0.25, div * 0.25)
        if rng.random() < 0.2 and ln > 200:      # a long indel in the middle
            c = int(rng.integers(50, ln - 50)); qy = np.concatenate([qy[:c], S.random_codes(rng, int(rng.integers(30, 300))), qy[c:]])
        if rng.random() < 0.25 and len(qy) > 500:   # an unrelated block of the same length: the walk's score falls by hundreds
            c = int(rng.integers(50, len(qy) - 450)); qy = qy.copy(); qy[c:c + 400] = S.random_codes(rng, 400)
        if rng.random() < 0.1:
            qy = qy.copy(); qy[int(rng.integers(0, len(qy)))] = 4
        qy = np.ascontiguousarray(qy, np.uint8); t = np.ascontiguousarray(t, np.uint8)
        ez = O.Extz()
        w = max(len(qy), len(t))
        L.mmo_ksw_extd2(len(qy), qy.ctypes.data, len(t), t.ctypes.data, 5, mat.ctypes.data, q, e, q2, e2, w, mo.zdrop, -1, 0x08, C.byref(ez))   # KSW_EZ_APPROX_MAX: the fills' flag
        if ez.n_cigar <= 0:
            continue
        cg = np.ctypeslib.as_array(ez.cigar, shape=(ez.n_cigar,)).copy()
        op, ln_ = cg & 0xf, (cg >> 4).astype(np.int64)
        M = int(ln_[op == 0].sum())
        gaps = ln_[(op == 1) | (op == 2)]
        G1 = int((q + e * gaps).sum()); G2 = int(np.minimum(q + e * gaps, q2 + e2 * gaps).sum())
        loss = a * M - G2 - ez.score
        assert loss >= 0, (it, loss)                               # `score` is the two-piece score of the CIGAR's path
        quiet = loss + G1 <= min(mo.zdrop, mo.zdrop_inv)
        code = L.mmo_test_zdrop(C.byref(mo), qy.ctypes.data, t.ctypes.data, ez.n_cigar, cg.ctypes.data, mat.ctypes.data)
        if quiet:
            assert code == 0, (it, loss, G1, code)
        n_quiet += quiet; n_zdrop += code != 0; n += 1
        L.free(ez.cigar)
    assert n > 350 and n_quiet > 100 and n_zdrop > 5, (n, n_quiet, n_zdrop)
