"""One level of U:ksort.h::rs_sort (the in-place cycle-leader permutation of radix_sort_128x) without moving elements one by one: the closed
form of mappy-rs_amd/csrc/mm355_wave.h::wave_rs_level_walk (and k_sort_level_mw), restated here step for step in numpy -- foreign elements and
their positions, the walk over the 1-byte label queues that yields arrival ranks, how many arrive before a bucket's own turn, the final
scatter -- and held against the LITERAL loop of the reference: exhaustively for short arrays, on random ones for every bucket structure the
GPU path meets (two non-empty buckets: the walk-free form; a few; all 256; one bucket holding almost everything).  The device code itself is
compared bit for bit through the anchor parity tests of the GPU suite; this is the argument it rests on, on the CPU."""
import itertools

import numpy as np


def literal_level(lab):
    """the element order after one level of rs_sort; lab[i] = bucket of element i.  Returns perm: out[j] = index of the element at j"""
    n = len(lab)
    arr = list(range(n))
    cnt = np.bincount(lab, minlength=256)
    be = np.cumsum(cnt); bb = list(be - cnt); be = list(be)
    k = 0
    while k < 256:
        if bb[k] != be[k]:
            l = lab[arr[bb[k]]]
            if l != k:
                tmp = arr[bb[k]]
                while True:
                    swap = tmp; tmp = arr[bb[l]]; arr[bb[l]] = swap; bb[l] += 1
                    l = lab[tmp]
                    if l == k:
                        break
                arr[bb[k]] = tmp; bb[k] += 1
            else:
                bb[k] += 1
        else:
            k += 1
    return arr


def closed_form_level(lab, two_bucket_shortcut=True):
    """wave_rs_level_walk, sequentially"""
    n = len(lab)
    lab = np.asarray(lab)
    cnt = np.bincount(lab, minlength=256)
    be = np.cumsum(cnt); bb = be - cnt
    rel = np.arange(n)
    foreign = ~((rel >= bb[lab]) & (rel < be[lab]))
    fpos = rel[foreign]; flab = lab[foreign]; nfor = len(fpos)
    fst = np.searchsorted(fpos, bb, side="left")          # first foreign slot at or after the start of region k
    fend = np.append(fst[1:], nfor)
    cur = fst.copy(); arr = np.zeros(256, np.int64); abef = np.zeros(256, np.int64); rank = np.zeros(max(nfor, 1), np.int64)
    if two_bucket_shortcut and int((cnt != 0).sum()) == 2:
        m = nfor >> 1
        rank[:nfor] = np.where(np.arange(nfor) < m, np.arange(nfor), np.arange(nfor) - m)
        abef = np.where((cnt != 0) & (bb != 0), m, 0)
    else:
        for k in range(256):
            abef[k] = arr[k]
            while cur[k] < fend[k]:
                c = k
                while True:
                    e = cur[c]; cur[c] += 1
                    g = flab[e]
                    rank[e] = arr[g]; arr[g] += 1
                    c = g
                    if c == k:
                        break
    out = [-1] * n
    pre_all = np.cumsum(foreign) - foreign                 # foreign elements before rel
    for i in range(n):
        g = lab[i]; al = abef[g]; f0 = fst[g]
        if foreign[i]:
            r = rank[pre_all[i]]
            dest = (bb[g] if r == 0 else fpos[f0 + r - 1] + 1) if r < al else fpos[f0 + r]
        else:
            dest = i + (1 if (pre_all[i] - f0) < al else 0)
        assert out[dest] == -1
        out[dest] = i
    return out


def test_exhaustive_short_arrays():
    n_cases = 0
    for n in range(1, 8):
        for lab in itertools.product((0, 1, 200), repeat=n):
            lab = np.array(lab)
            assert closed_form_level(lab) == literal_level(lab), lab
            n_cases += 1
    for lab in itertools.product((0, 1, 2, 3), repeat=6):
        lab = np.array(lab)
        assert closed_form_level(lab) == literal_level(lab), lab
        n_cases += 1
    assert n_cases > 7000


def test_random_arrays_of_every_bucket_structure():
    rng = np.random.default_rng(3)
    for it in range(300):
        n = int(rng.integers(1, 1500))
        kind = it % 6
        if kind == 0:   labs = rng.choice(256, 2, replace=False); lab = rng.choice(labs, n)                      # two buckets (strand byte): no walk
        elif kind == 1: labs = rng.choice(256, int(rng.integers(3, 30)), replace=False); lab = rng.choice(labs, n)   # a few (the rid byte)
        elif kind == 2: lab = rng.integers(0, 256, n)                                                             # all of them
        elif kind == 3: lab = np.where(rng.random(n) < 0.95, 7, rng.integers(0, 256, n))                          # one bucket holds nearly everything
        elif kind == 4: lab = np.sort(rng.integers(0, 256, n)); m = rng.random(n) < 0.05; lab[m] = rng.integers(0, 256, int(m.sum()))   # nearly sorted: few foreign elements
        else:           lab = np.sort(rng.integers(0, 40, n))[::-1].copy()                                        # reversed: everything foreign
        lab = np.asarray(lab)
        lit = literal_level(lab)
        assert closed_form_level(lab) == lit, (it, n, kind)
        if int((np.bincount(lab, minlength=256) != 0).sum()) == 2:
            assert closed_form_level(lab, two_bucket_shortcut=False) == lit, (it, n)      # the walk agrees with its own shortcut
