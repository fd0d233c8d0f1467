"""Size-independent properties of the mapping path that need no oracle (used at BASELINE.json's full sizes):
(1) the hits of a read do not depend on which other reads share its batch, (2) a second run is identical, (3) every CIGAR spans exactly
its query and target intervals, (4) cs agrees with the CIGAR and NM, (5) reads map back to where they were drawn."""
import ctypes as C
import re

import numpy as np


def run_batch(L, ctx, mo, sub, flags=1):
    from mappy_rs import _ffi
    import mappy_rs
    rarr, rlens, keep = _ffi.pack_reads(sub)
    hp = C.POINTER(_ffi.Hits)()
    _ffi.check(L.mm355_map_batch(ctx, C.byref(mo), len(sub), rarr, rlens, flags, C.byref(hp)))
    h = hp.contents
    off = np.ctypeslib.as_array(h.hit_off, shape=(len(sub) + 1,)).copy()
    hits = np.frombuffer(C.string_at(h.hits, int(h.n_hits) * C.sizeof(_ffi.Hit)), dtype=mappy_rs._HIT_DTYPE).copy()
    cig = np.ctypeslib.as_array(h.cigar, shape=(max(int(h.n_cigar), 1),)).copy()
    sbuf = C.string_at(h.str, int(h.n_str)) if h.n_str else b""
    L.mm355_free_hits(hp)
    return off, hits, cig, sbuf


def per_read(off, hits, cig, sbuf):
    out = []
    for i in range(len(off) - 1):
        rs = []
        for k in range(off[i], off[i + 1]):
            x = hits[k]
            rs.append((int(x["rid"]), int(x["target_start"]), int(x["target_end"]), int(x["query_start"]), int(x["query_end"]), int(x["strand"]),
                       int(x["mapq"]), int(x["NM"]), cig[x["cigar_off"]:x["cigar_off"] + x["n_cigar"]].tobytes(),
                       sbuf[x["cs_off"]:x["cs_off"] + x["cs_len"]], int(x["is_primary"])))
        out.append(rs)
    return out


def check_properties(L, ctx, mo, reads, truth, cuts):
    """returns (n_hits, n_right): hit records checked, reads whose first hit overlaps the locus they were drawn from"""
    whole = per_read(*run_batch(L, ctx, mo, reads))
    assert per_read(*run_batch(L, ctx, mo, reads)) == whole                        # (2)
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        parts += per_read(*run_batch(L, ctx, mo, reads[a:b]))
    assert parts == whole                                                          # (1)
    n_hits = n_right = 0
    for i, rs in enumerate(whole):
        for (rid, ts, te, qs, qe, strand, mapq, nm, cg, cs, pri) in rs:
            ops = np.frombuffer(cg, dtype=np.uint32)
            ln, op = ops >> 4, ops & 0xf
            assert int(ln[(op == 0) | (op == 1)].sum()) == qe - qs and int(ln[(op == 0) | (op == 2)].sum()) == te - ts   # (3)
            cs = cs.decode()
            n_match = sum(int(v) for v in re.findall(r":(\d+)", cs))
            n_sub = cs.count("*")
            n_ins = sum(len(v) for v in re.findall(r"\+([acgtn]+)", cs)); n_del = sum(len(v) for v in re.findall(r"-([acgtn]+)", cs))
            assert n_match + n_sub == int(ln[op == 0].sum()) and n_ins == int(ln[op == 1].sum()) and n_del == int(ln[op == 2].sum())   # (4)
            assert nm == n_sub + n_ins + n_del
            n_hits += 1
        if rs:
            ci, st, en, sd = truth[i]
            rid, ts, te, qs, qe, strand, *_ = rs[0]
            n_right += rid == ci and (min(te, en) - max(ts, st)) > 0.5 * (en - st) and (strand > 0) == (sd > 0)
    return n_hits, n_right
