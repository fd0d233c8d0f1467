"""ORACLE bindings (test infrastructure only).

ctypes view of oracle/libmm2oracle.so -- the plain-C CPU restatement of the
minimap2 2.26 per-read path that mappy-rs reaches at /root/reference/src/lib.rs:482
and :587.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product (mappy-rs_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class IdxOpt(C.Structure):
    _fields_ = [("k", C.c_short), ("w", C.c_short), ("flag", C.c_short), ("bucket_bits", C.c_short),
                ("mini_batch_size", C.c_int64), ("batch_size", C.c_uint64)]


class MapOpt(C.Structure):
    _fields_ = [
        ("flag", C.c_int64), ("seed", C.c_int), ("sdust_thres", C.c_int), ("max_qlen", C.c_int),
        ("bw", C.c_int), ("bw_long", C.c_int), ("max_gap", C.c_int), ("max_gap_ref", C.c_int),
        ("max_frag_len", C.c_int), ("max_chain_skip", C.c_int), ("max_chain_iter", C.c_int),
        ("min_cnt", C.c_int), ("min_chain_score", C.c_int), ("chain_gap_scale", C.c_float),
        ("chain_skip_scale", C.c_float), ("rmq_size_cap", C.c_int), ("rmq_inner_dist", C.c_int),
        ("rmq_rescue_size", C.c_int), ("rmq_rescue_ratio", C.c_float), ("mask_level", C.c_float),
        ("mask_len", C.c_int), ("pri_ratio", C.c_float), ("best_n", C.c_int), ("alt_drop", C.c_float),
        ("a", C.c_int), ("b", C.c_int), ("q", C.c_int), ("e", C.c_int), ("q2", C.c_int), ("e2", C.c_int),
        ("sc_ambi", C.c_int), ("noncan", C.c_int), ("junc_bonus", C.c_int), ("zdrop", C.c_int),
        ("zdrop_inv", C.c_int), ("end_bonus", C.c_int), ("min_dp_max", C.c_int), ("min_ksw_len", C.c_int),
        ("anchor_ext_len", C.c_int), ("anchor_ext_shift", C.c_int), ("max_clip_ratio", C.c_float),
        ("rank_min_len", C.c_int), ("rank_frac", C.c_float), ("pe_ori", C.c_int), ("pe_bonus", C.c_int),
        ("mid_occ_frac", C.c_float), ("q_occ_frac", C.c_float), ("min_mid_occ", C.c_int32),
        ("max_mid_occ", C.c_int32), ("mid_occ", C.c_int32), ("max_occ", C.c_int32),
        ("max_max_occ", C.c_int32), ("occ_dist", C.c_int32), ("mini_batch_size", C.c_int64),
        ("max_sw_mat", C.c_int64), ("cap_kalloc", C.c_int64)]


class IdxSeq(C.Structure):
    _fields_ = [("name", C.c_char_p), ("offset", C.c_uint64), ("len", C.c_uint32), ("is_alt", C.c_uint32)]


class Bucket(C.Structure):
    _fields_ = [("n", C.c_int32), ("p", C.POINTER(C.c_uint64)), ("n_keys", C.c_uint32), ("cap", C.c_uint32),
                ("keys", C.POINTER(C.c_uint64)), ("vals", C.POINTER(C.c_uint64)),
                ("a_n", C.c_size_t), ("a_m", C.c_size_t), ("a_a", C.c_void_p)]


class Idx(C.Structure):
    _fields_ = [("b", C.c_int32), ("w", C.c_int32), ("k", C.c_int32), ("flag", C.c_int32),
                ("n_seq", C.c_uint32), ("n_alt", C.c_int32), ("seq", C.POINTER(IdxSeq)),
                ("S", C.POINTER(C.c_uint32)), ("B", C.POINTER(Bucket))]


class Hit(C.Structure):
    _fields_ = [("query_start", C.c_int32), ("query_end", C.c_int32), ("strand", C.c_int32), ("rid", C.c_int32),
                ("target_len", C.c_int32), ("target_start", C.c_int32), ("target_end", C.c_int32),
                ("match_len", C.c_int32), ("block_len", C.c_int32), ("mapq", C.c_uint32),
                ("is_primary", C.c_int32), ("NM", C.c_int32), ("n_cigar", C.c_int32),
                ("cigar_off", C.c_int64), ("cs_off", C.c_int64), ("cs_len", C.c_int64),
                ("md_off", C.c_int64), ("md_len", C.c_int64),
                ("score0", C.c_int32), ("dp_max", C.c_int32), ("dp_max2", C.c_int32), ("dp_score", C.c_int32),
                ("cnt", C.c_int32), ("n_sub", C.c_int32), ("subsc", C.c_int32), ("sam_pri", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("n_hits", C.c_int), ("hits", C.POINTER(Hit)),
                ("n_cigar", C.c_size_t), ("m_cigar", C.c_size_t), ("cigar", C.POINTER(C.c_uint32)),
                ("n_str", C.c_size_t), ("m_str", C.c_size_t), ("str", C.POINTER(C.c_char))]


class Extz(C.Structure):
    _fields_ = [("max_zd", C.c_uint32), ("max_q", C.c_int), ("max_t", C.c_int), ("mqe", C.c_int), ("mqe_t", C.c_int),
                ("mte", C.c_int), ("mte_q", C.c_int), ("score", C.c_int), ("m_cigar", C.c_int), ("n_cigar", C.c_int),
                ("reach_end", C.c_int), ("cigar", C.POINTER(C.c_uint32))]


class Stats(C.Structure):
    _fields_ = [("n_mz", C.c_int64), ("n_hit", C.c_int64), ("n_a", C.c_int64), ("n_a_multi", C.c_int64),
                ("chain_pairs", C.c_int64), ("dp_cells", C.c_int64), ("n_dp_calls", C.c_int64),
                ("rep_len", C.c_int32), ("n_chain0", C.c_int32), ("n_chain1", C.c_int32), ("did_rmq", C.c_int32)]


class MM128V(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


def build(force=False):
    so = os.path.join(_HERE, "libmm2oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-j4"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.mmo_idx_load.restype = C.POINTER(Idx)
        L.mmo_idx_load.argtypes = [C.c_char_p, C.POINTER(IdxOpt)]
        L.mmo_idx_build_mem.restype = C.POINTER(Idx)
        L.mmo_idx_build_mem.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p),
                                        C.POINTER(C.c_int), C.POINTER(C.c_char_p)]
        L.mmo_idx_build_mem_mt.restype = C.POINTER(Idx)
        L.mmo_idx_build_mem_mt.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.c_int]
        L.mmo_idx_destroy.argtypes = [C.POINTER(Idx)]
        L.mmo_idx_get.restype = C.POINTER(C.c_uint64)
        L.mmo_idx_get.argtypes = [C.POINTER(Idx), C.c_uint64, C.POINTER(C.c_int)]
        L.mmo_idx_getseq.argtypes = [C.POINTER(Idx), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.mmo_idx_name2id.argtypes = [C.POINTER(Idx), C.c_char_p]
        L.mmo_idx_cal_max_occ.argtypes = [C.POINTER(Idx), C.c_float]
        L.mmo_idx_dump.argtypes = [C.POINTER(Idx), C.c_char_p]
        L.mmo_idx_n_minimizers.restype = C.c_int64
        L.mmo_idx_n_minimizers.argtypes = [C.POINTER(Idx), C.POINTER(C.c_int64)]
        L.mmo_set_opt.argtypes = [C.c_char_p, C.POINTER(IdxOpt), C.POINTER(MapOpt)]
        L.mmo_mapopt_update.argtypes = [C.POINTER(MapOpt), C.POINTER(Idx)]
        L.mmo_sketch.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.POINTER(MM128V)]
        L.mmo_seed_mz_flt.argtypes = [C.POINTER(MM128V), C.c_int32, C.c_float]
        L.mmo_collect_seed_hits.restype = C.c_void_p
        L.mmo_collect_seed_hits.argtypes = [C.POINTER(MapOpt), C.c_int, C.POINTER(Idx), C.POINTER(MM128V), C.c_int,
                                            C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                            C.POINTER(C.c_void_p), C.c_int]
        L.mmo_lchain_dp_fill.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                         C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mmo_radix_sort_128x.argtypes = [C.c_void_p, C.c_void_p]
        L.mmo_map_flat.argtypes = [C.POINTER(Idx), C.POINTER(MapOpt), C.c_char_p, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(Result)]
        L.mmo_result_free.argtypes = [C.POINTER(Result)]
        L.mmo_ksw_extd2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int8, C.c_void_p, C.c_int8, C.c_int8,
                                    C.c_int8, C.c_int8, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Extz)]
        L.mmo_ksw_gen_simple_mat.argtypes = [C.c_int, C.c_void_p, C.c_int8, C.c_int8, C.c_int8]
        L.free.argtypes = [C.c_void_p]
        L.malloc.restype = C.c_void_p
        L.malloc.argtypes = [C.c_size_t]
        L.mmo_lchain_dp.restype = C.c_void_p
        L.mmo_lchain_dp.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                    C.c_int, C.c_int, C.c_int64, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        L.mmo_cs_core.restype = C.c_void_p
        L.mmo_cs_core.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.mmo_md_core.restype = C.c_void_p
        L.mmo_md_core.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mmo_extra_walk.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int8, C.c_int8, C.c_int] + [C.c_void_p] * 6
        L.mmo_lchain_rmq.restype = C.c_void_p
        L.mmo_lchain_rmq.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                     C.c_int64, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        _LIB = L
    return _LIB


CIGAR_OPS = "MIDNSHP=X"


def update_extra_stage(qcodes, tcodes, ops, a, b, sc_ambi, q, e):
    """the oracle's own mm_update_extra walk (after mm_fix_cigar) and write_cs_core on given code strings and CIGAR [(op, len)]:
    (mlen, blen, n_ambi, dp_max, cs, MD) -- the stage oracle of the device walk k_extra"""
    L = lib()
    cig = np.array([ln << 4 | op for op, ln in ops] + [0], np.uint32)
    qc = np.ascontiguousarray(np.concatenate([qcodes, np.zeros(8, np.uint8)]), np.uint8)
    tc = np.ascontiguousarray(tcodes, np.uint8)
    mat = np.zeros(25, np.int8)
    L.mmo_ksw_gen_simple_mat(5, mat.ctypes.data, a, b, sc_ambi)
    out = np.zeros(6, np.int32)
    L.mmo_extra_walk(cig.ctypes.data, len(ops), qc.ctypes.data, tc.ctypes.data, mat.ctypes.data, q, e, 1, *[out[i:].ctypes.data for i in range(6)])
    p = L.mmo_cs_core(cig.ctypes.data, len(ops), qc.ctypes.data, tc.ctypes.data, 1, None, None)
    cs = C.string_at(p).decode()
    L.free(p)
    p = L.mmo_md_core(cig.ctypes.data, len(ops), qc.ctypes.data, tc.ctypes.data, None, None)
    md = C.string_at(p).decode()
    L.free(p)
    return int(out[0]), int(out[1]), int(out[2]), int(out[3]), cs, md


class OracleAligner:
    """CPU oracle with the mappy-rs option semantics of /root/reference/src/lib.rs:331-385."""

    def __init__(self, fn_idx_in=None, preset=None, k=None, w=None, min_cnt=None, min_chain_score=None,
                 min_dp_score=None, bw=None, best_n=None, max_frag_len=None, extra_flags=None, scoring=None,
                 seqs=None, names=None, codes=None, n_threads=1):
        L = lib()
        self.io, self.mo = IdxOpt(), MapOpt()
        L.mmo_set_opt(None, C.byref(self.io), C.byref(self.mo))
        if preset is not None:
            L.mmo_set_opt(preset.encode(), C.byref(self.io), C.byref(self.mo))
        self.mo.flag |= 4
        if k is not None: self.io.k = k
        if w is not None: self.io.w = w
        if min_cnt is not None: self.mo.min_cnt = min_cnt
        if min_chain_score is not None: self.mo.min_chain_score = min_chain_score
        if min_dp_score is not None: self.mo.min_dp_max = min_dp_score
        if bw is not None: self.mo.bw = bw
        if best_n is not None: self.mo.best_n = best_n
        if max_frag_len is not None: self.mo.max_frag_len = max_frag_len
        if extra_flags is not None: self.mo.flag |= extra_flags
        if scoring is not None and len(scoring) >= 4:
            self.mo.a, self.mo.b, self.mo.q, self.mo.e = scoring[:4]
            self.mo.q2, self.mo.e2 = self.mo.q, self.mo.e
            if len(scoring) >= 6:
                self.mo.q2, self.mo.e2 = scoring[4:6]
                if len(scoring) >= 7:
                    self.mo.sc_ambi = scoring[6]
        if codes is not None:      # contigs as numpy uint8 code arrays (0..4), no copy; threaded build (mmo_idx_build_mem_mt)
            n = len(codes)
            keep = [np.ascontiguousarray(c, dtype=np.uint8) for c in codes]
            arr = (C.c_void_p * n)(*[c.ctypes.data for c in keep])
            lens = (C.c_int * n)(*[len(c) for c in keep])
            nm = (C.c_char_p * n)(*[(names[i] if names else "ref%d" % i).encode() for i in range(n)])
            self.idx = L.mmo_idx_build_mem_mt(self.io.w, self.io.k, self.io.bucket_bits, self.io.flag, n, arr, lens, nm, int(n_threads))
        elif seqs is not None:
            n = len(seqs)
            bs = [s if isinstance(s, bytes) else s.encode() for s in seqs]
            arr = (C.c_char_p * n)(*bs)
            lens = (C.c_int * n)(*[len(s) for s in bs])
            nm = (C.c_char_p * n)(*[(names[i] if names else "ref%d" % i).encode() for i in range(n)])
            self.idx = L.mmo_idx_build_mem(self.io.w, self.io.k, self.io.bucket_bits, self.io.flag, n, arr, lens, nm)
        else:
            self.idx = L.mmo_idx_load(str(fn_idx_in).encode(), C.byref(self.io))
        if not self.idx:
            raise RuntimeError("Did not create or open an index")
        L.mmo_mapopt_update(C.byref(self.mo), self.idx)

    @property
    def k(self): return self.idx.contents.k
    @property
    def w(self): return self.idx.contents.w
    @property
    def n_seq(self): return self.idx.contents.n_seq
    @property
    def seq_names(self): return [self.idx.contents.seq[i].name.decode() for i in range(self.n_seq)]
    @property
    def seq_lens(self): return [self.idx.contents.seq[i].len for i in range(self.n_seq)]

    def seq(self, name, start=0, end=0x7fffffff):
        L = lib()
        rid = L.mmo_idx_name2id(self.idx, name.encode())
        if rid < 0: return None
        ln = self.idx.contents.seq[rid].len
        if start >= ln or start >= end: return None
        if end < 0 or end > ln: end = ln
        buf = np.zeros(end - start, dtype=np.uint8)
        n = L.mmo_idx_getseq(self.idx, rid, start, end, buf.ctypes.data)
        if n < 0: return None
        return "".join("ACGTN"[c] for c in buf[:n])

    def sketch(self, seq, rid=0):
        L = lib()
        b = seq if isinstance(seq, bytes) else seq.encode()
        v = MM128V()
        L.mmo_sketch(b, len(b), self.w, self.k, rid, self.idx.contents.flag & 1, C.byref(v))
        out = np.ctypeslib.as_array(C.cast(v.a, C.POINTER(C.c_uint64)), shape=(v.n, 2)).copy() if v.n else np.zeros((0, 2), np.uint64)
        L.free(v.a)
        return out

    def anchors(self, seq, sorted_=True):
        """(sorted anchors [n_a,2] u64, rep_len, mini_pos[u64], minimizers after mz_flt)"""
        L = lib()
        b = seq if isinstance(seq, bytes) else seq.encode()
        v = MM128V()
        L.mmo_sketch(b, len(b), self.w, self.k, 0, self.idx.contents.flag & 1, C.byref(v))
        if self.mo.q_occ_frac > 0:
            L.mmo_seed_mz_flt(C.byref(v), self.mo.mid_occ, self.mo.q_occ_frac)
        mz = np.ctypeslib.as_array(C.cast(v.a, C.POINTER(C.c_uint64)), shape=(v.n, 2)).copy() if v.n else np.zeros((0, 2), np.uint64)
        n_a, rep_len, n_mp, mp = C.c_int64(), C.c_int(), C.c_int(), C.c_void_p()
        a = L.mmo_collect_seed_hits(C.byref(self.mo), self.mo.mid_occ, self.idx, C.byref(v), len(b), C.byref(n_a),
                                    C.byref(rep_len), C.byref(n_mp), C.byref(mp), 1 if sorted_ else 0)
        arr = np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_uint64)), shape=(n_a.value, 2)).copy() if n_a.value else np.zeros((0, 2), np.uint64)
        mpa = np.ctypeslib.as_array(C.cast(mp, C.POINTER(C.c_uint64)), shape=(n_mp.value,)).copy() if n_mp.value else np.zeros((0,), np.uint64)
        L.free(a); L.free(mp); L.free(v.a)
        return arr, rep_len.value, mpa, mz

    def chain_fill(self, anchors, qlen):
        """f,p,v,t of mg_lchain_dp's fill loop on sorted anchors"""
        L = lib()
        n = anchors.shape[0]
        a = np.ascontiguousarray(anchors, dtype=np.uint64)
        f = np.zeros(n, np.int32); p = np.zeros(n, np.int64); v = np.zeros(n, np.int32); t = np.zeros(n, np.int32)
        mo = self.mo
        gap_ref = mo.max_gap_ref if mo.max_gap_ref > 0 else (max(mo.max_frag_len - qlen, mo.max_gap) if mo.max_frag_len > 0 else mo.max_gap)
        pen_gap = np.float32(np.float64(mo.chain_gap_scale) * 0.01 * self.k)
        pen_skip = np.float32(np.float64(mo.chain_skip_scale) * 0.01 * self.k)
        if n:
            L.mmo_lchain_dp_fill(gap_ref, mo.max_gap, mo.bw, mo.max_chain_skip, mo.max_chain_iter, pen_gap, pen_skip,
                                 n, a.ctypes.data, f.ctypes.data, p.ctypes.data, v.ctypes.data, t.ctypes.data)
        return f, p, v, t

    def chains(self, anchors, qlen):
        """u[] and compacted anchors after mg_lchain_dp (fill + backtrack + compact_a)"""
        L = lib()
        n = anchors.shape[0]
        if n == 0:
            return np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64)
        buf = L.malloc(n * 16)
        C.memmove(buf, np.ascontiguousarray(anchors, dtype=np.uint64).ctypes.data, n * 16)
        mo = self.mo
        gap_ref = mo.max_gap_ref if mo.max_gap_ref > 0 else (max(mo.max_frag_len - qlen, mo.max_gap) if mo.max_frag_len > 0 else mo.max_gap)
        pen_gap = np.float32(np.float64(mo.chain_gap_scale) * 0.01 * self.k)
        pen_skip = np.float32(np.float64(mo.chain_skip_scale) * 0.01 * self.k)
        n_u, u = C.c_int(), C.c_void_p()
        b = L.mmo_lchain_dp(gap_ref, mo.max_gap, mo.bw, mo.max_chain_skip, mo.max_chain_iter, mo.min_cnt, mo.min_chain_score,
                            pen_gap, pen_skip, 0, 1, n, buf, C.byref(n_u), C.byref(u))
        if n_u.value == 0:
            return np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64)
        ua = np.ctypeslib.as_array(C.cast(u, C.POINTER(C.c_uint64)), shape=(n_u.value,)).copy()
        nv = int((ua & np.uint64(0xffffffff)).sum())
        aa = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint64)), shape=(nv, 2)).copy()
        L.free(u); L.free(b)
        return ua, aa

    def _lchain_rmq(self, sorted_anchors, bw):
        """mg_lchain_rmq on anchors sorted by x: (u, compacted anchors)"""
        L = lib()
        n = sorted_anchors.shape[0]
        if n == 0:
            return np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64)
        buf = L.malloc(n * 16)
        C.memmove(buf, np.ascontiguousarray(sorted_anchors, dtype=np.uint64).ctypes.data, n * 16)
        mo = self.mo
        pen_gap = np.float32(np.float64(mo.chain_gap_scale) * 0.01 * self.k)
        pen_skip = np.float32(np.float64(mo.chain_skip_scale) * 0.01 * self.k)
        n_u, u = C.c_int(), C.c_void_p()
        b = L.mmo_lchain_rmq(mo.max_gap, mo.rmq_inner_dist, bw, mo.max_chain_skip, mo.rmq_size_cap, mo.min_cnt, mo.min_chain_score,
                             pen_gap, pen_skip, n, buf, C.byref(n_u), C.byref(u))
        if n_u.value == 0:
            return np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64)
        ua = np.ctypeslib.as_array(C.cast(u, C.POINTER(C.c_uint64)), shape=(n_u.value,)).copy()
        nv = int((ua & np.uint64(0xffffffff)).sum())
        aa = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint64)), shape=(nv, 2)).copy()
        L.free(u); L.free(b)
        return ua, aa

    def chains_final(self, anchors, qlen):
        """what mm_map_frag (mmo_map.c) holds before mm_gen_regs: (u, anchors, did) with did bit 0 = primary RMQ chainer ran, bit 1 = long-join re-chain ran.  MM_F_RMQ presets: mg_lchain_rmq is the primary
        chainer; otherwise mg_lchain_dp and, when the rescue test fires, the long-join re-chain (radix_sort_128x + mg_lchain_rmq, bw_long)"""
        L = lib()
        mo = self.mo
        did = 0
        if mo.flag & 0x80000000:
            u, a = self._lchain_rmq(anchors, mo.bw)
            did = 1
        else:
            u, a = self.chains(anchors, qlen)
        if mo.bw_long > mo.bw and (mo.flag & (0x080 | 0x1000 | 0x400)) == 0 and len(u) > 1:
            st = int(np.int32(a[0, 1] & np.uint64(0xffffffff))); en = int(np.int32(a[int(u[0] & np.uint64(0xffffffff)) - 1, 1] & np.uint64(0xffffffff)))
            if qlen - (en - st) > mo.rmq_rescue_size or np.float32(en - st) > np.float32(qlen) * np.float32(mo.rmq_rescue_ratio):
                srt = np.ascontiguousarray(a, dtype=np.uint64).copy()
                L.mmo_radix_sort_128x(srt.ctypes.data, srt.ctypes.data + srt.shape[0] * 16)
                u, a = self._lchain_rmq(srt, mo.bw_long)
                return u, a, did + 2
        return u, a, did

    def map(self, seq, cs=False, MD=False):
        L = lib()
        b = seq if isinstance(seq, bytes) else seq.encode()
        res = Result()
        rc = L.mmo_map_flat(self.idx, C.byref(self.mo), b, len(b), int(cs), int(MD), C.byref(res))
        if rc == -1: raise RuntimeError("No index")
        if rc == -2: raise RuntimeError("Sequence is empty")
        names = None
        out = []
        for i in range(res.n_hits):
            h = res.hits[i]
            cig = [(res.cigar[h.cigar_off + j] >> 4, res.cigar[h.cigar_off + j] & 0xf) for j in range(h.n_cigar)]
            d = dict(query_start=h.query_start, query_end=h.query_end, strand=h.strand, rid=h.rid,
                     target_name=self.idx.contents.seq[h.rid].name.decode() if self.idx.contents.seq[h.rid].name else None,
                     target_len=h.target_len, target_start=h.target_start, target_end=h.target_end,
                     match_len=h.match_len, block_len=h.block_len, mapq=h.mapq, is_primary=bool(h.is_primary),
                     cigar=cig, NM=h.NM,
                     cs=C.string_at(C.addressof(res.str.contents) + h.cs_off, h.cs_len).decode() if h.cs_len >= 0 else None,
                     MD=C.string_at(C.addressof(res.str.contents) + h.md_off, h.md_len).decode() if h.md_len >= 0 else None,
                     score0=h.score0, dp_max=h.dp_max, dp_max2=h.dp_max2, dp_score=h.dp_score, cnt=h.cnt,
                     n_sub=h.n_sub, subsc=h.subsc)
            d["cigar_str"] = "".join("%d%s" % (l, CIGAR_OPS[o]) for l, o in cig)
            out.append(d)
        L.mmo_result_free(C.byref(res))
        return out

    def stats(self):
        return Stats.in_dll(lib(), "mmo_stats")

    def __del__(self):
        try:
            if self.idx: lib().mmo_idx_destroy(self.idx)
        except Exception:
            pass
