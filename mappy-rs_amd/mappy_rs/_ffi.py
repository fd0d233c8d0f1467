"""ctypes binding of libmm355.so (the C-ABI declared in include/mm355.h).

This is the stub a maintainer of the reference would write in Rust (`extern "C"` block, see
INTEGRATION.md); here the host language above the C-ABI is Python because the image has no
Rust toolchain.  The library is built in-tree (mappy-rs_amd/csrc/libmm355.so) and must be
present: there is no fallback of any kind.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MM355_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "csrc", "libmm355.so")   # (override: A/B runs of two builds)

MM355_ENODEV, MM355_EINVAL, MM355_ENOMEM, MM355_EIO, MM355_ENOIDX, MM355_EEMPTY, MM355_EUNSUP, MM355_EHIP = \
    -1, -2, -3, -4, -5, -6, -7, -8
OUT_CS, OUT_MD = 1, 2


class IdxOpt(C.Structure):
    _fields_ = [("k", C.c_int16), ("w", C.c_int16), ("flag", C.c_int16), ("bucket_bits", C.c_int16),
                ("mini_batch_size", C.c_int64), ("batch_size", C.c_uint64)]


class MapOpt(C.Structure):
    _fields_ = [("flag", C.c_int64), ("seed", C.c_int32), ("sdust_thres", C.c_int32), ("max_qlen", C.c_int32),
                ("bw", C.c_int32), ("bw_long", C.c_int32), ("max_gap", C.c_int32), ("max_gap_ref", C.c_int32),
                ("max_frag_len", C.c_int32), ("max_chain_skip", C.c_int32), ("max_chain_iter", C.c_int32),
                ("min_cnt", C.c_int32), ("min_chain_score", C.c_int32), ("chain_gap_scale", C.c_float),
                ("chain_skip_scale", C.c_float), ("rmq_size_cap", C.c_int32), ("rmq_inner_dist", C.c_int32),
                ("rmq_rescue_size", C.c_int32), ("rmq_rescue_ratio", C.c_float), ("mask_level", C.c_float),
                ("mask_len", C.c_int32), ("pri_ratio", C.c_float), ("best_n", C.c_int32), ("alt_drop", C.c_float),
                ("a", C.c_int32), ("b", C.c_int32), ("q", C.c_int32), ("e", C.c_int32), ("q2", C.c_int32),
                ("e2", C.c_int32), ("sc_ambi", C.c_int32), ("zdrop", C.c_int32), ("zdrop_inv", C.c_int32),
                ("end_bonus", C.c_int32), ("min_dp_max", C.c_int32), ("min_ksw_len", C.c_int32),
                ("max_clip_ratio", C.c_float), ("mid_occ_frac", C.c_float), ("q_occ_frac", C.c_float),
                ("min_mid_occ", C.c_int32), ("max_mid_occ", C.c_int32), ("mid_occ", C.c_int32),
                ("max_occ", C.c_int32), ("max_max_occ", C.c_int32), ("occ_dist", C.c_int32),
                ("max_sw_mat", C.c_int64)]


class Hit(C.Structure):
    _fields_ = [("query_start", C.c_int32), ("query_end", C.c_int32), ("strand", C.c_int32), ("rid", C.c_int32),
                ("target_len", C.c_int32), ("target_start", C.c_int32), ("target_end", C.c_int32),
                ("match_len", C.c_int32), ("block_len", C.c_int32), ("mapq", C.c_uint32),
                ("is_primary", C.c_int32), ("NM", C.c_int32), ("n_cigar", C.c_int32),
                ("cigar_off", C.c_int64), ("cs_off", C.c_int64), ("cs_len", C.c_int64),
                ("md_off", C.c_int64), ("md_len", C.c_int64),
                ("score0", C.c_int32), ("dp_max", C.c_int32), ("dp_max2", C.c_int32), ("dp_score", C.c_int32),
                ("cnt", C.c_int32), ("n_sub", C.c_int32), ("subsc", C.c_int32), ("reserved", C.c_int32)]


class Hits(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("hit_off", C.POINTER(C.c_int64)), ("status", C.POINTER(C.c_int32)),
                ("hits", C.POINTER(Hit)), ("cigar", C.POINTER(C.c_uint32)), ("str", C.POINTER(C.c_char)),
                ("n_hits", C.c_int64), ("n_cigar", C.c_int64), ("n_str", C.c_int64)]


class Stats(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_bases", C.c_int64), ("n_mz", C.c_int64), ("n_hit", C.c_int64),
                ("n_a", C.c_int64), ("n_a_multi", C.c_int64), ("chain_pairs", C.c_int64), ("dp_cells", C.c_int64),
                ("n_dp_jobs", C.c_int64),
                ("ms_sketch", C.c_double), ("ms_seed", C.c_double), ("ms_sort", C.c_double), ("ms_chain", C.c_double),
                ("ms_backtrack", C.c_double), ("ms_dp", C.c_double), ("ms_host", C.c_double), ("ms_total", C.c_double),
                ("ms_seed_lookup", C.c_double), ("ms_seed_expand", C.c_double), ("n_launch_seed", C.c_int64), ("n_launch_dp", C.c_int64),
                ("ms_dp_group", C.c_double * 24), ("dp_cells_group", C.c_int64 * 24), ("n_launch_group", C.c_int64 * 24),
                ("n_ext_rounds", C.c_int64), ("n_sort_fast_reads", C.c_int64), ("n_sort_tie_reads", C.c_int64),
                ("ms_rmq", C.c_double), ("n_rmq_reads", C.c_int64), ("n_rmq_host", C.c_int64), ("rmq_scanned", C.c_int64),
                ("host_cpu_ms", C.c_double), ("n_a_kept", C.c_int64), ("ms_kernel", C.c_double * 24), ("chain_pairs_big", C.c_int64), ("n_a_literal", C.c_int64), ("n_v_rmq", C.c_int64), ("n_dp_band", C.c_int64), ("n_dp_band_redo", C.c_int64), ("n_rounds_split", C.c_int64)]


class DpJob(C.Structure):
    _fields_ = [("qlen", C.c_int32), ("tlen", C.c_int32), ("qoff", C.c_int64), ("toff", C.c_int64),
                ("w", C.c_int32), ("zdrop", C.c_int32), ("end_bonus", C.c_int32), ("flag", C.c_int32)]


class ExtraJob(C.Structure):
    _fields_ = [("q_off", C.c_int64), ("cigar_off", C.c_int64), ("rid", C.c_int32), ("t_st", C.c_int32), ("n_cigar", C.c_int32), ("pad", C.c_int32)]


class ExtraRes(C.Structure):
    _fields_ = [("mlen", C.c_int32), ("blen", C.c_int32), ("n_ambi", C.c_int32), ("dp_max", C.c_int32), ("cs_off", C.c_int64), ("cs_len", C.c_int32), ("pad", C.c_int32),
                ("md_off", C.c_int64), ("md_len", C.c_int32), ("pad2", C.c_int32)]


class DpRes(C.Structure):
    _fields_ = [("max", C.c_int32), ("zdropped", C.c_int32), ("max_q", C.c_int32), ("max_t", C.c_int32),
                ("mqe", C.c_int32), ("mqe_t", C.c_int32), ("mte", C.c_int32), ("mte_q", C.c_int32),
                ("score", C.c_int32), ("reach_end", C.c_int32), ("n_cigar", C.c_int32), ("cigar_off", C.c_int64)]


EXPORTS = [
    "mm355_set_opt", "mm355_mapopt_update", "mm355_index_load", "mm355_index_build", "mm355_index_build_device", "mm355_index_free",
    "mm355_index_info", "mm355_index_seq_name", "mm355_index_seq_len", "mm355_index_name2id", "mm355_index_getseq",
    "mm355_index_get", "mm355_index_stat", "mm355_upload", "mm355_ctx_create", "mm355_ctx_destroy", "mm355_map_batch",
    "mm355_free_hits", "mm355_batch_upload", "mm355_batch_select", "mm355_map_resident", "mm355_stage_sketch", "mm355_stage_anchors", "mm355_stage_chain", "mm355_stage_chains", "mm355_stage_rmq",
    "mm355_stage_dp", "mm355_stage_extra", "mm355_get_stats", "mm355_device_count", "mm355_device_synchronize", "mm355_strerror", "mm355_version",
]

_LIB = None


def lib():
    """Load libmm355.so; raises if the HIP extension has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError("libmm355.so is missing (%s): build it with `python __graft_entry__.py` / make -C mappy-rs_amd/csrc; "
                          "the MI355X mapping path has no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i64p, i32p = C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    L.mm355_set_opt.argtypes = [C.c_char_p, C.POINTER(IdxOpt), C.POINTER(MapOpt)]
    L.mm355_device_synchronize.argtypes = [C.c_int]
    L.mm355_mapopt_update.argtypes = [C.POINTER(MapOpt), vp]
    L.mm355_index_load.argtypes = [C.c_char_p, C.POINTER(IdxOpt), C.c_int, C.POINTER(vp)]
    L.mm355_index_build.argtypes = [C.POINTER(IdxOpt), C.c_int, C.POINTER(C.c_char_p), i64p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(vp)]
    L.mm355_index_build_device.argtypes = [C.POINTER(IdxOpt), C.c_int, C.POINTER(C.c_char_p), i64p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(vp)]
    L.mm355_index_free.argtypes = [vp]
    L.mm355_index_info.argtypes = [vp, i32p, i32p, i32p, i32p, C.POINTER(C.c_uint32)]
    L.mm355_index_seq_name.restype = C.c_char_p
    L.mm355_index_seq_name.argtypes = [vp, C.c_uint32]
    L.mm355_index_seq_len.restype = C.c_int64
    L.mm355_index_seq_len.argtypes = [vp, C.c_uint32]
    L.mm355_index_name2id.argtypes = [vp, C.c_char_p]
    L.mm355_index_getseq.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.mm355_index_get.argtypes = [vp, C.c_uint64, vp, C.c_int]
    L.mm355_index_stat.argtypes = [vp, i64p, i64p, i64p, i64p]
    L.mm355_upload.argtypes = [vp, i32p, C.c_int]
    L.mm355_ctx_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.mm355_ctx_destroy.argtypes = [vp]
    L.mm355_map_batch.argtypes = [vp, C.POINTER(MapOpt), C.c_int64, C.POINTER(C.c_char_p), i32p, C.c_int, C.POINTER(C.POINTER(Hits))]
    L.mm355_free_hits.argtypes = [C.POINTER(Hits)]
    L.mm355_batch_upload.argtypes = [vp, C.c_int64, C.POINTER(C.c_char_p), i32p]
    L.mm355_batch_select.argtypes = [vp, C.c_int]
    L.mm355_map_resident.argtypes = [vp, C.POINTER(MapOpt), C.c_int, C.POINTER(C.POINTER(Hits))]
    L.mm355_stage_sketch.argtypes = [vp, C.c_int64, C.POINTER(C.c_char_p), i32p, vp, vp, C.c_int64]
    L.mm355_stage_anchors.argtypes = [vp, C.POINTER(MapOpt), C.c_int64, C.POINTER(C.c_char_p), i32p, C.c_int, vp, vp, C.c_int64, vp, vp]
    L.mm355_stage_chain.argtypes = [vp, C.POINTER(MapOpt), C.c_int64, C.POINTER(C.c_char_p), i32p, vp, vp, vp, vp, vp, C.c_int64]
    L.mm355_stage_chains.argtypes = [vp, C.POINTER(MapOpt), C.c_int64, C.POINTER(C.c_char_p), i32p, vp, vp, C.c_int64, vp, vp, C.c_int64]
    L.mm355_stage_dp.argtypes = [vp, C.POINTER(MapOpt), C.c_int64, vp, vp, C.c_int64, vp, C.c_int64, vp, vp, C.c_int64]
    L.mm355_stage_rmq.argtypes = [vp, C.POINTER(MapOpt), C.c_int64, C.POINTER(C.c_char_p), i32p, vp, vp, C.c_int64, vp, vp, C.c_int64, vp]
    L.mm355_stage_extra.argtypes = [vp, C.POINTER(MapOpt), C.c_int64, vp, vp, C.c_int64, vp, C.c_int64, C.c_int, vp, vp, C.c_int64]
    L.mm355_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.mm355_strerror.restype = C.c_char_p
    L.mm355_strerror.argtypes = [C.c_int]
    L.mm355_version.restype = C.c_char_p
    _LIB = L
    return L


class Mm355Error(RuntimeError):
    def __init__(self, code):
        self.code = code
        RuntimeError.__init__(self, lib().mm355_strerror(code).decode())


def check(rc):
    if rc != 0:
        raise Mm355Error(rc)


_utf8 = C.pythonapi.PyUnicode_AsUTF8          # ASCII str: a pointer into the object itself (CPython's compact representation), no copy
_utf8.restype = C.c_void_p
_utf8.argtypes = [C.py_object]


def pack_reads(seqs):
    """list of str/bytes -> (char** array, int32 lens array, keepalive).  Plain ASCII str -- every read a sequencer or a FASTQ parser hands
    over -- is passed by the address of its own buffer: `s.encode()` costs 5-6 us per 8 kb read under the GIL (an allocation and a copy),
    more than everything else the interpreter does for a read."""
    n = len(seqs)
    if n and set(map(type, seqs)) == {str} and all(map(str.isascii, seqs)):
        import numpy as np
        ptrs = np.fromiter(map(_utf8, seqs), dtype=np.uint64, count=n)
        lens = np.fromiter(map(len, seqs), dtype=np.int32, count=n)
        # (ctypes arrays over the numpy buffers: same types as the copying path below; they hold the buffers, `seqs` holds the bytes)
        return (C.c_char_p * n).from_buffer(ptrs), (C.c_int32 * n).from_buffer(lens), seqs
    bs = [s if isinstance(s, (bytes, bytearray)) else s.encode() for s in seqs]
    n = len(bs)
    arr = (C.c_char_p * n)(*bs)
    lens = (C.c_int32 * n)(*[len(b) for b in bs])
    return arr, lens, bs


class StageRunner:
    """Per-stage view of the device path for the parity tests and the kernel bench."""

    def __init__(self, idx_handle, mapopt, device=0):
        self.L = lib()
        self.mo = mapopt
        self.ctx = C.c_void_p()
        check(self.L.mm355_ctx_create(idx_handle, device, C.byref(self.ctx)))

    def close(self):
        if self.ctx:
            self.L.mm355_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self):
        st = Stats()
        check(self.L.mm355_get_stats(self.ctx, C.byref(st)))
        return st

    def sketch(self, seqs):
        arr, lens, keep = pack_reads(seqs)
        n = len(seqs)
        cap = sum(len(b) for b in keep) + 16
        off = np.zeros(n + 1, np.int64)
        mz = np.zeros((cap, 2), np.uint64)
        check(self.L.mm355_stage_sketch(self.ctx, n, arr, lens, off.ctypes.data, mz.ctypes.data, cap))
        return [mz[off[i]:off[i + 1]].copy() for i in range(n)]

    def anchors(self, seqs, sorted_=True, cap=None):
        arr, lens, keep = pack_reads(seqs)
        n = len(seqs)
        cap = cap or (64 * sum(len(b) for b in keep) + 1024)
        off = np.zeros(n + 1, np.int64)
        a = np.zeros((cap, 2), np.uint64)
        rep = np.zeros(n, np.int32); nmp = np.zeros(n, np.int32)
        check(self.L.mm355_stage_anchors(self.ctx, C.byref(self.mo), n, arr, lens, int(sorted_), off.ctypes.data,
                                         a.ctypes.data, cap, rep.ctypes.data, nmp.ctypes.data))
        return [a[off[i]:off[i + 1]].copy() for i in range(n)], rep, nmp

    def chain(self, seqs, cap=None):
        arr, lens, keep = pack_reads(seqs)
        n = len(seqs)
        cap = cap or (64 * sum(len(b) for b in keep) + 1024)
        off = np.zeros(n + 1, np.int64)
        a = np.zeros((cap, 2), np.uint64)
        f = np.zeros(cap, np.int32); p = np.zeros(cap, np.int32); v = np.zeros(cap, np.int32)
        check(self.L.mm355_stage_chain(self.ctx, C.byref(self.mo), n, arr, lens, off.ctypes.data, a.ctypes.data,
                                       f.ctypes.data, p.ctypes.data, v.ctypes.data, cap))
        return [(a[off[i]:off[i + 1]].copy(), f[off[i]:off[i + 1]].copy(), p[off[i]:off[i + 1]].copy(),
                 v[off[i]:off[i + 1]].copy()) for i in range(n)]

    def chains(self, seqs, cap=None):
        arr, lens, keep = pack_reads(seqs)
        n = len(seqs)
        cap = cap or (64 * sum(len(b) for b in keep) + 1024)
        uoff = np.zeros(n + 1, np.int64); aoff = np.zeros(n + 1, np.int64)
        u = np.zeros(cap, np.uint64); a = np.zeros((cap, 2), np.uint64)
        check(self.L.mm355_stage_chains(self.ctx, C.byref(self.mo), n, arr, lens, uoff.ctypes.data, u.ctypes.data, cap,
                                        aoff.ctypes.data, a.ctypes.data, cap))
        return [(u[uoff[i]:uoff[i + 1]].copy(), a[aoff[i]:aoff[i + 1]].copy()) for i in range(n)]

    def rmq(self, seqs, cap=None):
        """chains after mg_lchain_rmq (long-join re-chain, or the primary chainer of MM_F_RMQ presets): [(u, anchors, state)] per read"""
        arr, lens, keep = pack_reads(seqs)
        n = len(seqs)
        cap = cap or (64 * sum(len(b) for b in keep) + 1024)
        uoff = np.zeros(n + 1, np.int64); aoff = np.zeros(n + 1, np.int64)
        u = np.zeros(cap, np.uint64); a = np.zeros((cap, 2), np.uint64)
        state = np.zeros(n, np.int32)
        check(self.L.mm355_stage_rmq(self.ctx, C.byref(self.mo), n, arr, lens, uoff.ctypes.data, u.ctypes.data, cap,
                                     aoff.ctypes.data, a.ctypes.data, cap, state.ctypes.data))
        return [(u[uoff[i]:uoff[i + 1]].copy(), a[aoff[i]:aoff[i + 1]].copy(), int(state[i])) for i in range(n)]
