"""mappy_rs -- drop-in surface of Adoni5/mappy-rs for the MI355X-native mapping path.

Mirrors the reference's PyO3 module (/root/reference/src/lib.rs:995-999: only `Aligner` is
exported; `Mapping` and the batch iterator are reachable as return values).  Same constructor
keywords (lib.rs:312), same properties (lib.rs:439-470, 651-670), same `map` / `map_batch` /
`enable_threading` semantics and the same exception types and messages (lib.rs:388-394, 435,
477-481, 777-792, 847-866, 889-896).  Where the reference dispatches reads to N OS threads that
each call minimap2's mm_map (lib.rs:541-636), this module hands whole batches to libmm355.so
(hand-written HIP kernels for gfx950) through the C-ABI of include/mm355.h.  The extension is
mandatory: importing works without a GPU (so the API can be inspected), but creating an
Aligner that maps reads requires the HIP library and a visible MI355X -- there is no CPU path.
"""
import collections.abc
import ctypes as C
import threading

from . import _ffi

__all__ = ["Aligner", "Mapping"]

_CIGAR_OPS = "MIDNSHP=X"

# capacity constants of the reference (lib.rs:429-430, 950)
WORK_QUEUE_CAP = 50000
RESULT_CHANNEL_CAP = 20000


class Mapping:
    """Result record; fields and aliases of mappy_rs::Mapping (lib.rs:109-154, 196-284)."""

    __slots__ = ("query_start", "query_end", "_strand", "target_name", "target_len", "target_start", "target_end",
                 "match_len", "block_len", "mapq", "is_primary", "cigar", "NM", "MD", "cs")

    def __init__(self, query_start, query_end, strand, target_name, target_len, target_start, target_end, match_len,
                 block_len, mapq, is_primary, cigar, NM, MD, cs):
        self.query_start = query_start
        self.query_end = query_end
        self._strand = strand
        self.target_name = target_name
        self.target_len = target_len
        self.target_start = target_start
        self.target_end = target_end
        self.match_len = match_len
        self.block_len = block_len
        self.mapq = mapq
        self.is_primary = is_primary
        self.cigar = cigar
        self.NM = NM
        self.MD = MD
        self.cs = cs

    # mappy aliases (lib.rs:196-284)
    ctg = property(lambda s: s.target_name)
    ctg_len = property(lambda s: s.target_len)
    r_st = property(lambda s: s.target_start)
    r_en = property(lambda s: s.target_end)
    q_st = property(lambda s: s.query_start)
    q_en = property(lambda s: s.query_end)
    strand = property(lambda s: s._strand)          # +1 / -1 (lib.rs:231-237)
    blen = property(lambda s: s.block_len)
    mlen = property(lambda s: s.match_len)

    @property
    def cigar_str(self):
        out = []
        for n, op in self.cigar:
            if op > 8:
                raise ValueError("Invalid CIGAR code `{op}`")
            out.append("%d%s" % (n, _CIGAR_OPS[op]))
        return "".join(out)

    def __str__(self):  # PAF-like, lib.rs:159-180
        tp = "tp:A:P" if self.is_primary else "tp:A:S"
        return "\t".join(str(x) for x in (self.query_start, self.query_end, "+" if self._strand > 0 else "-",
                                          self.target_name, self.target_len, self.target_start, self.target_end,
                                          self.match_len, self.block_len, self.mapq, tp, "cg:Z:" + self.cigar_str))

    def __repr__(self):
        return "Mapping(%s)" % ", ".join("%s=%r" % (k.lstrip("_"), getattr(self, k)) for k in self.__slots__)

    def __eq__(self, o):
        return isinstance(o, Mapping) and all(getattr(self, k) == getattr(o, k) for k in self.__slots__)


def _hits_to_mappings(L, idx, hp, lo, hi):
    out = []
    h = hp.contents
    for i in range(lo, hi):
        x = h.hits[i]
        cig = [(h.cigar[x.cigar_off + j] >> 4, h.cigar[x.cigar_off + j] & 0xf) for j in range(x.n_cigar)]
        cs = C.string_at(C.addressof(h.str.contents) + x.cs_off, x.cs_len).decode() if x.cs_len >= 0 else None
        md = C.string_at(C.addressof(h.str.contents) + x.md_off, x.md_len).decode() if x.md_len >= 0 else None
        nm = L.mm355_index_seq_name(idx, x.rid)
        out.append(Mapping(x.query_start, x.query_end, x.strand, nm.decode() if nm is not None else None, x.target_len,
                           x.target_start, x.target_end, x.match_len, x.block_len, x.mapq, bool(x.is_primary), cig, x.NM,
                           md, cs))
    return out


class AlignmentBatchResultIter:
    """Iterator returned by map_batch (lib.rs:923-991): yields (list[Mapping], original_dict)."""

    def __init__(self, results):
        self._results = results
        self._i = 0

    def __iter__(self):
        return self

    def __next__(self):
        if self._i >= len(self._results):
            raise StopIteration("Finished")
        r = self._results[self._i]
        self._results[self._i] = None
        self._i += 1
        return r


class Aligner:
    """mappy-compatible aligner (lib.rs:288-671) whose mapping path runs on one MI355X."""

    def __init__(self, fn_idx_in=None, preset=None, k=None, w=None, min_cnt=None, min_chain_score=None,
                 min_dp_score=None, bw=None, best_n=None, n_threads=3, fn_idx_out=None, max_frag_len=None,
                 extra_flags=None, seq=None, scoring=None, device=0):
        L = _ffi.lib()
        self._L = L
        self._idx = C.c_void_p()
        self._ctx = C.c_void_p()
        self._device = device
        self._n_threads = 0
        self._lock = threading.Lock()
        io, mo = _ffi.IdxOpt(), _ffi.MapOpt()
        L.mm355_set_opt(None, C.byref(io), C.byref(mo))
        self._preset_rc = 0
        if preset is not None:
            self._preset_rc = L.mm355_set_opt(str(preset).encode(), C.byref(io), C.byref(mo))
        mo.flag |= 4                       # MM_F_CIGAR, lib.rs:339
        io.batch_size |= 0x7fffffffffffffff  # lib.rs:340
        if k is not None: io.k = k
        if w is not None: io.w = w
        if min_cnt is not None: mo.min_cnt = min_cnt
        if min_chain_score is not None: mo.min_chain_score = min_chain_score
        if min_dp_score is not None: mo.min_dp_max = min_dp_score
        if bw is not None: mo.bw = bw
        if best_n is not None: mo.best_n = best_n
        if max_frag_len is not None: mo.max_frag_len = max_frag_len
        if extra_flags is not None: mo.flag |= extra_flags
        if scoring is not None and len(scoring) >= 4:
            mo.a, mo.b, mo.q, mo.e = (int(x) for x in scoring[:4])
            mo.q2, mo.e2 = mo.q, mo.e
            if len(scoring) >= 6:
                mo.q2, mo.e2 = int(scoring[4]), int(scoring[5])
                if len(scoring) >= 7:
                    mo.sc_ambi = int(scoring[6])
        self._io, self._mo = io, mo
        if seq is not None:
            raise NotImplementedError("Not Implemented")
        if fn_idx_out is not None:
            raise NotImplementedError("Not Implemented")
        if fn_idx_in is None:
            raise RuntimeError("Did not create or open an index")
        rc = L.mm355_index_load(str(fn_idx_in).encode(), C.byref(io), int(n_threads), C.byref(self._idx))
        if rc != 0 or not self._idx:
            raise RuntimeError("Did not create or open an index")
        L.mm355_mapopt_update(C.byref(mo), self._idx)

    # ---- properties (lib.rs:439-470, 651-670)
    def _info(self):
        k, w, b, fl, n = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint32()
        self._L.mm355_index_info(self._idx, C.byref(k), C.byref(w), C.byref(b), C.byref(fl), C.byref(n))
        return k.value, w.value, b.value, fl.value, n.value

    def __bool__(self):
        return bool(self._idx)

    @property
    def k(self): return self._info()[0]

    @property
    def w(self): return self._info()[1]

    @property
    def n_seq(self): return self._info()[4]

    @property
    def seq_names(self):
        if not self._idx:
            raise RuntimeError("Index hasn't loaded")
        return [self._L.mm355_index_seq_name(self._idx, i).decode() for i in range(self.n_seq)]

    def seq(self, name, start=0, end=0x7fffffff):
        """lib.rs:464-470 with the rules of lib.rs:706-766: None on any error."""
        L = self._L
        if not self._idx:
            return None
        rid = L.mm355_index_name2id(self._idx, name.encode())
        if rid < 0:
            return None
        ln = L.mm355_index_seq_len(self._idx, rid)
        if start >= ln or start >= end:
            return None
        if end < 0 or end > ln:
            end = ln
        buf = (C.c_uint8 * (end - start))()
        n = L.mm355_index_getseq(self._idx, rid, start, end, buf)
        if n < 0:
            return None
        return bytes(buf[:n]).translate(bytes.maketrans(b"\x00\x01\x02\x03\x04", b"ACGTN")).decode()

    # ---- device context
    def _context(self):
        if not self._ctx:
            rc = self._L.mm355_ctx_create(self._idx, self._device, C.byref(self._ctx))
            if rc != 0:
                raise RuntimeError("mm355: " + self._L.mm355_strerror(rc).decode())
        return self._ctx

    def _map_many(self, seqs, flags):
        """one mm355_map_batch call; returns list of list[Mapping]"""
        L = self._L
        arr, lens, keep = _ffi.pack_reads(seqs)
        hp = C.POINTER(_ffi.Hits)()
        with self._lock:
            rc = L.mm355_map_batch(self._context(), C.byref(self._mo), len(seqs), arr, lens, flags, C.byref(hp))
        if rc != 0:
            raise RuntimeError(L.mm355_strerror(rc).decode())
        try:
            h = hp.contents
            out = []
            for i in range(len(seqs)):
                if h.status[i] == _ffi.MM355_EEMPTY:
                    out.append(RuntimeError("Sequence is empty"))
                else:
                    out.append(_hits_to_mappings(L, self._idx, hp, h.hit_off[i], h.hit_off[i + 1]))
            return out
        finally:
            L.mm355_free_hits(hp)

    # ---- single read (lib.rs:473-514)
    def map(self, seq, seq2=None, cs=False, MD=False):
        if seq2 is not None:
            raise NotImplementedError("Using `seq2` is not implemented")
        if not isinstance(seq, str):
            raise TypeError("argument 'seq': 'bytes' object cannot be converted to 'PyString'" if isinstance(seq, bytes)
                            else "argument 'seq' must be str")
        flags = (_ffi.OUT_CS if cs else 0) | (_ffi.OUT_MD if MD else 0)
        r = self._map_many([seq], flags)[0]
        if isinstance(r, Exception):
            raise r
        return r

    def map_no_op(self, _seq, seq2=None, _cs=False, _MD=False):
        """canned record of lib.rs:675-693 (binding-overhead probe)"""
        if seq2 is not None:
            raise NotImplementedError("Using `seq2` is not implemented")
        return [Mapping(0, 0, 1, "No_op", 0, 0, 0, 0, 0, 0, True, [], 0, None, None)]

    # ---- batch path (lib.rs:541-648, 771-906)
    def enable_threading(self, n_threads):
        """In the reference this spawns N mm_map worker threads; here it arms the GPU batch path.
        n_threads is kept for API parity (it bounds nothing: one context drives one GPU)."""
        self._n_threads = int(n_threads)

    def map_batch(self, seqs, back_off=True):
        if self._n_threads == 0:
            raise RuntimeError("Multi threading not enabled on this instance. Please call `.enable_threading()`")
        # accepted iterables: list / tuple / iterator / generator / sequence -- not dict, not str (lib.rs:782-792, 910-920)
        if isinstance(seqs, (dict, str, bytes)) or not (isinstance(seqs, (list, tuple, collections.abc.Sequence)) or
                                                        isinstance(seqs, collections.abc.Iterator)):
            raise TypeError("Unsupported batch type, pass a list, iter, generator or tuple")
        items, reads = [], []
        for n_pending, item in enumerate(seqs):
            if not isinstance(item, dict):
                raise TypeError("Element in iterable is not a dictionary")
            if "seq" not in item:
                raise KeyError("AHHH Key \U0001F5DD️  not found in iterated dictionary")
            s = item["seq"]
            if not isinstance(s, str):
                raise ValueError("`seq` must be a string")
            # capacity rule made deterministic: without back-off more than 50 000 pending items is an error
            if not back_off and n_pending >= WORK_QUEUE_CAP:
                raise RuntimeError("Internal error adding data to work queue, without backoff. "
                                   "Is your fastq batch larger than 50000? Perhaps try `map_batch` with back_off=True?")
            items.append(item)
            reads.append(s)
        results = []
        # the whole iterable is consumed before the first result is yielded (lib.rs:845-903); batches bound device memory
        step_bases, lo = 64_000_000, 0
        while lo < len(reads):
            hi, nb = lo, 0
            while hi < len(reads) and (hi == lo or nb + len(reads[hi]) <= step_bases) and hi - lo < WORK_QUEUE_CAP:
                nb += len(reads[hi]); hi += 1
            maps = self._map_many(reads[lo:hi], _ffi.OUT_CS)          # cs=true, MD=false: lib.rs:589-590
            for j, m in enumerate(maps):
                if isinstance(m, Exception):
                    continue                                          # worker error => no result for that id (lib.rs:621-623)
                results.append((m, items[lo + j]))
            lo = hi
        return AlignmentBatchResultIter(results)

    def _stage_runner(self):
        """per-stage access to the same kernels (parity tests, kernel bench)"""
        return _ffi.StageRunner(self._idx, self._mo, self._device)

    def __del__(self):
        try:
            if self._ctx: self._L.mm355_ctx_destroy(self._ctx)
            if self._idx: self._L.mm355_index_free(self._idx)
        except Exception:
            pass
