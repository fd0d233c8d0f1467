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
import os
import threading

# ROCclr multiplexes all HIP streams over GPU_MAX_HW_QUEUES hardware queues (default 4); the pipelined map_batch keeps several contexts in
# flight and measures best with 8 (bench.py sets the same).  Only effective if the HIP runtime has not been started yet.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

from . import _ffi

__all__ = ["Aligner", "Mapping"]

_CIGAR_OPS = "MIDNSHP=X"

# capacity constants of the reference (lib.rs:429-430, 950)
SUB_BATCH_READS, SUB_BATCH_BASES = 4096, 32_000_000   # one GPU sub-batch of map_batch
WORK_QUEUE_CAP = 50000
RESULT_CHANNEL_CAP = 20000


class Mapping:
    """Result record; fields and aliases of mappy_rs::Mapping (lib.rs:109-154, 196-284)."""

    __slots__ = ("query_start", "query_end", "_strand", "target_name", "target_len", "target_start", "target_end",
                 "match_len", "block_len", "mapq", "is_primary", "_cig", "NM", "MD", "cs")

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
        self._cig = cigar          # list of (length, op) or the packed uint32 words (length << 4 | op) until first read
        self.NM = NM
        self.MD = MD
        self.cs = cs

    @property
    def cigar(self):
        """list of (length, op) tuples, as mappy-rs; unpacked on first access (the reference converts its Vec on access as well)"""
        c = self._cig
        if not isinstance(c, list):
            c = list(zip((c >> 4).tolist(), (c & 0xf).tolist()))
            self._cig = c
        return c

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
        return "Mapping(%s)" % ", ".join("%s=%r" % (k.lstrip("_"), getattr(self, "cigar" if k == "_cig" else k)) for k in self.__slots__)

    def __eq__(self, o):
        return isinstance(o, Mapping) and all(getattr(self, "cigar" if k == "_cig" else k) == getattr(o, "cigar" if k == "_cig" else k)
                                              for k in self.__slots__)


_HIT_DTYPE = np.dtype([(k, np.dtype(t)) for k, t in _ffi.Hit._fields_], align=True)


def _batch_to_mappings(hp, n_reads, names):
    """all hits of one mm355_hits_t -> list (per read) of list[Mapping] or RuntimeError.  One bulk copy per array; CIGARs stay packed
    (unpacked when .cigar is first read), strings are sliced from one bytes object."""
    h = hp.contents
    nh = int(h.n_hits)
    off = np.ctypeslib.as_array(h.hit_off, shape=(n_reads + 1,)).tolist()
    status = np.ctypeslib.as_array(h.status, shape=(max(n_reads, 1),)).tolist()
    if nh:
        assert _HIT_DTYPE.itemsize == C.sizeof(_ffi.Hit)
        rows = np.frombuffer(C.string_at(h.hits, nh * C.sizeof(_ffi.Hit)), dtype=_HIT_DTYPE).tolist()
        cig = np.ctypeslib.as_array(h.cigar, shape=(max(int(h.n_cigar), 1),)).copy()
        sbuf = C.string_at(h.str, int(h.n_str)) if h.n_str else b""
    F = {k: i for i, (k, _t) in enumerate(_ffi.Hit._fields_)}
    qs, qe, st, rid, tl, ts, te, ml, bl, mq, pr, nm, nc, co, cso, csl, mdo, mdl = (F[k] for k in (
        "query_start", "query_end", "strand", "rid", "target_len", "target_start", "target_end", "match_len", "block_len", "mapq",
        "is_primary", "NM", "n_cigar", "cigar_off", "cs_off", "cs_len", "md_off", "md_len"))
    out = []
    for i in range(n_reads):
        if status[i] == _ffi.MM355_EEMPTY:
            out.append(RuntimeError("Sequence is empty"))
            continue
        ms = []
        for k in range(off[i], off[i + 1]):
            x = rows[k]
            cs = sbuf[x[cso]:x[cso] + x[csl]].decode() if x[csl] >= 0 else None
            md = sbuf[x[mdo]:x[mdo] + x[mdl]].decode() if x[mdl] >= 0 else None
            ms.append(Mapping(x[qs], x[qe], x[st], names[x[rid]], x[tl], x[ts], x[te], x[ml], x[bl], x[mq], bool(x[pr]),
                              cig[x[co]:x[co] + x[nc]], x[nm], md, cs))
        out.append(ms)
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
        self._wctx = []                      # contexts of the map_batch pipeline workers (one per host thread)
        self._name_cache = None
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

    def _map_many(self, seqs, flags, ctx=None):
        """one mm355_map_batch call; returns list of list[Mapping].  ctx: a pipeline worker's own context (no lock needed)"""
        L = self._L
        arr, lens, keep = _ffi.pack_reads(seqs)
        hp = C.POINTER(_ffi.Hits)()
        if ctx is None:
            with self._lock:
                rc = L.mm355_map_batch(self._context(), C.byref(self._mo), len(seqs), arr, lens, flags, C.byref(hp))
        else:
            rc = L.mm355_map_batch(ctx, C.byref(self._mo), len(seqs), arr, lens, flags, C.byref(hp))
        if rc != 0:
            raise RuntimeError(L.mm355_strerror(rc).decode())
        try:
            return _batch_to_mappings(hp, len(seqs), self._names())
        finally:
            L.mm355_free_hits(hp)

    def _names(self):
        if self._name_cache is None:
            self._name_cache = [nm.decode() if nm is not None else None for nm in (self._L.mm355_index_seq_name(self._idx, i) for i in range(self.n_seq))]
        return self._name_cache

    def _worker_contexts(self, n):
        while len(self._wctx) < n:
            ctx = C.c_void_p()
            rc = self._L.mm355_ctx_create(self._idx, self._device, C.byref(ctx))
            if rc != 0:
                raise RuntimeError(self._L.mm355_strerror(rc).decode())
            self._wctx.append(ctx)
        return self._wctx[:n]

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
        """In the reference this spawns N mm_map worker threads; here it arms the GPU batch path: map_batch drives up to
        min(n_threads, 8) host threads, each with its own context (HIP streams + buffers) on the one GPU."""
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
        # The whole iterable is consumed before the first result is yielded (lib.rs:845-903).  The reads then go through the GPU in
        # sub-batches of a few thousand reads: `n_threads` host threads (enable_threading; at most 8 are useful) each drive their own
        # context, so that the front kernels, the host tail and the extension rounds of different sub-batches overlap.
        # sub-batch size: at most SUB_BATCH_READS, but small inputs are cut finer so that every worker gets about two sub-batches
        # (16384 reads: 1024 per sub-batch maps 20 % faster than 4096, 4096 reads 50 % faster)
        sb_reads = min(SUB_BATCH_READS, max(1024, -(-len(reads) // (2 * max(1, min(self._n_threads, 8))))))
        subs, lo = [], 0
        while lo < len(reads):
            hi, nb = lo, 0
            while hi < len(reads) and (hi == lo or nb + len(reads[hi]) <= SUB_BATCH_BASES) and hi - lo < sb_reads:
                nb += len(reads[hi]); hi += 1
            subs.append((lo, hi))
            lo = hi
        n_workers = max(1, min(self._n_threads, 8, len(subs)))
        out = [None] * len(subs)
        if n_workers == 1:
            for k, (a, b) in enumerate(subs):
                out[k] = self._map_many(reads[a:b], _ffi.OUT_CS)      # cs=true, MD=false: lib.rs:589-590
        else:
            ctxs = self._worker_contexts(n_workers)
            nxt = [0]
            pick = threading.Lock()
            errors = []

            def work(ctx):
                while True:
                    with pick:
                        k = nxt[0]; nxt[0] += 1
                    if k >= len(subs) or errors:
                        return
                    a, b = subs[k]
                    try:
                        out[k] = self._map_many(reads[a:b], _ffi.OUT_CS, ctx)
                    except Exception as e:   # surfaced after the join, like a worker panic in the reference
                        errors.append(e)
                        return
            ths = [threading.Thread(target=work, args=(ctx,)) for ctx in ctxs]
            for t in ths: t.start()
            for t in ths: t.join()
            if errors:
                raise errors[0]
        results = []
        for (a, b), maps in zip(subs, out):
            for j, m in enumerate(maps):
                if isinstance(m, Exception):
                    continue                                          # worker error => no result for that id (lib.rs:621-623)
                results.append((m, items[a + j]))
        return AlignmentBatchResultIter(results)

    def _stage_runner(self):
        """per-stage access to the same kernels (parity tests, kernel bench)"""
        return _ffi.StageRunner(self._idx, self._mo, self._device)

    def __del__(self):
        try:
            if self._ctx: self._L.mm355_ctx_destroy(self._ctx)
            for ctx in self._wctx: self._L.mm355_ctx_destroy(ctx)
            if self._idx: self._L.mm355_index_free(self._idx)
        except Exception:
            pass
