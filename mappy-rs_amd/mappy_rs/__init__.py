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
import collections
import collections.abc
import ctypes as C
import os
import queue
import threading
import time
import weakref

# ROCclr multiplexes all HIP streams over GPU_MAX_HW_QUEUES hardware queues (default 4); the pipelined map_batch keeps several contexts in
# flight and measures best with 8 (bench.py sets the same).  Only effective if the HIP runtime has not been started yet.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

from . import _ffi

__all__ = ["Aligner", "Mapping", "shard_by_bases", "order_by_length"]

_CIGAR_OPS = "MIDNSHP=X"

# capacity constants of the reference (lib.rs:429-430, 950)
SUB_BATCH_READS, SUB_BATCH_BASES = 9216, 96_000_000   # one GPU sub-batch of map_batch (bench.py's default shape: a 73 728-read block over eight contexts)
WORK_QUEUE_CAP = 50000
# results a batch may hold before its workers wait for the consumer: the reference's results_queue (ArrayQueue of 50 000, lib.rs:430) plus its
# bounded(20000) channel (lib.rs:950) -- here one channel.  (With 20 000 alone the workers stall while map_batch is still consuming its
# iterable, which nobody reads during: -12 % on 262 144 reads.)
RESULT_CHANNEL_CAP = 70000


_HIT_FIELDS = [k for k, _t in _ffi.Hit._fields_]
_F = {k: i for i, k in enumerate(_HIT_FIELDS)}
_HIT_DTYPE = np.dtype([(k, np.dtype(t)) for k, t in _ffi.Hit._fields_], align=True)
(_QS, _QE, _ST, _RID, _TL, _TS, _TE, _ML, _BL, _MQ, _PR, _NM, _NC, _CO, _CSO, _CSL, _MDO, _MDL) = (_F[k] for k in (
    "query_start", "query_end", "strand", "rid", "target_len", "target_start", "target_end", "match_len", "block_len", "mapq",
    "is_primary", "NM", "n_cigar", "cigar_off", "cs_off", "cs_len", "md_off", "md_len"))


class _HitBatch:
    """what the Mapping records of one mm355_hits_t share: the packed CIGAR words, the string arena, the contig names"""
    __slots__ = ("cig", "sbuf", "names")

    def __init__(self, cig, sbuf, names):
        self.cig, self.sbuf, self.names = cig, sbuf, names


class Mapping:
    """Result record; fields and aliases of mappy_rs::Mapping (lib.rs:109-154, 196-284).

    A record that comes out of the mapping path is a view: `_r` is the row of the C-ABI hit array (a tuple, converted in bulk), `_b`
    the arenas it points into; fields are read on access (the reference converts its Vec / Strings on access as well, lib.rs:196-284).
    Creating a record is two slot stores -- the per-hit cost under the GIL that used to cap `map_batch` below the C-ABI rate.
    Retention: an untouched view keeps its sub-batch's arenas alive (CIGAR words + cs / MD bytes of up to SUB_BATCH_READS reads, a few MB).
    The first access to `cigar`, `cs` or `MD` -- or `detach()` -- copies the record's own slices and drops that reference, so a caller that
    keeps a few records of a large `map_batch` (the reference's records own their Vec / Strings) holds a few hundred bytes each."""

    __slots__ = ("_b", "_r", "_cig", "_own")
    FIELDS = ("query_start", "query_end", "strand", "target_name", "target_len", "target_start", "target_end",
              "match_len", "block_len", "mapq", "is_primary", "cigar", "NM", "MD", "cs")

    def __init__(self, query_start, query_end, strand, target_name, target_len, target_start, target_end, match_len,
                 block_len, mapq, is_primary, cigar, NM, MD, cs):
        self._b = None
        self._r = None
        self._cig = list(cigar)
        self._own = (query_start, query_end, strand, target_name, target_len, target_start, target_end, match_len, block_len, mapq,
                     bool(is_primary), NM, MD, cs)

    @classmethod
    def _view(cls, batch, row):
        m = object.__new__(cls)
        m._b = batch
        m._r = row
        m._cig = None
        m._own = None
        return m

    query_start = property(lambda s: s._r[_QS] if s._own is None else s._own[0])
    query_end = property(lambda s: s._r[_QE] if s._own is None else s._own[1])
    strand = property(lambda s: s._r[_ST] if s._own is None else s._own[2])          # +1 / -1 (lib.rs:231-237)
    target_name = property(lambda s: s._b.names[s._r[_RID]] if s._own is None else s._own[3])
    target_len = property(lambda s: s._r[_TL] if s._own is None else s._own[4])
    target_start = property(lambda s: s._r[_TS] if s._own is None else s._own[5])
    target_end = property(lambda s: s._r[_TE] if s._own is None else s._own[6])
    match_len = property(lambda s: s._r[_ML] if s._own is None else s._own[7])
    block_len = property(lambda s: s._r[_BL] if s._own is None else s._own[8])
    mapq = property(lambda s: s._r[_MQ] if s._own is None else s._own[9])
    is_primary = property(lambda s: bool(s._r[_PR]) if s._own is None else s._own[10])
    NM = property(lambda s: s._r[_NM] if s._own is None else s._own[11])

    def detach(self):
        """make the record independent of its sub-batch: copy its fields, CIGAR and cs / MD out of the shared arenas and drop the reference"""
        if self._own is None:
            r, b = self._r, self._b
            w = b.cig[r[_CO]:r[_CO] + r[_NC]]
            self._cig = list(zip((w >> 4).tolist(), (w & 0xf).tolist()))
            md = b.sbuf[r[_MDO]:r[_MDO] + r[_MDL]].decode() if r[_MDL] >= 0 else None
            cs = b.sbuf[r[_CSO]:r[_CSO] + r[_CSL]].decode() if r[_CSL] >= 0 else None
            self._own = (r[_QS], r[_QE], r[_ST], b.names[r[_RID]], r[_TL], r[_TS], r[_TE], r[_ML], r[_BL], r[_MQ], bool(r[_PR]), r[_NM], md, cs)
            self._b = self._r = None
        return self

    @property
    def MD(self):
        if self._own is None:
            self.detach()
        return self._own[12]

    @property
    def cs(self):
        if self._own is None:
            self.detach()
        return self._own[13]

    @property
    def cigar(self):
        """list of (length, op) tuples, as mappy-rs; unpacked from the packed uint32 words (length << 4 | op) on first access"""
        if self._own is None:
            self.detach()
        return self._cig

    # mappy aliases (lib.rs:196-284)
    ctg = property(lambda s: s.target_name)
    ctg_len = property(lambda s: s.target_len)
    r_st = property(lambda s: s.target_start)
    r_en = property(lambda s: s.target_end)
    q_st = property(lambda s: s.query_start)
    q_en = property(lambda s: s.query_end)
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
        return "\t".join(str(x) for x in (self.query_start, self.query_end, "+" if self.strand > 0 else "-",
                                          self.target_name, self.target_len, self.target_start, self.target_end,
                                          self.match_len, self.block_len, self.mapq, tp, "cg:Z:" + self.cigar_str))

    def __repr__(self):
        return "Mapping(%s)" % ", ".join("%s=%r" % (k, getattr(self, k)) for k in Mapping.FIELDS)

    def __eq__(self, o):
        return isinstance(o, Mapping) and all(getattr(self, k) == getattr(o, k) for k in Mapping.FIELDS)

    __hash__ = None


def _batch_to_mappings(hp, n_reads, names):
    """all hits of one mm355_hits_t -> list (per read) of list[Mapping] or RuntimeError.  One bulk copy per array (hit rows, CIGAR words,
    string arena); every Mapping is a view of its row (fields, cs / MD strings and the CIGAR list are produced on access)."""
    h = hp.contents
    nh = int(h.n_hits)
    off = np.ctypeslib.as_array(h.hit_off, shape=(n_reads + 1,)).tolist()
    status = np.ctypeslib.as_array(h.status, shape=(max(n_reads, 1),))
    empty = np.flatnonzero(status[:n_reads] == _ffi.MM355_EEMPTY).tolist() if n_reads else []
    if nh:
        assert _HIT_DTYPE.itemsize == C.sizeof(_ffi.Hit)
        rows = np.frombuffer(C.string_at(h.hits, nh * C.sizeof(_ffi.Hit)), dtype=_HIT_DTYPE).tolist()
        cig = np.ctypeslib.as_array(h.cigar, shape=(max(int(h.n_cigar), 1),)).copy()
        sbuf = C.string_at(h.str, int(h.n_str)) if h.n_str else b""
        B = _HitBatch(cig, sbuf, names)
        view = Mapping._view
        ms = [view(B, r) for r in rows]
        out = [ms[off[i]:off[i + 1]] for i in range(n_reads)]
    else:
        out = [[] for _ in range(n_reads)]
    for i in empty:
        out[i] = RuntimeError("Sequence is empty")
    return out


def shard_by_bases(lengths, n_shards):
    """a batch is cut into `n_shards` CONTIGUOUS shards balanced by cumulative bases (not by read count): a read belongs to
    the shard that contains the midpoint of its span on the cumulative-bases axis.  Returns n_shards+1 boundaries b with shard
    s = items[b[s]:b[s+1]]; every item belongs to exactly one shard."""
    n_shards = max(1, int(n_shards))
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    if n == 0:
        return [0] * (n_shards + 1)
    cum = np.cumsum(lengths)
    total = max(1, int(cum[-1]))
    mid2 = 2 * cum - lengths                      # twice the midpoint, exact in integers
    shard = np.minimum(n_shards - 1, (mid2 * n_shards) // (2 * total))
    return [int(np.searchsorted(shard, s, side="left")) for s in range(n_shards)] + [n]


def order_by_length(lengths):
    """indices of the reads, longest first (stable).  The per-read kernels of a sub-batch cost the latency of its longest read, so a
    dispatcher that has a window of pending reads cuts its sub-batches from this order: sub-batches of similar reads, the few very long
    ones together"""
    return np.argsort(-np.asarray(lengths, dtype=np.int64), kind="stable").tolist()


class _Channel:
    """bounded multi-producer channel (the reference's ArrayQueue(50000) + crossbeam `bounded(20000)`, lib.rs:430, 950): producers block while it is full, the consumer blocks
    while it is empty; both waits release the GIL."""

    def __init__(self, cap):
        self.cap = cap
        self.d = collections.deque()
        self.cv = threading.Condition()

    def put_many(self, items, cancel):
        """appends the entries in order, waiting whenever the channel is full; False when `cancel` was set while waiting"""
        i, n = 0, len(items)
        with self.cv:
            while i < n:
                while len(self.d) >= self.cap:
                    if cancel.is_set():
                        return False
                    self.cv.wait(0.2)
                room = self.cap - len(self.d)
                self.d.extend(items[i:i + room])
                i += room
                self.cv.notify_all()
        return True

    def get(self):
        with self.cv:
            while not self.d:
                self.cv.wait()
            x = self.d.popleft()
            if len(self.d) == self.cap - 1:
                self.cv.notify_all()      # a producer may be waiting for room
            return x

    def get_many(self, limit):
        """blocks until at least one entry is there, then takes up to `limit` of them under ONE lock acquisition (a lock per result was a
        microsecond and a half of the consumer's GIL time per read)"""
        with self.cv:
            while not self.d:
                self.cv.wait()
            was_full = len(self.d) >= self.cap
            n = min(limit, len(self.d))
            out = [self.d.popleft() for _ in range(n)]
            if was_full:
                self.cv.notify_all()      # producers may be waiting for room
            return out

    def __len__(self):
        return len(self.d)


class _BatchState:
    """Everything the worker threads and the collector of one map_batch call share.  The threads hold THIS object, never the iterator
    the caller gets: when the caller drops the iterator (or calls close()) a weakref finalizer sets `cancel` / `abandoned` here, the
    workers blocked on the full result channel give up, return their GPU contexts to the Aligner's pool and exit -- the reference's
    workers stop the same way when their `tx.send` fails because the receiver is gone."""

    def __init__(self):
        self.ch = _Channel(RESULT_CHANNEL_CAP)
        self.cancel = threading.Event()      # workers stop taking sub-batches (worker error, validation error, abandoned iterator)
        self.abandoned = threading.Event()   # nobody will ever read the channel again
        self.errors = []
        self.n_sub_batches = 0
        self.t_sub_done = []      # wall-clock time each sub-batch left the GPU pipeline (tests / latency diagnostics)
        self.threads = []         # workers + collector (tests: they must all exit once the iterator is gone)

    def close(self):
        self.cancel.set()
        self.abandoned.set()


class AlignmentBatchResultIter:
    """Iterator returned by map_batch (lib.rs:923-991): yields (list[Mapping], dict) in COMPLETION order.

    Mirrors the reference's plumbing: workers push results into a bounded channel (RESULT_CHANNEL_CAP entries, lib.rs:430 + 950) and `__next__`
    blocks on it (`rx.recv()`, lib.rs:973) -- here with the GIL released.  A full channel blocks the workers: a slow consumer holds
    back the GPU pipeline instead of growing memory (back-pressure).  `Finished` arrives after every worker is done (lib.rs:804-815)."""

    _FINISHED = object()

    def __init__(self, state=None):
        self._st = state if state is not None else _BatchState()
        self._done = False
        self._buf = []            # results taken from the channel in one go, handed out one by one (reversed: pop() from the end)
        self.t_first_yield = None
        self._fin = weakref.finalize(self, _BatchState.close, self._st)   # the state, not the iterator, is what the threads keep alive

    n_sub_batches = property(lambda s: s._st.n_sub_batches)
    t_sub_done = property(lambda s: s._st.t_sub_done)

    def __iter__(self):
        return self

    def __next__(self):
        if self._done:
            raise StopIteration("Finished")
        if not self._buf:
            self._buf = self._st.ch.get_many(2048)
            self._buf.reverse()
        r = self._buf.pop()
        if r is AlignmentBatchResultIter._FINISHED:
            self._done = True
            if self._st.errors:
                raise self._st.errors[0]
            raise StopIteration("Finished")
        if self.t_first_yield is None:
            self.t_first_yield = time.perf_counter()
        return r

    def close(self):
        """stop mapping what has not been started yet; results already produced are dropped"""
        self._st.close()


class Aligner:
    """mappy-compatible aligner (lib.rs:288-671) whose mapping path runs on MI355X GPUs.

    `device` / `devices` are the only additions to the reference's constructor: the GPU (or list of GPUs of one node) that map
    this Aligner's reads.  With several devices the index is replicated into each one's HBM (mm355_upload) and `map_batch` deals
    its sub-batches to the contexts of all of them -- the GPU analogue of the reference's N worker threads over one shared index
    (lib.rs:541-636).  Reads are independent: no collective."""

    def __init__(self, fn_idx_in=None, preset=None, k=None, w=None, min_cnt=None, min_chain_score=None,
                 min_dp_score=None, bw=None, best_n=None, n_threads=3, fn_idx_out=None, max_frag_len=None,
                 extra_flags=None, seq=None, scoring=None, device=0, devices=None):
        L = _ffi.lib()
        self._L = L
        self._idx = C.c_void_p()
        self._ctx = C.c_void_p()
        self._wctx = {}                      # device -> free contexts of the map_batch pipeline workers (one per host thread)
        self._name_cache = None
        self._devices = [int(d) for d in devices] if devices is not None else [int(device)]
        if not self._devices:
            raise ValueError("`devices` must name at least one GPU")
        self._device = self._devices[0]
        self._n_threads = 0
        self._lock = threading.Lock()
        io, mo = _ffi.IdxOpt(), _ffi.MapOpt()
        L.mm355_set_opt(None, C.byref(io), C.byref(mo))
        if preset is not None:
            rc = L.mm355_set_opt(str(preset).encode(), C.byref(io), C.byref(mo))
            # The reference ignores mm_set_opt's return value (lib.rs:336): an unknown name leaves the defaults in place, silently, and so
            # does this mirror (MM355_EINVAL).  A preset minimap2 knows but this path does not implement (sr, splice ...) must not be
            # mapped with other parameters behind the caller's back: refuse.
            if rc == _ffi.MM355_EUNSUP:
                raise NotImplementedError("preset %r is not implemented by the MI355X mapping path (long-read presets only: map-ont, "
                                          "map-hifi, map-pb, asm5/asm10/asm20, ava-ont, ava-pb)" % (preset,))
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
        if len(self._devices) > 1:         # replicate now, so that the first map_batch does not pay for it
            arr = (C.c_int32 * len(self._devices))(*self._devices)
            rc = L.mm355_upload(self._idx, arr, len(self._devices))
            if rc != 0:
                raise RuntimeError("mm355: " + L.mm355_strerror(rc).decode())

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
        """lib.rs:464-470 with the rules of lib.rs:706-766: None on any error (incl. "No sequence in this index", lib.rs:710-714)."""
        L = self._L
        if not self._idx:
            return None
        if (self._mo.flag & 4) and (self._info()[3] & 2):      # MM_F_CIGAR && MM_I_NO_SEQ
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

    def _ctx_acquire(self, slot):
        """a free worker context on GPU devices[slot % n_devices] (created on demand; contexts are pooled per device and never shared
        between two running workers, also not between two overlapping map_batch calls)"""
        dev = self._devices[slot % len(self._devices)]
        with self._lock:
            free = self._wctx.setdefault(dev, [])
            if free:
                return dev, free.pop()
        ctx = C.c_void_p()
        rc = self._L.mm355_ctx_create(self._idx, dev, C.byref(ctx))
        if rc != 0:
            raise RuntimeError(self._L.mm355_strerror(rc).decode())
        return dev, ctx

    def _ctx_release(self, dev_ctx):
        with self._lock:
            self._wctx.setdefault(dev_ctx[0], []).append(dev_ctx[1])

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
        return [Mapping(0, 1000, 1, "Hello", 101010, 10, 1010, 1000, 1000, 60, True, [], 0, None, "Cigar string")]

    # ---- batch path (lib.rs:541-648, 771-906)
    def enable_threading(self, n_threads):
        """In the reference this spawns N mm_map worker threads; here it arms the GPU batch path: map_batch drives up to
        min(n_threads, 8) host threads per GPU, each with its own context (HIP streams + buffers)."""
        self._n_threads = int(n_threads)

    def map_batch(self, seqs, back_off=True):
        """lib.rs:639-648, 771-906.  The iterable is consumed completely before the iterator is returned (lib.rs:845-903), but -- as in the
        reference, whose workers pop the queue while `_map_batch` is still pushing -- mapping starts as soon as the first sub-batch of reads
        has been taken from it, and results stream out of the returned iterator in completion order while later sub-batches are still on the
        GPU (lib.rs:793-839: collector thread -> bounded channel -> `__next__`)."""
        if self._n_threads == 0:
            raise RuntimeError("Multi threading not enabled on this instance. Please call `.enable_threading()`")
        # accepted iterables: list / tuple / iterator / generator / sequence -- not dict, not str (lib.rs:782-792, 910-920)
        if isinstance(seqs, (dict, str, bytes)) or not (isinstance(seqs, (list, tuple, collections.abc.Sequence)) or
                                                        isinstance(seqs, collections.abc.Iterator)):
            raise TypeError("Unsupported batch type, pass a list, iter, generator or tuple")
        st = _BatchState()
        max_workers = max(1, min(self._n_threads, 8)) * len(self._devices)
        try:
            n_known = len(seqs)
        except TypeError:
            n_known = None
        # sub-batch size: at most SUB_BATCH_READS; small inputs are cut finer so that every worker gets about two sub-batches (16384 reads:
        # 1024 per sub-batch maps 20 % faster than 4096).  Unknown length (iterators): ramp up, so that the first results leave early.
        sb_fixed = None if n_known is None else min(SUB_BATCH_READS, max(1024, -(-n_known // (2 * max_workers))))

        def sb_size(k):
            return sb_fixed if sb_fixed is not None else min(SUB_BATCH_READS, 512 << min(4, k // max_workers))

        work = collections.deque()          # sub-batches (reads, items) waiting for a worker
        cv = threading.Condition()
        state = {"closed": False, "n_sub": 0, "pending": 0}     # pending: reads in `work`, not yet taken by a worker
        workers = st.threads
        self._names()                       # fill the name cache before the workers read it
        map_many, acquire, release = self._map_many, self._ctx_acquire, self._ctx_release

        # (the closures below capture `st`, never the iterator handed to the caller)
        def worker(slot):
            ctx = None
            try:
                ctx = acquire(slot)
                while True:
                    with cv:
                        while not work and not state["closed"] and not st.cancel.is_set():
                            cv.wait(0.2)
                        if st.cancel.is_set() or not work:
                            return
                        reads, items = work.popleft()
                        state["pending"] -= len(reads)
                        cv.notify_all()                                   # the producer may be waiting for room (back-off)
                    maps = map_many(reads, _ffi.OUT_CS, ctx[1])           # cs=true, MD=false: lib.rs:589-590
                    st.t_sub_done.append(time.perf_counter())
                    # a worker error on one read => no result for that id (lib.rs:621-623)
                    out = [(m, it) for m, it in zip(maps, items) if not isinstance(m, Exception)]
                    if not st.ch.put_many(out, st.cancel):
                        return
            except Exception as e:   # surfaced by the iterator when it finishes, like a worker panic in the reference
                st.errors.append(e)
                st.cancel.set()
            finally:
                if ctx is not None:
                    release(ctx)

        def dispatch(reads, items):
            with cv:
                # the reference's work queue holds 50 000 reads (lib.rs:429): with back-off the producer sleeps until the workers have
                # made room (lib.rs:870-888), so a huge iterable never sits in memory as pending sub-batches
                # (not while the result channel is full: nobody reads it before map_batch has returned, so the workers cannot make room)
                while (back_off and state["pending"] > 0 and state["pending"] + len(reads) > WORK_QUEUE_CAP and not st.cancel.is_set()
                       and len(st.ch) < RESULT_CHANNEL_CAP):
                    cv.wait(0.05)
                work.append((reads, items))
                state["pending"] += len(reads)
                state["n_sub"] += 1
                cv.notify()
            if len(workers) < max_workers and len(workers) < state["n_sub"]:
                t = threading.Thread(target=worker, args=(len(workers),), daemon=True)
                workers.append(t)
                t.start()

        def close_and_join():
            with cv:
                state["closed"] = True
                cv.notify_all()
            for t in list(workers):
                t.join()

        cur_reads, cur_items, cur_bases = [], [], 0
        sb_limit = sb_size(0)
        n_fast = 0
        try:
            # lists and tuples of plain dicts with str sequences -- what a FASTQ reader hands over -- are cut a sub-batch at a time with
            # C-level loops (0.4 us per read instead of 2 under the interpreter: the producer shares the GIL with the workers' result
            # building).  Anything else -- another element type, a missing key, a sub-batch over the base limit, the capacity rule without
            # back-off -- leaves the remainder to the element-wise loop below, which raises what the reference raises, at the same element.
            if isinstance(seqs, (list, tuple)) and (back_off or len(seqs) <= WORK_QUEUE_CAP):
                n_all = len(seqs)
                while n_fast < n_all:
                    chunk = seqs[n_fast:n_fast + sb_limit]
                    if set(map(type, chunk)) != {dict}:
                        break
                    try:
                        reads = [it["seq"] for it in chunk]
                    except KeyError:
                        break
                    if set(map(type, reads)) != {str} or sum(map(len, reads)) > SUB_BATCH_BASES:
                        break
                    dispatch(reads, list(map(dict, chunk)))   # the reference hands back its own copy of the dict (lib.rs:849-855, 977-979)
                    n_fast += len(chunk)
                    sb_limit = sb_size(state["n_sub"])
                if n_fast:
                    seqs = seqs[n_fast:]
            for n_pending, item in enumerate(seqs, n_fast):
                if not isinstance(item, dict):
                    raise TypeError("Element in iterable is not a dictionary")
                if "seq" not in item:
                    raise KeyError("AHHH Key \U0001F5DD\uFE0F  not found in iterated dictionary")
                s = item["seq"]
                if not isinstance(s, str):
                    raise ValueError("`seq` must be a string")
                # capacity rule made deterministic: without back-off more than 50 000 pending items is an error
                if not back_off and n_pending >= WORK_QUEUE_CAP:
                    raise RuntimeError("Internal error adding data to work queue, without backoff. "
                                       "Is your fastq batch larger than 50000? Perhaps try `map_batch` with back_off=True?")
                if len(cur_reads) >= sb_limit or cur_bases + len(s) > SUB_BATCH_BASES:
                    if cur_reads:
                        dispatch(cur_reads, cur_items)
                        cur_reads, cur_items, cur_bases = [], [], 0
                        sb_limit = sb_size(state["n_sub"])
                cur_reads.append(s)
                cur_items.append(dict(item))      # the reference hands back its own copy of the dict (lib.rs:849-855, 977-979)
                cur_bases += len(s)
            if cur_reads:
                dispatch(cur_reads, cur_items)
        except BaseException:
            st.close()                            # nothing is yielded: the workers stop after their current sub-batch
            close_and_join()
            raise
        st.n_sub_batches = state["n_sub"]

        def finalize():                           # the reference's collector thread: `Finished` once every worker is done (lib.rs:804-815)
            close_and_join()
            while not st.ch.put_many([AlignmentBatchResultIter._FINISHED], st.abandoned):
                if st.abandoned.is_set():
                    return

        tf = threading.Thread(target=finalize, daemon=True)
        tf.start()
        st.threads = list(workers) + [tf]
        return AlignmentBatchResultIter(st)

    def _stage_runner(self):
        """per-stage access to the same kernels (parity tests, kernel bench)"""
        return _ffi.StageRunner(self._idx, self._mo, self._device)

    def __del__(self):
        try:
            if self._ctx: self._L.mm355_ctx_destroy(self._ctx)
            for ctxs in self._wctx.values():
                for ctx in ctxs: self._L.mm355_ctx_destroy(ctx)
            if self._idx: self._L.mm355_index_free(self._idx)
        except Exception:
            pass
