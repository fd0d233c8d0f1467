// mm355_dev.h -- device-side structs and kernel launch prototypes (host <-> .hip boundary inside the library)
#pragma once
#include "mm355_core.h"

struct DevIndex {
	const mm355_slot *slots;   // n_lines * 8 slots, 128-B lines
	uint64_t line_mask;        // n_lines - 1
	const uint64_t *pos;       // y words of multi-occurrence minimizers, ascending inside a run
	const uint32_t *S2;        // 2-bit packed reference, 16 bases per u32 (an ambiguous base is stored as 0 ...
	const uint64_t *nr;        // ... and listed here: n_nr runs [nr[2i], nr[2i+1]) of ambiguous bases, global offsets, ascending)
	uint32_t n_nr;
	const uint64_t *seq_off;   // per contig offset into S (bases)
	const uint32_t *seq_len;
	int32_t k, w, b, flag;
	uint32_t n_seq;
};

#ifdef __HIPCC__
// U:index.c::mm_idx_getseq on the 2-bit image: the N runs that can touch [lo, hi) -- `first` = the first run that ends after lo, `cnt` = how
// many of them start before hi (0 almost always: a genome has a few hundred runs) -- and the code (0..4) of the base at global offset o
__device__ __forceinline__ void ref_window(const DevIndex &ix, uint64_t lo, uint64_t hi, uint32_t &first, uint32_t &cnt)
{
	uint32_t a = 0, b = ix.n_nr;
	while (a < b) { const uint32_t m = (a + b) >> 1; if (ix.nr[2 * m + 1] > lo) b = m; else a = m + 1; }
	first = a;
	uint32_t c = 0;
	while (a + c < ix.n_nr && ix.nr[2 * (a + c)] < hi) ++c;
	cnt = c;
}
__device__ __forceinline__ uint32_t ref_code(const DevIndex &ix, uint32_t first, uint32_t cnt, uint64_t o)
{
	uint32_t c = ix.S2[o >> 4] >> ((o & 15) << 1) & 3u;
	for (uint32_t k = 0; k < cnt; ++k) if (o >= ix.nr[2 * (first + k)] && o < ix.nr[2 * (first + k) + 1]) c = 4;
	return c;
}
#endif

struct DevParams {           // subset of mm_mapopt_t the kernels read
	int64_t flag;
	int32_t mid_occ, max_max_occ, occ_dist;
	float q_occ_frac;
	int32_t max_gap, max_gap_ref, max_frag_len;
	int32_t bw, max_chain_skip, max_chain_iter, min_cnt, min_chain_score;
	float pen_gap, pen_skip;
};

struct DevBatch {            // one batch of reads resident in HBM
	int32_t n_reads;
	const uint8_t *seq;        // ASCII (or raw codes), each read starts 16-B aligned
	const int64_t *roff;       // byte offset of read r in seq; also its slot offset in mz/seed arrays
	const int32_t *rlen;
	const int32_t *order;      // reads sorted by length (desc) for the lane-per-read kernels
	unsigned long long *prof;  // MM355_KPROF=1: [0..31] summed / [32..63] maximal shader cycles per kernel phase (diagnostics), else null
};

struct DevSeeds {
	mm128 *mz;                 // [total padded bases] minimizers, read r at roff[r]
	mm128 *mz_tmp;             // scratch of the same size (mz_flt sort)
	int32_t *n_mz;             // [n_reads]
	uint32_t *sn;              // per minimizer: #occurrences in the index (0 = absent)
	uint64_t *sv;              // per minimizer: y word (n==1) or offset into pos[]
	uint8_t *sflt;             // per minimizer: 1 = filtered by mm_seed_select
	int32_t *hl;               // per minimizer scratch: hit list
	uint32_t *soff;            // per minimizer: exclusive prefix of kept n within the read
	int32_t *n_a;              // [n_reads] anchors per read
	int32_t *rep_len;          // [n_reads]
	int32_t *n_mini;           // [n_reads] kept seeds
	uint64_t *mini_pos;        // per minimizer slot: q_span<<32 | q_pos>>1 of kept seeds (compacted per read)
	unsigned long long *counters; // [8]: n_hit, n_a_multi, probes, chain_pairs ...
};

struct DevAnchors {
	const int64_t *aoff;       // [n_reads+1] anchor offsets
	mm128 *a;                  // anchors (generation order, then sorted in place)
	int32_t *f, *p, *v;        // chaining DP arrays
	uint64_t *z;               // backtrack scratch: f<<32|i
	uint8_t *t8;               // backtrack marks
	int32_t *vi;               // backtrack chain member list
	mm128 *b;                  // compacted anchors (output of compact_a)
	mm128 *wk;                 // chain sort keys scratch
	uint64_t *u;               // [total anchors / 1] chain descriptors, read r at aoff[r]
	uint64_t *u2;
	int32_t *n_u, *n_v;        // [n_reads]
	const int32_t *tcnt;       // optional (fast sort): equal-key pair counts of the sorted array, see WalkScratch
};

// Per-kernel HIP-event timers (mm355_stats_t::ms_kernel): a begin / end pair of events around ONE kernel (or a run of kernels with no host wait
// between them) on the stream it is launched on.  kt = the context, or null for "not timed".  bench.py's `roofline` is the entry with the
// largest summed duration among these and the extension kernels' own timers (ms_dp_group) -- no exclusion list.
enum { KT_SKETCH = 0, KT_MZFLT, KT_LOOKUP, KT_SELECT, KT_EXPAND, KT_CULL, KT_ASORT, KT_LITERAL, KT_CHAIN_SEG, KT_CHAIN_BIG, KT_CHAIN_SMALL, KT_BACKTRACK,
       KT_RMQ_SORT, KT_RMQ_DP, KT_RMQ_BT, KT_DP_GATHER, KT_DP_BACKTRACK, KT_EXTRA, KT_CODES, KT_PACK, KT_LIT_MED, KT_LIT_TASKS, KT_TIE_AUX, KT_N };
#ifdef __HIPCC__
void mm355_kt(void *kt, int slot, int end, hipStream_t st);
struct KtScope { void *kt; int slot; hipStream_t st; KtScope(void *k, int s, hipStream_t t) : kt(k), slot(s), st(t) { if (kt) mm355_kt(kt, slot, 0, st); } ~KtScope() { if (kt) mm355_kt(kt, slot, 1, st); } };
int mm355_sketch_chunk_size(void);
void mm355_launch_sketch(const DevIndex &ix, const DevBatch &bt, DevSeeds &sd, const int32_t *chunk_read, const int32_t *chunk_start, int n_chunks,
                         const int64_t *read_chunk0, int32_t *chunk_n, hipStream_t st, void *kt = 0);
void mm355_launch_mzflt(const DevParams &pr, const DevBatch &bt, DevSeeds &sd, hipStream_t st, void *kt = 0);
void mm355_launch_seed_lookup(const DevIndex &ix, const DevBatch &bt, DevSeeds &sd, const int32_t *chunk_read, const int32_t *chunk_start, int n_chunks,
                              unsigned long long *hit_ctr, unsigned int *tile_ctr, hipStream_t st, void *kt = 0);
void mm355_launch_seed_select(const DevIndex &ix, const DevParams &pr, const DevBatch &bt, DevSeeds &sd, hipStream_t st, void *kt = 0);
void mm355_launch_seed_expand(const DevIndex &ix, const DevParams &pr, const DevBatch &bt, DevSeeds &sd, DevAnchors &an, hipStream_t st, void *kt = 0);
struct SortTask { int32_t read; uint32_t beg, end; int32_t s; };   // a bucket [beg, end) of one read's array, to be sorted from byte shift s
int mm355_launch_sort(const DevBatch &bt, DevAnchors &an, int *err, const void *h_tasks, int n_big, int n_med, int n_small, size_t n_elems, void *task_buf, size_t task_cap, hipStream_t st, void *kt = 0, int n_levels = 0);
int mm355_sort_heavy_threshold(void);
size_t mm355_sort_buf_bytes(size_t task_cap);   // device scratch mm355_launch_sort needs for `task_cap`
int mm355_sort_medium_threshold(void);
int mm355_chain_chunk(void);
int mm355_launch_chain(const DevParams &pr, const DevBatch &bt, DevAnchors &an, unsigned long long *pairs, void *seg_small, void *seg_big, unsigned int *ctr,
                       const void *chunks, int n_chunks, hipStream_t st, void *kt = 0);
void mm355_launch_backtrack(const DevParams &pr, const DevBatch &bt, DevAnchors &an, int *err, const int32_t *heavy_first, unsigned int *qctr, hipStream_t st, void *kt = 0);
int mm355_resident_blocks(void);
#endif
hipError_t mm355_wait_stream(hipStream_t st);   // like hipStreamSynchronize, but the calling thread sleeps
