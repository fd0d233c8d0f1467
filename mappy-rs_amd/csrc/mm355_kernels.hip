// mm355_kernels.hip -- hand-written gfx950 kernels of the seeding + chaining half of the path.
//
// Rows of SURVEY.md section 8(a) implemented here (U: = minimap2 2.26 unit reproduced; the reference
// reaches all of them through mm_map at /root/reference/src/lib.rs:482 and :587):
//   a1  k_sketch        U:sketch.c::mm_sketch            one lane per read (order-dependent window logic), ring in LDS
//   a2  k_mzflt         U:seed.c::mm_seed_mz_flt         one wave per read
//   a3  k_seed_lookup   U:index.c::mm_idx_get            one lane per minimizer, 128-B-line flat table in HBM
//   a4  k_seed_select   U:seed.c::mm_seed_select / mm_collect_matches   one wave per read
//   a5  k_seed_expand   U:map.c::collect_seed_hits       one lane per anchor
//   a6  k_sort_anchors  U:ksort.h::radix_sort_128x       one wave per read, literal in-place MSD radix (unstable!)
//   a7  k_chain         U:lchain.c::mg_lchain_dp         one wave per read, 64 predecessors per step, ballot/prefix-max
//   a8  k_backtrack     U:lchain.c::mg_chain_backtrack + compact_a    one wave per read
// These are integer/byte, HBM- and latency-bound kernels: no MFMA anywhere (BASELINE.json north_star).
// wave = 64 lanes; every workgroup below is exactly one wave unless stated.
#include <hip/hip_runtime.h>
#include "mm355_dev.h"

#define WAVE 64
#define LANE_LT_MASK(lane) ((lane) == 0? 0ULL : (~0ULL >> (64 - (lane))))

// ------------------------------------------------------------------ a1: sketch
// mm_sketch is a sequential state machine (ring of the last w k-mer records, rightmost minimum, run length l since
// the last ambiguous base).  Its state after base i is a pure function of a bounded suffix of the read, so a read is
// cut into chunks of SK_CHUNK bases and every chunk is sketched by its own lane:
//   * the lane starts `warm` bases before its chunk with an empty machine and PROVES, while running, that by the time
//     it reaches its first owned base the machine state equals the sequential one: both k-mers hold k real bases
//     (j*), then either an ambiguous base resets l on both sides or >= w+k k-mers were counted (every test on l is
//     saturated), then w ring writes happened (ring and rightmost minimum are then functions of exact records only);
//     if the proof is not complete at the chunk start the lane retries with a 4x longer warm-up (0 = read start, which
//     is exact by definition);
//   * a lane emits only minimizers whose own position lies in its chunk, and runs past the chunk end until w ring
//     writes lie beyond it (no record of the chunk can be emitted later); only the lane that reaches the end of the
//     read performs the final flush.
// mm_sketch emits minimizers in increasing position, so chunk outputs concatenated in chunk order are the sequential
// output.  k_sketch_compact then packs the chunk outputs of a read to the front of its slot range.
#include "mm355_sketch.h"
// kernels whose waves are serial dependence chains (one read or one chain segment per wave): raise their issue priority inside the
// SIMD so that they are not starved by the wide, throughput-bound extension grids of other contexts sharing the CU
#define MM355_LATENCY_KERNEL() __builtin_amdgcn_s_setprio(3)
// MM355_KPROF diagnostics: shader cycles of a kernel phase, summed over reads and the maximum over reads (lane 0 of a wave)
#define KPROF_BEGIN(bt) unsigned long long *kp_ = (bt).prof; unsigned long long kp_t_ = kp_? (unsigned long long)clock64() : 0
#define KPROF(i) do { if (kp_ && (threadIdx.x & 63) == 0) { const unsigned long long t_ = (unsigned long long)clock64(); atomicAdd(&kp_[i], t_ - kp_t_); atomicMax(&kp_[32 + (i)], t_ - kp_t_); kp_t_ = t_; } } while (0)
template <bool HPC>
__device__ __forceinline__ void sketch_lane_per_chunk(const DevIndex &ix, const DevBatch &bt, const DevSeeds &sd, const int32_t *chunk_read, const int32_t *chunk_start, int n_chunks, int32_t *chunk_n)
{
	MM355_LATENCY_KERNEL();
	extern __shared__ mm128 ring[];   // w entries per lane, lane-interleaved
	int t = blockIdx.x * WAVE + threadIdx.x;
	if (t >= n_chunks) return;
	const int r = chunk_read[t], cs = chunk_start[t];
	const int len = bt.rlen[r];
	const int64_t off = bt.roff[r];
	int ce = cs + SK_CHUNK;
	if (ce > len) ce = len;
	chunk_n[t] = sketch_chunk<HPC>(bt.seq + off, len, ix.w, ix.k, cs, ce, sd.mz + off + cs, ring + threadIdx.x, WAVE);
}
__global__ __launch_bounds__(WAVE) void k_sketch(DevIndex ix, DevBatch bt, DevSeeds sd, const int32_t *chunk_read, const int32_t *chunk_start, int n_chunks, int32_t *chunk_n)
{
	sketch_lane_per_chunk<false>(ix, bt, sd, chunk_read, chunk_start, n_chunks, chunk_n);
}
// the same on a homopolymer-compressed index (MM_I_HPC: map-pb / ava-pb)
__global__ __launch_bounds__(WAVE) void k_sketch_hpc(DevIndex ix, DevBatch bt, DevSeeds sd, const int32_t *chunk_read, const int32_t *chunk_start, int n_chunks, int32_t *chunk_n)
{
	sketch_lane_per_chunk<true>(ix, bt, sd, chunk_read, chunk_start, n_chunks, chunk_n);
}

// Small batches (a single read is ~20 chunks): lane 0 of a wave alone.  With a lane per chunk the wave executes the union of every lane's
// branches -- both ring rescans on almost every base -- which is what a full grid amortises and a 20-chunk grid does not.  A chunk is cut
// once more, into SKS_WAVES pieces of SK_CHUNK / SKS_WAVES bases with a wave each (a piece pays the ~45-base warm-up of the exactness
// proof again: 32 + 45 base steps on the critical path instead of 384 + 45); the pieces write at their own offsets of the chunk's slot
// range and wave 0 then packs them to its front, so the chunk looks as if one lane had done it.
#define SKS_WAVES 12
#define SKS_PIECE (SK_CHUNK / SKS_WAVES)
template <bool HPC>
__device__ __forceinline__ void sketch_block_per_chunk(const DevIndex &ix, const DevBatch &bt, const DevSeeds &sd, const int32_t *chunk_read, const int32_t *chunk_start, int n_chunks, int32_t *chunk_n)
{
	MM355_LATENCY_KERNEL();
	extern __shared__ mm128 ring[];              // w entries per wave
	__shared__ int s_n[SKS_WAVES];
	const int t = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
	if (t >= n_chunks) return;
	const int r = chunk_read[t], cs = chunk_start[t];
	const int len = bt.rlen[r];
	const int64_t off = bt.roff[r];
	int ce = cs + SK_CHUNK;
	if (ce > len) ce = len;
	KPROF_BEGIN(bt);
	if (lane == 0) {
		const int ps = cs + wv * SKS_PIECE;
		int pe = ps + SKS_PIECE;
		if (pe > ce) pe = ce;
		s_n[wv] = ps < pe? sketch_chunk<HPC>(bt.seq + off, len, ix.w, ix.k, ps, pe, sd.mz + off + ps, ring + wv * ix.w, 1) : 0;
	}
	KPROF(24);
	__syncthreads();
	if (wv != 0) return;
	mm128 *mz = sd.mz + off + cs;
	int m = s_n[0];
	for (int k = 1; k < SKS_WAVES; ++k) {        // a piece holds at most SKS_PIECE <= 64 minimizers: one wave-wide move, source read before the write
		const int n = s_n[k];
		mm128 v; v.x = v.y = 0;
		if (lane < n) v = mz[k * SKS_PIECE + lane];
		if (lane < n && m != k * SKS_PIECE) mz[m + lane] = v;
		m += n;
	}
	if (lane == 0) chunk_n[t] = m;
	KPROF(25);
}
__global__ __launch_bounds__(WAVE * SKS_WAVES) void k_sketch_sparse(DevIndex ix, DevBatch bt, DevSeeds sd, const int32_t *chunk_read, const int32_t *chunk_start, int n_chunks, int32_t *chunk_n)
{
	sketch_block_per_chunk<false>(ix, bt, sd, chunk_read, chunk_start, n_chunks, chunk_n);
}
__global__ __launch_bounds__(WAVE * SKS_WAVES) void k_sketch_sparse_hpc(DevIndex ix, DevBatch bt, DevSeeds sd, const int32_t *chunk_read, const int32_t *chunk_start, int n_chunks, int32_t *chunk_n)
{
	sketch_block_per_chunk<true>(ix, bt, sd, chunk_read, chunk_start, n_chunks, chunk_n);
}

// packs the per-chunk outputs of a read to the front of its slot range (in place; destination never passes the source)
__global__ __launch_bounds__(WAVE) void k_sketch_compact(DevBatch bt, DevSeeds sd, const int64_t *read_chunk0, const int32_t *chunk_n)
{
	const int r = blockIdx.x, lane = threadIdx.x;
	const int len = bt.rlen[r];
	const int64_t off = bt.roff[r], c0 = read_chunk0[r];
	const int nch = (len + SK_CHUNK - 1) / SK_CHUNK;
	mm128 *mz = sd.mz + off;
	int m = 0;
	for (int c = 0; c < nch; ++c) {
		const int n = chunk_n[c0 + c], src = c * SK_CHUNK;
		if (m != src) {
			for (int b = 0; b < n; b += WAVE) {
				mm128 v; v.x = v.y = 0;
				if (b + lane < n) v = mz[src + b + lane];
				__syncthreads();
				if (b + lane < n) mz[m + b + lane] = v;
				__syncthreads();
			}
		}
		m += n;
	}
	if (lane == 0) sd.n_mz[r] = m;
}

#include "mm355_wave.h"

// ------------------------------------------------------------------ a2: mm_seed_mz_flt
#define MZ_STAGE 2048
__global__ __launch_bounds__(WAVE) void k_mzflt(DevParams pr, DevBatch bt, DevSeeds sd)
{
	MM355_LATENCY_KERNEL();
	__shared__ SortLds L;
	__shared__ mm128 stage[MZ_STAGE];
	const int r = blockIdx.x, lane = threadIdx.x;
	const int n = sd.n_mz[r];
	const int q_occ_max = pr.mid_occ;
	if (n <= q_occ_max || pr.q_occ_frac <= 0.0f || q_occ_max <= 0) return;
	const int64_t off = bt.roff[r];
	mm128 *mz = sd.mz + off, *tmp = sd.mz_tmp + off;
	{   // cheap proof that nothing can be filtered: a count-sketch bucket holds at least every occurrence of a value,
		// so if no bucket exceeds q_occ_max no minimizer does (the common case: the exact sort below is skipped)
		uint32_t *cs = (uint32_t*)stage;   // 8192 counters
		for (int i = lane; i < 8192; i += WAVE) cs[i] = 0;
		__syncthreads();
		for (int i = lane; i < n; i += WAVE) atomicAdd(&cs[(uint32_t)(mm_table_hash(mz[i].x) >> 20) & 8191u], 1u);
		__syncthreads();
		uint32_t mx = 0;
		for (int i = lane; i < 8192; i += WAVE) mx = mx > cs[i]? mx : cs[i];
		for (int o2 = 32; o2 > 0; o2 >>= 1) { uint32_t y2 = __shfl_xor(mx, o2); mx = mx > y2? mx : y2; }
		__syncthreads();
		if (mx <= (uint32_t)q_occ_max) return;
	}
	for (int i = lane; i < n; i += WAVE) { tmp[i].x = mz[i].x; tmp[i].y = (uint64_t)i; }
	__syncthreads();
	wave_radix_sort(tmp, (uint32_t)n, mm_key_x(), &L, stage, (uint32_t)MZ_STAGE);
	__syncthreads();
	const float thr = (float)n * pr.q_occ_frac;
	for (int i = lane; i < n; i += WAVE) {
		if (i == 0 || tmp[i].x != tmp[i-1].x) {   // run start
			int e = i + 1;
			while (e < n && tmp[e].x == tmp[i].x) ++e;
			int cnt = e - i;
			if (cnt > q_occ_max && (float)cnt > thr)
				for (int j = i; j < e; ++j) mz[tmp[j].y].x = 0;
		}
	}
	__syncthreads();
	int m = 0;   // order-preserving in-place compaction
	for (int base = 0; base < n; base += WAVE) {
		int i = base + lane;
		mm128 v; v.x = 0; v.y = 0;
		if (i < n) v = mz[i];
		bool keep = i < n && v.x != 0;
		unsigned long long mask = __ballot(keep);
		__syncthreads();
		if (keep) mz[m + __popcll(mask & LANE_LT_MASK(lane))] = v;
		m += __popcll(mask);
		__syncthreads();
	}
	if (lane == 0) sd.n_mz[r] = m;
}

// ------------------------------------------------------------------ a3: seed lookup (the HBM-gather kernel)
// Algorithmic bytes per minimizer: 16 (minimizer read) + 16 (one slot, when present).
// A probe is ONE line fetch: the eight lanes of a group load the eight 16-B slots of the minimizer's 128-B line with one
// global_load_dwordx4 each -- a single coalesced 128-B request per minimizer -- and a ballot finds the matching slot.  Occupied slots
// form a prefix of a line (the builders fill the first empty slot and never delete), so: a match anywhere in the line = hit, else an
// empty slot = absent, else (line full, ~1 line in 12 at load 0.55) the next line.  A wave keeps LK_UNROLL x 8 independent line fetches
// in flight per iteration; nothing in the loop depends on an earlier fetch.
// Grid: the chunk table of the sketch (one block per SK_CHUNK bases of a read; a read's minimizers sit packed at the front of its slot range,
// so block (read, start) owns the minimizer slots [start, start + SK_CHUNK) below n_mz[read] and most blocks of a read leave at once).  Work
// is flat over the minimizers of the batch -- a 100 kb read is 260 blocks, not one block that loops 140 times -- and every 8-lane group
// has all its LK_PER_GROUP line fetches in flight before it looks at the first.  The roof of this access pattern, measured on the same chip
// (tools/ubench/linebench.hip, profiles/r03_random_line_roof.json): 5.9 TB/s of uniformly random 128-byte lines out of 16 GB = 46 G lookups/s;
// 64-byte buckets are served at the same request rate (3.1 TB/s), so a narrower bucket would not buy lookups.
#define LK_PER_GROUP (SK_CHUNK / 32)
// the table entries that hold minimizers: entry (read, start) is live when start < n_mz[read]; four fifths are not (a read has ~0.19
// minimizers per base).  A live entry becomes a 16-byte tile descriptor {read offset, start, n_mz}; one atomic per 1024 entries (a single
// word takes ~88 atomics per microsecond); the order of the list does not matter.
__global__ __launch_bounds__(1024) void k_lookup_tiles(DevBatch bt, DevSeeds sd, const int32_t *chunk_read, const int32_t *chunk_start, int n_chunks, int4 *tiles, unsigned int *n_tiles)
{
	__shared__ unsigned int s_cnt[16], s_base;
	const int ck = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	int r = 0, s0 = 0, n = 0;
	if (ck < n_chunks) { r = chunk_read[ck]; s0 = chunk_start[ck]; n = sd.n_mz[r]; }
	const bool live = s0 < n;
	const unsigned long long m = __ballot(live);
	if (lane == 0) s_cnt[w] = (unsigned int)__popcll(m);
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned int tot = 0;
		for (int i = 0; i < 16; ++i) { const unsigned int c = s_cnt[i]; s_cnt[i] = tot; tot += c; }
		s_base = tot? atomicAdd(n_tiles, tot) : 0;
	}
	__syncthreads();
	if (live) {
		const int64_t off = bt.roff[r];
		tiles[s_base + s_cnt[w] + __popcll(m & LANE_LT_MASK(lane))] = make_int4((int)(uint32_t)off, (int)(off >> 32), s0, n);
	}
}

typedef unsigned int lk_u32x4 __attribute__((ext_vector_type(4)));   // one 16-byte slot as ONE load: as a struct of four words the compiler
                                                                      // sinks the value half into the match branch -- a second round trip
struct LkTile { int64_t off; int32_t s0, n; uint64_t minier[LK_PER_GROUP]; };
__device__ __forceinline__ void lk_fetch_tile(LkTile &T, const DevSeeds &sd, const int4 d, int grp)
{
	T.off = (int64_t)((uint64_t)(uint32_t)d.x | (uint64_t)(uint32_t)d.y << 32); T.s0 = d.z; T.n = d.w;
	const mm128 *mz = sd.mz + T.off;
#pragma unroll
	for (int u = 0; u < LK_PER_GROUP; ++u) {
		const int j = T.s0 + u * 32 + grp;
		T.minier[u] = mz[j < T.n? j : T.n - 1].x;             // unconditional (a branch here costs a full wait per load); a live tile has s0 < n;
		                                                      // the raw word: nothing here may wait for the load
	}
}

// one probe of slot u's line: true when the line was full without the key (the caller reads the next line)
__device__ __forceinline__ bool lk_probe(const DevSeeds &sd, const LkTile &cu, int u, int grp, int sl, int gsh, const lk_u32x4 raw, unsigned int &hits, bool act = true)
{
	const int j = cu.s0 + u * 32 + grp;
	const bool live = act && cu.minier[u] != ~0ULL;        // (act: the groups of a wave that are not walking on must not count or store twice)
	const uint64_t key = (uint64_t)raw.y << 32 | raw.x, v = (uint64_t)raw.w << 32 | raw.z;
	const bool match = live && (key >> 1) == cu.minier[u] && key != UINT64_MAX;
	const bool empty = key == UINT64_MAX;
	const unsigned int mm = (unsigned int)(__ballot(match) >> gsh) & 0xffu, em = (unsigned int)(__ballot(empty) >> gsh) & 0xffu;
	if (match) {
		uint32_t cnt; uint64_t val;
		if (key & 1) cnt = 1, val = v;
		else cnt = (uint32_t)v, val = v >> 32;
		sd.sn[cu.off + j] = cnt; sd.sv[cu.off + j] = val;
		++hits;
	}
	if (mm) return false;
	if (em || !live) {
		if (live && sl == 0) { sd.sn[cu.off + j] = 0; sd.sv[cu.off + j] = 0; }
		return false;
	}
	return true;
}

// Persistent form: a block walks the live tiles b, b + G, b + 2G ...  Three tiles are in flight per block: the table lines of tile t, the
// minimizers of tile t + G and the descriptor of tile t + 2G, each a single round trip issued behind the older ones (the vector memory
// counter is in order, so waiting for the line fetches leaves the younger loads in flight).
__global__ __launch_bounds__(256) void k_seed_lookup(DevIndex ix, DevBatch bt, DevSeeds sd, const int4 *__restrict__ tiles, const unsigned int *__restrict__ n_tiles, unsigned long long *hit_ctr)
{
	const int grp = threadIdx.x >> 3, sl = threadIdx.x & 7;          // 32 groups of 8 lanes per block; lane sl owns slot sl of the line
	const int gsh = (threadIdx.x & 63) & ~7;                          // bit position of this group's 8 lanes in a wave ballot
	const int n_act = (int)*n_tiles, G = gridDim.x;
	unsigned int hits = 0;
	int t = blockIdx.x;
	if (t >= n_act) return;                                           // (no barrier below is reached by part of a block: t is uniform)
	LkTile nx;
	lk_fetch_tile(nx, sd, tiles[t], grp);
	int4 dn = tiles[min(t + G, n_act - 1)];
	while (t < n_act) {
		LkTile cu = nx;
		uint64_t line[LK_PER_GROUP]; lk_u32x4 raw[LK_PER_GROUP];
#pragma unroll
		for (int u = 0; u < LK_PER_GROUP; ++u) {
			const bool in = cu.s0 + u * 32 + grp < cu.n;
			const uint64_t h = mm_table_hash(cu.minier[u] >> 8) & ix.line_mask;
			cu.minier[u] = in? cu.minier[u] >> 8 : ~0ULL;             // ~0: no minimizer in this slot
			line[u] = in? h : 0;                                      // (slots past the end read line 0: cached)
		}
#pragma unroll
		for (int u = 0; u < LK_PER_GROUP; ++u)                        // every line fetch of the group issued back to back, no branch between them
			raw[u] = *(const lk_u32x4*)(ix.slots + line[u] * MM355_SLOTS_PER_LINE + sl);
		t += G;
		lk_fetch_tile(nx, sd, dn, grp);                               // behind the line fetches in the memory queue; unconditional, so that the
		dn = tiles[min(t + G, n_act - 1)];                            // waits below are by count (past the end: the last tile again, unused)
		unsigned int pend = 0;                                        // bit u: the first line was full and did not hold the key
#pragma unroll
		for (int u = 0; u < LK_PER_GROUP; ++u)                        // first probe: straight-line code, so that each line is waited for by count
			pend |= (unsigned int)lk_probe(sd, cu, u, grp, sl, gsh, raw[u], hits) << u;
		if (__ballot(pend != 0) != 0) {                               // rare (a table at most 55 % full): the groups concerned walk on alone
#pragma unroll
			for (int u = 0; u < LK_PER_GROUP; ++u) {
				bool more = pend >> u & 1;
				while (__ballot(more) != 0) {
					if (more) line[u] = (line[u] + 1) & ix.line_mask;
					const lk_u32x4 r2 = *(const lk_u32x4*)(ix.slots + line[u] * MM355_SLOTS_PER_LINE + sl);
					const bool again = lk_probe(sd, cu, u, grp, sl, gsh, r2, hits, more);
					more = more && again;
				}
			}
		}
	}
	// one atomic per block, spread over 64 words (one word takes ~88 atomics per microsecond)
	__shared__ unsigned int s_hits[4];
	for (int o = 32; o > 0; o >>= 1) hits += __shfl_down(hits, o);
	if ((threadIdx.x & 63) == 0) s_hits[threadIdx.x >> 6] = hits;
	__syncthreads();
	if (threadIdx.x == 0) { const unsigned int h = s_hits[0] + s_hits[1] + s_hits[2] + s_hits[3]; if (h) atomicAdd(hit_ctr + (blockIdx.x & 63), (unsigned long long)h); }
}

// ------------------------------------------------------------------ a4: mm_seed_select + mm_collect_matches
__device__ inline void heapdown_u64(uint32_t i, uint32_t n, uint64_t *l)
{
	uint32_t k = i;
	uint64_t tmp = l[i];
	while ((k = (k << 1) + 1) < n) {
		if (k != n - 1 && l[k] < l[k+1]) ++k;
		if (l[k] < tmp) break;
		l[i] = l[k]; i = k;
	}
	l[i] = tmp;
}

// U:map.c::skip_seed (qname is NULL through the reference, so only the strand filters remain): 1 = the hit becomes an anchor
__device__ __forceinline__ uint32_t mm355_keep_strand(int64_t flag, bool forward)
{
	return forward? !(flag & MMF_REV_ONLY) : !(flag & MMF_FOR_ONLY);
}

// One wave per read; nothing is done lane by lane except the heap of U:seed.c::mm_seed_select on streaks that are LONGER than the
// number of high-occurrence seeds they may keep (rare):
//   1. hit list, order preserving, as dense per-hit arrays (occurrence count, query word, index of the minimizer) + a bit mask of the
//      high-occurrence hits (c > mid_occ) in LDS, one 64-bit word per tile of 64 hits;
//   2. mm_seed_select walks the STREAKS of high-occurrence hits (maximal runs of set bits, found with scalar bit scans on the LDS mask):
//      a streak that may keep nothing is filtered, one that may keep all of its hits is kept, by all lanes; only a streak longer than
//      its quota runs the reference's max-heap selection (lane 0, the streak's dense counts);
//   3. mm_collect_matches' tail as wave scans over tiles of 64 hits: kept list (ballot compaction), anchor offsets (prefix sum of the
//      occurrence counts), mini_pos, and rep_len as "every filtered seed that starts a new covered run adds prev_end - start" with the
//      last end added once (the union length of the filtered seeds' query intervals, accumulated exactly like the sequential loop).
// The dense arrays alias the outputs (soff <- counts, mini_pos <- query words, hl <- minimizer index): a tile is read completely before
// its (never more numerous) kept entries are written at or below its own range.
#define SEL_MASK_TILES 1024          // tiles of 64 hits whose mask is kept in LDS (reads up to ~0.6 Mb); later tiles are recomputed on the fly
__global__ __launch_bounds__(WAVE) void k_seed_select(DevIndex ix, DevParams pr, DevBatch bt, DevSeeds sd)
{
	MM355_LATENCY_KERNEL();
	__shared__ uint64_t heap[128];
	__shared__ unsigned long long hmask[SEL_MASK_TILES];
	const int r = blockIdx.x, lane = threadIdx.x;
	const int n = sd.n_mz[r], qlen = bt.rlen[r];
	const int64_t off = bt.roff[r];
	const mm128 *mz = sd.mz + off;
	const uint32_t *sn = sd.sn + off;
	uint8_t *hflt = sd.sflt + off;                 // per HIT ordinal: 1 = filtered
	int32_t *hl = sd.hl + off;                     // per hit ordinal: index of the minimizer; rewritten as the kept list
	uint32_t *hc = sd.soff + off;                  // per hit ordinal: occurrence count; rewritten as the anchor offsets of the kept seeds
	uint64_t *hy = sd.mini_pos + off;              // per hit ordinal: q_span << 32 | (q_pos << 1 | strand); rewritten as mini_pos
	const int max_occ = pr.mid_occ;
	int n_m0 = 0, n_high = 0;
	KPROF_BEGIN(bt);
	for (int base = 0; base < n; base += WAVE) {   // 1. hit list, order preserving
		const int j = base + lane;
		const uint32_t c = j < n? sn[j] : 0;
		const bool hit = c > 0;
		const unsigned long long mask = __ballot(hit);
		if (hit) {
			const int e = n_m0 + __popcll(mask & LANE_LT_MASK(lane));
			const mm128 m = mz[j];
			hl[e] = j; hc[e] = c; hy[e] = (m.x & 0xff) << 32 | (uint32_t)m.y; hflt[e] = 0;
		}
		n_m0 += __popcll(mask);
	}
	__syncthreads();
	const int n_tiles = (n_m0 + WAVE - 1) / WAVE;
	for (int t = 0; t < n_tiles; ++t) {            // high-occurrence mask (dense order)
		const int i = t * WAVE + lane;
		const bool high = i < n_m0 && hc[i] > (uint32_t)max_occ;
		const unsigned long long m = __ballot(high);
		if (lane == 0 && t < SEL_MASK_TILES) hmask[t] = m;
		n_high += __popcll(m);
	}
	__syncthreads();
	KPROF(4);
	if (pr.occ_dist > 0 && pr.max_max_occ > max_occ) {   // 2. mm_seed_select
		if (n_m0 >= 2 && n_high > 0) {
			const int dist = pr.occ_dist;
			int pos = 0;                           // scan position (hit ordinal), wave-uniform
			auto tile_mask = [&](int t) -> unsigned long long {   // wave-uniform calls only
				if (t < SEL_MASK_TILES) return hmask[t];
				const int i = t * WAVE + lane;
				return __ballot(i < n_m0 && hc[i] > (uint32_t)max_occ);
			};
			while (pos < n_m0) {
				// next streak [st, en): first set bit at or after pos, then the first clear bit after it
				int t = pos >> 6;
				unsigned long long w = tile_mask(t) & (~0ULL << (pos & 63));
				while (w == 0 && ++t < n_tiles) w = tile_mask(t);
				if (w == 0) break;
				const int st = t * 64 + __builtin_ctzll(w);
				unsigned long long z = ~tile_mask(t) & (~0ULL << (st & 63));
				while (z == 0 && ++t < n_tiles) z = ~tile_mask(t);
				int en = z? t * 64 + __builtin_ctzll(z) : n_m0;
				if (en > n_m0) en = n_m0;
				pos = en;
				const int32_t ps = st == 0? 0 : (int32_t)((uint32_t)hy[st - 1] >> 1);
				const int32_t pe = en == n_m0? qlen : (int32_t)((uint32_t)hy[en] >> 1);
				int32_t max_high_occ = (int32_t)((double)(pe - ps) / dist + .499);
				const int L = en - st;
				if (max_high_occ > 128) max_high_occ = 128;
				if (max_high_occ <= 0) {              // nothing may stay: flt = 0 ^ 1
					for (int i = st + lane; i < en; i += WAVE) hflt[i] = 1;
				} else if (L <= max_high_occ) {        // the heap would hold the whole streak: flt = 1 ^ 1, then the max_max_occ cut
					for (int i = st + lane; i < en; i += WAVE) hflt[i] = hc[i] > (uint32_t)pr.max_max_occ? 1 : 0;
				} else {                               // the reference's selection, literally (max-heap of the max_high_occ smallest)
					if (lane == 0) {
						int j, k;
						for (j = st, k = 0; j < en && k < max_high_occ; ++j, ++k) heap[k] = (uint64_t)hc[j] << 32 | (uint32_t)j;
						for (uint32_t h = ((uint32_t)k >> 1) - 1; h != (uint32_t)-1; --h) heapdown_u64(h, (uint32_t)k, heap);
						for (; j < en; ++j) {
							if ((int32_t)hc[j] < (int32_t)(heap[0] >> 32)) {
								heap[0] = (uint64_t)hc[j] << 32 | (uint32_t)j;
								heapdown_u64(0, (uint32_t)k, heap);
							}
						}
						for (j = st; j < en; ++j) hflt[j] = 1;
						for (j = 0; j < k; ++j) hflt[(uint32_t)heap[j]] = 0;
						for (j = st; j < en; ++j) if (hc[j] > (uint32_t)pr.max_max_occ) hflt[j] = 1;
					}
				}
				__syncthreads();
			}
		}
	} else {
		for (int i = lane; i < n_m0; i += WAVE) if (hc[i] > (uint32_t)max_occ) hflt[i] = 1;
	}
	__syncthreads();
	KPROF(5);
	// 3. mm_collect_matches tail
	const bool strand_flt = (pr.flag & (MMF_FOR_ONLY | MMF_REV_ONLY)) != 0;
	int n_kept = 0;
	uint32_t n_a = 0;
	unsigned long long multi = 0;
	long long rep_acc = 0;                         // sum over run starts of (previous end - start)
	int32_t last_en = 0; bool any_flt = false;     // end of the last filtered seed so far
	for (int t = 0; t < n_tiles; ++t) {
		const int i = t * WAVE + lane;
		const bool in = i < n_m0;
		uint32_t c = 0; uint64_t yw = 0; int j = 0; bool flt = false;
		if (in) { c = hc[i]; yw = hy[i]; j = hl[i]; flt = hflt[i] != 0; }
		const uint32_t q_pos = (uint32_t)yw, q_span = (uint32_t)(yw >> 32) & 0xff;
		const int32_t en_i = (int32_t)(q_pos >> 1) + 1, st_i = en_i - (int32_t)q_span;
		// rep_len: previous filtered seed's end (inside the tile: the nearest lower filtered lane; else the carry)
		const unsigned long long fm = __ballot(in && flt);
		if (fm) {
			const unsigned long long lower = fm & LANE_LT_MASK(lane);
			const int src = lower? 63 - __builtin_clzll(lower) : lane;
			int32_t prev_en = __shfl(en_i, src);
			if (!lower) prev_en = any_flt? last_en : 0;
			long long contrib = 0;
			if (in && flt && st_i > prev_en) contrib = (long long)prev_en - st_i;
			for (int o2 = 32; o2 > 0; o2 >>= 1) contrib += __shfl_xor(contrib, o2);
			rep_acc += contrib;
			last_en = __shfl(en_i, 63 - __builtin_clzll(fm)); any_flt = true;
		}
		// kept seeds
		const bool keep = in && !flt;
		uint32_t c_eff = keep? c : 0;
		if (keep && strand_flt) {                  // U:map.c::skip_seed: hits of the excluded strand produce no anchor
			const uint64_t v = sd.sv[off + j];
			c_eff = 0;
			for (uint32_t kq = 0; kq < c; ++kq) {
				const uint64_t rk = c == 1? v : ix.pos[v + kq];
				c_eff += mm355_keep_strand(pr.flag, (rk & 1) == (q_pos & 1));
			}
		}
		const unsigned long long km = __ballot(keep);
		const uint32_t incl = (uint32_t)wave_incl_scan_add((int32_t)c_eff);
		const uint32_t tile_sum = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		unsigned long long mc = keep && c > 1? (unsigned long long)c : 0ULL;
		for (int o2 = 32; o2 > 0; o2 >>= 1) mc += __shfl_xor(mc, o2);
		multi += mc;
		__syncthreads();                           // the whole tile has been read: its kept entries may now overwrite it
		if (keep) {
			const int e = n_kept + __popcll(km & LANE_LT_MASK(lane));
			hl[e] = j; hc[e] = n_a + incl - c_eff; hy[e] = (uint64_t)q_span << 32 | q_pos >> 1;
		}
		n_kept += __popcll(km); n_a += tile_sum;
		__syncthreads();
	}
	if (lane == 0) {
		sd.n_a[r] = (int32_t)n_a; sd.rep_len[r] = (int32_t)(rep_acc + (any_flt? last_en : 0)); sd.n_mini[r] = n_kept;
		if (multi) atomicAdd(&sd.counters[1], multi);
	}
	KPROF(6);
}

// ------------------------------------------------------------------ a5: collect_seed_hits (anchor expansion)
// One lane per anchor, 256-thread block per read.  The kept seeds of the read are staged through LDS in tiles of EX_TILE (prefix of
// their hit counts, occurrence count, pos[] offset / singleton word, query word, tandem flag), so that finding the seed of anchor t
// (binary search over the prefix) and reading its record never leave the CU; HBM sees the pos[] gather (8 B per multi-occurrence
// anchor, consecutive lanes read consecutive entries of a run) and the coalesced 16-B anchor stores.
#define EX_TILE 1024
__global__ __launch_bounds__(256) void k_seed_expand(DevIndex ix, DevParams pr, DevBatch bt, DevSeeds sd, DevAnchors an)
{
	__shared__ uint32_t s_off[EX_TILE + 1], s_c[EX_TILE];
	__shared__ uint64_t s_v[EX_TILE], s_y[EX_TILE];      // s_y: q_span << 32 | q_pos word (strand in bit 0) | tandem flag in bit 63
	const int r = blockIdx.x;
	const int na = sd.n_a[r], nk = sd.n_mini[r], nmz = sd.n_mz[r], qlen = bt.rlen[r];
	if (na == 0) return;
	const int64_t off = bt.roff[r];
	const mm128 *mz = sd.mz + off;
	const int32_t *hl = sd.hl + off;
	const uint32_t *soff = sd.soff + off;
	mm128 *a = an.a + an.aoff[r];
	const bool strand_flt = (pr.flag & (MMF_FOR_ONLY | MMF_REV_ONLY)) != 0;
	for (int k0 = 0; k0 < nk; k0 += EX_TILE) {
		const int kn = nk - k0 < EX_TILE? nk - k0 : EX_TILE;
		__syncthreads();
		for (int i = threadIdx.x; i < kn; i += 256) {
			const int j = hl[k0 + i];
			const mm128 m = mz[j];
			const bool tandem = (j > 0 && (mz[j-1].x >> 8) == (m.x >> 8)) || (j < nmz - 1 && (mz[j+1].x >> 8) == (m.x >> 8));
			s_off[i] = soff[k0 + i]; s_c[i] = sd.sn[off + j]; s_v[i] = sd.sv[off + j];
			s_y[i] = (m.x & 0xff) << 32 | (uint32_t)m.y | (tandem? 1ULL << 63 : 0);
		}
		if (threadIdx.x == 0) s_off[kn] = k0 + kn < nk? soff[k0 + kn] : (uint32_t)na;
		__syncthreads();
		const int t0 = (int)s_off[0], t1 = (int)s_off[kn];
		for (int t = t0 + (int)threadIdx.x; t < t1; t += 256) {
			int lo = 0, hi = kn;
			while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= (uint32_t)t) lo = mid; else hi = mid; }
			const uint32_t kq = (uint32_t)t - s_off[lo];
			const uint32_t c = s_c[lo];
			const uint64_t v = s_v[lo], yw = s_y[lo];
			const uint32_t q_pos = (uint32_t)yw, q_span = (uint32_t)(yw >> 32) & 0xff;
			uint64_t rk = c == 1? v : ix.pos[v + kq];
			if (strand_flt && c > 1) {   // kq counts the hits of the admitted strand only (k_seed_select)
				uint32_t seen = 0;
				for (uint32_t kk = 0; kk < c; ++kk) {
					rk = ix.pos[v + kk];
					if (mm355_keep_strand(pr.flag, (rk & 1) == (q_pos & 1)) && seen++ == kq) break;
				}
			}
			const uint32_t rpos = (uint32_t)rk >> 1;
			mm128 o;
			if ((rk & 1) == (q_pos & 1)) {   // forward strand
				o.x = (rk & 0xffffffff00000000ULL) | rpos;
				o.y = (uint64_t)q_span << 32 | q_pos >> 1;
			} else {                          // reverse strand
				o.x = 1ULL << 63 | (rk & 0xffffffff00000000ULL) | rpos;
				o.y = (uint64_t)q_span << 32 | (uint32_t)(qlen - (int)((q_pos >> 1) + 1 - q_span) - 1);
			}
			if (yw >> 63) o.y |= MM355_SEED_TANDEM;
			a[t] = o;
		}
	}
}

// ------------------------------------------------------------------ a6: radix_sort_128x on the anchors
#define A_STAGE 2048
__global__ __launch_bounds__(WAVE) void k_sort_anchors(DevBatch bt, DevAnchors an, int *err, const int32_t *heavy_first)
{
	MM355_LATENCY_KERNEL();
	__shared__ SortLds L;
	__shared__ mm128 stage[A_STAGE];
	const int r = heavy_first[blockIdx.x];
	const int64_t o = an.aoff[r];
	const uint32_t n = (uint32_t)(an.aoff[r+1] - o);
	KPROF_BEGIN(bt);
	WalkScratch ws; ws.out = an.b + o; ws.fpos = (uint32_t*)an.f + o; ws.rank = (uint32_t*)an.p + o; ws.flab = an.t8 + o;   // all free before chaining
	ws.tcnt = an.tcnt? an.tcnt + o : 0;
	wave_radix_sort(an.a + o, n, mm_key_x(), &L, stage, (uint32_t)A_STAGE, &ws);
	if (threadIdx.x == 0 && n > MM355_RS_MIN_SIZE && L.overflow) *err = 1;
	KPROF(8);
}

// ---- heavy reads: the top radix levels of an array with tens/hundreds of thousands of elements are done by a whole
// 1024-thread block (k_sort_level_mw: same label-walk level as wave_rs_level_walk, block-wide ordered prefix), which
// emits the resulting buckets as independent tasks; tasks that are still large go through another block level, the rest
// is finished by one wave each (k_sort_tasks).  The latency of the slowest read drops from O(n) wave-serial steps per
// level to O(n/1024) + the 1-byte label walk.
#define MW_NT 1024
#ifndef MW_STK
#define MW_STK 128                   // entries of a big-class block's own stack of big buckets (depth first inside the block)
#endif
#define MW_LAB_CAP 122880   // 120 KB of labels in LDS
#define MW_BIG 16384        // buckets larger than this take a 1024-thread level
#define MW_MED 2048         // ... larger than this a 256-thread level; smaller ones are finished by one wave in LDS

int mm355_sort_heavy_threshold(void);
int mm355_sort_medium_threshold(void);
template <typename T> struct SortArr;   // per-read base pointers of the array being sorted and of its scratch
template <> struct SortArr<mm128> {
	__device__ static mm128 *arr(const DevAnchors &an, int64_t o) { return an.a + o; }
	__device__ static WalkScratch ws(const DevAnchors &an, int64_t o) { WalkScratch w; w.out = an.b + o; w.fpos = (uint32_t*)an.f + o; w.rank = (uint32_t*)an.p + o; w.flab = an.t8 + o; w.tcnt = an.tcnt? an.tcnt + o : 0; return w; }
};
template <> struct SortArr<uint64_t> {   // z[] of the backtrack: v[] and vi[] are free at that point
	__device__ static uint64_t *arr(const DevAnchors &an, int64_t o) { return an.z + o; }
	__device__ static WalkScratch ws(const DevAnchors &an, int64_t o) { WalkScratch w; w.out = an.u2 + o; w.fpos = (uint32_t*)an.vi + o; w.rank = (uint32_t*)an.v + o; w.flab = an.t8 + o; w.tcnt = 0; return w; }
};

template <int NT> struct MwLds {
	uint32_t cnt[256], bb[256], be[256], cur[256], fend[256], arr[256], abef[256], fst[256];
	uint32_t wtot[NT / WAVE];
	uint32_t nfor, single;
};

// block-wide exclusive count of `flag` over the threads of this iteration, in thread order; returns the offset and adds the total to *base
template <int NT>
__device__ inline uint32_t block_ordered_prefix(bool flag, uint32_t &base, MwLds<NT> *L)
{
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const unsigned long long mask = __ballot(flag);
	if (lane == 0) L->wtot[wv] = (uint32_t)__popcll(mask);
	__syncthreads();
	uint32_t before = 0, tot = 0;
	for (uint32_t w2 = 0; w2 < NT / WAVE; ++w2) { const uint32_t c = L->wtot[w2]; if (w2 < wv) before += c; tot += c; }
	const uint32_t off = base + before + (uint32_t)__popcll(mask & LANE_LT_MASK(lane));
	base += tot;
	__syncthreads();
	return off;
}

// One block per task (a bucket of one read at byte shift s).  NT threads, LABCAP label bytes in LDS: <1024, 120 KB> for buckets of more
// than MW_BIG elements, <256, MW_BIG + 64 B> for the medium ones (six blocks per CU instead of one).  Children go to the list of their
// own size class; ctr = { next-level big, next-level medium, wave tasks (running total) }.
template <typename T, typename Key, int NT, int LABCAP>
__device__ inline void mw_level_task(MwLds<NT> &L, uint8_t *lds_lab, const DevAnchors &an, const SortTask tk, SortTask *out_big, SortTask *out_med, SortTask *out_small,
                                     unsigned int *ctr, unsigned int *ctr_small, uint32_t big_min, uint32_t med_min, SortTask *stk = 0, unsigned int *stk_n = 0, uint32_t stk_cap = 0)
{
	const int64_t o = an.aoff[tk.read];
	T *a = SortArr<T>::arr(an, o);
	const WalkScratch ws = SortArr<T>::ws(an, o);
	Key key;
	const uint32_t tid = threadIdx.x, beg = tk.beg, end = tk.end, tot = end - beg;
	int s = tk.s;
	// skip the levels where every key has the same byte (identity permutation)
	for (;;) {
		for (uint32_t i = tid; i < 256; i += NT) L.cnt[i] = 0;
		if (tid == 0) L.single = 0;
		__syncthreads();
		for (uint32_t i = beg + tid; i < end; i += NT) atomicAdd(&L.cnt[(uint32_t)(key(a[i]) >> s) & 255u], 1u);
		__syncthreads();
		if (tid < 256 && L.cnt[tid] == tot) L.single = 1;
		__syncthreads();
		if (!L.single) break;
		if (s == 0) return;          // all keys equal: nothing moves
		s -= 8;
		__syncthreads();
	}
	if (tid == 0) { uint32_t acc = 0; for (int k = 0; k < 256; ++k) { L.bb[k] = acc; acc += L.cnt[k]; L.be[k] = acc; } }
	__syncthreads();
	T *out = (T*)ws.out + beg;
	uint32_t *fpos = ws.fpos + beg, *rank = ws.rank + beg;
	uint8_t *flab = ws.flab + beg;
	uint32_t nfor = 0;
	for (uint32_t base = 0; base < tot; base += NT) {
		const uint32_t rel = base + tid;
		uint32_t g = 0; bool foreign = false;
		if (rel < tot) { g = (uint32_t)(key(a[beg + rel]) >> s) & 255u; foreign = !(rel >= L.bb[g] && rel < L.be[g]); }
		const uint32_t e = block_ordered_prefix<NT>(foreign, nfor, &L);
		if (foreign) { fpos[e] = rel; flab[e] = (uint8_t)g; if (e < (uint32_t)LABCAP) lds_lab[e] = (uint8_t)g; }
	}
	__syncthreads();
	if (tid < 256) {
		uint32_t lo = 0, hi = nfor; const uint32_t target = L.bb[tid];
		while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (fpos[mid] < target) lo = mid + 1; else hi = mid; }
		L.cur[tid] = lo; L.fst[tid] = lo; L.arr[tid] = 0;
	}
	__syncthreads();
	if (tid < 256) L.fend[tid] = tid < 255? L.fst[tid + 1] : nfor;
	__syncthreads();
	// Two non-empty buckets k1 < k2 (the strand level of every read that maps to both strands -- the level with the most elements):
	// every cycle is k1 -> k2 -> k1, so the i-th foreign element of either region is the i-th arrival at the other bucket, all arrivals
	// at k2 precede its turn and none at k1 do.  No walk.
	if (tid < 256) { const unsigned long long ne = __ballot(L.cnt[tid] != 0); if ((tid & 63) == 0) L.wtot[tid >> 6] = (uint32_t)__popcll(ne); }
	__syncthreads();
	const bool two = L.wtot[0] + L.wtot[1] + L.wtot[2] + L.wtot[3] == 2;
	__syncthreads();
	if (two) {
		const uint32_t m = nfor >> 1;   // foreign elements per region
		for (uint32_t e = tid; e < nfor; e += NT) rank[e] = e < m? e : e - m;
		if (tid < 256) L.abef[tid] = (L.cnt[tid] != 0 && L.bb[tid] != 0)? m : 0;   // bb != 0: the second of the two regions
	} else if (nfor < (uint32_t)LABCAP) {
		if (tid < 256) { const uint32_t f0 = L.fst[tid]; L.cur[tid] = f0 << 8 | lds_lab[f0]; }
		__syncthreads();
		if (tid == 0) rs_walk_packed(L.cur, L.fend, L.arr, L.abef, lds_lab, rank);
	} else if (tid == 0) {           // labels do not fit the LDS: walk them in HBM
		for (uint32_t k = 0; k < 256; ++k) {
			L.abef[k] = L.arr[k];
			while (L.cur[k] < L.fend[k]) {
				uint32_t c = k;
				do {
					const uint32_t e = L.cur[c]++;
					const uint32_t g = flab[e];
					rank[e] = L.arr[g]++;
					c = g;
				} while (c != k);
			}
		}
	}
	__syncthreads();
	uint32_t nfb = 0;
	for (uint32_t base = 0; base < tot; base += NT) {
		const uint32_t rel = base + tid;
		uint32_t g = 0; bool foreign = false; T el;
		if (rel < tot) { el = a[beg + rel]; g = (uint32_t)(key(el) >> s) & 255u; foreign = !(rel >= L.bb[g] && rel < L.be[g]); }
		const uint32_t pre = block_ordered_prefix<NT>(foreign, nfb, &L);
		if (rel < tot) {
			uint32_t dest;
			const uint32_t al = L.abef[g], f0 = L.fst[g];
			if (foreign) { const uint32_t r = rank[pre]; dest = r < al? (r == 0? L.bb[g] : fpos[f0 + r - 1] + 1) : fpos[f0 + r]; }
			else dest = rel + ((pre - f0) < al? 1u : 0u);
			out[dest] = el;
		}
	}
	__syncthreads();
	for (uint32_t i = tid; i < tot; i += NT) a[beg + i] = out[i];
	__syncthreads();
	if (s > 0 && tid < 256) {   // children: big / medium -> another block level, small -> one wave each, <= 64 -> insertion sort right here
		const uint32_t b0 = L.bb[tid], sz = L.cnt[tid];
		if (sz > 1 && !ws_has_tie(ws.tcnt, beg + b0, beg + b0 + sz)) { /* unique content, restored by the caller */ }
		else if (sz > MM355_RS_MIN_SIZE) {
			SortTask c; c.read = tk.read; c.beg = beg + b0; c.end = beg + b0 + sz; c.s = s - 8;
			if (sz > big_min) {   // the block's own stack first (it goes on with this read's big buckets itself), the next level's list when that is full
				unsigned int k = stk? atomicAdd(stk_n, 1u) : stk_cap;
				if (k < stk_cap) stk[k] = c;
				else { if (stk) atomicSub(stk_n, 1u); out_big[atomicAdd(&ctr[0], 1u)] = c; }
			}
			else if (sz > med_min) out_med[atomicAdd(&ctr[1], 1u)] = c;
			else out_small[atomicAdd(ctr_small, 1u)] = c;
		} else if (sz > 1) mm_rs_insertsort(a + beg + b0, a + beg + b0 + sz, key);
	}
}
// The list length stays on the device (no host round trip per level) and the grid is a bound on it: a block takes the tasks blockIdx.x,
// blockIdx.x + gridDim.x, ...  A level that turns out empty costs a few dozen blocks that leave at once -- not hundreds of 120-KB-LDS
// blocks that each wait for a whole free CU beside the other contexts' kernels.
template <typename T, typename Key, int NT, int LABCAP>
__global__ __launch_bounds__(NT) void k_sort_level_mw(DevAnchors an, const SortTask *tasks, const unsigned int *n_tasks_p, SortTask *out_big, SortTask *out_med, SortTask *out_small,
                                                      unsigned int *ctr, unsigned int *ctr_small, uint32_t big_min, uint32_t med_min, int *err, SortTask *stacks)
{
	__shared__ MwLds<NT> L;
	__shared__ unsigned int s_stk_n;
	extern __shared__ uint8_t lds_lab[];   // LABCAP labels
	const unsigned int n_tasks = *n_tasks_p;
	// stacks != 0 (the big class): a block keeps the big buckets its task produces on a stack of its own (MW_STK entries in HBM) and works them
	// off itself, depth first, before it takes another task: the 1024-thread / 120-KB block got its CU once, instead of once per level beside the
	// other contexts' kernels.  Medium and small buckets go to the lists as before; a full stack spills to the next level's list.
	SortTask *stk = stacks? stacks + (size_t)blockIdx.x * MW_STK : 0;
	for (unsigned int t = blockIdx.x; t < n_tasks; t += gridDim.x) {
		if (threadIdx.x == 0) s_stk_n = 0;
		__syncthreads();
		mw_level_task<T, Key, NT, LABCAP>(L, lds_lab, an, tasks[t], out_big, out_med, out_small, ctr, ctr_small, big_min, med_min, stk, &s_stk_n, stk? MW_STK : 0);
		for (;;) {
			__threadfence_block();
			__syncthreads();
			const unsigned int n = s_stk_n;
			if (n == 0) break;
			const SortTask tk = stk[n - 1];
			__syncthreads();
			if (threadIdx.x == 0) s_stk_n = n - 1;
			__syncthreads();
			mw_level_task<T, Key, NT, LABCAP>(L, lds_lab, an, tk, out_big, out_med, out_small, ctr, ctr_small, big_min, med_min, stk, &s_stk_n, MW_STK);
		}
	}
	(void)err;
}

// one wave per task: finishes a bucket exactly as rs_sort(beg, end, 8, s) would
template <typename T, typename Key>
__global__ __launch_bounds__(WAVE) void k_sort_tasks(DevAnchors an, const SortTask *tasks, const unsigned int *n_tasks_p, int *err)
{
	__shared__ SortLds L;
	__shared__ mm128 stage[2048];
	const unsigned int n_tasks = *n_tasks_p;
	for (unsigned int t = blockIdx.x; t < n_tasks; t += gridDim.x) {   // (the grid is a bound on the list length, which stays on the device)
		const SortTask tk = tasks[t];
		const int64_t o = an.aoff[tk.read];
		if (tk.end - tk.beg <= MM355_RS_MIN_SIZE) {   // only a whole array can be this short (children are larger): radix_sort = insertion sort
			wave_rank_sort_small(SortArr<T>::arr(an, o) + tk.beg, tk.end - tk.beg, Key());
		} else {
			WalkScratch ws = SortArr<T>::ws(an, o);
			ws.out = (T*)ws.out + tk.beg; ws.fpos += tk.beg; ws.rank += tk.beg; ws.flab += tk.beg; if (ws.tcnt) ws.tcnt += tk.beg;
			if (threadIdx.x == 0) { L.stk_n = 0; L.overflow = 0; }
			__syncthreads();
			wave_rs_core<true>(SortArr<T>::arr(an, o) + tk.beg, tk.end - tk.beg, tk.s, Key(), &L, (T*)stage, (uint32_t)(sizeof(stage) / sizeof(T)), &ws);
			if (threadIdx.x == 0 && L.overflow) *err = 1;
		}
		__syncthreads();
	}
}

// host side: the initial tasks (whole arrays of the reads to sort, at byte 56) are already in the three size-class lists on the device.
// A level = one launch per block-level class; the lengths of the lists a level leaves behind stay ON THE DEVICE (ctr[level + 1][big, medium],
// one running total for the wave tasks) and the next level's grids are upper bounds -- disjoint buckets of more than big_min / med_min
// elements out of n_elems -- whose surplus blocks leave at once: the whole emulation is one asynchronous chain of launches (round 3 read the
// counters back after every level: up to nine host round trips per sub-batch, each a few ms under the bench load).
#define MW_MED_LAB (MW_BIG + 64)
#define MW_LEVELS 9
#define MW_STACK_BLOCKS 4096          // most blocks of a big-class launch that get a stack (grids are capped below this)
size_t mm355_sort_buf_bytes(size_t task_cap);
#define MW_STK 128                   // entries of a big-class block's own stack of big buckets
template <typename T, typename Key>
static int sort_tasks_run(DevAnchors &an, SortTask *d_big[2], SortTask *d_med[2], SortTask *d_small, unsigned int *d_ctr, int n_big, int n_med, int n_small, size_t n_elems, size_t task_cap, int n_levels, int *err, hipStream_t st, void *kt, SortTask *stacks)
{
	(void)hipFuncSetAttribute((const void*)k_sort_level_mw<T, Key, MW_NT, MW_LAB_CAP>, hipFuncAttributeMaxDynamicSharedMemorySize, MW_LAB_CAP);
	const uint32_t big_min = (uint32_t)mm355_sort_heavy_threshold(), med_min = (uint32_t)mm355_sort_medium_threshold();
	// d_ctr: [level][2] list lengths, then the wave-task total
	unsigned int h[2 * (MW_LEVELS + 1) + 1];
	memset(h, 0, sizeof(h));
	h[0] = (unsigned int)n_big; h[1] = (unsigned int)n_med; h[2 * (MW_LEVELS + 1)] = (unsigned int)n_small;
	if (hipMemcpyAsync(d_ctr, h, sizeof(h), hipMemcpyHostToDevice, st) != hipSuccess) return -1;   // (pageable source: the copy is staged before the call returns)
	unsigned int *d_small_ctr = d_ctr + 2 * (MW_LEVELS + 1);
	// grids behind level 0: a bound on the list length, capped -- the blocks stride over the list (under the bench load: 48 big-class blocks 435 ms per
	// step in the emulation, 96: 392, 192: 366, 400: 359 -- a block per task beats a short grid although every block wants a whole CU)
	static const size_t lim_big = [] { const char *e = getenv("MM355_SORT_CAP_BIG"); return (size_t)(e && atoi(e) > 0? atoi(e) : 512); }();
	static const size_t lim_med = [] { const char *e = getenv("MM355_SORT_CAP_MED"); return (size_t)(e && atoi(e) > 0? atoi(e) : 1024); }();
	const size_t cap_big = std::min<size_t>(std::min(task_cap, n_elems / big_min + 1), lim_big), cap_med = std::min<size_t>(std::min(task_cap, n_elems / med_min + 1), lim_med);
	int cur = 0;
	for (int level = 0; level < MW_LEVELS && level < n_levels; ++level) {
		// (a big-class block wants a whole CU -- 1024 threads, 120 KB of LDS -- even to find its list empty: behind the first two levels the buckets
		// of more than 16384 elements are few, and a short grid strides over them)
		static const size_t deep_big = [] { const char *e = getenv("MM355_SORT_CAP_DEEP"); return (size_t)(e && atoi(e) > 0? atoi(e) : 48); }();
		const size_t gb = level == 0? (size_t)n_big : level == 1? cap_big : std::min(cap_big, deep_big), gm = level == 0? (size_t)n_med : cap_med;
		if (level == 0 && n_big + n_med == 0) break;
		unsigned int *c_in = d_ctr + 2 * level, *c_out = d_ctr + 2 * (level + 1);
		if (gb) { KtScope ks(kt, KT_LITERAL, st); hipLaunchKernelGGL((k_sort_level_mw<T, Key, MW_NT, MW_LAB_CAP>), dim3((unsigned)std::min<size_t>(gb, MW_STACK_BLOCKS)), dim3(MW_NT), MW_LAB_CAP, st, an, d_big[cur], c_in, d_big[cur ^ 1], d_med[cur ^ 1], d_small, c_out, d_small_ctr, big_min, med_min, err, stacks); }
		if (gm) { KtScope ks(kt, KT_LIT_MED, st); hipLaunchKernelGGL((k_sort_level_mw<T, Key, 256, MW_MED_LAB>), dim3((unsigned)gm), dim3(256), MW_MED_LAB, st, an, d_med[cur], c_in + 1, d_big[cur ^ 1], d_med[cur ^ 1], d_small, c_out, d_small_ctr, big_min, med_min, err, (SortTask*)0); }
		cur ^= 1;
	}
	const size_t gs = n_big + n_med == 0? (size_t)n_small : std::min<size_t>(task_cap, 16384);
	if (gs) { KtScope ks(kt, KT_LIT_TASKS, st); hipLaunchKernelGGL((k_sort_tasks<T, Key>), dim3((unsigned)gs), dim3(WAVE), 0, st, an, d_small, d_small_ctr, err); }
	return 0;
}

// ------------------------------------------------------------------ a7: mg_lchain_dp (fill)
#define TW_SIZE 8192
#define TW_MASK (TW_SIZE - 1)

// Chaining is independent between "segments": maximal runs of the sorted anchor array with the same strand|rid whose
// consecutive x differ by at most max_dist_x.  For the first anchor of a segment the window start `st` reaches the anchor
// itself (every earlier anchor is on another strand/rid or farther than max_dist_x), so no score, no t[] mark and no
// max_ii of an earlier segment can influence it.  k_chain_segments cuts the reads into segments; 1-anchor segments are
// finished on the spot, short ones go to one lane each (k_chain_small), long ones to one wave each (k_chain_big).

__device__ inline void chain_dist(const DevParams &pr, int qlen, int32_t &max_dist_x, int32_t &max_dist_y)
{
	max_dist_y = pr.max_gap;
	if (pr.max_gap_ref > 0) max_dist_x = pr.max_gap_ref;
	else if (pr.max_frag_len > 0) { max_dist_x = pr.max_frag_len - qlen; if (max_dist_x < pr.max_gap) max_dist_x = pr.max_gap; }
	else max_dist_x = pr.max_gap;
	if (max_dist_x < pr.bw) max_dist_x = pr.bw;
	if (max_dist_y < pr.bw) max_dist_y = pr.bw;
}

struct ChainSeg { int32_t read, i0, len, pad; };

// One 256-thread block per CHUNK of SEG_CHUNK anchors of a read (chunk table from the host, which knows the anchor counts); a wave takes
// the chunk's 64-anchor strips one after the other.  A lane flags "segment start" for its anchor (only the left neighbour is looked at);
// a start's segment ends at the next start of the strip (bit scan on the ballot), and for the LAST start of a strip the wave looks ahead
// 64 anchors at a time until it meets the next start -- no lane ever walks a segment, no strip waits for another one.
#define SEG_CHUNK 4096
__global__ __launch_bounds__(256) void k_chain_segments(DevParams pr, DevBatch bt, DevAnchors an, const int2 *chunks, int n_chunks, ChainSeg *small, ChainSeg *big, unsigned int *ctr, int small_max)
{
	// The two lists are appended to with ONE atomic per block and list: with ~100 M anchors per batch (GRCh38-scale ONT reads) one atomic
	// per 64-anchor strip was still 1.6 M strips x 2 on two words -- at ~88 atomics/us per address that WAS the kernel (17 ms of 17.5).
	// Pass 1 leaves every start's segment length in LDS and the strip's two counts; wave 0 turns the counts into offsets (one scan, one
	// atomic per list); pass 2 writes the segments, in anchor order over the whole chunk.
	__shared__ int32_t s_len[SEG_CHUNK];
	__shared__ uint32_t s_ns[SEG_CHUNK / 64], s_nb[SEG_CHUNK / 64];
	if ((int)blockIdx.x >= n_chunks) return;
	const int2 ck = chunks[blockIdx.x];
	const int r = ck.x, c0 = ck.y;
	const int64_t o = an.aoff[r];
	const int n = (int)(an.aoff[r+1] - o);
	const mm128 *a = an.a + o;
	int32_t mdx, mdy;
	chain_dist(pr, bt.rlen[r], mdx, mdy);
	const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int c1 = c0 + SEG_CHUNK < n? c0 + SEG_CHUNK : n;
	if (threadIdx.x < SEG_CHUNK / 64) s_ns[threadIdx.x] = s_nb[threadIdx.x] = 0;
	__syncthreads();
	for (int base = c0 + wv * 64; base < c1; base += 256) {
		const int i = base + lane;
		uint64_t xi = 0;
		if (i < n) xi = a[i].x;
		uint64_t xp = __shfl_up(xi, 1);
		if (lane == 0 && base > 0) xp = a[base - 1].x;
		const bool start = i < n && (i == 0 || (xi >> 32 != xp >> 32) || xi > xp + (uint64_t)(int64_t)mdx);
		const unsigned long long m = __ballot(start);
		s_len[i - c0] = 0;
		if (m == 0) continue;                                  // (wave-uniform)
		int end = -1;
		if (start) { const unsigned long long rest = lane == 63? 0ULL : (m >> (lane + 1)) << (lane + 1); if (rest) end = base + __builtin_ctzll(rest); }
		// the last start of the strip: look ahead for the next start, 512 anchors per step (eight independent loads per lane in flight:
		// a segment of 100 000 anchors is ~200 steps, not 1600)
		const int last = 63 - __builtin_clzll(m);
		int fend = n;
		uint64_t xl = __shfl(xi, 63);                          // x of the anchor just before the look-ahead position
		for (int pos = base + 64; pos < n && fend == n; pos += 512) {
			uint64_t xj[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) { const int j = pos + 64 * u + lane; xj[u] = j < n? a[j].x : 0; }
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const int j = pos + 64 * u + lane;
				uint64_t xq = __shfl_up(xj[u], 1);
				if (lane == 0) xq = xl;
				const bool st2 = j < n && ((xj[u] >> 32 != xq >> 32) || xj[u] > xq + (uint64_t)(int64_t)mdx);
				const unsigned long long m2 = __ballot(st2);
				if (m2) { fend = pos + 64 * u + __builtin_ctzll(m2); break; }
				xl = __shfl(xj[u], 63);
			}
		}
		if (lane == last) end = fend;
		const int len = end - i;
		if (start) {
			s_len[i - c0] = len;
			if (len == 1) {   // finished on the spot
				const int32_t sp = (int32_t)(a[i].y >> 32 & 0xff);
				an.f[o + i] = sp; an.p[o + i] = -1; an.v[o + i] = sp;
			}
		}
		const unsigned long long ms = __ballot(start && len > 1 && len <= small_max), mb = __ballot(start && len > small_max);
		if (lane == 0) { s_ns[(base - c0) >> 6] = (uint32_t)__popcll(ms); s_nb[(base - c0) >> 6] = (uint32_t)__popcll(mb); }
	}
	__syncthreads();
	if (wv == 0) {   // strip counts -> list offsets
		const uint32_t cs = s_ns[lane], cb = s_nb[lane];
		uint32_t is = cs, ib = cb;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const uint32_t us = __shfl_up(is, d), ub = __shfl_up(ib, d); if (lane >= d) { is += us; ib += ub; } }
		uint32_t bs = 0, bb = 0;
		if (lane == 63) { if (is) bs = atomicAdd(&ctr[0], is); if (ib) bb = atomicAdd(&ctr[1], ib); }
		bs = __shfl(bs, 63); bb = __shfl(bb, 63);
		s_ns[lane] = bs + is - cs; s_nb[lane] = bb + ib - cb;
	}
	__syncthreads();
	for (int base = c0 + wv * 64; base < c1; base += 256) {
		const int i = base + lane;
		const int len = s_len[i - c0];
		const bool to_small = len > 1 && len <= small_max, to_big = len > small_max;
		const unsigned long long ms = __ballot(to_small), mb = __ballot(to_big);
		if (to_small || to_big) {
			ChainSeg sg; sg.read = r; sg.i0 = i; sg.len = len; sg.pad = 0;
			if (to_small) small[s_ns[(base - c0) >> 6] + __popcll(ms & LANE_LT_MASK(lane))] = sg;
			else big[s_nb[(base - c0) >> 6] + __popcll(mb & LANE_LT_MASK(lane))] = sg;
		}
	}
}

#define CHAIN_SMALL 32
// one lane per short segment: the literal sequential recurrence (U:lchain.c::mg_lchain_dp)
__global__ __launch_bounds__(256) void k_chain_small(DevParams pr, DevBatch bt, DevAnchors an, const ChainSeg *segs, unsigned int n_segs, unsigned long long *pairs_ctr)
{
	MM355_LATENCY_KERNEL();
	__shared__ int32_t tl[CHAIN_SMALL][256];
	const unsigned int sidx = blockIdx.x * 256 + threadIdx.x;
	unsigned long long pairs = 0;
	if (sidx < n_segs) {
		const ChainSeg sg = segs[sidx];
		const int64_t o = an.aoff[sg.read];
		const mm128 *a = an.a + o;
		int32_t *f = an.f + o, *p = an.p + o, *v = an.v + o;
		int32_t mdx, mdy;
		chain_dist(pr, bt.rlen[sg.read], mdx, mdy);
		const int i0 = sg.i0, i1 = sg.i0 + sg.len, bw = pr.bw, max_skip = pr.max_chain_skip, max_iter = pr.max_chain_iter;
		for (int k = 0; k < sg.len; ++k) tl[k][threadIdx.x] = -1;
		int st = i0, max_ii = -1;
		for (int i = i0; i < i1; ++i) {
			const mm128 ai = a[i];
			int max_j = -1, j, end_j;
			int32_t max_f = (int32_t)(ai.y >> 32 & 0xff), n_skip = 0;
			while (st < i && (ai.x >> 32 != a[st].x >> 32 || ai.x > a[st].x + (uint64_t)(int64_t)mdx)) ++st;
			if (i - st > max_iter) st = i - max_iter;
			for (j = i - 1; j >= st; --j) {
				const mm128 aj = a[j];
				int32_t sc = mm_comput_sc(ai.x, ai.y, aj.x, aj.y, mdx, mdy, bw, pr.pen_gap, pr.pen_skip);
				++pairs;
				if (sc == MM355_SC_NONE) continue;
				sc += f[j];
				if (sc > max_f) { max_f = sc, max_j = j; if (n_skip > 0) --n_skip; }
				else if (tl[j - i0][threadIdx.x] == i) { if (++n_skip > max_skip) break; }
				const int32_t pj = p[j];
				if (pj >= 0) tl[pj - i0][threadIdx.x] = i;
			}
			end_j = j;
			if (max_ii < 0 || ai.x - a[max_ii].x > (uint64_t)(int64_t)mdx) {
				int32_t mx = INT32_MIN;
				max_ii = -1;
				for (j = i - 1; j >= st; --j) if (mx < f[j]) mx = f[j], max_ii = j;
			}
			if (max_ii >= 0 && max_ii < end_j) {
				const mm128 am = a[max_ii];
				int32_t tmp = mm_comput_sc(ai.x, ai.y, am.x, am.y, mdx, mdy, bw, pr.pen_gap, pr.pen_skip);
				if (tmp != MM355_SC_NONE && max_f < tmp + f[max_ii]) max_f = tmp + f[max_ii], max_j = max_ii;
			}
			f[i] = max_f; p[i] = max_j;
			v[i] = max_j >= 0 && v[max_j] > max_f? v[max_j] : max_f;
			if (max_ii < 0 || (ai.x - a[max_ii].x <= (uint64_t)(int64_t)mdx && f[max_ii] < f[i])) max_ii = i;
		}
	}
	for (int o2 = 32; o2 > 0; o2 >>= 1) pairs += __shfl_down(pairs, o2);
	if ((threadIdx.x & 63) == 0 && pairs) atomicAdd(pairs_ctr + (blockIdx.x & 31), pairs);          // words 0..31: short segments
}

// one wave per long segment
#ifndef CH_RING
#define CH_RING 256                  // anchors (and their f/p/v) kept in LDS behind the current one
#endif
#define CH_RMASK (CH_RING - 1)
#define CH_XRING 1024                // reference coordinates only (window start search): a longer ring
#define CH_XMASK (CH_XRING - 1)
#define CH_XNEAR (CH_XRING - 2 * WAVE)
#define CH_NEAR (CH_RING - 2 * WAVE)    // j is served from the rings when i - j <= CH_NEAR (the a[] ring also holds one chunk ahead)
__device__ __forceinline__ void chain_big_segment(const DevParams &pr, const DevBatch &bt, const DevAnchors &an, const ChainSeg *segs, unsigned int n_segs, unsigned long long *pairs_ctr, const unsigned int bid)
{
	// t[] marks of the active window, circular by anchor index: 15 bits of the marking anchor + a valid bit (a stale mark
	// would need an index distance that is a multiple of 32768, larger than window + ring size)
	__shared__ uint16_t tw[TW_SIZE];
	// The last CH_RING anchors and their f/p/v live in LDS rings: the loop-carried dependence of the chain never waits for HBM.
	// a[] enters the ring one 64-anchor chunk ahead of the sweep (coalesced load, issued a chunk earlier); f/p/v are written by
	// lane 0 when an anchor is finished.  Predecessors further back than CH_NEAR (repeat-dense windows) come from HBM.
	__shared__ uint64_t rax[CH_XRING], ray[CH_RING];
	__shared__ int32_t rf[CH_RING], rp[CH_RING], rv[CH_RING];
#define CH_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
	const int lane = threadIdx.x;
	if (bid >= n_segs) return;
	KPROF_BEGIN(bt);
	const ChainSeg sg = segs[bid];
	const int r = sg.read;
	const int64_t o = an.aoff[r];
	const int i_begin = sg.i0, n = sg.i0 + sg.len;
	const mm128 *a = an.a + o;
	int32_t *f = an.f + o, *p = an.p + o, *v = an.v + o;
	const int qlen = bt.rlen[r];
	int32_t max_dist_x, max_dist_y = pr.max_gap;
	if (pr.max_gap_ref > 0) max_dist_x = pr.max_gap_ref;
	else if (pr.max_frag_len > 0) { max_dist_x = pr.max_frag_len - qlen; if (max_dist_x < pr.max_gap) max_dist_x = pr.max_gap; }
	else max_dist_x = pr.max_gap;
	const int bw = pr.bw, max_skip = pr.max_chain_skip, max_iter = pr.max_chain_iter;
	if (max_dist_x < bw) max_dist_x = bw;
	if (max_dist_y < bw) max_dist_y = bw;
	const float pen_gap = pr.pen_gap, pen_skip = pr.pen_skip;
	for (int i = lane; i < TW_SIZE; i += WAVE) tw[i] = 0;
	mm128 nx; nx.x = nx.y = 0;
	if (i_begin + lane < n) nx = a[i_begin + lane];   // first chunk
	__syncthreads();
	int st = i_begin, max_ii = -1;
	uint64_t max_ii_x = 0;                      // a[max_ii].x
	unsigned long long pairs = 0;
	for (int i = i_begin; i < n; ++i) {
		if (((i - i_begin) & (WAVE - 1)) == 0) {   // chunk boundary: publish the prefetched chunk, fetch the next one
			if (i + lane < n) { rax[(i + lane) & CH_XMASK] = nx.x; ray[(i + lane) & CH_RMASK] = nx.y; }
			if (i + WAVE + lane < n) nx = a[i + WAVE + lane];
			CH_SYNC();
		}
		const uint64_t aix = rax[i & CH_XMASK], aiy = ray[i & CH_RMASK];
		// advance st (U: while (st < i && (other rid/strand || too far)) ++st)
		for (;;) {
			int idx = st + lane;
			bool c = false;
			if (idx < i) {
				const uint64_t xs = (i - st <= CH_XNEAR)? rax[idx & CH_XMASK] : a[idx].x;
				c = (aix >> 32 != xs >> 32) || aix > xs + (uint64_t)(int64_t)max_dist_x;
			}
			unsigned long long m = __ballot(c);
			if (m == ~0ULL) { st += 64; continue; }
			st += __builtin_ctzll(~m);
			break;
		}
		if (i - st > max_iter) st = i - max_iter;
		int32_t max_f = (int32_t)(aiy >> 32 & 0xff), max_j = -1, n_skip = 0;
		int end_j = st - 1;
		const uint16_t mark = (uint16_t)((i & 0x7fff) | 0x8000);
		for (int jb = i - 1; jb >= st; jb -= WAVE) {
			const int j = jb - lane;
			const bool active = j >= st;
			const bool near_ = i - (jb - (WAVE - 1)) <= CH_NEAR;   // the whole batch is inside the rings
			int32_t sc = MM355_SC_NONE, pj = -1;
			if (!near_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // older f/p come from HBM: their stores have landed
			if (active) {
				uint64_t ajx, ajy;
				if (near_) { ajx = rax[j & CH_XMASK]; ajy = ray[j & CH_RMASK]; } else { const mm128 aj = a[j]; ajx = aj.x; ajy = aj.y; }
				sc = mm_comput_sc(aix, aiy, ajx, ajy, max_dist_x, max_dist_y, bw, pen_gap, pen_skip);
				if (sc != MM355_SC_NONE) { sc += near_? rf[j & CH_RMASK] : f[j]; pj = near_? rp[j & CH_RMASK] : p[j]; }
			}
			const bool valid = sc != MM355_SC_NONE;
			pairs += __popcll(__ballot(active));
			if (valid && pj >= st) tw[pj & TW_MASK] = mark;   // t[p[j]] = i; marks below st are never tested
			CH_SYNC();
			const bool marked = valid && tw[j & TW_MASK] == mark;
			const int32_t scv = valid? sc : INT32_MIN;
			int32_t pm = wave_excl_prefix_max(scv, lane);
			pm = pm > max_f? pm : max_f;
			const bool improved = valid && sc > pm;
			// n_skip over the lanes in scan order: +1 on a marked lane, -1 (not below 0) on an improving one; the scan stops at the
			// first marked lane that lifts it above max_skip.  A walk reflected at 0 has the closed form c_k = S_k - min(0, min_{m<=k} S_m)
			// with S the plain prefix sums (Lindley), so two DPP scans replace the lane-by-lane replay -- in a co-linear chain nearly
			// every lane is marked and the replay was ~10 scalar instructions per lane.
			const int32_t dstep = improved? -1 : marked? 1 : 0;
			const int32_t S = n_skip + wave_incl_scan_add(dstep);
			const int32_t m0 = wave_incl_scan_min(S);
			const int32_t ck = S - (m0 < 0? m0 : 0);
			const unsigned long long bmask = __ballot(dstep == 1 && ck > max_skip);
			int brk = -1;
			if (bmask) brk = __builtin_ctzll(bmask);
			else n_skip = __builtin_amdgcn_readlane(ck, 63);
			const bool considered = brk < 0 || lane <= brk;
			int32_t cv = (valid && considered)? sc : INT32_MIN;
			const int32_t cmax = wave_reduce_max(cv);
			if (cmax > max_f) {
				unsigned long long w = __ballot(cv == cmax);
				int wl = __builtin_ctzll(w);   // lowest lane = highest j = first met by the sequential scan
				max_f = cmax; max_j = jb - wl;
			}
			CH_SYNC();
			if (brk >= 0) { end_j = jb - brk; break; }
		}
		// max_ii rescue
		bool need = max_ii < 0;
		if (!need) need = aix - max_ii_x > (uint64_t)(int64_t)max_dist_x;
		if (need) {
			int32_t bf = INT32_MIN, bj = -1;
			const bool near_ = i - st <= CH_NEAR;
			if (!near_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			for (int j = i - 1 - lane; j >= st; j -= WAVE) { int32_t fj = near_? rf[j & CH_RMASK] : f[j]; if (bf < fj) bf = fj, bj = j; }
			// best f, ties -> larger j: one 64-bit key (f in the high word, j >= -1 biased to be non-negative in the low word)
			const long long best = wave_reduce_max64((long long)(((unsigned long long)(uint32_t)bf << 32) | (uint32_t)(bj + 1)));
			max_ii = (int)(uint32_t)best - 1;
			if (max_ii >= 0) max_ii_x = (i - max_ii <= CH_XNEAR)? rax[max_ii & CH_XMASK] : a[max_ii].x;
		}
		if (max_ii >= 0 && max_ii < end_j) {
			const bool near_ = i - max_ii <= CH_NEAR;
			uint64_t amx, amy;
			if (near_) { amx = rax[max_ii & CH_XMASK]; amy = ray[max_ii & CH_RMASK]; } else { const mm128 am = a[max_ii]; amx = am.x; amy = am.y; }
			int32_t tmp = mm_comput_sc(aix, aiy, amx, amy, max_dist_x, max_dist_y, bw, pen_gap, pen_skip);
			if (tmp != MM355_SC_NONE) {
				int32_t fm;
				if (near_) fm = rf[max_ii & CH_RMASK]; else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); fm = f[max_ii]; }
				if (max_f < tmp + fm) max_f = tmp + fm, max_j = max_ii;
			}
		}
		int32_t vi = max_f;
		if (max_j >= 0) {
			int32_t vm;
			if (i - max_j <= CH_NEAR) vm = rv[max_j & CH_RMASK]; else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); vm = v[max_j]; }
			if (vm > max_f) vi = vm;
		}
		if (max_ii < 0) { max_ii = i; max_ii_x = aix; }
		else {
			const uint64_t d = aix - max_ii_x;
			int32_t fm;
			if (i - max_ii <= CH_NEAR) fm = rf[max_ii & CH_RMASK]; else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); fm = f[max_ii]; }
			if (d <= (uint64_t)(int64_t)max_dist_x && fm < max_f) { max_ii = i; max_ii_x = aix; }
		}
		CH_SYNC();   // every lane has finished reading the ring slots this anchor overwrites
		if (lane == 0) { f[i] = max_f; p[i] = max_j; v[i] = vi; rf[i & CH_RMASK] = max_f; rp[i & CH_RMASK] = max_j; rv[i & CH_RMASK] = vi; }
		CH_SYNC();
	}
#undef CH_SYNC
	KPROF(12);
	if (lane == 0 && pairs) atomicAdd(pairs_ctr + 32 + (bid & 31), pairs);                     // words 32..63: long segments
}


// One wave per block, a RESIDENT grid: a block takes the next entry of the work list when it is free (one atomic per entry) instead of one block
// per entry.  These kernels hold 30-50 KB of LDS per wave for as long as their read lasts; with a block per entry the first blocks of a launch
// (the heaviest reads: the lists are sorted) took every CU's LDS for milliseconds, and the LDS kernels of the other contexts waited behind them.
// A few hundred resident blocks leave LDS for the others, and the tail is no longer: the heaviest entries start first and a free block
// always takes the next one (longest-processing-time order).  Exit: the shared counter has run past the list -- every block gets there.
#define MM355_DEQUEUE(ctr_ptr, n_items, idx_var) \
	for (unsigned int idx_var = mm355_next_item(ctr_ptr); idx_var < (unsigned int)(n_items); idx_var = mm355_next_item(ctr_ptr))
__device__ __forceinline__ unsigned int mm355_next_item(unsigned int *ctr)
{
	__syncthreads();                                        // the block's LDS is free again
	unsigned int v = 0;
	if (threadIdx.x == 0) v = atomicAdd(ctr, 1u);
	return (unsigned int)__builtin_amdgcn_readfirstlane((int)v);
}
__global__ __launch_bounds__(WAVE) void k_chain_big(DevParams pr, DevBatch bt, DevAnchors an, const ChainSeg *segs, unsigned int n_segs, unsigned long long *pairs_ctr, unsigned int *qctr)
{
	MM355_LATENCY_KERNEL();
	MM355_DEQUEUE(qctr, n_segs, bid) chain_big_segment(pr, bt, an, segs, n_segs, pairs_ctr, bid);
}

#include "mm355_btcore.h"
__global__ __launch_bounds__(WAVE) void k_backtrack(DevParams pr, DevBatch bt, DevAnchors an, int *err, const int32_t *heavy_first, unsigned int *qctr)
{
	MM355_LATENCY_KERNEL();
	__shared__ BtLds S;
	MM355_DEQUEUE(qctr, bt.n_reads, bid) {
		const int r = heavy_first[bid];
		wave_backtrack_read(pr, bt, an, err, &S, r, (int)(an.aoff[r+1] - an.aoff[r]), pr.bw);
	}
}

// ------------------------------------------------------------------ launchers
int mm355_sketch_chunk_size(void) { return SK_CHUNK; }
// resident blocks of the one-wave-per-read kernels that hold tens of KB of LDS each (MM355_DEQUEUE): MM355_RESIDENT_BLOCKS, default 768 = 3 per CU
int mm355_resident_blocks(void) { static const int n = [] { const char *e = getenv("MM355_RESIDENT_BLOCKS"); return e && atoi(e) > 0? atoi(e) : 768; }(); return n; }
void mm355_launch_sketch(const DevIndex &ix, const DevBatch &bt, DevSeeds &sd, const int32_t *chunk_read, const int32_t *chunk_start, int n_chunks,
                         const int64_t *read_chunk0, int32_t *chunk_n, hipStream_t st, void *kt)
{
	if (bt.n_reads == 0) return;
	KtScope ks(kt, KT_SKETCH, st);
	if (n_chunks > 0) {
		const char *e = getenv("MM355_SKETCH_SPARSE_MAX");   // read per launch: the parity tests force either form
		const int sparse_max = e? atoi(e) : 2048;             // chunks
		const bool hpc = (ix.flag & 1) != 0;                  // MM_I_HPC
		if (n_chunks <= sparse_max) {
			hipLaunchKernelGGL(hpc? k_sketch_sparse_hpc : k_sketch_sparse, dim3(n_chunks), dim3(WAVE * SKS_WAVES), (size_t)ix.w * SKS_WAVES * sizeof(mm128), st, ix, bt, sd, chunk_read, chunk_start, n_chunks, chunk_n);
		} else {
			int blocks = (n_chunks + WAVE - 1) / WAVE;
			size_t lds = (size_t)ix.w * WAVE * sizeof(mm128);
			hipLaunchKernelGGL(hpc? k_sketch_hpc : k_sketch, dim3(blocks), dim3(WAVE), lds, st, ix, bt, sd, chunk_read, chunk_start, n_chunks, chunk_n);
		}
	}
	hipLaunchKernelGGL(k_sketch_compact, dim3(bt.n_reads), dim3(WAVE), 0, st, bt, sd, read_chunk0, chunk_n);
}
void mm355_launch_mzflt(const DevParams &pr, const DevBatch &bt, DevSeeds &sd, hipStream_t st, void *kt)
{
	if (bt.n_reads == 0) return;
	KtScope ks(kt, KT_MZFLT, st);
	hipLaunchKernelGGL(k_mzflt, dim3(bt.n_reads), dim3(WAVE), 0, st, pr, bt, sd);
}
void mm355_launch_seed_lookup(const DevIndex &ix, const DevBatch &bt, DevSeeds &sd, const int32_t *chunk_read, const int32_t *chunk_start, int n_chunks,
                              unsigned long long *hit_ctr, unsigned int *tile_ctr, hipStream_t st, void *kt)
{
	if (bt.n_reads == 0 || n_chunks == 0) return;
	KtScope ks(kt, KT_LOOKUP, st);
	int4 *tiles = (int4*)sd.hl;   // (the hit list of k_seed_select: free until then; 4 bytes per minimizer slot >= 16 per 384-slot table entry)
	(void)hipMemsetAsync(tile_ctr, 0, 4, st);
	hipLaunchKernelGGL(k_lookup_tiles, dim3((n_chunks + 1023) / 1024), dim3(1024), 0, st, bt, sd, chunk_read, chunk_start, n_chunks, tiles, tile_ctr);
	static const int lk_grid = []{ const char *e = getenv("MM355_LK_GRID"); return e && atoi(e) > 0? atoi(e) : 2048; }();
	const int grid = n_chunks < lk_grid? n_chunks : lk_grid;   // resident blocks walk the live tiles
	hipLaunchKernelGGL(k_seed_lookup, dim3(grid), dim3(256), 0, st, ix, bt, sd, (const int4*)tiles, (const unsigned int*)tile_ctr, hit_ctr);
}
void mm355_launch_seed_select(const DevIndex &ix, const DevParams &pr, const DevBatch &bt, DevSeeds &sd, hipStream_t st, void *kt)
{
	if (bt.n_reads == 0) return;
	KtScope ks(kt, KT_SELECT, st);
	hipLaunchKernelGGL(k_seed_select, dim3(bt.n_reads), dim3(WAVE), 0, st, ix, pr, bt, sd);
}
void mm355_launch_seed_expand(const DevIndex &ix, const DevParams &pr, const DevBatch &bt, DevSeeds &sd, DevAnchors &an, hipStream_t st, void *kt)
{
	if (bt.n_reads == 0) return;
	KtScope ks(kt, KT_EXPAND, st);
	hipLaunchKernelGGL(k_seed_expand, dim3(bt.n_reads), dim3(256), 0, st, ix, pr, bt, sd, an);
}
// Literal radix_sort_128x of the listed reads.  h_tasks: the initial whole-read tasks (byte 56) grouped by size class -- n_big entries
// (> MW_BIG anchors: 1024-thread levels), then n_med (> MW_MED: 256-thread levels), then n_small (one wave each) -- in pinned or otherwise
// stable host memory until the stream has consumed it.  n_elems: elements of all listed arrays together.  task_buf: device scratch for 5 task
// lists of `task_cap` entries + 64 counters.  n_levels (0 = all nine): a bound on the number of list levels -- one per byte of the key that is not the
// same for every element the index can produce (a level whose byte is constant is skipped inside the kernel and opens no new list).
int mm355_launch_sort(const DevBatch &bt, DevAnchors &an, int *err, const void *h_tasks, int n_big, int n_med, int n_small, size_t n_elems, void *task_buf, size_t task_cap, hipStream_t st, void *kt, int n_levels)
{
	(void)bt;
	const int n = n_big + n_med + n_small;
	if (n == 0) return 0;
	SortTask *base = (SortTask*)task_buf;
	SortTask *big[2] = { base, base + task_cap }, *med[2] = { base + 2 * task_cap, base + 3 * task_cap };
	SortTask *small = base + 4 * task_cap;
	unsigned int *ctr = (unsigned int*)(base + 5 * task_cap);
	// stacks of the big-class blocks (depth first inside a block): MW_STK entries per block of the widest grid, behind the counters
	// (MM355_SORT_DFS=1; off by default: measured under the bench load the big levels take 300-310 ms per step with it against 233 without -- a read's two
	// strand buckets then run one after the other in ONE block instead of side by side in two, and that costs more than the launches it saves)
	static const bool use_stacks = [] { const char *e = getenv("MM355_SORT_DFS"); return e && atoi(e) != 0; }();
	SortTask *stacks = use_stacks? (SortTask*)((char*)task_buf + mm355_sort_buf_bytes(task_cap) - (size_t)MW_STACK_BLOCKS * MW_STK * sizeof(SortTask)) : 0;
	const SortTask *ht = (const SortTask*)h_tasks;
	if (n_big && hipMemcpyAsync(big[0], ht, (size_t)n_big * sizeof(SortTask), hipMemcpyHostToDevice, st) != hipSuccess) return -1;
	if (n_med && hipMemcpyAsync(med[0], ht + n_big, (size_t)n_med * sizeof(SortTask), hipMemcpyHostToDevice, st) != hipSuccess) return -1;
	if (n_small && hipMemcpyAsync(small, ht + n_big + n_med, (size_t)n_small * sizeof(SortTask), hipMemcpyHostToDevice, st) != hipSuccess) return -1;
	return sort_tasks_run<mm128, mm_key_x>(an, big, med, small, ctr, n_big, n_med, n_small, n_elems, task_cap, n_levels > 0? n_levels : MW_LEVELS, err, st, kt, stacks);
}
int mm355_sort_task_bytes(void) { return (int)sizeof(SortTask); }
// device scratch of mm355_launch_sort for `task_cap`: five task lists, 64 counters, the stacks of the big-class blocks
size_t mm355_sort_buf_bytes(size_t task_cap) { return task_cap * 5 * sizeof(SortTask) + 512 + (size_t)MW_STACK_BLOCKS * MW_STK * sizeof(SortTask); }
int mm355_sort_heavy_threshold(void)   // MM355_SORT_HEAVY_MIN: test hook that pushes ordinary reads through the 1024-thread path
{
	static int thr = [] { const char *e = getenv("MM355_SORT_HEAVY_MIN"); int v = e? atoi(e) : MW_BIG; return v < 65? 65 : v; }();
	return thr;
}
int mm355_sort_medium_threshold(void)   // buckets above this (and up to the heavy threshold) take a 256-thread level; MM355_SORT_MEDIUM_MIN: test hook
{
	static int thr = [] { const char *e = getenv("MM355_SORT_MEDIUM_MIN"); int v = e? atoi(e) : MW_MED; return v < 65? 65 : v; }();
	const int hv = mm355_sort_heavy_threshold();
	return thr < hv? thr : hv;
}
// seg_small / seg_big: scratch lists of at least tot_a/2 + 1 entries each; ctr: 2 zeroed u32 on the device; chunks: (read, first anchor)
// of every mm355_chain_chunk()-anchor piece of every read
int mm355_chain_chunk(void) { return SEG_CHUNK; }
int mm355_launch_chain(const DevParams &pr, const DevBatch &bt, DevAnchors &an, unsigned long long *pairs, void *seg_small, void *seg_big, unsigned int *ctr,
                       const void *chunks, int n_chunks, hipStream_t st, void *kt)
{
	if (bt.n_reads == 0 || n_chunks == 0) return 0;
	if (hipMemsetAsync(ctr, 0, 8, st) != hipSuccess) return -1;
	if (kt) mm355_kt(kt, KT_CHAIN_SEG, 0, st);
	hipLaunchKernelGGL(k_chain_segments, dim3(n_chunks), dim3(256), 0, st, pr, bt, an, (const int2*)chunks, n_chunks, (ChainSeg*)seg_small, (ChainSeg*)seg_big, ctr, CHAIN_SMALL);
	if (kt) mm355_kt(kt, KT_CHAIN_SEG, 1, st);
	unsigned int h[2] = {0, 0};
	if (hipMemcpyAsync(h, ctr, 8, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
	if (mm355_wait_stream(st) != hipSuccess) return -1;
	if (hipMemsetAsync(ctr + 1, 0, 4, st) != hipSuccess) return -1;   // (read back: the word becomes k_chain_big's work-list cursor)
	if (h[1]) { KtScope ks(kt, KT_CHAIN_BIG, st); hipLaunchKernelGGL(k_chain_big, dim3(h[1] < (unsigned)mm355_resident_blocks()? h[1] : (unsigned)mm355_resident_blocks()), dim3(WAVE), 0, st, pr, bt, an, (const ChainSeg*)seg_big, h[1], pairs, ctr + 1); }
	if (h[0]) { KtScope ks(kt, KT_CHAIN_SMALL, st); hipLaunchKernelGGL(k_chain_small, dim3((h[0] + 255) / 256), dim3(256), 0, st, pr, bt, an, (const ChainSeg*)seg_small, h[0], pairs); }
	return 0;
}
void mm355_launch_backtrack(const DevParams &pr, const DevBatch &bt, DevAnchors &an, int *err, const int32_t *heavy_first, unsigned int *qctr, hipStream_t st, void *kt)
{
	if (bt.n_reads == 0) return;
	(void)hipMemsetAsync(qctr, 0, 4, st);
	KtScope ks(kt, KT_BACKTRACK, st);
	hipLaunchKernelGGL(k_backtrack, dim3(bt.n_reads < mm355_resident_blocks()? bt.n_reads : mm355_resident_blocks()), dim3(WAVE), 0, st, pr, bt, an, err, heavy_first, qctr);
}
