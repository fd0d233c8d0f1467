// mm355_cullsort.hip -- row a6 for anchor-rich reads (GRCh38-scale): cull, then sort what is left, one block per read, in LDS.
//
// What the reference does (U:map.c::mm_map_frag, reached from /root/reference/src/lib.rs:587): radix_sort_128x over ALL anchors of a read by
// x = strand | rid | rpos (U:ksort.h), then mg_lchain_dp over the sorted array.  Two facts about that pair:
//  (1) mg_lchain_dp is local to an "x-component" -- a maximal run of the sorted array on one strand / contig whose consecutive reference
//      positions differ by at most max_dist_x: the window start `st` of the first anchor of a component is that anchor itself, so no score,
//      no t[] mark and no max_ii crosses a component boundary (the same argument that lets k_chain_segments cut the array).  A component
//      with n anchors cannot reach f >= min_chain_score unless n * k >= min_chain_score (every step adds at most the k-mer span: k, or up to
//      255 on a homopolymer-compressed index, where 255 takes k's place), and its
//      chains have at most n anchors, so with n < T = ceil(min_chain_score / k) nothing of it enters z[], mg_chain_backtrack
//      or compact_a.  Deleting such a component from the sorted array changes neither u[] nor the compacted anchors: indices only appear
//      as p[] / z[].y inside the chainer, and the (unstable) sort of z[] by score looks at the scores alone.  n_a, rep_len and mini_pos are
//      fixed before the sort.  On the GRCh38-scale workload nine anchors out of ten sit in such components: lone repeat hits.
//  (2) the unstable sort is only observable through the order of EQUAL keys, and that order depends on the whole array.
// So: anchors are binned by position (bins of 2^sh >= max_dist_x bases of the concatenated, strand-doubled reference; three bitmaps in LDS
// count a bin's anchors up to 3), an anchor is dropped when the run of non-empty bins around its own holds fewer than T anchors (a run of
// bins is a union of whole x-components: two anchors at most max_dist_x apart are in the same or in adjacent bins), the survivors -- 8-byte
// words position << ib | index in generation order -- are sorted by a bitonic network in LDS, and a read whose SURVIVORS contain equal keys
// goes through the literal emulation of radix_sort_128x on its whole generation-order array (k_sort_level_mw / k_sort_tasks in
// mm355_kernels.hip, on a copy), from which the survivors are then taken in order.  No library sort: the round-3 path ran
// rocprim::radix_sort_pairs over every anchor of every read (16 % of the GPU time of the bench).
#include <cstring>
#include <cstdio>
#include <algorithm>
#include <numeric>
#include <hip/hip_runtime.h>
#include "mm355_pipeline.h"
#include "mm355_wave.h"

#define CS_WPL_MAX 12288               // most words per bitmap level: 393216 bins, 3 x 48 KB of LDS (the kernel takes the number as an argument)
#define CS_WPL_DEFAULT 6144            // ... and the default: 3 x 24 KB, so that a block needs half a CU, not a whole one, beside the other contexts' kernels (+1.5 % in the bench; twice the passes)
#define CS_GUARD 3                     // bins a run is followed to either side of an anchor's own bin


struct CullPar {
	uint64_t tot_len;                  // bases of all contigs; position word = strand * tot_len + seq_off[rid] + rpos < 2 * tot_len
	int32_t ib, sh, n_pass, T;         // ib: bits of (index in generation order << 1 | kept) below the position
	int32_t wpl, bpp;                  // words per bitmap level (LDS: 3 * 4 * wpl bytes), bins decided per pass = 32 * wpl - 2 * CS_GUARD
};

__device__ __forceinline__ uint64_t cs_pos(const DevIndex &ix, uint64_t x, uint64_t tot_len)
{
	const uint32_t rid = (uint32_t)(x >> 32) & 0x7fffffffu;
	return (x >> 63) * tot_len + ix.seq_off[rid] + (uint32_t)x;
}

__device__ __forceinline__ void cs_add(uint32_t *bm, uint32_t rel, const uint32_t CS_WPL)
{
	const uint32_t w = rel >> 5, m = 1u << (rel & 31);
	if (bm[2 * CS_WPL + w] & m) return;                       // (bits only ever get set: a stale read costs an atomic, not correctness)
	if (!(atomicOr(&bm[w], m) & m)) return;
	if (!(atomicOr(&bm[CS_WPL + w], m) & m)) return;
	atomicOr(&bm[2 * CS_WPL + w], m);
}
__device__ __forceinline__ uint32_t cs_cnt(const uint32_t *bm, uint32_t rel, const uint32_t CS_WPL)   // 0, 1, 2, 3 (= three or more)
{
	const uint32_t w = rel >> 5, b = rel & 31;
	uint32_t c = bm[w] >> b & 1u;
	if (c) { c += bm[CS_WPL + w] >> b & 1u; if (c == 2) c += bm[2 * CS_WPL + w] >> b & 1u; }
	return c;
}
// does the run of non-empty bins around table entry `rel` hold at least T anchors?  Conservative: "yes" whenever a count is saturated
// or the walk stops at the guard distance without having met an empty bin.
__device__ __forceinline__ bool cs_keep(const uint32_t *bm, uint32_t rel, uint32_t T, const uint32_t CS_WPL)
{
	uint32_t tot = cs_cnt(bm, rel, CS_WPL);
	if (tot >= 3 || tot >= T) return true;
#pragma unroll
	for (int dir = -1; dir <= 1; dir += 2)
		for (int s = 1; s <= CS_GUARD; ++s) {
			const uint32_t c = cs_cnt(bm, (uint32_t)((int)rel + dir * s), CS_WPL);
			if (c == 0) break;
			tot += c;
			if (c == 3 || tot >= T || s == CS_GUARD) return true;
		}
	return false;
}

// One block per read.  keys[o + i] = position word << ib | i << 1 | kept for every anchor (generation order); surv[o ..] = the words of the
// anchors that are kept, in no particular order; n_keep[r] = how many.
__global__ __launch_bounds__(1024) void k_cull(DevIndex ix, const int64_t *aoff, const mm128 *a, uint64_t *keys, uint64_t *surv, int32_t *n_keep, int n_reads, CullPar cp, const int32_t *heavy_first)
{
	extern __shared__ uint32_t bm[];   // 3 * CS_WPL
	__shared__ uint32_t s_cur;
	if ((int)blockIdx.x >= n_reads) return;
	const int r = heavy_first[blockIdx.x];   // the reads with the most anchors first: the longest block starts at t = 0
	const int64_t o = aoff[r];
	const uint32_t n = (uint32_t)(aoff[r + 1] - o), tid = threadIdx.x, lane = tid & 63, CS_NT = blockDim.x, CS_WPL = (uint32_t)cp.wpl;
	const int64_t CS_BPP = cp.bpp;
	if (tid == 0) s_cur = 0;
	if (n == 0) { if (tid == 0) n_keep[r] = 0; return; }
	for (int p = 0; p < cp.n_pass; ++p) {
		const int64_t lo = (int64_t)p * CS_BPP - CS_GUARD;          // bin of table entry 0
		{ uint4 *b4 = (uint4*)bm; const uint4 z4 = make_uint4(0, 0, 0, 0); for (uint32_t i = tid; i < 3 * CS_WPL / 4; i += CS_NT) b4[i] = z4; }
		__syncthreads();
		for (uint32_t i = tid; i < n; i += CS_NT) {
			uint64_t k;
			if (p == 0) { k = cs_pos(ix, a[o + i].x, cp.tot_len); keys[o + i] = k << cp.ib | (uint64_t)i << 1; }
			else k = keys[o + i] >> cp.ib;
			const int64_t rel = (int64_t)(k >> cp.sh) - lo;
			if (rel >= 0 && rel < CS_BPP + 2 * CS_GUARD) cs_add(bm, (uint32_t)rel, CS_WPL);
		}
		__syncthreads();
		for (uint32_t base = 0; base < n; base += CS_NT) {
			const uint32_t i = base + tid;
			bool keep = false; uint64_t kw = 0;
			if (i < n) {
				kw = keys[o + i] | 1ULL;                                // (written by this very thread in pass 0)
				const int64_t rel = (int64_t)((kw >> cp.ib) >> cp.sh) - lo;
				if (rel >= CS_GUARD && rel < CS_BPP + CS_GUARD) keep = cs_keep(bm, (uint32_t)rel, (uint32_t)cp.T, CS_WPL);
				if (keep) keys[o + i] = kw;                             // the decision in bit 0, for the reads that are sorted literally
			}
			const unsigned long long mk = __ballot(keep);
			if (mk) {
				uint32_t at = 0;
				if (lane == 0) at = atomicAdd(&s_cur, (uint32_t)__popcll(mk));
				at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
				if (keep) surv[o + at + (uint32_t)__popcll(mk & LANE_LT_MASK(lane))] = kw;
			}
		}
		__syncthreads();
	}
	if (tid == 0) n_keep[r] = (int32_t)s_cur;
}

// culling off (stage entry of the full sorted array; T < 2): every anchor survives
__global__ __launch_bounds__(256) void k_keys_all(DevIndex ix, const int64_t *aoff, const mm128 *a, uint64_t *surv, int32_t *n_keep, int n_reads, CullPar cp)
{
	const int r = blockIdx.x;
	if (r >= n_reads) return;
	const int64_t o = aoff[r];
	const uint32_t n = (uint32_t)(aoff[r + 1] - o);
	for (uint32_t i = threadIdx.x; i < n; i += 256) surv[o + i] = cs_pos(ix, a[o + i].x, cp.tot_len) << cp.ib | (uint64_t)i << 1 | 1ULL;
	if (threadIdx.x == 0) n_keep[r] = (int32_t)n;
}

// ------------------------------------------------------------------ bitonic network, ascending comparators only
// Stage k merges sorted runs of k / 2: first every element i of a run's lower half meets its mirror image in the upper half, then the
// half-cleaners j = k / 4 ... 1 (i against i + j).  Every comparator puts the smaller word at the lower index, so an array of any length n
// sorts as if it were padded with +infinity: a comparator whose upper index is >= n is skipped.
template <int NT>
__device__ __forceinline__ void cs_ce(uint64_t *s, uint32_t i, uint32_t l) { const uint64_t x = s[i], y = s[l]; if (x > y) { s[i] = y; s[l] = x; } }

template <int NT>
__device__ inline void bitonic_steps_lds(uint64_t *s, uint32_t n, uint32_t m, uint32_t j_first)   // half-cleaners j_first, j_first / 2, ... 1 over s[0, n)
{
	const uint32_t tid = threadIdx.x;
	for (uint32_t j = j_first; j > 0; j >>= 1) {
		__syncthreads();
		for (uint32_t t = tid; t < (m >> 1); t += NT) {
			const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;
			if (l < n) cs_ce<NT>(s, i, l);
		}
	}
}
template <int NT>
__device__ inline void bitonic_sort_lds(uint64_t *s, uint32_t n)
{
	const uint32_t tid = threadIdx.x;
	uint32_t m = 2; while (m < n) m <<= 1;
	for (uint32_t k = 2; k <= m; k <<= 1) {
		__syncthreads();
		const uint32_t hk = k >> 1;
		for (uint32_t t = tid; t < (m >> 1); t += NT) {
			const uint32_t blk = t / hk, off = t & (hk - 1), i = blk * k + off, l = blk * k + k - 1 - off;
			if (l < n) cs_ce<NT>(s, i, l);
		}
		bitonic_steps_lds<NT>(s, n, m, k >> 2);
	}
	__syncthreads();
}

// One block per listed read: sorts surv[o, o + nk) in place; flag[r] = the survivors contain equal positions and the reference would not
// have insertion-sorted the read (n_a > 64: U:ksort.h::radix_sort_128x); otherwise the read's sorted anchors are written to
// out[aoff2[r] ..] (gathered from the generation-order array by the index in the low bits of the word).
// CAP words of LDS; a read with more survivors has its CAP-word tiles sorted in LDS and merged through HBM (the same network: the
// comparators of distance >= CAP run on the global array, the rest of a stage on one tile at a time).
template <int NT, int CAP>
__global__ __launch_bounds__(NT) void k_asort(const int32_t *list, int n_list, const int64_t *aoff, const int64_t *aoff2, const int32_t *n_keep, const mm128 *a,
                                              uint64_t *surv, mm128 *out, uint8_t *flag, int ib)
{
	extern __shared__ uint64_t s[];    // CAP
	if ((int)blockIdx.x >= n_list) return;
	const int r = list[blockIdx.x];
	const int64_t o = aoff[r];
	const uint32_t n_all = (uint32_t)(aoff[r + 1] - o), n = n_keep? (uint32_t)n_keep[r] : n_all, tid = threadIdx.x;   // (n_keep = null: the whole array, sort only)
	uint64_t *g = surv + o;
	bool tie = false;
	if (n <= (uint32_t)CAP) {
		for (uint32_t i = tid; i < n; i += NT) s[i] = g[i];
		if (n > 1) bitonic_sort_lds<NT>(s, n); else __syncthreads();
		for (uint32_t i = tid; i < n; i += NT) { const uint64_t w = s[i]; g[i] = w; if (i > 0 && (w >> ib) == (s[i - 1] >> ib)) tie = true; }
	} else {
		uint32_t m = 2; while (m < n) m <<= 1;
		for (uint32_t t0 = 0; t0 < n; t0 += CAP) {
			const uint32_t cnt = n - t0 < (uint32_t)CAP? n - t0 : (uint32_t)CAP;
			__syncthreads();
			for (uint32_t i = tid; i < cnt; i += NT) s[i] = g[t0 + i];
			bitonic_sort_lds<NT>(s, cnt);
			for (uint32_t i = tid; i < cnt; i += NT) g[t0 + i] = s[i];
		}
		for (uint32_t k = 2u * CAP; (k >> 1) < n; k <<= 1) {
			__syncthreads();
			const uint32_t hk = k >> 1;
			for (uint32_t t = tid; t < (m >> 1); t += NT) {
				const uint32_t blk = t / hk, off = t & (hk - 1), i = blk * k + off, l = blk * k + k - 1 - off;
				if (l < n) cs_ce<NT>(g, i, l);
			}
			for (uint32_t j = k >> 2; j >= (uint32_t)CAP; j >>= 1) {
				__syncthreads();
				for (uint32_t t = tid; t < (m >> 1); t += NT) {
					const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;
					if (l < n) cs_ce<NT>(g, i, l);
				}
			}
			for (uint32_t t0 = 0; t0 < n; t0 += CAP) {
				const uint32_t cnt = n - t0 < (uint32_t)CAP? n - t0 : (uint32_t)CAP;
				__syncthreads();
				for (uint32_t i = tid; i < cnt; i += NT) s[i] = g[t0 + i];
				bitonic_steps_lds<NT>(s, cnt, (uint32_t)CAP, (uint32_t)CAP >> 1);
				__syncthreads();
				for (uint32_t i = tid; i < cnt; i += NT) g[t0 + i] = s[i];
			}
		}
		__syncthreads();
		for (uint32_t i = tid; i < n; i += NT) if (i > 0 && (g[i] >> ib) == (g[i - 1] >> ib)) tie = true;
	}
	if (out == 0) return;                                   // sort only
	const int any = __syncthreads_or(tie);
	const bool literal = any && n_all > MM355_RS_MIN_SIZE;
	if (tid == 0) flag[r] = literal? 1 : 0;
	if (literal) return;
	const uint64_t im = (1ULL << (ib - 1)) - 1;
	mm128 *dst = out + aoff2[r];
	const mm128 *src = a + o;
	if (n <= (uint32_t)CAP) { for (uint32_t i = tid; i < n; i += NT) dst[i] = src[s[i] >> 1 & im]; }
	else for (uint32_t i = tid; i < n; i += NT) dst[i] = src[g[i] >> 1 & im];
}

// ------------------------------------------------------------------ reads with equal keys among their survivors
// the generation-order anchors of the listed reads, copied to the dense arrays the literal emulation works on; bit 63 of y (unused: y is
// flags | span << 32 | query position) carries k_cull's decision through the sort
__global__ __launch_bounds__(256) void k_tie_copy(const int32_t *list, int n_list, const int64_t *aoff, const int64_t *toff, const mm128 *a, const uint64_t *keys, mm128 *ta, int keep_all)
{
	const int t = blockIdx.x;
	if (t >= n_list) return;
	const int r = list[t];
	const int64_t o = aoff[r], d = toff[t];
	const uint32_t n = (uint32_t)(aoff[r + 1] - o);
	for (uint32_t i = threadIdx.x; i < n; i += 256) {
		mm128 el = a[o + i];
		if (keep_all || (keys[o + i] & 1ULL)) el.y |= 1ULL << 63;
		ta[d + i] = el;
	}
}
// tcnt[d + i] = equal-position neighbour pairs among the first i + 1 elements of the read's plainly sorted array (sorted[o ..]): the literal
// emulation skips every bucket without such a pair -- its content is unique and comes from the plain sort (WalkScratch::tcnt)
__global__ __launch_bounds__(256) void k_tie_tcnt(const int32_t *list, int n_list, const int64_t *aoff, const int64_t *toff, const uint64_t *sorted, int32_t *tcnt, int ib, int kept_only)
{
	__shared__ uint32_t s_w[4];
	const int t = blockIdx.x;
	if (t >= n_list) return;
	const int r = list[t];
	const int64_t o = aoff[r], d = toff[t];
	const uint32_t n = (uint32_t)(aoff[r + 1] - o), tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	uint32_t base = 0;
	for (uint32_t b0 = 0; b0 < n; b0 += 256) {
		const uint32_t i = b0 + tid;
		// (kept_only: equal positions share a bin of k_cull, hence its decision -- a pair of culled anchors is dropped whatever its order, and
		// the emulation need not descend into a bucket for it)
		const bool tie = i > 0 && i < n && (sorted[o + i] >> ib) == (sorted[o + i - 1] >> ib) && (!kept_only || (sorted[o + i] & 1ULL));
		const unsigned long long mk = __ballot(tie);
		if (lane == 0) s_w[wv] = (uint32_t)__popcll(mk);
		__syncthreads();
		uint32_t before = 0, tot = 0;
		for (uint32_t w2 = 0; w2 < 4; ++w2) { const uint32_t c = s_w[w2]; if (w2 < wv) before += c; tot += c; }
		if (i < n) tcnt[d + i] = (int32_t)(base + before + (uint32_t)__popcll(mk & (LANE_LT_MASK(lane) | 1ULL << lane)));
		base += tot;
		__syncthreads();
	}
}
// ... and after the emulation: the kept anchors of the read's sorted array, in order.  Inside a run of equal positions the emulation's
// result (and the decision it carried); everywhere else the plain sort's element
__global__ __launch_bounds__(256) void k_tie_emit(const int32_t *list, int n_list, const int64_t *aoff, const int64_t *toff, const int64_t *aoff2, const mm128 *a, const mm128 *ta,
                                                  const uint64_t *sorted, mm128 *out, int ib, int kept_only)
{
	__shared__ uint32_t s_w[4];
	const int t = blockIdx.x;
	if (t >= n_list) return;
	const int r = list[t];
	const int64_t o = aoff[r], d = toff[t];
	const uint32_t n = (uint32_t)(aoff[r + 1] - o), tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint64_t im = (1ULL << (ib - 1)) - 1;
	mm128 *dst = out + aoff2[r];
	uint32_t base_out = 0;
	for (uint32_t base = 0; base < n; base += 256) {
		const uint32_t i = base + tid;
		bool keep = false; mm128 el; el.x = el.y = 0;
		if (i < n) {
			const uint64_t w = sorted[o + i], k = w >> ib;
			const bool run = ((i > 0 && (sorted[o + i - 1] >> ib) == k) || (i + 1 < n && (sorted[o + i + 1] >> ib) == k)) && (!kept_only || (w & 1ULL));   // (a culled run: its bucket may have been skipped, and it is dropped anyway)
			if (run) { el = ta[d + i]; keep = el.y >> 63 != 0; el.y &= ~(1ULL << 63); }
			else { el = a[o + (w >> 1 & im)]; keep = (w & 1ULL) != 0; }
		}
		const unsigned long long mk = __ballot(keep);
		if (lane == 0) s_w[wv] = (uint32_t)__popcll(mk);
		__syncthreads();
		uint32_t before = 0, tot = 0;
		for (uint32_t w2 = 0; w2 < 4; ++w2) { const uint32_t c = s_w[w2]; if (w2 < wv) before += c; tot += c; }
		if (keep) dst[base_out + before + (uint32_t)__popcll(mk & LANE_LT_MASK(lane))] = el;
		base_out += tot;
		__syncthreads();
	}
}

// ------------------------------------------------------------------ host side
static int bits_for(uint64_t v) { int b = 1; while (b < 64 && (1ULL << b) <= v) ++b; return b; }   // bits that hold the values 0 .. v

#define CS_SMALL_CAP 2048
#define CS_MID_CAP   8192
#define CS_BIG_CAP   16384

bool mm355_cull_sort_fits(const mm355_ctx *c)
{
	const mm355_index *mi = c->mi;
	const uint64_t tot_len = mi->n_seq? mi->seq_off[mi->n_seq - 1] + mi->seq_len[mi->n_seq - 1] : 1;
	int32_t max_na = 1;
	for (int64_t i = 0; i < c->hb.n_reads; ++i) if (c->hb.n_a[i] > max_na) max_na = c->hb.n_a[i];
	return bits_for(2 * tot_len - 1) + bits_for((uint64_t)max_na - 1) + 1 <= 64;
}

// Sorts the anchors of every read of the batch (c->a, offsets c->aoff / hb.aoff) and, with cull != 0, drops the anchors that cannot chain.
// On return c->a / c->aoff / c->n_a, hb.aoff, hb.n_a and hb.tot_a describe the new (dense) array; the old buffers are scratch again.
int mm355_cull_sort(mm355_ctx *c, const DevParams &pr, int cull)
{
	HostBatch &hb = c->hb;
	const int n_reads = (int)hb.n_reads;
	const int64_t tot = hb.tot_a;
	if (n_reads <= 0 || tot <= 0) return 0;
	const mm355_index *mi = c->mi;
	CullPar cp; memset(&cp, 0, sizeof(cp));
	cp.tot_len = mi->n_seq? mi->seq_off[mi->n_seq - 1] + mi->seq_len[mi->n_seq - 1] : 1;
	int32_t max_na = 1;
	for (int i = 0; i < n_reads; ++i) if (hb.n_a[i] > max_na) max_na = hb.n_a[i];
	cp.ib = bits_for((uint64_t)max_na - 1) + 1;   // index in generation order, and the cull's decision in bit 0
	if (bits_for(2 * cp.tot_len - 1) + cp.ib > 64) return MM355_EUNSUP;   // (a 2^40-base reference with 2^22 anchors on one read)
	// T: the fewest anchors a chain that survives mg_chain_backtrack can have; D: the largest max_dist_x of any read (chain_dist)
	// T: the fewest anchors with which a component can reach f >= min_chain_score, i.e. put anything into z[] (a step adds at most a seed's
	// span: k, or a sum of k run lengths below 256 on an HPC index -- U:sketch.c keeps no record with kmer_span >= 256 -- where nothing can
	// be culled: T = 1).  NOT max(min_cnt, ..): a component of fewer than min_cnt anchors leaves no chain, but an anchor of it with
	// f >= min_chain_score is an element of z[] while mg_chain_backtrack sorts z[] by score with the UNSTABLE radix sort, and the order of
	// equal scores decides which chain end claims a shared anchor first (tests/test_gpu_hpc.py::test_hpc_cull_threshold_counts_spans_not_k:
	// 3 reads of 40 differed with min_cnt in the rule).  The presets' values are the same either way (map-ont, map-hifi: 3).
	const int max_span = (mi->flag & 1)? 255 : mi->k;
	const int T = (pr.min_chain_score + max_span - 1) / max_span;
	int64_t D = pr.max_gap_ref > 0? pr.max_gap_ref : pr.max_frag_len > 0? std::max(pr.max_frag_len, pr.max_gap) : pr.max_gap;
	if (D < pr.bw) D = pr.bw;
	if (D < 1) D = 1;
	cp.T = T;
	const bool do_cull = cull && T >= 2;
	// MM355_CULL_WPL / _NT / _SH (experiments): words per bitmap level, threads per block, least bin shift
	static const int env_wpl = [] { const char *e = getenv("MM355_CULL_WPL"); return e? atoi(e) : 0; }();
	static const int env_nt = [] { const char *e = getenv("MM355_CULL_NT"); return e? atoi(e) : 0; }();
	static const int env_sh = [] { const char *e = getenv("MM355_CULL_SH"); return e? atoi(e) : 0; }();
	static const int max_pass = [] { const char *e = getenv("MM355_CULL_MAX_PASS"); return e && atoi(e) > 0? atoi(e) : 8; }();
	cp.wpl = env_wpl >= 256 && env_wpl <= CS_WPL_MAX? (env_wpl & ~3) : CS_WPL_DEFAULT;
	cp.bpp = cp.wpl * 32 - 2 * CS_GUARD;
	const int cull_nt = env_nt == 256 || env_nt == 512? env_nt : 1024;
	if (do_cull) {
		cp.sh = 0; while ((1LL << cp.sh) < D) ++cp.sh;
		if (env_sh > cp.sh) cp.sh = env_sh;
		for (;; ++cp.sh) {   // at most max_pass passes over a read's anchors: wider bins beyond that
			const uint64_t bins = ((2 * cp.tot_len) >> cp.sh) + 1;
			cp.n_pass = (int)((bins + cp.bpp - 1) / cp.bpp);
			if (cp.n_pass <= max_pass) break;
		}
	}
	const size_t nr = (size_t)n_reads;
	if (c->n_keep.ensure(nr * 4 + 64) || c->aoff2.ensure((nr + 1) * 8 + 64) || c->sort_flag.ensure(nr + 64) || c->cs_list.ensure(nr * 4 + (nr + 1) * 8 + 64)) return MM355_ENOMEM;
	if (c->h_cs.ensure(nr * 4 + (nr + 1) * 8 + nr + nr * 4 + 256)) return MM355_ENOMEM;
	uint64_t *keys = c->z.as<uint64_t>(), *surv = c->u.as<uint64_t>();
	const int64_t *aoff = c->aoff.as<int64_t>();
	int32_t *d_nk = c->n_keep.as<int32_t>();
	const double t0 = mm355_now_ms();
	if (do_cull) {
		if (hipFuncSetAttribute((const void*)k_cull, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * CS_WPL_MAX * 4) != hipSuccess) return MM355_EHIP;
		KtScope ks(c, KT_CULL, c->st);
		hipLaunchKernelGGL(k_cull, dim3((unsigned)n_reads), dim3(cull_nt), 3 * cp.wpl * 4, c->st, c->dix, aoff, c->a.as<mm128>(), keys, surv, d_nk, n_reads, cp, c->heavy.as<int32_t>());
	} else hipLaunchKernelGGL(k_keys_all, dim3((unsigned)n_reads), dim3(256), 0, c->st, c->dix, aoff, c->a.as<mm128>(), surv, d_nk, n_reads, cp);
	// pinned staging: [n_keep: nr x i32][aoff2: (nr + 1) x i64][flags: nr x u8][lists: nr x i32]
	int32_t *h_nk = (int32_t*)c->h_cs.p;
	int64_t *h_off2 = (int64_t*)((char*)c->h_cs.p + ((nr * 4 + 63) & ~(size_t)63));
	uint8_t *h_flag = (uint8_t*)(h_off2 + nr + 1);
	int32_t *h_list = (int32_t*)((char*)h_flag + ((nr + 63) & ~(size_t)63));
	HIPCHK(hipMemcpyAsync(h_nk, d_nk, nr * 4, hipMemcpyDeviceToHost, c->st));
	HIPCHK(mm355_wait_stream(c->st));
	mm355_trace_add(c, "s:cull", t0, mm355_now_ms());
	int64_t tk = 0;
	int n_small = 0, n_mid = 0, n_big = 0;
	for (int i = 0; i < n_reads; ++i) { h_off2[i] = tk; tk += h_nk[i]; if (h_nk[i] > CS_MID_CAP) ++n_big; else if (h_nk[i] > CS_SMALL_CAP) ++n_mid; else if (h_nk[i] > 0) ++n_small; }
	h_off2[n_reads] = tk;
	{   // three size classes (16 KB / 64 KB / 128 KB of LDS per block): small reads in index order, the others by size (longest first)
		int is = 0, ib2 = n_small, im = n_small + n_big;   // [small][big][mid]
		for (int i = 0; i < n_reads; ++i) { if (h_nk[i] > CS_MID_CAP) h_list[ib2++] = i; else if (h_nk[i] > CS_SMALL_CAP) h_list[im++] = i; else if (h_nk[i] > 0) h_list[is++] = i; }
		std::stable_sort(h_list + n_small, h_list + n_small + n_big, [&](int32_t x, int32_t y) { return h_nk[x] > h_nk[y]; });
		std::stable_sort(h_list + n_small + n_big, h_list + n_small + n_big + n_mid, [&](int32_t x, int32_t y) { return h_nk[x] > h_nk[y]; });
	}
	// No 128-KB class by default: reads with more than CS_MID_CAP survivors go through the 64-KB class and its HBM stages (two blocks per CU, and
	// no block that waits for a whole free CU beside the other contexts' kernels: k_asort 84 -> 52 ms per step in the bench, 1380 against 1351
	// Mbases/s over three alternating pairs).  MM355_ASORT_BIG=1 brings the <1024, 16384> class back.
	static const bool big_class = [] { const char *e = getenv("MM355_ASORT_BIG"); return e && atoi(e) != 0; }();
	if (!big_class) { n_mid += n_big; n_big = 0; }
	if (c->b.ensure(((size_t)tk + 64) * 16)) return MM355_ENOMEM;
	int32_t *d_list = c->cs_list.as<int32_t>();
	int64_t *d_off2 = c->aoff2.as<int64_t>();
	HIPCHK(hipMemcpyAsync(d_off2, h_off2, (nr + 1) * 8, hipMemcpyHostToDevice, c->st));
	if (n_small + n_mid + n_big) HIPCHK(hipMemcpyAsync(d_list, h_list, (size_t)(n_small + n_mid + n_big) * 4, hipMemcpyHostToDevice, c->st));
	const double t1 = mm355_now_ms();
	HIPCHK(hipMemsetAsync(c->sort_flag.p, 0, nr, c->st));   // (reads without survivors are not listed: their flag stays 0)
	mm355_kt(c, KT_ASORT, 0, c->st);
	if (n_small) hipLaunchKernelGGL((k_asort<256, CS_SMALL_CAP>), dim3((unsigned)n_small), dim3(256), CS_SMALL_CAP * 8, c->st, d_list, n_small, aoff, d_off2, d_nk, c->a.as<mm128>(), surv,
	                                c->b.as<mm128>(), c->sort_flag.as<uint8_t>(), cp.ib);
	// (per call, not once per process: the attribute belongs to the function on the CURRENT device, and one process may drive several)
	if (hipFuncSetAttribute((const void*)k_asort<1024, CS_BIG_CAP>, hipFuncAttributeMaxDynamicSharedMemorySize, CS_BIG_CAP * 8) != hipSuccess) return MM355_EHIP;
	if (n_mid) {   // (64 KB: two blocks per CU, and a block that does not need a whole CU to itself beside the other contexts' kernels)
		if (hipFuncSetAttribute((const void*)k_asort<512, CS_MID_CAP>, hipFuncAttributeMaxDynamicSharedMemorySize, CS_MID_CAP * 8) != hipSuccess) return MM355_EHIP;
		hipLaunchKernelGGL((k_asort<512, CS_MID_CAP>), dim3((unsigned)n_mid), dim3(512), CS_MID_CAP * 8, c->st, d_list + n_small + n_big, n_mid, aoff, d_off2, d_nk, c->a.as<mm128>(), surv,
		                   c->b.as<mm128>(), c->sort_flag.as<uint8_t>(), cp.ib);
	}
	if (n_big) {
		hipLaunchKernelGGL((k_asort<1024, CS_BIG_CAP>), dim3((unsigned)n_big), dim3(1024), CS_BIG_CAP * 8, c->st, d_list + n_small, n_big, aoff, d_off2, d_nk, c->a.as<mm128>(), surv,
		                   c->b.as<mm128>(), c->sort_flag.as<uint8_t>(), cp.ib);
	}
	mm355_kt(c, KT_ASORT, 1, c->st);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(h_flag, c->sort_flag.p, nr, hipMemcpyDeviceToHost, c->st));
	HIPCHK(mm355_wait_stream(c->st));
	mm355_trace_add(c, "s:sort", t1, mm355_now_ms());
	// ---- reads whose survivors contain equal positions: literal radix_sort_128x of the whole generation-order array (on a copy)
	int n_tie = 0; int64_t tt = 0;
	for (int i = 0; i < n_reads; ++i) if (h_nk[i] > 0 && h_flag[i]) { ++n_tie; tt += hb.n_a[i]; }
	c->stats.n_sort_fast_reads += n_reads; c->stats.n_sort_tie_reads += n_tie;
	c->stats.n_a_kept += tk; c->stats.n_a_literal += tt;
	if (n_tie) {
		const double t2 = mm355_now_ms();
		const size_t na = (size_t)tt + 64;
		if (c->tie_a.ensure(na * 16) || c->tie_b.ensure(na * 16) || c->tie_f.ensure(na * 4) || c->tie_p.ensure(na * 4) || c->tie_t8.ensure(na) || c->tie_tcnt.ensure(na * 4) ||
		    c->tie_list.ensure((size_t)n_tie * 4 + ((size_t)n_tie + 1) * 8 + 64)) return MM355_ENOMEM;
		if (c->h_tasks.ensure((size_t)n_tie * sizeof(SortTask) + (size_t)n_tie * 4 + ((size_t)n_tie + 1) * 8 + 256)) return MM355_ENOMEM;
		// pinned: [tasks][toff: (n_tie + 1) x i64][list: n_tie x i32]
		SortTask *ht = (SortTask*)c->h_tasks.p;
		int64_t *h_toff = (int64_t*)((char*)c->h_tasks.p + (((size_t)n_tie * sizeof(SortTask) + 63) & ~(size_t)63));
		int32_t *h_tl = (int32_t*)(h_toff + n_tie + 1);
		const int big_min = mm355_sort_heavy_threshold(), med_min = mm355_sort_medium_threshold();
		int nb = 0, nm = 0, ns = 0, k = 0; int64_t to = 0;
		for (int i = 0; i < n_reads; ++i) if (h_nk[i] > 0 && h_flag[i]) { h_tl[k] = i; h_toff[k] = to; to += hb.n_a[i]; ++k; const int v = hb.n_a[i]; if (v > big_min) ++nb; else if (v > med_min) ++nm; else ++ns; }
		h_toff[n_tie] = to;
		{
			int ibg = 0, im = nb, is = nb + nm;
			for (int t = 0; t < n_tie; ++t) {
				const int v = hb.n_a[h_tl[t]];
				SortTask tk2; tk2.read = t; tk2.beg = 0; tk2.end = (uint32_t)v; tk2.s = 56;
				if (v > big_min) ht[ibg++] = tk2; else if (v > med_min) ht[im++] = tk2; else ht[is++] = tk2;
			}
			std::stable_sort(ht, ht + nb, [](const SortTask &x, const SortTask &y) { return x.end > y.end; });
		}
		int64_t *d_toff = c->tie_list.as<int64_t>(); int32_t *d_tl = (int32_t*)(d_toff + n_tie + 1);
		HIPCHK(hipMemcpyAsync(d_toff, h_toff, ((size_t)n_tie + 1) * 8, hipMemcpyHostToDevice, c->st));
		HIPCHK(hipMemcpyAsync(d_tl, h_tl, (size_t)n_tie * 4, hipMemcpyHostToDevice, c->st));
		mm355_kt(c, KT_TIE_AUX, 0, c->st);
		hipLaunchKernelGGL(k_tie_copy, dim3((unsigned)n_tie), dim3(256), 0, c->st, d_tl, n_tie, aoff, d_toff, c->a.as<mm128>(), keys, c->tie_a.as<mm128>(), do_cull? 0 : 1);
		// the plain sort of the WHOLE array of these reads (with the cull off the survivors are the whole array, sorted already): where its equal
		// positions are, so that the emulation only descends into the buckets that hold some (MM355_TIE_SKIP=0: it sorts everything)
		static const bool tie_skip = [] { const char *e = getenv("MM355_TIE_SKIP"); return !(e && atoi(e) == 0); }();
		const uint64_t *full_sorted = do_cull? keys : surv;
		if (do_cull && big_class) hipLaunchKernelGGL((k_asort<1024, CS_BIG_CAP>), dim3((unsigned)n_tie), dim3(1024), CS_BIG_CAP * 8, c->st, d_tl, n_tie, aoff, d_off2, (const int32_t*)0, c->a.as<mm128>(), keys,
		                                             (mm128*)0, (uint8_t*)0, cp.ib);
		else if (do_cull) {
			if (hipFuncSetAttribute((const void*)k_asort<512, CS_MID_CAP>, hipFuncAttributeMaxDynamicSharedMemorySize, CS_MID_CAP * 8) != hipSuccess) return MM355_EHIP;
			hipLaunchKernelGGL((k_asort<512, CS_MID_CAP>), dim3((unsigned)n_tie), dim3(512), CS_MID_CAP * 8, c->st, d_tl, n_tie, aoff, d_off2, (const int32_t*)0, c->a.as<mm128>(), keys,
			                   (mm128*)0, (uint8_t*)0, cp.ib);
		}
		if (tie_skip) hipLaunchKernelGGL(k_tie_tcnt, dim3((unsigned)n_tie), dim3(256), 0, c->st, d_tl, n_tie, aoff, d_toff, full_sorted, c->tie_tcnt.as<int32_t>(), cp.ib, do_cull? 1 : 0);
		mm355_kt(c, KT_TIE_AUX, 1, c->st);
		DevAnchors at; memset(&at, 0, sizeof(at));
		at.aoff = d_toff; at.a = c->tie_a.as<mm128>(); at.b = c->tie_b.as<mm128>(); at.f = c->tie_f.as<int32_t>(); at.p = c->tie_p.as<int32_t>(); at.t8 = c->tie_t8.as<uint8_t>(); at.tcnt = tie_skip? c->tie_tcnt.as<int32_t>() : 0;
		const size_t task_cap = (size_t)tt / 64 + (size_t)n_tie + 1024;
		if (c->sort_tasks.ensure(mm355_sort_buf_bytes(task_cap))) return MM355_ENOMEM;
		DevBatch bt; memset(&bt, 0, sizeof(bt));
		if (mm355_launch_sort(bt, at, c->err.as<int>(), ht, nb, nm, ns, (size_t)tt, c->sort_tasks.p, task_cap, c->st, c, mm355_sort_levels(c->mi))) return MM355_EHIP;
		hipLaunchKernelGGL(k_tie_emit, dim3((unsigned)n_tie), dim3(256), 0, c->st, d_tl, n_tie, aoff, d_toff, d_off2, c->a.as<mm128>(), c->tie_a.as<mm128>(), full_sorted, c->b.as<mm128>(), cp.ib, do_cull && tie_skip? 1 : 0);
		HIPCHK(hipGetLastError());
		HIPCHK(mm355_wait_stream(c->st));   // (the pinned lists above are reused by the next call)
		mm355_trace_add(c, "s:levels", t2, mm355_now_ms());
	}
	// ---- the culled, sorted array becomes the batch's anchor array
	std::swap(c->a, c->b); std::swap(c->aoff, c->aoff2);   // (both pairs are sized anew by every call; n_a is sized per resident batch: copied, not swapped)
	HIPCHK(hipMemcpyAsync(c->n_a.p, d_nk, nr * 4, hipMemcpyDeviceToDevice, c->st));
	for (int i = 0; i < n_reads; ++i) { hb.n_a[i] = h_nk[i]; hb.aoff[i] = h_off2[i]; }
	hb.aoff[n_reads] = tk; hb.tot_a = tk;
	return 0;
}
