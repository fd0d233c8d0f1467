// mm355_wave.h -- single-wave building blocks shared by the per-read kernels (mm355_kernels.hip, mm355_rmq.hip):
// the literal U:ksort.h::radix_sort_128x emulation for one wave (LDS-staged, label walk) and the DPP scans / reductions.
#pragma once
#include <hip/hip_runtime.h>
#include "mm355_dev.h"

#ifndef WAVE
#define WAVE 64
#endif
#ifndef LANE_LT_MASK
#define LANE_LT_MASK(lane) ((lane) == 0? 0ULL : (~0ULL >> (64 - (lane))))
#endif

// ------------------------------------------------------------------ literal radix_sort_128x, one wave
#define RS_STK 1024
struct SortLds {
	uint32_t cnt[256], bb[256], be[256];
	uint32_t cur[256], fend[256], arr[256], abef[256], fst[256];   // label-walk state (wave_rs_level_walk)
	uint32_t stk_beg[RS_STK], stk_end[RS_STK];
	uint32_t stk_n, overflow;
};

// per-read HBM scratch of the parallel permutation (all indexed like the array being sorted)
// tcnt (optional): running count of equal-key neighbour pairs of the SORTED array (mm355_fastsort.hip).  A bucket [b, e) without such a
// pair has a unique sorted content that the caller restores from the plain sort afterwards, so the literal recursion skips it.
struct WalkScratch { void *out; uint32_t *fpos; uint32_t *rank; uint8_t *flab; const int32_t *tcnt; };
__device__ inline bool ws_has_tie(const int32_t *tc, uint32_t b, uint32_t e) { return tc == 0 || tc[e - 1] - tc[b] > 0; }

// The sequential part of a level (see wave_rs_level_walk): one thread follows the cycles over the 1-byte label queues.  The state of a
// bucket is ONE LDS word cn[c] = cursor << 8 | label of the element at the cursor, so a step of the chase is a single dependent LDS read
// (cn of the bucket the popped element goes to); the refill of cn (label of the next element), the arrival counter and the rank store are
// issued beside it.  A foreign element never carries the label of its own region, so cn[g2] may be read before cn[g] is written back.
// lab[] must be readable one entry past the last foreign element.
__device__ inline void rs_walk_packed(uint32_t *cn, const uint32_t *fend, uint32_t *arr, uint32_t *abef, const uint8_t *lab, uint32_t *rank)
{
	for (uint32_t k = 0; k < 256; ++k) {
		abef[k] = arr[k];
		uint32_t ck = cn[k] >> 8;
		const uint32_t fk = fend[k];
		while (ck < fk) {
			uint32_t e = ck++, g = lab[e];
			uint32_t w = cn[g], r = arr[g];
			while (g != k) {
				const uint32_t e2 = w >> 8, g2 = w & 255u;
				const uint32_t nl = lab[e2 + 1];
				const uint32_t w2 = cn[g2], r2 = arr[g2];      // g2 != g
				arr[g] = r + 1; rank[e] = r;
				cn[g] = (e2 + 1) << 8 | nl;
				e = e2; g = g2; w = w2; r = r2;
			}
			arr[k] = r + 1; rank[e] = r;
		}
	}
}

// One level of the in-place cycle-leader permutation of rs_sort, reproduced WITHOUT moving elements one by one.
// The permutation only depends on the byte labels: inside the region R_k of bucket k an element is "home" (label k) or
// "foreign".  Foreign elements leave their region in position order; an element arriving at bucket l before l's own
// turn is inserted at l's cursor and pushes the following run of home elements right by one, an element arriving during
// l's turn fills the next foreign slot.  So the final position of every element follows from (a) the order in which
// foreign elements arrive at each bucket and (b) how many arrive before the bucket's turn.  (a)/(b) are produced by a
// sequential walk over the 1-byte label queues only (lane 0, labels in LDS); everything else -- histogram, compaction
// of foreign elements, the final scatter -- is done by all 64 lanes with coalesced traffic.  The argument is checked on the
// CPU against the literal loop (tests/test_level_walk_model.py: exhaustively for short arrays, random ones of every bucket
// structure), the kernel itself bit for bit through the anchor parity tests.
template <typename T, typename Key>
__device__ void wave_rs_level_walk(T *a, uint32_t beg, uint32_t end, int s, Key key, SortLds *L, const WalkScratch &ws, uint8_t *lds_lab, uint32_t lds_cap)
{
	const uint32_t lane = threadIdx.x & 63, tot = end - beg;
	if (lane == 0) { uint32_t acc = 0; for (int k = 0; k < 256; ++k) { L->bb[k] = acc; acc += L->cnt[k]; L->be[k] = acc; } }
	__syncthreads();
	T *out = (T*)ws.out + beg;
	uint32_t *fpos = ws.fpos + beg, *rank = ws.rank + beg;
	uint8_t *flab = ws.flab + beg;
	uint32_t nfor = 0;
	for (uint32_t base = 0; base < tot; base += WAVE) {
		const uint32_t rel = base + lane;
		uint32_t g = 0; bool foreign = false;
		if (rel < tot) { g = (uint32_t)(key(a[beg + rel]) >> s) & 255u; foreign = !(rel >= L->bb[g] && rel < L->be[g]); }
		const unsigned long long mask = __ballot(foreign);
		if (foreign) { const uint32_t e = nfor + __popcll(mask & LANE_LT_MASK(lane)); fpos[e] = rel; flab[e] = (uint8_t)g; if (e < lds_cap) lds_lab[e] = (uint8_t)g; }
		nfor += __popcll(mask);
	}
	__syncthreads();
	for (uint32_t k = lane; k < 256; k += WAVE) {   // first foreign slot at or after the start of region k
		uint32_t lo = 0, hi = nfor; const uint32_t target = L->bb[k];
		while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (fpos[mid] < target) lo = mid + 1; else hi = mid; }
		L->cur[k] = lo; L->fst[k] = lo; L->arr[k] = 0;
	}
	__syncthreads();
	for (uint32_t k = lane; k < 256; k += WAVE) L->fend[k] = k < 255? L->fst[k + 1] : nfor;
	__syncthreads();
	unsigned long long ne = 0;
	for (uint32_t k = lane; k < 256; k += WAVE) ne += __popcll(__ballot(L->cnt[k] != 0));
	if (ne == 2) {   // two non-empty buckets: closed form, see k_sort_level_mw
		const uint32_t m = nfor >> 1;
		for (uint32_t e = lane; e < nfor; e += WAVE) rank[e] = e < m? e : e - m;
		for (uint32_t k = lane; k < 256; k += WAVE) L->abef[k] = (L->cnt[k] != 0 && L->bb[k] != 0)? m : 0;
	} else if (nfor < lds_cap) {   // the only sequential part: one step per foreign element, on 1-byte labels
		for (uint32_t k = lane; k < 256; k += WAVE) { const uint32_t f0 = L->fst[k]; L->cur[k] = f0 << 8 | lds_lab[f0]; }
		__syncthreads();
		if (lane == 0) rs_walk_packed(L->cur, L->fend, L->arr, L->abef, lds_lab, rank);
	} else if (lane == 0) {
		const bool in_lds = false;
		for (uint32_t k = 0; k < 256; ++k) {
			L->abef[k] = L->arr[k];
			while (L->cur[k] < L->fend[k]) {
				uint32_t c = k;
				do {
					const uint32_t e = L->cur[c]++;
					const uint32_t g = in_lds? lds_lab[e] : flab[e];
					rank[e] = L->arr[g]++;
					c = g;
				} while (c != k);
			}
		}
	}
	__syncthreads();
	uint32_t nfb = 0;
	for (uint32_t base = 0; base < tot; base += WAVE) {
		const uint32_t rel = base + lane;
		uint32_t g = 0; bool foreign = false; T el;
		if (rel < tot) { el = a[beg + rel]; g = (uint32_t)(key(el) >> s) & 255u; foreign = !(rel >= L->bb[g] && rel < L->be[g]); }
		const unsigned long long mask = __ballot(foreign);
		const uint32_t pre = nfb + __popcll(mask & LANE_LT_MASK(lane));
		if (rel < tot) {
			uint32_t dest;
			const uint32_t al = L->abef[g], f0 = L->fst[g];
			if (foreign) {
				const uint32_t r = rank[pre];
				dest = r < al? (r == 0? L->bb[g] : fpos[f0 + r - 1] + 1) : fpos[f0 + r];
			} else dest = rel + ((pre - f0) < al? 1u : 0u);
			out[dest] = el;
		}
		nfb += __popcll(mask);
	}
	__syncthreads();
	for (uint32_t i = lane; i < tot; i += WAVE) a[beg + i] = out[i];
	__syncthreads();
}

template <typename T, typename Key>
__device__ void wave_rank_sort_small(T *a, uint32_t n, Key key)   // n <= 64: stable == rs_insertsort's result
{
	const uint32_t lane = threadIdx.x & 63;
	T mine; uint64_t kx = 0;
	if (lane < n) { mine = a[lane]; kx = key(mine); }
	uint32_t rank = 0;
	for (uint32_t j = 0; j < n; ++j) {
		uint64_t kj = __shfl(kx, (int)j);
		rank += (kj < kx) || (kj == kx && j < lane);
	}
	__syncthreads();
	if (lane < n) a[rank] = mine;
	__syncthreads();
}

// sort a[0..n0) starting at byte shift s0, exactly as rs_sort_128x(beg,end,8,s0) would.
template <bool STAGE, typename T, typename Key>
__device__ void wave_rs_core(T *a, uint32_t n0, int s0, Key key, SortLds *L, T *stage, uint32_t stage_cap, const WalkScratch *ws = 0)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t base = L->stk_n;
	__syncthreads();
	if (lane == 0) { L->stk_beg[base] = 0; L->stk_end[base] = n0 | ((uint32_t)(s0 >> 3) << 28); L->stk_n = base + 1; }
	__syncthreads();
	for (;;) {
		uint32_t sn = L->stk_n;
		if (sn <= base) break;
		uint32_t beg = L->stk_beg[sn - 1], e = L->stk_end[sn - 1];
		uint32_t end = e & 0x0fffffffu, tot = end - beg;
		int s = (int)(e >> 28) * 8;
		__syncthreads();
		if (lane == 0) L->stk_n = sn - 1;
		__syncthreads();
		if (STAGE && tot <= stage_cap) {   // the whole sub-problem fits the LDS stage: finish it there
			for (uint32_t i = lane; i < tot; i += WAVE) stage[i] = a[beg + i];
			__syncthreads();
			wave_rs_core<false>(stage, tot, s, key, L, (T*)0, 0u);
			__syncthreads();
			for (uint32_t i = lane; i < tot; i += WAVE) a[beg + i] = stage[i];
			__syncthreads();
			continue;
		}
		for (uint32_t i = lane; i < 256; i += WAVE) L->cnt[i] = 0;
		__syncthreads();
		for (uint32_t i = beg + lane; i < end; i += WAVE) atomicAdd(&L->cnt[(uint32_t)(key(a[i]) >> s) & 255u], 1u);
		__syncthreads();
		bool single = false;
		for (uint32_t i = lane; i < 256; i += WAVE) if (L->cnt[i] == tot) single = true;
		if (__any(single)) {   // one bucket holds everything: the cycle-leader pass is the identity
			if (s > 0 && lane == 0) {
				uint32_t slot = L->stk_n;
				L->stk_beg[slot] = beg; L->stk_end[slot] = end | ((uint32_t)((s - 8) >> 3) << 28); L->stk_n = slot + 1;
			}
			__syncthreads();
			continue;
		}
		if (STAGE && ws) wave_rs_level_walk(a, beg, end, s, key, L, *ws, (uint8_t*)stage, (uint32_t)(stage_cap * sizeof(T)));
		else if (lane == 0) mm_rs_permute(a + beg, (int64_t)tot, s, L->cnt, L->bb, L->be, key);
		__syncthreads();
		if (s > 0) {
			uint32_t s2 = (uint32_t)((s - 8) >> 3);
			for (uint32_t k = lane; k < 256; k += WAVE) {
				uint32_t b0 = L->bb[k], sz = L->cnt[k];
				if (sz > 1 && !ws_has_tie(ws? ws->tcnt : 0, beg + b0, beg + b0 + sz)) continue;   // unique content, restored by the caller
				if (sz > MM355_RS_MIN_SIZE) {
					uint32_t slot = atomicAdd(&L->stk_n, 1u);
					if (slot < RS_STK) { L->stk_beg[slot] = beg + b0; L->stk_end[slot] = (beg + b0 + sz) | (s2 << 28); }
					else L->overflow = 1;
				} else if (sz > 1) mm_rs_insertsort(a + beg + b0, a + beg + b0 + sz, key);
			}
		}
		__syncthreads();
		if (L->overflow) { if (lane == 0 && L->stk_n > RS_STK) L->stk_n = RS_STK; __syncthreads(); }
	}
	__syncthreads();
}

template <typename T, typename Key>
__device__ void wave_radix_sort(T *a, uint32_t n, Key key, SortLds *L, T *stage, uint32_t stage_cap, const WalkScratch *ws = 0, int s0 = 56)
{
	// s0 < 56: the caller knows that no key has a bit at or above s0 + 8.  A level on which every key has the same digit is the identity
	// (one bucket holds everything, rs_sort's cycle-leader pass moves nothing and recurses into that bucket), so starting below such
	// levels gives the same array -- and saves a histogram pass over all n keys per level (chain scores: six of eight levels).
	if (n <= 1) return;
	if (n <= MM355_RS_MIN_SIZE) { wave_rank_sort_small(a, n, key); return; }
	if ((threadIdx.x & 63) == 0) { L->stk_n = 0; L->overflow = 0; }
	__syncthreads();
	wave_rs_core<true>(a, n, s0, key, L, stage, stage_cap, ws);
}

// Cross-lane scans and reductions by DPP register moves (row_shr / row_bcast): ~12 VALU operations, no LDS crossbar round trips
// (a __shfl is a ds_bpermute, >100 cycles each -- seven of them were the longest part of a chaining step)
#define DPP_I32(old, src, ctrl, rmask) __builtin_amdgcn_update_dpp((int)(old), (int)(src), (ctrl), (rmask), 0xf, false)
__device__ inline int32_t wave_incl_scan_max(int32_t x)   // inclusive prefix max over lanes 0..lane
{
	int32_t y;
	y = DPP_I32(INT32_MIN, x, 0x111, 0xf); x = x > y? x : y;   // row_shr:1
	y = DPP_I32(INT32_MIN, x, 0x112, 0xf); x = x > y? x : y;   // row_shr:2
	y = DPP_I32(INT32_MIN, x, 0x114, 0xf); x = x > y? x : y;   // row_shr:4
	y = DPP_I32(INT32_MIN, x, 0x118, 0xf); x = x > y? x : y;   // row_shr:8
	y = DPP_I32(INT32_MIN, x, 0x142, 0xa); x = x > y? x : y;   // row_bcast:15 into rows 1 and 3
	y = DPP_I32(INT32_MIN, x, 0x143, 0xc); x = x > y? x : y;   // row_bcast:31 into rows 2 and 3
	return x;
}
__device__ inline int32_t wave_incl_scan_add(int32_t x)   // inclusive prefix sum over lanes 0..lane
{
	x += DPP_I32(0, x, 0x111, 0xf); x += DPP_I32(0, x, 0x112, 0xf); x += DPP_I32(0, x, 0x114, 0xf); x += DPP_I32(0, x, 0x118, 0xf);
	x += DPP_I32(0, x, 0x142, 0xa); x += DPP_I32(0, x, 0x143, 0xc);
	return x;
}
__device__ inline int32_t wave_incl_scan_min(int32_t x)   // inclusive prefix minimum over lanes 0..lane
{
	int32_t y;
	y = DPP_I32(INT32_MAX, x, 0x111, 0xf); x = x < y? x : y;
	y = DPP_I32(INT32_MAX, x, 0x112, 0xf); x = x < y? x : y;
	y = DPP_I32(INT32_MAX, x, 0x114, 0xf); x = x < y? x : y;
	y = DPP_I32(INT32_MAX, x, 0x118, 0xf); x = x < y? x : y;
	y = DPP_I32(INT32_MAX, x, 0x142, 0xa); x = x < y? x : y;
	y = DPP_I32(INT32_MAX, x, 0x143, 0xc); x = x < y? x : y;
	return x;
}
__device__ inline int32_t wave_excl_prefix_max(int32_t v, int lane)   // exclusive prefix max over lanes, INT32_MIN identity
{
	(void)lane;
	const int32_t x = wave_incl_scan_max(v);
	return DPP_I32(INT32_MIN, x, 0x138, 0xf);   // wave_shr:1, lane 0 keeps the identity
}
__device__ inline int32_t wave_reduce_max(int32_t v) { return __builtin_amdgcn_readlane(wave_incl_scan_max(v), 63); }
__device__ inline long long wave_reduce_max64(long long v)   // maximum of a 64-bit key over the wave (uniform result)
{
	const int ctrl[6] = { 0x111, 0x112, 0x114, 0x118, 0x142, 0x143 }, rm[6] = { 0xf, 0xf, 0xf, 0xf, 0xa, 0xc };
#pragma unroll
	for (int k = 0; k < 6; ++k) {
		int lo = (int)(uint32_t)v, hi = (int)(v >> 32), ylo, yhi;
		switch (k) {   // the builtin wants immediate control words
		case 0: ylo = DPP_I32(0, lo, 0x111, 0xf); yhi = DPP_I32(INT32_MIN, hi, 0x111, 0xf); break;
		case 1: ylo = DPP_I32(0, lo, 0x112, 0xf); yhi = DPP_I32(INT32_MIN, hi, 0x112, 0xf); break;
		case 2: ylo = DPP_I32(0, lo, 0x114, 0xf); yhi = DPP_I32(INT32_MIN, hi, 0x114, 0xf); break;
		case 3: ylo = DPP_I32(0, lo, 0x118, 0xf); yhi = DPP_I32(INT32_MIN, hi, 0x118, 0xf); break;
		case 4: ylo = DPP_I32(0, lo, 0x142, 0xa); yhi = DPP_I32(INT32_MIN, hi, 0x142, 0xa); break;
		default: ylo = DPP_I32(0, lo, 0x143, 0xc); yhi = DPP_I32(INT32_MIN, hi, 0x143, 0xc); break;
		}
		(void)ctrl; (void)rm;
		const long long y = (long long)(((unsigned long long)(uint32_t)yhi << 32) | (uint32_t)ylo);
		v = v > y? v : y;
	}
	const uint32_t rlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63), rhi = (uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), 63);
	return (long long)(((unsigned long long)rhi << 32) | rlo);
}

struct key_hi32 { __host__ __device__ uint64_t operator()(const uint64_t &v) const { return v >> 32; } };
