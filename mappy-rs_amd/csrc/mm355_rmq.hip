// mm355_rmq.hip -- row a9 of SURVEY.md section 8: U:lchain.c::mg_lchain_rmq (+ U:krmq.h) on the device.
// Reference call sites (un-vendored minimap2 2.26, reached through mm_map at /root/reference/src/lib.rs:482 and :587):
//   * the long-join re-chain of U:map.c::mm_map_frag: when a read has more than one chain and the first chain leaves more than
//     rmq_rescue_size query bases uncovered (or covers more than rmq_rescue_ratio of the read), its chained anchors are sorted by
//     reference position (radix_sort_128x) and chained again with bw_long by mg_lchain_rmq;
//   * MM_F_RMQ presets (asm5 / asm10 / asm20): mg_lchain_rmq is the primary chainer over all sorted anchors.
//
// What mg_lchain_rmq computes for anchor i (anchors sorted by x = strand | rid | ref position):
//   outer set  = anchors j in [st, i0): i0 = start of i's run of equal x (an anchor only becomes visible once x has moved on),
//                st = first anchor of the same strand / contig within max_dist of x_i (and at most cap_rmq_size behind i0);
//   candidate  = the element of the outer set with y in (y_i - max_dist, y_i) (or y == y_i for j == 0: the closed upper end of the key
//                range (y_i, 0)) that minimises pri_j = -(f_j + 0.5 * pen_gap * (x_j + y_j)) -- a double;
//   inner walk = unless that candidate continues i exactly, the anchors of [st_inner, i0) with y in [y_i - max_dist_inner, y_i - 1], in
//                descending (y, j) order, with the skip / mark heuristic of mg_lchain_dp.
// The reference keeps both sets in AVL trees; only two things of a tree are observable: the set it holds (above) and, for the outer
// tree, WHICH element a range-minimum query returns when two elements share the minimal priority -- that depends on the tree's shape
// and on the order in which its subtree minima were refreshed.  This kernel therefore answers the query by looking at every element of
// the window (64 per step) and PROVES the answer unique: if the minimal priority is attained twice the read is handed to the literal
// host implementation (flag 2), like every other capacity limit below.  Everything else is order-free or reproduced literally.
//
// One wave per read.  State of the last RQ_RING anchors (y, pri, f, p) lives in LDS rings so that the loop-carried dependence never
// waits for HBM; the inner set is a sorted key array in LDS (insert / erase shift it with all 64 lanes); marks t[] are an LDS window.
#include <hip/hip_runtime.h>
#include "mm355_btcore.h"
#include "mm355_rmq.h"

#define MM355_LATENCY_KERNEL() __builtin_amdgcn_s_setprio(3)
#define RQ_RING 1024
#define RQ_RMASK (RQ_RING - 1)
#define RQ_INN 1024                 // capacity of the inner set
#define RQ_TW 2048                  // mark window (> RQ_RING + one wave)
#define RQ_TWMASK (RQ_TW - 1)
#define RQ_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
#define A_STAGE 2048

// ---- U:lchain.c::comput_sc_simple
__device__ inline int32_t rq_comput_sc(uint64_t xi, uint64_t yi, uint64_t xj, uint64_t yj, float pen_gap, float pen_skip, bool *exact, int32_t *width)
{
	const int32_t dq = (int32_t)yi - (int32_t)yj, dr = (int32_t)(xi - xj);
	const int32_t dd = dr > dq? dr - dq : dq - dr, dg = dr < dq? dr : dq;
	const int32_t q_span = (int32_t)(yj >> 32 & 0xff);
	int32_t sc = q_span < dg? q_span : dg;
	*width = dd;
	*exact = dd == 0 && dg <= q_span;
	if (dd || dq > q_span) {
		float a1 = pen_gap * (float)dd, a2 = pen_skip * (float)dg;
		const float lin_pen = a1 + a2;
		const float log_pen = dd >= 1? mm_log2f_approx((float)(dd + 1)) : 0.0f;
		a1 = .5f * log_pen;
		a1 = lin_pen + a1;
		sc -= (int)a1;
	}
	return sc;
}

__device__ inline uint64_t rq_bcast64(uint64_t v, int l)
{
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l);
	return (uint64_t)hi << 32 | lo;
}
// loads of arrays this wave has written itself (f, p, y, pri older than the rings): served by L2, never by a stale L1 line
__device__ inline int32_t rq_ld32(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double rq_ldf64(const double *p)
{
	const long long v = __hip_atomic_load((const long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return __longlong_as_double(v);
}

// ---- sorted key array in LDS (the reference's root_inner): keys (int64)y << 32 | j, ascending
__device__ inline int rq_count(const long long *inn, int n, long long key, bool or_equal)   // #entries < key (<= key)
{
	const int lane = threadIdx.x & 63;
	int pos = 0;
	for (int base = 0; base < n; base += WAVE) {
		const int k = base + lane;
		const bool c = k < n && (or_equal? inn[k] <= key : inn[k] < key);
		const unsigned long long m = __ballot(c);
		pos += __popcll(m);
		if (m != ~0ULL) break;
	}
	return pos;
}
__device__ inline void rq_insert(long long *inn, int &n, long long key)
{
	const int lane = threadIdx.x & 63;
	const int pos = rq_count(inn, n, key, false);
	if (n > pos) {
		for (int b = pos + ((n - 1 - pos) / WAVE) * WAVE; b >= pos; b -= WAVE) {   // top chunk first: its first slot is free when the chunk below writes into it
			const int k = b + lane;
			long long v = 0;
			if (k < n) v = inn[k];
			RQ_SYNC();
			if (k < n) inn[k + 1] = v;
			RQ_SYNC();
		}
	}
	if (lane == 0) inn[pos] = key;
	RQ_SYNC();
	++n;
}
__device__ inline void rq_erase(long long *inn, int &n, long long key)
{
	const int lane = threadIdx.x & 63;
	const int pos = rq_count(inn, n, key, false);
	if (pos >= n || inn[pos] != key) return;   // (U: krmq_find misses -> nothing erased; cannot happen for members)
	for (int b = pos + 1; b < n; b += WAVE) {
		const int k = b + lane;
		long long v = 0;
		if (k < n) v = inn[k];
		RQ_SYNC();
		if (k < n) inn[k - 1] = v;
		RQ_SYNC();
	}
	--n;
}

// ---- stage 1 (long-join re-chain only): the rescue test of U:map.c::mm_map_frag and radix_sort_128x of the chained anchors
__global__ __launch_bounds__(WAVE) void k_rmq_sort(RmqParams rp, DevBatch bt, DevAnchors an, const int32_t *list, uint8_t *flag, int *err)
{
	MM355_LATENCY_KERNEL();
	__shared__ SortLds L;
	__shared__ mm128 stage[A_STAGE];
	const int r = list[blockIdx.x], lane = threadIdx.x;
	const int64_t o = an.aoff[r];
	const int n_u = an.n_u[r], n_v = an.n_v[r];
	mm128 *a = an.a + o;
	bool go = false;
	if (n_u > 1) {
		const int32_t qlen = bt.rlen[r];
		const int32_t st = (int32_t)a[0].y, en = (int32_t)a[(int32_t)an.u[o] - 1].y;
		go = qlen - (en - st) > rp.rescue_size || (float)(en - st) > (float)qlen * rp.rescue_ratio;
	}
	if (lane == 0) flag[r] = go? MM355_RMQ_DONE : MM355_RMQ_KEEP;
	if (!go) return;
	WalkScratch ws; ws.out = an.b + o; ws.fpos = (uint32_t*)an.f + o; ws.rank = (uint32_t*)an.p + o; ws.flab = an.t8 + o; ws.tcnt = 0;   // f / p are rewritten by k_rmq_dp
	wave_radix_sort(a, (uint32_t)n_v, mm_key_x(), &L, stage, (uint32_t)A_STAGE, &ws);
	if (lane == 0 && n_v > MM355_RS_MIN_SIZE && L.overflow) *err = 1;
}

// ---- stage 2: the chaining recurrence
__global__ __launch_bounds__(WAVE) void k_rmq_dp(RmqParams rp, DevBatch bt, DevAnchors an, const int32_t *list, uint8_t *flag, unsigned long long *ctr)
{
	MM355_LATENCY_KERNEL();
	__shared__ long long inn[RQ_INN];
	__shared__ uint32_t tw[RQ_TW];
	__shared__ double rpri[RQ_RING];
	__shared__ int32_t ry[RQ_RING], rf[RQ_RING], rpp[RQ_RING];
	const int r = list[blockIdx.x], lane = threadIdx.x;
	if (flag[r] != MM355_RMQ_DONE) return;
	const int64_t o = an.aoff[r];
	const int n = rp.primary? (int)(an.aoff[r + 1] - o) : an.n_v[r];
	if (n <= 0) return;
	const mm128 *a = an.a + o;
	int32_t *f = an.f + o, *p = an.p + o, *yy = an.vi + o;
	double *pri = (double*)(an.z + o);
	int32_t max_dist = rp.max_dist, max_dist_inner = rp.max_dist_inner;
	const int32_t bw = rp.bw, max_skip = rp.max_chn_skip, cap = rp.cap;
	if (max_dist < bw) max_dist = bw;
	if (max_dist_inner < 0) max_dist_inner = 0;
	if (max_dist_inner > max_dist) max_dist_inner = max_dist;
	if (cap < 0) { if (lane == 0) flag[r] = MM355_RMQ_HOST; return; }   // (a negative cap empties the trees before anything is inserted: left to the literal code)
	const float pen_gap = rp.pen_gap, pen_skip = rp.pen_skip;
	const double half_gap = 0.5 * (double)pen_gap;
	for (int k = lane; k < RQ_TW; k += WAVE) tw[k] = 0;
	RQ_SYNC();
	int i0 = 0, st = 0, st_in = 0, inn_n = 0;
	uint64_t x_i0 = 0, cx = 0, cy = 0;
	unsigned long long n_scan = 0;
	bool bail = false;
	for (int i = 0; i < n; ++i) {
		if ((i & (WAVE - 1)) == 0) { cx = cy = 0; if (i + lane < n) { const mm128 t = a[i + lane]; cx = t.x; cy = t.y; } }
		const uint64_t xi = rq_bcast64(cx, i & (WAVE - 1)), yi = rq_bcast64(cy, i & (WAVE - 1));
		const int32_t yi32 = (int32_t)yi;
		// the anchors of the previous run of equal x become visible
		if (i == 0) x_i0 = xi;
		if (i0 < i && x_i0 != xi) {
			if (max_dist_inner > 0) {
				for (int j = i0; j < i; ++j) {
					if (inn_n >= RQ_INN) { bail = true; break; }
					rq_insert(inn, inn_n, (long long)(((uint64_t)(int64_t)ry[j & RQ_RMASK]) << 32 | (uint32_t)j));   // i - j <= run length: still in the ring (checked below)
				}
				if (bail) break;
			}
			i0 = i; x_i0 = xi;
		}
		// st: first anchor of the same strand / contig within max_dist; at most cap elements stay
		for (;;) {
			const int idx = st + lane;
			bool c = false;
			if (idx < i) { const uint64_t xs = a[idx].x; c = (xi >> 32 != xs >> 32) || xi > xs + (uint64_t)(int64_t)max_dist; }
			const unsigned long long m = __ballot(c);
			if (m == ~0ULL) { st += WAVE; continue; }
			st += __builtin_ctzll(~m);
			break;
		}
		if (st < i0 && i0 - st > cap) st = i0 - cap;
		if (max_dist_inner > 0) {
			int s2 = st_in;
			for (;;) {
				const int idx = s2 + lane;
				bool c = false;
				if (idx < i) { const uint64_t xs = a[idx].x; c = (xi >> 32 != xs >> 32) || xi > xs + (uint64_t)(int64_t)max_dist_inner; }
				const unsigned long long m = __ballot(c);
				if (m == ~0ULL) { s2 += WAVE; continue; }
				s2 += __builtin_ctzll(~m);
				break;
			}
			if (s2 < i0 && i0 - s2 > cap) s2 = i0 - cap;
			if (s2 >= i0) inn_n = 0;                       // every member left
			else if (s2 > st_in) {
				if (i - st_in > RQ_RING) { bail = true; break; }   // (members older than the ring: not reachable with the capacity checks below)
				for (int j = st_in; j < s2; ++j) rq_erase(inn, inn_n, (long long)(((uint64_t)(int64_t)ry[j & RQ_RMASK]) << 32 | (uint32_t)j));
			}
			st_in = s2;
			if (i - st_in > RQ_RING - WAVE) { bail = true; break; }   // the rings (and the mark window) must cover the inner set and the current run
		}
		// range-minimum query over the outer set, 64 elements per step
		const int32_t lo_y = yi32 - max_dist;
		double best = 0.0; int bj = -1; bool tie = false;
		const int near0 = i - RQ_RING > st? i - RQ_RING : st;   // [near0, i0) is served by the rings
		for (int j = st + lane; j < near0; j += WAVE) {
			const int32_t yj = rq_ld32(yy + j);
			if ((yj > lo_y && yj < yi32) || (yj == yi32 && j == 0)) {
				const double pj = rq_ldf64(pri + j);
				if (bj < 0 || pj < best) { best = pj; bj = j; tie = false; } else if (pj == best) tie = true;
			}
		}
		for (int j = near0 + lane; j < i0; j += WAVE) {
			const int32_t yj = ry[j & RQ_RMASK];
			if ((yj > lo_y && yj < yi32) || (yj == yi32 && j == 0)) {
				const double pj = rpri[j & RQ_RMASK];
				if (bj < 0 || pj < best) { best = pj; bj = j; tie = false; } else if (pj == best) tie = true;
			}
		}
		n_scan += (unsigned long long)(i0 - st);
		int32_t max_f = (int32_t)(yi >> 32 & 0xff), max_j = -1;
		{   // wave minimum of an order-preserving integer image of the double; the minimum must be attained exactly once
			long long sk = __double_as_longlong(best);
			sk = sk >= 0? sk : (sk ^ 0x7fffffffffffffffLL);
			const long long mine = bj >= 0? ~sk : (long long)0x8000000000000000ULL;
			const long long mx = wave_reduce_max64(mine);
			const unsigned long long who = __ballot(bj >= 0 && mine == mx);
			if (who) {
				if (__popcll(who) > 1 || __ballot(tie && mine == mx)) { bail = true; break; }
				const int wl = __builtin_ctzll(who);
				const int j = __builtin_amdgcn_readlane(bj, wl);
				const mm128 aj = a[j];
				const int32_t fj = i - j <= RQ_RING? rf[j & RQ_RMASK] : rq_ld32(f + j);
				bool exact; int32_t width;
				const int32_t sc = fj + rq_comput_sc(xi, yi, aj.x, aj.y, pen_gap, pen_skip, &exact, &width);
				if (width <= bw && sc > max_f) { max_f = sc; max_j = j; }
				if (!exact && inn_n > 0 && yi32 > 0) {
					// largest key <= (y_i - 1, n), then downwards while y >= y_i - max_dist_inner
					const int top = rq_count(inn, inn_n, (long long)(((uint64_t)(int64_t)(yi32 - 1)) << 32 | (uint32_t)n), true);
					int32_t n_skip = 0;
					const uint32_t mark = (uint32_t)i + 1u;
					for (int kb = top - 1; kb >= 0; kb -= WAVE) {
						const int k = kb - lane;
						bool active = k >= 0;
						long long key = 0;
						if (active) key = inn[k];
						const int32_t y2 = (int32_t)(key >> 32);
						const unsigned long long mstop = __ballot(active && y2 < yi32 - max_dist_inner);
						if (mstop) active = active && lane < __builtin_ctzll(mstop);
						const int jj = (int)(uint32_t)key;
						int32_t sc2 = 0, pj = -1, width2 = 0;
						bool valid = false;
						if (active) {
							const mm128 aj2 = a[jj];
							bool ex2;
							sc2 = rf[jj & RQ_RMASK] + rq_comput_sc(xi, yi, aj2.x, aj2.y, pen_gap, pen_skip, &ex2, &width2);
							valid = width2 <= bw;
							if (valid) pj = rpp[jj & RQ_RMASK];
						}
						if (valid && pj >= st_in) tw[pj & RQ_TWMASK] = mark;   // t[p[j]] = i; marks below st_inner are never tested
						RQ_SYNC();
						const bool marked = valid && tw[jj & RQ_TWMASK] == mark;
						const int32_t scv = valid? sc2 : INT32_MIN;
						int32_t pm = wave_excl_prefix_max(scv, lane);
						pm = pm > max_f? pm : max_f;
						const bool improved = valid && sc2 > pm;
						// n_skip along the scan order (see k_chain_big): +1 on a marked lane, -1 (not below 0) on an improving one
						const int32_t dstep = improved? -1 : marked? 1 : 0;
						const int32_t S = n_skip + wave_incl_scan_add(dstep);
						const int32_t m0 = wave_incl_scan_min(S);
						const int32_t ck = S - (m0 < 0? m0 : 0);
						const unsigned long long bmask = __ballot(dstep == 1 && ck > max_skip);
						int brk = -1;
						if (bmask) brk = __builtin_ctzll(bmask);
						else n_skip = __builtin_amdgcn_readlane(ck, 63);
						const bool considered = brk < 0 || lane <= brk;
						const int32_t cv = (valid && considered)? sc2 : INT32_MIN;
						const int32_t cmax = wave_reduce_max(cv);
						if (cmax > max_f) {
							const unsigned long long w = __ballot(cv == cmax);
							max_f = cmax; max_j = __builtin_amdgcn_readlane(jj, __builtin_ctzll(w));   // lowest lane = first met by the sequential walk
						}
						RQ_SYNC();
						if (brk >= 0 || mstop) break;
					}
				}
			}
		}
		const double pv = -((double)max_f + half_gap * (double)((int32_t)xi + yi32));
		RQ_SYNC();   // every lane has finished reading the ring slots this anchor overwrites
		if (lane == 0) {
			f[i] = max_f; p[i] = max_j; yy[i] = yi32; pri[i] = pv;
			rf[i & RQ_RMASK] = max_f; rpp[i & RQ_RMASK] = max_j; ry[i & RQ_RMASK] = yi32; rpri[i & RQ_RMASK] = pv;
		}
		RQ_SYNC();
	}
	if (bail) { if (lane == 0) flag[r] = MM355_RMQ_HOST; return; }
	if (lane == 0 && ctr) atomicAdd(ctr + (blockIdx.x & 63), n_scan);
}

// ---- stage 3: mg_chain_backtrack + compact_a of the re-chained reads (max_drop = the band width mg_lchain_rmq was given)
__global__ __launch_bounds__(WAVE) void k_rmq_backtrack(RmqParams rp, DevParams pr, DevBatch bt, DevAnchors an, const int32_t *list, const uint8_t *flag, int *err)
{
	MM355_LATENCY_KERNEL();
	__shared__ BtLds S;
	const int r = list[blockIdx.x];
	if (flag[r] != MM355_RMQ_DONE) return;
	const int n = rp.primary? (int)(an.aoff[r + 1] - an.aoff[r]) : an.n_v[r];
	wave_backtrack_read(pr, bt, an, err, &S, r, n, rp.bw);
}

int mm355_launch_rmq(const RmqParams &rp, const DevParams &pr, const DevBatch &bt, DevAnchors &an, const int32_t *d_list, int n_list, uint8_t *d_flag,
                     int *err, unsigned long long *ctr, hipStream_t st)
{
	if (n_list <= 0) return 0;
	if (!rp.primary) hipLaunchKernelGGL(k_rmq_sort, dim3(n_list), dim3(WAVE), 0, st, rp, bt, an, d_list, d_flag, err);
	hipLaunchKernelGGL(k_rmq_dp, dim3(n_list), dim3(WAVE), 0, st, rp, bt, an, d_list, d_flag, ctr);
	hipLaunchKernelGGL(k_rmq_backtrack, dim3(n_list), dim3(WAVE), 0, st, rp, pr, bt, an, d_list, (const uint8_t*)d_flag, err);
	return hipGetLastError() == hipSuccess? 0 : -1;
}
