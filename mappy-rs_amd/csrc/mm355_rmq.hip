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
#define RQ_RING 512                 // x, y, f, p of the last anchors: covers the inner set (checked) and, usually, the range-minimum winner
#define RQ_RMASK (RQ_RING - 1)
#define RQ_WRING 2048               // (y, pri) of the last anchors: the range-minimum window (20 kb of reference at ~0.07 chained anchors per base)
#define RQ_WMASK (RQ_WRING - 1)
#define RQ_NB (RQ_WRING / 64)       // block summaries of the (y, pri) ring
#define RQ_TW 1024                  // mark window (> RQ_RING + one wave)
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

// ---- stage 1 (long-join re-chain only): the rescue test of U:map.c::mm_map_frag and radix_sort_128x of the chained anchors
__device__ __forceinline__ void rmq_sort_read(const RmqParams &rp, const DevBatch &bt, const DevAnchors &an, const int32_t *list, uint8_t *flag, int *err, const unsigned int bid)
{
	__shared__ SortLds L;
	__shared__ mm128 stage[A_STAGE];
	const int r = list[bid], lane = threadIdx.x;
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
// Nothing on the loop-carried path of an anchor waits for HBM: the window starts st / st_inner are monotone functions of the sorted x and are
// computed for all anchors up front (binary searches, 64 anchors at a time); x, y, f, p of the last RQ_RING anchors and (y, pri) of the last
// RQ_WRING anchors live in LDS rings.  Only a range-minimum window longer than RQ_WRING anchors (or a winner further back than RQ_RING) reads HBM.
//
// Range minimum: the window [st, i0) is cut at multiples of 64.  Every complete block of 64 anchors has a summary (min y, max y, minimal pri, its
// index, "attained twice"); a block whose y range lies inside the query's is answered by its summary, one lane per block; only the two ragged
// ends of the window and the blocks that straddle a y bound are looked at element by element -- a 20 kb window of a long read is ~4 steps, not ~30.
//
// Inner walk: the reference walks the members of [st_inner, i0) with y in [y_i - max_dist_inner, y_i - 1] in descending (y, j) order.  Among
// anchors of one chain y grows with j, so descending j IS that order unless chains of different diagonals interleave; the walk runs over j,
// checks on the way that y never increases, and only when it does sorts the candidates of this anchor (rank by counting, in LDS).
struct RqWalk { int32_t max_f, max_j, n_skip; };
// one step of the inner walk: 64 candidates in walk order (lane 0 first), `valid` lanes take part.  Returns true when the walk stops here.
__device__ inline bool rq_walk_step(RqWalk &w, uint32_t *tw, uint32_t mark, int jj, bool valid, int32_t sc2, int32_t pj, int st_in, int32_t max_skip)
{
	const int lane = threadIdx.x & 63;
	if (valid && pj >= st_in) tw[pj & RQ_TWMASK] = mark;   // t[p[j]] = i; marks below st_inner are never tested
	RQ_SYNC();
	const bool marked = valid && tw[jj & RQ_TWMASK] == mark;
	const int32_t scv = valid? sc2 : INT32_MIN;
	int32_t pm = wave_excl_prefix_max(scv, lane);
	pm = pm > w.max_f? pm : w.max_f;
	const bool improved = valid && sc2 > pm;
	// n_skip along the walk (see k_chain_big): +1 on a marked lane, -1 (not below 0) on an improving one; stop at the first marked lane above max_skip
	const int32_t dstep = improved? -1 : marked? 1 : 0;
	const int32_t S = w.n_skip + wave_incl_scan_add(dstep);
	const int32_t m0 = wave_incl_scan_min(S);
	const int32_t ck = S - (m0 < 0? m0 : 0);
	const unsigned long long bmask = __ballot(dstep == 1 && ck > max_skip);
	int brk = -1;
	if (bmask) brk = __builtin_ctzll(bmask);
	else w.n_skip = __builtin_amdgcn_readlane(ck, 63);
	const bool considered = brk < 0 || lane <= brk;
	const int32_t cv = (valid && considered)? sc2 : INT32_MIN;
	const int32_t cmax = wave_reduce_max(cv);
	if (cmax > w.max_f) {
		const unsigned long long m = __ballot(cv == cmax);
		w.max_f = cmax; w.max_j = __builtin_amdgcn_readlane(jj, __builtin_ctzll(m));   // lowest lane = first met by the sequential walk
	}
	RQ_SYNC();
	return brk >= 0;
}

struct RqBest { double best; int bj; bool tie; };
__device__ inline void rq_offer(RqBest &b, double pj, int j, bool tie_in)
{
	if (b.bj < 0 || pj < b.best) { b.best = pj; b.bj = j; b.tie = tie_in; } else if (pj == b.best) b.tie = true;
}

__device__ __forceinline__ void rmq_dp_read(const RmqParams &rp, const DevBatch &bt, const DevAnchors &an, const int32_t *list, uint8_t *flag, unsigned long long *ctr, const unsigned int bid)
{
	__shared__ long long cand[RQ_RING], sorted[RQ_RING];   // slow path of the inner walk
	__shared__ uint32_t tw[RQ_TW];
	__shared__ uint64_t rx[RQ_RING], ryy[RQ_RING];
	__shared__ int32_t rf[RQ_RING], rpp[RQ_RING];
	__shared__ int32_t rbad[RQ_RING];                        // number of anchors k' <= k with y[k'] < y[k' - 1] (a ring like rx / ryy)
	__shared__ double wpri[RQ_WRING];
	__shared__ int32_t wy[RQ_WRING];
	__shared__ double bpri[RQ_NB];
	__shared__ int32_t bmin[RQ_NB], bmax[RQ_NB], bjj[RQ_NB], btie[RQ_NB];
	const int r = list[bid], lane = threadIdx.x;
	if (flag[r] != MM355_RMQ_DONE) return;
	const int64_t o = an.aoff[r];
	const int n = rp.primary? (int)(an.aoff[r + 1] - o) : an.n_v[r];
	if (n <= 0) return;
	const mm128 *a = an.a + o;
	int32_t *f = an.f + o, *p = an.p + o, *yy = an.vi + o;
	int32_t *lb_out = an.v + o, *lbi_out = (int32_t*)(an.u2 + o);   // window starts of every anchor (v[] and u2[] are free until the backtrack)
	double *pri = (double*)(an.z + o);
	int32_t max_dist = rp.max_dist, max_dist_inner = rp.max_dist_inner;
	const int32_t bw = rp.bw, max_skip = rp.max_chn_skip, cap = rp.cap;
	if (max_dist < bw) max_dist = bw;
	if (max_dist_inner < 0) max_dist_inner = 0;
	if (max_dist_inner > max_dist) max_dist_inner = max_dist;
	if (cap < 0) { if (lane == 0) flag[r] = MM355_RMQ_HOST; return; }   // (a negative cap empties the trees before anything is inserted: left to the literal code)
	const float pen_gap = rp.pen_gap, pen_skip = rp.pen_skip;
	const double half_gap = 0.5 * (double)pen_gap;
	KPROF_BEGIN(bt);
	// ---- window starts: lb(i) = first idx <= i of i's strand / contig with x_idx + dist >= x_i.  The reference's `while (st < i && ...) ++st`
	// stops exactly there (its condition is true on a prefix of [0, i) and false behind it, and the prefix only grows with i).
	{
		int lo_all = 0, lo_all_in = 0;
		for (int base = 0; base < n; base += WAVE) {
			const int i = base + lane;
			int lo = lo_all, hi = i < n? i : n - 1, lo2 = lo_all_in, hi2 = hi;
			const uint64_t xi = a[i < n? i : n - 1].x;
			while (__any(lo < hi || lo2 < hi2)) {
				const int mid = (lo + hi) >> 1, mid2 = (lo2 + hi2) >> 1;
				const uint64_t xs = a[mid].x, xs2 = a[mid2].x;
				if (lo < hi) { if ((xi >> 32 != xs >> 32) || xi > xs + (uint64_t)(int64_t)max_dist) lo = mid + 1; else hi = mid; }
				if (lo2 < hi2) { if ((xi >> 32 != xs2 >> 32) || xi > xs2 + (uint64_t)(int64_t)max_dist_inner) lo2 = mid2 + 1; else hi2 = mid2; }
			}
			if (i < n) { lb_out[i] = lo; lbi_out[i] = lo2; }
			lo_all = __builtin_amdgcn_readlane(lo, 0); lo_all_in = __builtin_amdgcn_readlane(lo2, 0);   // lb is monotone: the next 64 anchors start from this chunk's first
		}
	}
	for (int k = lane; k < RQ_TW; k += WAVE) tw[k] = 0;
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	RQ_SYNC();
	KPROF(17);
	int i0 = 0, st = 0, st_in = 0;
	int32_t bad_tot = 0, y_before = INT32_MIN;               // descents so far; y of the anchor in front of the current chunk
	uint64_t x_i0 = 0, cx = 0, cy = 0;
	int32_t clb = 0, clbi = 0;
	unsigned long long n_scan = 0;
	bool bail = false;
	for (int i = 0; i < n; ++i) {
		if ((i & (WAVE - 1)) == 0) {
			// chunk boundary: this chunk's anchors enter the x / y rings (nothing older than i - RQ_RING + 64 may be read from them) ...
			cx = cy = 0; clb = clbi = 0;
			if (i + lane < n) {
				const mm128 t = a[i + lane]; cx = t.x; cy = t.y;
				clb = rq_ld32(lb_out + i + lane); clbi = rq_ld32(lbi_out + i + lane);
				rx[(i + lane) & RQ_RMASK] = cx; ryy[(i + lane) & RQ_RMASK] = cy;
			}
			{   // ... with the running count of descents of y in index order: a window without one is ascending in y, and the inner walk over
				// it needs neither the order check nor the candidates behind its stopping point
				const int32_t y32 = (int32_t)cy;
				int32_t yp = __builtin_amdgcn_update_dpp(0, y32, 0x138, 0xf, 0xf, false);   // wave_shr:1
				if (lane == 0) yp = y_before;
				const int32_t bad = (i + lane < n && y32 < yp)? 1 : 0;
				const int32_t incl = bad_tot + wave_incl_scan_add(bad);
				if (i + lane < n) rbad[(i + lane) & RQ_RMASK] = incl;
				bad_tot = __builtin_amdgcn_readlane(incl, 63);
				const int last = n - i < WAVE? n - i - 1 : WAVE - 1;
				y_before = __builtin_amdgcn_readlane(y32, last);
			}
			if (i > 0) {   // ... and the block of 64 anchors that has just been finished gets its summary
				const int jb = i - WAVE + lane;
				const int32_t yj = wy[jb & RQ_WMASK];
				const double pj = wpri[jb & RQ_WMASK];
				long long sk = __double_as_longlong(pj);
				sk = sk >= 0? sk : (sk ^ 0x7fffffffffffffffLL);
				const long long mx = wave_reduce_max64(~sk);
				const unsigned long long who = __ballot(~sk == mx);
				const int32_t ymx = wave_reduce_max(yj), ymn = -wave_reduce_max(-yj);
				if (lane == 0) {
					const int sl = ((i >> 6) - 1) & (RQ_NB - 1);
					const int wl = __builtin_ctzll(who);
					bmin[sl] = ymn; bmax[sl] = ymx; bjj[sl] = i - WAVE + wl; btie[sl] = __popcll(who) > 1;
				}
				const double pmin = wpri[(i - WAVE + __builtin_ctzll(who)) & RQ_WMASK];
				if (lane == 0) bpri[((i >> 6) - 1) & (RQ_NB - 1)] = pmin;
			}
			RQ_SYNC();
		}
		const uint64_t xi = rq_bcast64(cx, i & (WAVE - 1)), yi = rq_bcast64(cy, i & (WAVE - 1));
		const int32_t yi32 = (int32_t)yi;
		// the anchors of the previous run of equal x become visible
		if (i == 0) x_i0 = xi;
		if (i0 < i && x_i0 != xi) { i0 = i; x_i0 = xi; }
		// st: first anchor of the same strand / contig within max_dist; at most cap elements stay
		{
			const int lbv = __builtin_amdgcn_readlane(clb, i & (WAVE - 1));
			if (lbv > st) st = lbv;
			if (st < i0 && i0 - st > cap) st = i0 - cap;
		}
		if (max_dist_inner > 0) {
			const int lbv = __builtin_amdgcn_readlane(clbi, i & (WAVE - 1));
			if (lbv > st_in) st_in = lbv;
			if (st_in < i0 && i0 - st_in > cap) st_in = i0 - cap;
			if (st_in < i0 && i - st_in > RQ_RING - 2 * WAVE) { bail = true; break; }   // the rings (and the mark window) must cover the inner set and the current run
		}
		// ---- range-minimum query over the outer set [st, i0)
		const int32_t lo_y = yi32 - max_dist;
		RqBest B; B.best = 0.0; B.bj = -1; B.tie = false;
		const int near0 = i - RQ_WRING + WAVE > st? i - RQ_WRING + WAVE : st;   // [near0, i0) is served by the (y, pri) ring and the block summaries
		for (int j = st + lane; j < near0; j += WAVE) {
			const int32_t yj = rq_ld32(yy + j);
			const double pj = rq_ldf64(pri + j);
			if ((yj > lo_y && yj < yi32) || (yj == yi32 && j == 0)) rq_offer(B, pj, j, false);
		}
		if (near0 < i0) {
			const int fb0 = (near0 + WAVE - 1) >> 6, fb1 = i0 >> 6;     // complete blocks: fb0 .. fb1 - 1
			int e0 = near0, e1 = i0;                                    // element-wise: [near0, 64 fb0) and [64 fb1, i0), or everything
			unsigned long long straddle = 0;
			if (fb0 < fb1) {
				e1 = fb0 << 6;
				const int bb = fb0 + lane;
				bool strad = false;
				if (bb < fb1) {
					const int sl = bb & (RQ_NB - 1);
					const int32_t ymn = bmin[sl], ymx = bmax[sl];
					if (ymn > lo_y && ymx < yi32) rq_offer(B, bpri[sl], bjj[sl], btie[sl] != 0);        // every member inside the query's y range
					else if (!(ymx <= lo_y || ymn > yi32 || (ymn == yi32 && bb != 0))) strad = true;       // some member may be inside
				}
				straddle = __ballot(strad);
			}
			for (int j = e0 + lane; j < e1; j += WAVE) {     // leading ragged end (or the whole window when it holds no complete block)
				const int32_t yj = wy[j & RQ_WMASK];
				const double pj = wpri[j & RQ_WMASK];
				if ((yj > lo_y && yj < yi32) || (yj == yi32 && j == 0)) rq_offer(B, pj, j, false);
			}
			if (fb0 < fb1) {
				for (int j = (fb1 << 6) + lane; j < i0; j += WAVE) {   // trailing ragged end
					const int32_t yj = wy[j & RQ_WMASK];
					const double pj = wpri[j & RQ_WMASK];
					if ((yj > lo_y && yj < yi32) || (yj == yi32 && j == 0)) rq_offer(B, pj, j, false);
				}
				while (straddle) {                                       // blocks cut by a y bound
					const int j = ((fb0 + __builtin_ctzll(straddle)) << 6) + lane; straddle &= straddle - 1;
					const int32_t yj = wy[j & RQ_WMASK];
					const double pj = wpri[j & RQ_WMASK];
					if ((yj > lo_y && yj < yi32) || (yj == yi32 && j == 0)) rq_offer(B, pj, j, false);
				}
			}
		}
		n_scan += (unsigned long long)(i0 - st);
		RqWalk w; w.max_f = (int32_t)(yi >> 32 & 0xff); w.max_j = -1; w.n_skip = 0;
		{   // wave minimum of an order-preserving integer image of the double; the minimum must be attained exactly once
			long long sk = __double_as_longlong(B.best);
			sk = sk >= 0? sk : (sk ^ 0x7fffffffffffffffLL);
			const long long mine = B.bj >= 0? ~sk : (long long)0x8000000000000000ULL;
			const long long mx = wave_reduce_max64(mine);
			const unsigned long long who = __ballot(B.bj >= 0 && mine == mx);
			if (who) {
				if (__popcll(who) > 1 || __ballot(B.tie && mine == mx)) { bail = true; break; }
				const int wl = __builtin_ctzll(who);
				const int j = __builtin_amdgcn_readlane(B.bj, wl);
				uint64_t ajx, ajy; int32_t fj;
				if (i - j <= RQ_RING - WAVE) { ajx = rx[j & RQ_RMASK]; ajy = ryy[j & RQ_RMASK]; fj = rf[j & RQ_RMASK]; }
				else { const mm128 aj = a[j]; ajx = aj.x; ajy = aj.y; fj = rq_ld32(f + j); }
				bool exact; int32_t width;
				const int32_t sc = fj + rq_comput_sc(xi, yi, ajx, ajy, pen_gap, pen_skip, &exact, &width);
				if (width <= bw && sc > w.max_f) { w.max_f = sc; w.max_j = j; }
				if (!exact && max_dist_inner > 0 && st_in < i0 && yi32 > 0) {
					const uint32_t mark = (uint32_t)i + 1u;
					const int32_t y_lo = yi32 - max_dist_inner;
					// walk in descending j; on the way check that this IS descending (y, j): y must never increase along the walk
					// (the check runs over ALL candidates, also behind the point where the walk stops: a candidate further down in j with a
					// larger y would have been met earlier by the reference)
					const RqWalk w_saved = w;
					int32_t carry = INT32_MAX; bool sortit = false, stopped = false;
					const bool ascending = rbad[(i0 - 1) & RQ_RMASK] == rbad[st_in & RQ_RMASK];   // no k in (st_in, i0) with y[k] < y[k - 1]: the proof below in O(1)
					for (int jb = i0 - 1; jb >= st_in; jb -= WAVE) {
						const int jj = jb - lane;
						bool in = false, valid = false; int32_t sc2 = 0, pj = -1, y2 = 0;
						uint64_t yj = 0;
						if (jj >= st_in) { yj = ryy[jj & RQ_RMASK]; y2 = (int32_t)yj; in = y2 >= y_lo && y2 < yi32; }
						if (!ascending) {
							const int32_t neg = in? -y2 : INT32_MIN;                      // prefix minimum of y = - prefix maximum of -y
							int32_t pmn = wave_excl_prefix_max(neg, lane);
							pmn = pmn == INT32_MIN? INT32_MAX : -pmn;
							pmn = pmn < carry? pmn : carry;
							if (__ballot(in && y2 > pmn)) { sortit = true; break; }
							const int32_t cm = wave_reduce_max(neg);
							if (cm != INT32_MIN && -cm < carry) carry = -cm;
						}
						if (stopped) { if (ascending) break; continue; }
						if (in) {
							bool ex2; int32_t width2;
							sc2 = rf[jj & RQ_RMASK] + rq_comput_sc(xi, yi, rx[jj & RQ_RMASK], yj, pen_gap, pen_skip, &ex2, &width2);
							valid = width2 <= bw;
							if (valid) pj = rpp[jj & RQ_RMASK];
						}
						stopped = rq_walk_step(w, tw, mark, jj, valid, sc2, pj, st_in, max_skip);
					}
					if (sortit) {
						w = w_saved;                                                   // (the marks of the abandoned attempt carry `mark`; the redo uses another value)
						const uint32_t mark2 = mark | 0x80000000u;
						// chains of different diagonals interleave here: order the candidates of this anchor by (y, j), descending
						int m = 0;
						for (int jb = i0 - 1; jb >= st_in; jb -= WAVE) {
							const int jj = jb - lane;
							bool in = false; int32_t y2 = 0;
							if (jj >= st_in) { y2 = (int32_t)ryy[jj & RQ_RMASK]; in = y2 >= y_lo && y2 < yi32; }
							const unsigned long long mk = __ballot(in);
							if (in) cand[m + __popcll(mk & LANE_LT_MASK(lane))] = (long long)(((uint64_t)(int64_t)y2) << 32 | (uint32_t)jj);
							m += __popcll(mk);
						}
						RQ_SYNC();
						for (int c = lane; c < m; c += WAVE) {
							const long long kc = cand[c];
							int rank = 0;
							for (int t = 0; t < m; ++t) rank += cand[t] > kc;
							sorted[rank] = kc;          // keys are unique (j is part of them)
						}
						RQ_SYNC();
						for (int kb = 0; kb < m; kb += WAVE) {
							const int k = kb + lane;
							bool valid = false; int32_t sc2 = 0, pj = -1; int jj = 0;
							if (k < m) {
								jj = (int)(uint32_t)sorted[k];
								bool ex2; int32_t width2;
								sc2 = rf[jj & RQ_RMASK] + rq_comput_sc(xi, yi, rx[jj & RQ_RMASK], ryy[jj & RQ_RMASK], pen_gap, pen_skip, &ex2, &width2);
								valid = width2 <= bw;
								if (valid) pj = rpp[jj & RQ_RMASK];
							}
							if (rq_walk_step(w, tw, mark2, jj, valid, sc2, pj, st_in, max_skip)) break;
						}
					}
				}
			}
		}
		const int32_t max_f = w.max_f, max_j = w.max_j;
		const double pv = -((double)max_f + half_gap * (double)((int32_t)xi + yi32));
		RQ_SYNC();   // every lane has finished reading the ring slots this anchor overwrites
		if (lane == 0) {
			f[i] = max_f; p[i] = max_j; yy[i] = yi32; pri[i] = pv;
			rf[i & RQ_RMASK] = max_f; rpp[i & RQ_RMASK] = max_j; wy[i & RQ_WMASK] = yi32; wpri[i & RQ_WMASK] = pv;
		}
		RQ_SYNC();
	}
	KPROF(18);
	if (bail) { if (lane == 0) flag[r] = MM355_RMQ_HOST; return; }
	if (lane == 0 && ctr) atomicAdd(ctr + (bid & 63), n_scan);
}

// ---- stage 3: mg_chain_backtrack + compact_a of the re-chained reads (max_drop = the band width mg_lchain_rmq was given)
// the resident-grid forms (MM355_DEQUEUE, mm355_kernels.hip): a block takes the next listed read when it is free
__device__ __forceinline__ unsigned int rmq_next_item(unsigned int *ctr)
{
	__syncthreads();
	unsigned int v = 0;
	if (threadIdx.x == 0) v = atomicAdd(ctr, 1u);
	return (unsigned int)__builtin_amdgcn_readfirstlane((int)v);
}
__global__ __launch_bounds__(WAVE) void k_rmq_sort(RmqParams rp, DevBatch bt, DevAnchors an, const int32_t *list, int n_list, uint8_t *flag, int *err, unsigned int *qctr)
{
	MM355_LATENCY_KERNEL();
	for (unsigned int bid = rmq_next_item(qctr); bid < (unsigned int)n_list; bid = rmq_next_item(qctr)) rmq_sort_read(rp, bt, an, list, flag, err, bid);
}
__global__ __launch_bounds__(WAVE) void k_rmq_dp(RmqParams rp, DevBatch bt, DevAnchors an, const int32_t *list, int n_list, uint8_t *flag, unsigned long long *ctr, unsigned int *qctr)
{
	MM355_LATENCY_KERNEL();
	for (unsigned int bid = rmq_next_item(qctr); bid < (unsigned int)n_list; bid = rmq_next_item(qctr)) rmq_dp_read(rp, bt, an, list, flag, ctr, bid);
}
__global__ __launch_bounds__(WAVE) void k_rmq_backtrack(RmqParams rp, DevParams pr, DevBatch bt, DevAnchors an, const int32_t *list, int n_list, const uint8_t *flag, int *err, unsigned int *qctr)
{
	MM355_LATENCY_KERNEL();
	__shared__ BtLds S;
	for (unsigned int bid = rmq_next_item(qctr); bid < (unsigned int)n_list; bid = rmq_next_item(qctr)) {
		const int r = list[bid];
		if (flag[r] != MM355_RMQ_DONE) continue;
		const int n = rp.primary? (int)(an.aoff[r + 1] - an.aoff[r]) : an.n_v[r];
		wave_backtrack_read(pr, bt, an, err, &S, r, n, rp.bw, 20);
	}
}

int mm355_launch_rmq(const RmqParams &rp, const DevParams &pr, const DevBatch &bt, DevAnchors &an, const int32_t *d_list, int n_list, uint8_t *d_flag,
                     int *err, unsigned long long *ctr, unsigned int *qctr, hipStream_t st, void *kt)
{
	if (n_list <= 0) return 0;
	// qctr: three u32 work-list cursors (one per kernel); resident grids (k_rmq_dp holds 52 KB of LDS per wave: two per CU)
	if (hipMemsetAsync(qctr, 0, 16, st) != hipSuccess) return -1;
	const int rb = mm355_resident_blocks(), g2 = n_list < rb * 2 / 3? n_list : rb * 2 / 3, g3 = n_list < rb? n_list : rb;
	if (!rp.primary) { KtScope ks(kt, KT_RMQ_SORT, st); hipLaunchKernelGGL(k_rmq_sort, dim3(g2), dim3(WAVE), 0, st, rp, bt, an, d_list, n_list, d_flag, err, qctr); }
	{ KtScope ks(kt, KT_RMQ_DP, st); hipLaunchKernelGGL(k_rmq_dp, dim3(g2), dim3(WAVE), 0, st, rp, bt, an, d_list, n_list, d_flag, ctr, qctr + 1); }
	{ KtScope ks(kt, KT_RMQ_BT, st); hipLaunchKernelGGL(k_rmq_backtrack, dim3(g3), dim3(WAVE), 0, st, rp, pr, bt, an, d_list, n_list, (const uint8_t*)d_flag, err, qctr + 2); }
	return hipGetLastError() == hipSuccess? 0 : -1;
}
