// mm355_dp.hip -- banded gap-affine extension on gfx950: row a12 of SURVEY.md section 8(a).
//
// Reproduces minimap2 2.26's U:ksw2_extd2_sse.c::ksw_extd2_sse (two-piece affine, the kernel map-ont and
// map-hifi execute; reached from /root/reference/src/lib.rs:482 and :587 via mm_map -> mm_align_skeleton ->
// mm_align_pair) including U:ksw2.h::ksw_backtrack / ksw_apply_zdrop, bit for bit:
//   * anti-diagonal sweep in the Suzuki-Kasahara difference form, int8 wrap-around arithmetic;
//   * the SIMD kernel's cell set: whole 16-lane blocks [st/16*16, (en+16)/16*16-1] per diagonal, i.e. cells outside
//     the band are evaluated on stale inputs and can feed in-band cells at the band edge -- the same cells are
//     evaluated here, on the same stale values, so band-edge results agree;
//   * the SSE4.1 4-lane strided H-max (tie order), exact and approximate (KSW_EZ_APPROX_MAX) score tracking,
//     z-drop, and the 1-byte/cell direction matrix that ksw_backtrack walks.
// One wave = one alignment; lane = target position t inside a 64-cell chunk of the diagonal; the per-t state
// (u,v,x,y,x2,y2,s packed in 8 bytes, plus int32 H) lives in LDS when the target fits, else in an HBM work area.
// x[t-1]/v[t-1] of the previous diagonal arrive by a one-lane wave shuffle.  No MFMA: this is an int8 recurrence.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <mutex>
#include <condition_variable>
#include <algorithm>
#include "mm355_pipeline.h"
#include "mm355_dp.h"

#define WAVE 64
#define KSW_NEG_INF (-0x40000000)
#define EZ_SCORE_ONLY  0x01
#define EZ_RIGHT       0x02
#define EZ_APPROX_MAX  0x08
#define EZ_APPROX_DROP 0x10
#define EZ_EXTZ_ONLY   0x40
#define EZ_REV_CIGAR   0x80

struct EzState { int32_t max, max_q, max_t, mqe, mqe_t, mte, mte_q, score, zdropped, reach_end; };

__device__ inline int8_t I8(int v) { return (int8_t)v; }

__device__ __forceinline__ bool apply_zdrop(EzState &ez, int32_t H, int r, int t, int zdrop, int e2)
{
	if (H > ez.max) {
		ez.max = H, ez.max_t = t, ez.max_q = r - t;
	} else if (t >= ez.max_t && r - t >= ez.max_q) {
		int tl = t - ez.max_t, ql = (r - t) - ez.max_q, l;
		l = tl > ql? tl - ql : ql - tl;
		if (zdrop >= 0 && ez.max - H > zdrop + l * e2) { ez.zdropped = 1; return true; }
	}
	return false;
}

__device__ inline uint32_t *push_cigar(uint32_t *cigar, int &n, uint32_t op, int len)
{
	if (n == 0 || op != (cigar[n - 1] & 0xf)) cigar[n++] = (uint32_t)len << 4 | op;
	else cigar[n - 1] += (uint32_t)len << 4;
	return cigar;
}

// state word: byte0 u, 1 v, 2 x, 3 y, 4 x2, 5 y2, 6 s
__device__ inline int8_t SB(uint64_t w, int i) { return (int8_t)(w >> (8 * i)); }

// A workgroup is ONE wave: LDS traffic of a wave is executed in order, so lanes only need their own LDS operations to have
// completed (lgkmcnt) -- __syncthreads() would also wait for every outstanding global store (vmcnt(0)), i.e. for the
// direction-matrix bytes of the current anti-diagonal to reach HBM, which serialises the sweep on store latency.
#define DP_SYNC() do { if (NT == WAVE && state_in_lds) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } else __syncthreads(); } while (0)
// NT = threads per alignment: 64 (one wave; x[t-1]/v[t-1] by a lane shuffle) for the short problems, 512 (eight waves share
// one anti-diagonal; neighbours are re-read from LDS between two barriers) for long ones, whose single-wave sweep would
// otherwise be the tail of the whole launch.
// per-launch-group cell counters are spread over DP_CTR_SPREAD words (slot = block & 15): every alignment adds once, and a single word
// takes only ~88 atomics per microsecond -- 150 000 alignments of one launch on one word were 1.7 ms of atomics
#define DP_CTR_SPREAD CTR_SPREAD
template <int NT>
__global__ __launch_bounds__(NT) void k_ksw_extd2(DpConst dc, const DpJobDev *jobs, const int32_t *job_ids, int n_jobs,
                                                    const uint8_t *qbase, const uint8_t *tbase, uint8_t *pbase, int32_t *offbase,
                                                    uint32_t *cigbase, uint64_t *stbase, int32_t *Hbase, mm355_dpres_t *res,
                                                    int lds_cap, unsigned long long *cells_ctr, uint32_t *dense, unsigned long long *dense_ctr, int cls_base)
{
	extern __shared__ uint64_t lds[];
	// [lds_cap] state words, then [lds_cap] int32 H
	__shared__ uint64_t s_carry[2];
	__shared__ int32_t s_rh[NT / WAVE][4], s_rt[NT / WAVE][4];
	const int lane = threadIdx.x;
	if ((int)blockIdx.x >= n_jobs) return;
	if (NT > WAVE) __builtin_amdgcn_s_setprio(3);      // the long-target launches: a few barrier-chained blocks beside the wide grids of the round
	const int jid = job_ids[blockIdx.x];
	const DpJobDev jb = jobs[jid];
	const int qlen = jb.qlen, tlen = jb.tlen, flag = jb.flag, zdrop = jb.zdrop, end_bonus = jb.end_bonus;
	EzState ez;
	ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1;
	ez.max = 0; ez.score = ez.mqe = ez.mte = KSW_NEG_INF; ez.zdropped = 0; ez.reach_end = 0;
	(void)cigbase; (void)dense; (void)dense_ctr;
	if (qlen <= 0 || tlen <= 0 || jb.skip) {   // skip: tlen*qlen > max_sw_mat => treated as z-dropped by mm_align_pair
		if (lane == 0) {
			mm355_dpres_t o; o.max = 0; o.zdropped = jb.skip? 1 : 0; o.max_q = o.max_t = o.mqe_t = o.mte_q = -1;
			o.mqe = o.mte = o.score = KSW_NEG_INF; o.reach_end = 0; o.n_cigar = -1; o.cigar_off = -1;
			res[jid] = o;
		}
		return;
	}
	const uint8_t *query = qbase + jb.qoff, *target = tbase + jb.toff;
	const int q = dc.q, e = dc.e, q2 = dc.q2, e2 = dc.e2, qe = dc.qe_preswap;
	const int8_t qe8 = I8(q + e), qe28 = I8(q2 + e2), sc_mch = dc.sc_mch, sc_mis = dc.sc_mis, sc_N = dc.sc_N;
	const int long_thres = dc.long_thres, long_diff = dc.long_diff;
	const bool approx_max = flag & EZ_APPROX_MAX, right = flag & EZ_RIGHT;
	int w = jb.w;
	if (w < 0) w = tlen > qlen? tlen : qlen;
	const int tlen_ = (tlen + 15) / 16, T = tlen_ * 16;
	int n_col_ = qlen < tlen? qlen : tlen;
	n_col_ = ((n_col_ < w + 1? n_col_ : w + 1) + 15) / 16 + 1;
	const int n_col = n_col_ * 16;
	uint64_t *S; int32_t *H;
	const bool state_in_lds = T <= lds_cap;
	if (state_in_lds) { S = lds; H = (int32_t*)(lds + lds_cap); }
	else { S = stbase + jb.st_off; H = Hbase + jb.st_off; }
	uint8_t *p = pbase + jb.p_off;
	(void)offbase;
	{   // memset(u,v,x,y = -q-e; x2,y2 = -q2-e2); s = 0; H = NEG_INF
		const uint8_t a = (uint8_t)I8(-q - e), b = (uint8_t)I8(-q2 - e2);
		const uint64_t init = (uint64_t)a | (uint64_t)a << 8 | (uint64_t)a << 16 | (uint64_t)a << 24 | (uint64_t)b << 32 | (uint64_t)b << 40;
		for (int t = lane; t < T; t += NT) { S[t] = init; if (!approx_max) H[t] = KSW_NEG_INF; }
	}
	DP_SYNC();
	int last_st = -1, last_en = -1;
	int32_t H0 = 0, last_H0_t = 0;
	unsigned long long cells = 0;
	for (int r = 0; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1, st0, en0;
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
		if (en > (r + w) >> 1) en = (r + w) >> 1;
		if (st > en) { ez.zdropped = 1; break; }
		st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		int8_t x1, x21, v1;
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) {
				const uint64_t wv = S[st - 1];
				x1 = SB(wv, 2), x21 = SB(wv, 4), v1 = SB(wv, 1);
			} else { x1 = I8(-q - e), x21 = I8(-q2 - e2); v1 = I8(-q - e); }
		} else {
			x1 = I8(-q - e), x21 = I8(-q2 - e2);
			v1 = r == 0? I8(-q - e) : r < long_thres? I8(-e) : r == long_thres? I8(long_diff) : I8(-e2);
		}
		const bool edge = en >= r;
		const int8_t edge_u = r == 0? I8(-q - e) : r < long_thres? I8(-e) : r == long_thres? I8(long_diff) : I8(-e2);
		const int sc_end = st0 + ((en0 - st0) / 16 + 1) * 16;   // scores are (re)written for t in [st0, sc_end)
		int8_t cx = x1, cv = v1, cx2 = x21;
		uint8_t *pr = p + (size_t)r * n_col - st;
		int rnd = 0;
		for (int c0 = st; c0 <= en; c0 += NT, ++rnd) {
			const int t = c0 + lane;
			const bool act = t <= en;
			uint64_t old = act? S[t] : 0;
			uint64_t oldm = 0;
			if (NT > WAVE) {   // previous diagonal's cell t-1: LDS (or the word the previous round saved before overwriting it)
				if (act && t > st) oldm = (lane == 0)? s_carry[(rnd + 1) & 1] : S[t - 1];
				__syncthreads();
				if (act && lane == NT - 1) s_carry[rnd & 1] = old;
			}
			int8_t ou = SB(old, 0), ov = SB(old, 1), ox = SB(old, 2), oy = SB(old, 3), ox2 = SB(old, 4), oy2 = SB(old, 5), os = SB(old, 6);
			if (edge && t == r) { oy = I8(-q - e); oy2 = I8(-q2 - e2); ou = edge_u; }
			// x[t-1], v[t-1], x2[t-1] of the previous diagonal: neighbour lane, or the carry from the previous chunk
			int8_t xt1, vt1, x2t1;
			if (NT == WAVE) {
				int pk = (uint8_t)ox | (uint8_t)ov << 8 | (uint8_t)ox2 << 16;
				int nb = __shfl_up(pk, 1);
				xt1 = (int8_t)nb, vt1 = (int8_t)(nb >> 8), x2t1 = (int8_t)(nb >> 16);
				if (lane == 0) { xt1 = cx; vt1 = cv; x2t1 = cx2; }
				int last = __shfl(pk, 63);
				cx = (int8_t)last; cv = (int8_t)(last >> 8); cx2 = (int8_t)(last >> 16);
			} else {
				xt1 = SB(oldm, 2), vt1 = SB(oldm, 1), x2t1 = SB(oldm, 4);
				if (t == st) { xt1 = x1; vt1 = v1; x2t1 = x21; }
			}
			int8_t z;
			if (t >= st0 && t < sc_end) {
				const uint8_t sq = t < tlen? target[t] : 0;
				const int qi = r - t;
				const uint8_t sqq = qi >= 0? query[qi] : 0;   // qr[] is zero beyond the query
				z = sq == sqq? sc_mch : sc_mis;
				if (sq == 4 || sqq == 4) z = sc_N;
			} else z = os;
			const int8_t sc = z;
			int8_t a = I8(xt1 + vt1), b = I8(oy + ou), a2 = I8(x2t1 + vt1), b2 = I8(oy2 + ou);
			uint8_t d;
			int8_t nx, ny, nx2, ny2, tmp;
			if (!right) {
				d = a > z? 1 : 0;   z = z > a? z : a;
				d = b > z? 2 : d;   z = z > b? z : b;
				d = a2 > z? 3 : d;  z = z > a2? z : a2;
				d = b2 > z? 4 : d;  z = z > b2? z : b2;
				z = z < sc_mch? z : sc_mch;
				tmp = I8(z - q);  a = I8(a - tmp);  b = I8(b - tmp);
				tmp = I8(z - q2); a2 = I8(a2 - tmp); b2 = I8(b2 - tmp);
				nx  = I8((a  > 0? a  : 0) - qe8);  if (a  > 0) d |= 0x08;
				ny  = I8((b  > 0? b  : 0) - qe8);  if (b  > 0) d |= 0x10;
				nx2 = I8((a2 > 0? a2 : 0) - qe28); if (a2 > 0) d |= 0x20;
				ny2 = I8((b2 > 0? b2 : 0) - qe28); if (b2 > 0) d |= 0x40;
			} else {
				d = z > a? 0 : 1;   z = z > a? z : a;
				d = z > b? d : 2;   z = z > b? z : b;
				d = z > a2? d : 3;  z = z > a2? z : a2;
				d = z > b2? d : 4;  z = z > b2? z : b2;
				z = z < sc_mch? z : sc_mch;
				tmp = I8(z - q);  a = I8(a - tmp);  b = I8(b - tmp);
				tmp = I8(z - q2); a2 = I8(a2 - tmp); b2 = I8(b2 - tmp);
				nx  = I8((0 > a?  0 : a)  - qe8);  if (!(0 > a))  d |= 0x08;
				ny  = I8((0 > b?  0 : b)  - qe8);  if (!(0 > b))  d |= 0x10;
				nx2 = I8((0 > a2? 0 : a2) - qe28); if (!(0 > a2)) d |= 0x20;
				ny2 = I8((0 > b2? 0 : b2) - qe28); if (!(0 > b2)) d |= 0x40;
			}
			const int8_t nu = I8(z - vt1), nv = I8(z - ou);
			if (act) {
				S[t] = (uint64_t)(uint8_t)nu | (uint64_t)(uint8_t)nv << 8 | (uint64_t)(uint8_t)nx << 16 | (uint64_t)(uint8_t)ny << 24 |
				       (uint64_t)(uint8_t)nx2 << 32 | (uint64_t)(uint8_t)ny2 << 40 | (uint64_t)(uint8_t)sc << 48;
				pr[t] = d;
			}
			if (NT > WAVE) __syncthreads();
		}
		{   // the last unaligned 16-byte score store may reach past `en`: those s[] bytes persist for later diagonals
			const int t = en + 1 + lane;
			if (t < sc_end && t < T) {
				const uint8_t sq = t < tlen? target[t] : 0;
				const int qi = r - t;
				const uint8_t sqq = (qi >= 0 && qi < qlen)? query[qi] : 0;
				int8_t z = sq == sqq? sc_mch : sc_mis;
				if (sq == 4 || sqq == 4) z = sc_N;
				S[t] = (S[t] & ~(0xffULL << 48)) | (uint64_t)(uint8_t)z << 48;
			}
		}
		cells += (unsigned long long)(en0 - st0 + 1);
		DP_SYNC();
		if (!approx_max) {
			int32_t max_H, max_t;
			if (r > 0) {
				const int en1 = st0 + (en0 - st0) / 4 * 4;
				int32_t hen = en0 > 0? H[en0 - 1] + SB(S[en0], 0) : H[en0] + SB(S[en0], 1);
				DP_SYNC();
				max_H = hen; max_t = en0;
				// 4-lane strided maxima over [st0,en1): class = (t - st0) & 3, strict > keeps the first t of a class
				int32_t bh = INT32_MIN, bt = 0x7fffffff;
				for (int t = st0 + lane; t < en1; t += NT) {
					int32_t h = H[t] + (int32_t)SB(S[t], 1);
					H[t] = h;
					if (h > bh) bh = h, bt = t;
				}
				for (int o = 4; o < 64; o <<= 1) {   // reduce lanes of the same class (lane & 3): max, tie -> smaller t
					int32_t oh = __shfl_xor(bh, o), ot = __shfl_xor(bt, o);
					if (oh > bh || (oh == bh && ot < bt)) bh = oh, bt = ot;
				}
				if (NT > WAVE) {   // combine the waves (same class = same lane & 3, NT is a multiple of 4)
					if ((lane & 63) < 4) { s_rh[lane >> 6][lane & 3] = bh; s_rt[lane >> 6][lane & 3] = bt; }
					__syncthreads();
					bh = INT32_MIN, bt = 0x7fffffff;
					const int cl = lane & 3;
					for (int wv = 0; wv < NT / WAVE; ++wv) { int32_t oh = s_rh[wv][cl], ot = s_rt[wv][cl]; if (oh > bh || (oh == bh && ot < bt)) bh = oh, bt = ot; }
					__syncthreads();
				}
				for (int i = 0; i < 4; ++i) {        // HH[i] starts at H[en0]; combine in class order with strict <
					int32_t hh = __shfl(bh, i), tt = __shfl(bt, i);   // lanes 0..3 of every wave hold the class results
					if (hh > hen) { if (max_H < hh) max_H = hh, max_t = tt; }
				}
				if (lane == 0) H[en0] = hen;
				for (int t = en1; t < en0; ++t) {    // scalar tail (at most 3 cells), strict >
					int32_t h = H[t] + (int32_t)SB(S[t], 1);
					DP_SYNC();
					if (lane == 0) H[t] = h;
					if (h > max_H) max_H = h, max_t = t;
				}
				DP_SYNC();
			} else {
				int32_t h0 = (int32_t)SB(S[0], 1) - qe;
				if (lane == 0) H[0] = h0;
				max_H = h0; max_t = 0;
				DP_SYNC();
			}
			const int32_t Hen0 = H[en0], Hst0 = H[st0];
			if (en0 == tlen - 1 && Hen0 > ez.mte) ez.mte = Hen0, ez.mte_q = r - en;
			if (r - st0 == qlen - 1 && Hst0 > ez.mqe) ez.mqe = Hst0, ez.mqe_t = st0;
			if (apply_zdrop(ez, max_H, r, max_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H[tlen - 1];
		} else {
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = SB(S[last_H0_t], 1);
					int32_t d1 = SB(S[last_H0_t + 1], 0);
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += SB(S[last_H0_t], 1);
				} else {
					++last_H0_t, H0 += SB(S[last_H0_t], 0);
				}
			} else H0 = (int32_t)SB(S[0], 1) - qe, last_H0_t = 0;
			if ((flag & EZ_APPROX_DROP) && apply_zdrop(ez, H0, r, last_H0_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H0;
		}
		last_st = st, last_en = en;
		DP_SYNC();
	}
	// the sweep only records where the backtrack starts; k_ksw_backtrack walks the direction matrix (one LANE per alignment)
	if (lane == 0) {
		int i0 = -1, j0 = -1;
		if (!ez.zdropped && !(flag & EZ_EXTZ_ONLY)) { i0 = tlen - 1; j0 = qlen - 1; }
		else if (!ez.zdropped && (flag & EZ_EXTZ_ONLY) && ez.mqe + end_bonus > ez.max) { ez.reach_end = 1; i0 = ez.mqe_t; j0 = qlen - 1; }
		else if (ez.max_t >= 0 && ez.max_q >= 0) { i0 = ez.max_t; j0 = ez.max_q; }
		mm355_dpres_t o;
		o.max = ez.max; o.zdropped = ez.zdropped; o.max_q = ez.max_q; o.max_t = ez.max_t; o.mqe = ez.mqe; o.mqe_t = ez.mqe_t;
		o.mte = ez.mte; o.mte_q = ez.mte_q; o.score = ez.score; o.reach_end = ez.reach_end;
		o.n_cigar = i0;            // start cell, replaced by the CIGAR length / offset in k_ksw_backtrack
		o.cigar_off = j0;
		res[jid] = o;
		// cells_ctr: one counter per launch group, or (cls_base >= 0: the merged launch of every long-target class) the base of the
		// per-group counters, indexed by the job's own size class
		if (cells) atomicAdd((cls_base >= 0? cells_ctr + DP_CTR_SPREAD * (2 * (cls_base + (T > 4096) + (T > 12288)) + (approx_max? 0 : 1)) : cells_ctr) + (blockIdx.x & (DP_CTR_SPREAD - 1)), cells);
	}
}

#include "mm355_dpreg.h"
#include "mm355_dpmw.h"
#include "mm355_dprow.h"
#include "mm355_dpband.h"

// U:ksw2.h::ksw_backtrack (is_rot = 1).  The walk is a chain of dependent 1-byte loads (one per CIGAR column), i.e. pure
// latency: with one lane per alignment a wave keeps 64 independent chains in flight instead of one.  off[]/off_end[] of
// the reference are pure functions of (r, qlen, tlen, w) and are recomputed instead of being stored and re-loaded.
__global__ __launch_bounds__(WAVE) void k_ksw_backtrack(const DpJobDev *jobs, const int32_t *job_ids, int n_jobs, const uint8_t *pbase, const uint8_t *pbase2, uint32_t *cigbase,
                                                        mm355_dpres_t *res, uint32_t *dense, unsigned long long *dense_ctr)
{
	const int t = blockIdx.x * WAVE + threadIdx.x;
	if (t >= n_jobs) return;
	const int jid = job_ids[t];
	const DpJobDev jb = jobs[jid];
	mm355_dpres_t o = res[jid];
	const int i0 = o.n_cigar, j0 = (int)o.cigar_off;
	int n_cigar = 0;
	uint32_t *cigar = cigbase + jb.cig_off;
	const int qlen = jb.qlen, tlen = jb.tlen;
	if (qlen > 0 && tlen > 0 && !jb.skip && (i0 >= 0 || j0 >= 0)) {
		int w = jb.w;
		if (w < 0) w = tlen > qlen? tlen : qlen;
		int n_col_ = qlen < tlen? qlen : tlen;
		n_col_ = ((n_col_ < w + 1? n_col_ : w + 1) + 15) / 16 + 1;
		const int n_col = n_col_ * 16;
		const uint8_t *p = (jb.pad & 4? pbase2 : pbase) + jb.p_off;
		const int lay = jb.pad & 3, dlo = jb.dlo;
		const int row_stride = lay == 2? jb.bw : (tlen + 15) / 16 * 16 + 16;
		int i = i0, j = j0, state = 0;
		// the open CIGAR run lives in registers and is stored once, when the operation changes (it used to be a read-modify-write of the
		// scratch entry on every column: one more memory round trip per step, and most of the kernel's write traffic)
		uint32_t run_op = 0, run_len = 0;
#define BT_PUSH(op, len) do { if (run_len && run_op == (uint32_t)(op)) run_len += (uint32_t)(len); else { if (run_len) cigar[n_cigar++] = run_len << 4 | run_op; run_op = (uint32_t)(op); run_len = (uint32_t)(len); } } while (0)
		while (i >= 0 && j >= 0) {
			int force_state = -1, rr = i + j;
			int st = 0, en = tlen - 1;
			if (st < rr - qlen + 1) st = rr - qlen + 1;
			if (en > rr) en = rr;
			if (st < (rr - w + 1) >> 1) st = (rr - w + 1) >> 1;
			if (en > (rr + w) >> 1) en = (rr + w) >> 1;
			st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;   // off[rr], off_end[rr]
			uint32_t tmp;
			if (i < st) force_state = 2;
			if (i > en) force_state = 1;
			// lay = 1: the row sweep's tiled matrix (mm355_dprow.h); 2: the band's (mm355_dpband.h; the walk cannot leave the band: the kernel's proof)
			tmp = force_state < 0? (lay == 1? p[row_cell_off(j, i, row_stride)] : lay == 2? p[row_cell_off(j, min(max(i - j - dlo, 0), row_stride - 1), row_stride)] : p[(size_t)rr * n_col + i - st]) : 0;
			if (state == 0) state = tmp & 7;
			else if (!(tmp >> (state + 2) & 1)) state = 0;
			if (state == 0) state = tmp & 7;
			if (force_state >= 0) state = force_state;
			if (state == 0) { BT_PUSH(0, 1); --i; --j; }
			else if (state == 1 || state == 3) { BT_PUSH(2, 1); --i; }
			else { BT_PUSH(1, 1); --j; }
		}
		if (i >= 0) BT_PUSH(2, i + 1);
		if (j >= 0) BT_PUSH(1, j + 1);
		if (run_len) cigar[n_cigar++] = run_len << 4 | run_op;
#undef BT_PUSH
	}
	// dense arena (only real ops travel to the host): one atomic per wave -- the lanes' counts are prefix-summed first (the lanes that left
	// above are the top ones of the last wave, never read by a lower lane)
	long long dst;
	{
		const int lane = threadIdx.x & 63;
		unsigned int inc = (unsigned int)n_cigar;
		for (int d = 1; d < 64; d <<= 1) { const unsigned int u = __shfl_up(inc, d); if (lane >= d) inc += u; }
		const int last = 63 - __builtin_clzll(__ballot(1));
		unsigned long long base = 0;
		if (lane == last) base = atomicAdd(dense_ctr, (unsigned long long)inc);
		base = __shfl(base, last);
		dst = (long long)(base + inc - (unsigned int)n_cigar);
	}
	if (jb.flag & EZ_REV_CIGAR) for (int k = 0; k < n_cigar; ++k) dense[dst + k] = cigar[k];
	else for (int k = 0; k < n_cigar; ++k) dense[dst + k] = cigar[n_cigar - 1 - k];
	o.n_cigar = n_cigar; o.cigar_off = dst;
	res[jid] = o;
}

// gather kernel: materialise query / target code strings of each job (optionally reversed) from the read batch and
// the 2-bit packed reference + its N-run table (U:index.c::mm_idx_getseq semantics)
__global__ __launch_bounds__(256) void k_dp_gather(DevIndex ix, const DpGather *g, int n_jobs, const uint8_t *rq, uint8_t *qbuf, uint8_t *tbuf)
{
	const int j = blockIdx.x;
	if (j >= n_jobs) return;
	const DpGather gj = g[j];
	for (int i = threadIdx.x; i < gj.qlen; i += 256) {
		const int src = gj.rev? gj.qlen - 1 - i : i;
		qbuf[gj.qoff + i] = rq[gj.q_src + src];
	}
	const uint64_t base = ix.seq_off[gj.rid] + (uint64_t)gj.t_st;
	__shared__ uint32_t s_first, s_cnt;
	if (threadIdx.x == 0) { uint32_t f = 0, n = 0; if (ix.n_nr) ref_window(ix, base, base + (uint64_t)(gj.tlen > 0? gj.tlen : 0), f, n); s_first = f; s_cnt = n; }
	__syncthreads();
	const uint32_t first = s_first, cnt = s_cnt;
	for (int i = threadIdx.x; i < gj.tlen; i += 256) {
		const int src = gj.rev? gj.tlen - 1 - i : i;
		tbuf[gj.toff + i] = (uint8_t)ref_code(ix, first, cnt, base + (uint64_t)src);
	}
}

// per-read query codes: [0,len) forward codes, [len,2len) reverse-complement codes (U:align.c::mm_align_skeleton)
__global__ __launch_bounds__(256) void k_read_codes(DevBatch bt, uint8_t *rq)
{
	const int r = blockIdx.x;
	const int len = bt.rlen[r];
	const int64_t off = bt.roff[r];
	for (int i = threadIdx.x; i < len; i += 256) {
		const int c = mm_nt4(bt.seq[off + i]);
		rq[2 * off + i] = (uint8_t)c;
		rq[2 * off + len + (len - 1 - i)] = (uint8_t)(c < 4? 3 - c : 4);
	}
}

// ------------------------------------------------------------------ host driver
DpConst mm355_dp_const(const mm355_mapopt_t *mo)
{
	DpConst c;
	int q = mo->q, e = mo->e, q2 = mo->q2, e2 = mo->e2, t;
	int a = mo->a < 0? -mo->a : mo->a, b = mo->b > 0? -mo->b : mo->b, amb = mo->sc_ambi > 0? -mo->sc_ambi : mo->sc_ambi;
	c.qe_preswap = (int8_t)q + (int8_t)e;
	if (q2 + e2 < q + e) t = q, q = q2, q2 = t, t = e, e = e2, e2 = t;
	c.q = q; c.e = e; c.q2 = q2; c.e2 = e2;
	c.sc_mch = (int8_t)a; c.sc_mis = (int8_t)b; c.sc_N = amb == 0? (int8_t)(-e2) : (int8_t)amb;
	c.long_thres = e != e2? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + c.long_thres * e2 > q + e + c.long_thres * e) ++c.long_thres;
	c.long_diff = c.long_thres * (e - e2) - (q2 - q) - e2;
	int min_sc = b < amb? b : amb;
	c.valid = !(-min_sc > 2 * (q + e));
	return c;
}

// Size classes.  Targets up to 1024 bases run register-resident (k_ksw_reg, 2 * NP cells per lane), split by score
// tracking mode (KSW_EZ_APPROX_MAX or exact); longer ones keep their per-target state in LDS (12 B per position) or, beyond
// 12288 positions, in HBM, with eight waves per alignment.  MM355_DP_LEGACY=1 routes everything through the LDS kernel.
struct DpClass { int cap, kind, np; };   // kind 0: k_ksw_reg, 1: k_ksw_extd2<64> (LDS), 2: k_ksw_extd2<512>
static const DpClass DP_CLASSES[] = { {128, 0, 1}, {256, 0, 2}, {512, 0, 4}, {1024, 0, 8}, {4096, 2, 0}, {12288, 2, 0}, {0, 2, 0} };
static const DpClass DP_CLASSES_LEGACY[] = { {256, 1, 0}, {512, 1, 0}, {1024, 1, 0}, {1024, 1, 0}, {4096, 2, 0}, {12288, 2, 0}, {0, 2, 0} };
#define DP_N_CLASS 7
#define DP_N_GROUP 23                    // group = class * 2 + exact for the seven classes; 14 / 15 / 16 = k_ksw_row<2> / <4> / <8> (full-band approximate fills), 17 = k_ksw_rowl (the same, targets 1025..8192)
#define DP_G_ROW2 14
#define DP_G_ROW4 15
#define DP_G_ROW8 16
#define DP_G_ROWL 17
#define DP_G_BAND1 19                   // k_ksw_band<1 | 2 | 4>: the row sweep on a band of 128 / 256 / 512 diagonals with a sufficiency proof (mm355_dpband.h)
#define DP_G_BAND2 20
#define DP_G_BAND4 21
#define DP_G_BANDH 22                   // k_ksw_band2: two problems per wave, 64 diagonals each
#define DP_G_REDO 23                    // (not a class: the second, full-matrix run of band problems whose proof failed -- timer / counter slot)
#define DP_G_REGW 18                    // k_ksw_regw: exact, narrow band (w <= DP_WIN_MAX_W), targets > 1024 -- the register kernel with a moving window

// value range of the row sweep on a qlen x tlen problem (int16 halves; ROW_NEG must stay below every real value, differences of two real
// values and the shifted prefix terms G + t e must not wrap)
static bool rowl_range_ok(const DpConst &dc, int qlen, int tlen)
{
	const int tl = (tlen + 127) & ~127;       // the lanes beyond the target compute cells of a wider matrix
	auto cost = [&](int k) { const int c1 = dc.q + k * dc.e, c2 = dc.q2 + k * dc.e2; return c1 < c2? c1 : c2; };
	const int emax = dc.e > dc.e2? dc.e : dc.e2;
	const int hi = dc.sc_mch * (qlen < tl? qlen : tl);
	int step = dc.q + dc.e > dc.q2 + dc.e2? dc.q + dc.e : dc.q2 + dc.e2;
	if (-dc.sc_mis > step) step = -dc.sc_mis;
	if (-dc.sc_N > step) step = -dc.sc_N;
	const int lo = cost(qlen + 1) + cost(tl + 1) + step + emax + 64;
	return hi + emax * (tl + 1) <= 32000 && lo <= -ROW_NEG - 64 && hi + lo <= 32000;
}

template <int NP>
static void launch_reg(bool exact, unsigned n, hipStream_t st, const DpConst &dc, const DpJobDev *jobs, const int32_t *ids, const uint8_t *d_q, const uint8_t *d_t,
                       uint8_t *bt, mm355_dpres_t *res, unsigned long long *cells)
{
	// MM355_DP_LDS_PAD: bytes of (unused) LDS requested per wave, a cap on the extension waves per CU (160 KB / pad) that keeps wave slots
	// free for the latency-bound front kernels of the other contexts
	static const size_t pad = [] { const char *e = getenv("MM355_DP_LDS_PAD"); return (size_t)(e? atoi(e) : 0); }();
	if (exact) hipLaunchKernelGGL((k_ksw_reg<NP, true>), dim3(n), dim3(64), pad, st, dc, jobs, ids, (int)n, d_q, d_t, bt, res, cells);
	else hipLaunchKernelGGL((k_ksw_reg<NP, false>), dim3(n), dim3(64), pad, st, dc, jobs, ids, (int)n, d_q, d_t, bt, res, cells);
}


// which full-matrix row-sweep kernel takes a problem: 0 none (the anti-diagonal kernels), 1 k_ksw_row, 2 k_ksw_rowl -- ONE definition for the
// launch layout (mm355_dp_run) and for the HBM budget of a round (mm355_dp_matrix_bytes)
static int row_class(const DpConst &dc, int qlen, int tlen, int w_in, int flag)
{
	static const bool legacy = [] { const char *e = getenv("MM355_DP_LEGACY"); return e && atoi(e) != 0; }();
	static const bool use_row = [] { const char *e = getenv("MM355_DP_ROW"); return !legacy && !(e && atoi(e) == 0); }();   // MM355_DP_ROW=0: anti-diagonal kernels only
	static const bool use_rowl = [] { const char *e = getenv("MM355_DP_ROWL"); return !(e && atoi(e) == 0); }();                           // MM355_DP_ROWL=0: no eight-wave row sweep
	const int w = w_in < 0? std::max(qlen, tlen) : w_in;
	// ... only for a regular two-piece cost (after ksw2's ordering: e > e2, or two identical pieces).  Otherwise the boundary row and
	// column of U:ksw2_extd2_sse.c follow the dearer piece (long_thres <= 1), H(t,q) - H(t-1,q-1) can exceed the match score next to
	// them, the kernel's clamp `z = min(z, sc_mch)` becomes active and the result is no longer the plain recurrence the row sweep
	// computes (found by the option fuzzer: scoring=(4,10,3,3,12,3)); those options keep the literal anti-diagonal kernels.
	const bool regular = dc.e > dc.e2 || (dc.e == dc.e2 && dc.q == dc.q2);
	const bool row_kind = dc.valid && use_row && regular && (flag & EZ_APPROX_MAX) && !(flag & (EZ_APPROX_DROP | EZ_EXTZ_ONLY | EZ_SCORE_ONLY)) && w >= qlen + tlen;
	if (!row_kind) return 0;
	// (the int16 value range is checked per problem: a user scoring such as e2 >= 7 or a large match score would wrap the packed halves)
	if (tlen <= ROW_MAX_T) return qlen + tlen <= ROW_MAX_QT && rowl_range_ok(dc, qlen, tlen <= 256? 256 : tlen <= 512? 512 : 1024)? 1 : 0;
	return use_rowl && tlen <= ROWL_MAX_T && qlen <= ROWL_MAX_Q && rowl_range_ok(dc, qlen, tlen)? 2 : 0;
}

__global__ void k_dp_patch(DpJobDev *jobs, const int32_t *ids, const DpJobDev *nj, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) jobs[ids[i]] = nj[i]; }

// ---- band kernels (mm355_dpband.h): which problems take them, with which band
struct BandPlan { int nsb, dlo, lmin, bw; };   // nsb = 0: the full-matrix kernels; bw = 64 (two problems per wave, nsb = 1), 128, 256, 512
static int band_ubound(const DpConst &dc, int qlen, int tlen, int d)   // no path through diagonal d (outside [min(0, D0), max(0, D0)]) scores more
{
	const int D0 = tlen - qlen, g1 = d < 0? -d : d, g2 = D0 - d < 0? d - D0 : D0 - d;
	auto cost = [&](int g) { if (g <= 0) return 0; const int c1 = dc.q + g * dc.e, c2 = dc.q2 + g * dc.e2; return c1 < c2? c1 : c2; };
	int a = dc.sc_mch; if (dc.sc_mis > a) a = dc.sc_mis; if (dc.sc_N > a) a = dc.sc_N; if (a < 0) a = 0;
	int m = (qlen + tlen - g1 - g2) / 2; if (m < 0) m = 0;
	return a * m - cost(g1) - cost(g2);
}
static BandPlan band_plan(const DpConst &dc, int qlen, int tlen, int full_sets)
{
	BandPlan bp; bp.nsb = 0; bp.dlo = 0; bp.lmin = 0; bp.bw = 0;
	static const bool use_band = [] { const char *e = getenv("MM355_DP_BAND"); return !(e && atoi(e) == 0); }();        // MM355_DP_BAND=0: full-matrix kernels only
	static const double thr = [] { const char *e = getenv("MM355_DP_BAND_THR"); return e? atof(e) : 0.65; }();          // a band is tried when a score of thr x (all matches) would prove it
	static const int force = [] { const char *e = getenv("MM355_DP_BAND_FORCE"); return e? atoi(e) : 0; }();            // test hook: this many register sets whenever the geometry allows
	if (!use_band) return bp;
	const int D0 = tlen - qlen, lo = D0 < 0? D0 : 0, hi = D0 > 0? D0 : 0, span = hi - lo + 1;
	const int emax = dc.e > dc.e2? dc.e : dc.e2;
	auto cost = [&](int g) { const int c1 = dc.q + g * dc.e, c2 = dc.q2 + g * dc.e2; return c1 < c2? c1 : c2; };
	int a = dc.sc_mch > 0? dc.sc_mch : 0;
	const int n = qlen < tlen? qlen : tlen;
	static const double thr_half = [] { const char *e = getenv("MM355_DP_BAND_THR_HALF"); return e? atof(e) : 0.72; }();   // ... for the 64-diagonal band (0 = never)
	for (int nsb = 0; nsb <= 4; nsb = nsb? nsb * 2 : 1) {   // nsb = 0: half a register set (64 diagonals, k_ksw_band2)
		const int W = nsb? 128 * nsb : 64;
		if (force && (nsb? nsb : 64) != force) continue;
		if (nsb && nsb >= full_sets && !force) break;                   // no narrower than the full matrix
		if (W < span + 2 * BAND_MIN_MARGIN) continue;
		// int16 range: real values stay above -(cost(qlen + 1) + cost(tlen + 1)) - ..., the cells left of the border column start at ROW_NEG and
		// drift by at most a per row for W rows; prefix terms G + d e with |d| <= W + span
		if (cost(qlen + 1) + cost(tlen + 1) + 256 > 7000 || a * (W + 8) > 4000 || (W + span + 8) * emax > 4000 || a * n + (W + span + 8) * emax > 30000) continue;
		const int dlo = lo - (W - span) / 2;
		int u1 = band_ubound(dc, qlen, tlen, dlo - 1), u2 = band_ubound(dc, qlen, tlen, dlo + W);
		const int lmin = (u1 > u2? u1 : u2) + 1;
		if (!force && (double)lmin > (nsb? thr : thr_half) * (double)(a * n)) continue;   // not worth a try: only a nearly perfect alignment would prove this band
		bp.nsb = nsb? nsb : 1; bp.dlo = dlo; bp.lmin = lmin; bp.bw = W;
		return bp;
	}
	return bp;
}

// runs n jobs whose code strings are already on the device (d_q/d_t).  `jobs` is host memory that stays valid until the call
// returns (pinned when it comes from the mapping path).  Results: res_out -> c->h_res (pinned, valid until the next call),
// bytes of direction matrix mm355_dp_run lays out for one extension problem (the caller cuts a round so that a launch fits its HBM budget):
// the row sweep's tiled matrix (mm355_dprow.h) for the full-band approximate fills it takes, the reference's anti-diagonal layout otherwise
size_t mm355_dp_matrix_bytes(const mm355_mapopt_t *mo, const DpConst &dc, int qlen, int tlen, int w_in, int flag)
{
	if (qlen <= 0 || tlen <= 0) return 0;
	if (mo->max_sw_mat > 0 && (int64_t)tlen * qlen > mo->max_sw_mat) return 0;
	const int w = w_in < 0? std::max(qlen, tlen) : w_in;
	int n_col_ = std::min(qlen, tlen);
	n_col_ = ((n_col_ < w + 1? n_col_ : w + 1) + 15) / 16 + 1;
	const int T = (tlen + 15) / 16 * 16;
	const int rk = row_class(dc, qlen, tlen, w_in, flag);
	const bool row = rk == 1, rowl = rk == 2;
	if (row || rowl) {
		const BandPlan bp = band_plan(dc, qlen, tlen, rowl? 64 : tlen <= 256? 2 : tlen <= 512? 4 : 8);
		if (bp.nsb) return band_matrix_bytes(qlen, bp.bw) + 64;
		return row_matrix_bytes(qlen, T) + 64;   // (+ the alignment of its first tile)
	}
	return ((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16;
}

// cigar_out -> dense CIGAR arena in *arena (pinned, owned by the caller's batch).
int mm355_dp_run(mm355_ctx *c, const mm355_mapopt_t *mo, DpJobDev *jobs, size_t n, const uint8_t *d_q, const uint8_t *d_t, HBuf *arena,
                 const mm355_dpres_t **res_out, const uint32_t **cigar_out)
{
	*res_out = 0; *cigar_out = 0;
	if (n == 0) return 0;
	DpConst dc = mm355_dp_const(mo);
	static const bool legacy = [] { const char *e = getenv("MM355_DP_LEGACY"); return e && atoi(e) != 0; }();
	static const bool use_regw = [] { const char *e = getenv("MM355_DP_REGW"); return !legacy && !(e && atoi(e) == 0); }();               // MM355_DP_REGW=0: no windowed register kernel
	static const bool regw8 = [] { const char *e = getenv("MM355_DP_REGW8"); return !(e && atoi(e) == 0); }();                             // MM355_DP_REGW8=0: the single-wave kernel of round 2 (same results)
	size_t regw_seq = 0;                       // LDS bytes of the longest (query, target) pair of the eight-wave kernel in this round
	const DpClass *classes = legacy? DP_CLASSES_LEGACY : DP_CLASSES;
	// lay out per-job work areas; group = size class * 2 + exact
	size_t p_tot = 0, off_tot = 0, cig_tot = 0, st_tot = 0;
	std::vector<uint8_t> grp(n);
	size_t n_grp[DP_N_GROUP + 1] = {0};
	const int LB = 1024;                       // backtrack order: buckets of (qlen + tlen) / 8, longest first (approximate is enough)
	std::vector<uint32_t> lcnt(LB + 1, 0);
	for (size_t i = 0; i < n; ++i) {
		DpJobDev &j = jobs[i];
		j.skip = (mo->max_sw_mat > 0 && (int64_t)j.tlen * j.qlen > mo->max_sw_mat) || !dc.valid;
		int w = j.w < 0? std::max(j.qlen, j.tlen) : j.w;
		int n_col_ = std::min(j.qlen, j.tlen);
		n_col_ = ((n_col_ < w + 1? n_col_ : w + 1) + 15) / 16 + 1;
		int T = (j.tlen + 15) / 16 * 16;
		j.p_off = (int64_t)p_tot; j.off_off = (int64_t)off_tot; j.cig_off = (int64_t)cig_tot; j.st_off = 0;
		int g = 0;
		j.pad = 0; j.dlo = 0; j.lmin = 0; j.bw = 0; j.rsv = 0;
		if (j.qlen > 0 && j.tlen > 0 && !j.skip) {
			cig_tot += (size_t)j.qlen + j.tlen + 2;
			// gap fills whose band never binds, without z-drop on the approximate score: the row sweep (mm355_dprow.h; row_class above)
			const int rk = row_class(dc, j.qlen, j.tlen, j.w, j.flag);
			const bool row = rk == 1, rowl = rk == 2;
			if (row || rowl) {
				const BandPlan bp = band_plan(dc, j.qlen, j.tlen, rowl? 64 : j.tlen <= 256? 2 : j.tlen <= 512? 4 : 8);
				p_tot = (p_tot + 63) & ~(size_t)63;            // tiles are 64-byte lines
				j.p_off = (int64_t)p_tot;
				if (bp.nsb) {                                  // a band of 128 * nsb diagonals with a sufficiency proof (mm355_dpband.h)
					j.pad = 2; j.dlo = bp.dlo; j.lmin = bp.lmin; j.bw = bp.bw;
					p_tot += band_matrix_bytes(j.qlen, bp.bw);
					g = bp.bw == 64? DP_G_BANDH : bp.nsb == 1? DP_G_BAND1 : bp.nsb == 2? DP_G_BAND2 : DP_G_BAND4;
				} else {
					j.pad = 1;
					p_tot += row_matrix_bytes(j.qlen, T);
					g = rowl? DP_G_ROWL : j.tlen <= 256? DP_G_ROW2 : j.tlen <= 512? DP_G_ROW4 : DP_G_ROW8;
				}
			} else {
				p_tot += ((size_t)(j.qlen + j.tlen - 1) * n_col_ + 1) * 16;
				int cls = 0;
				while (cls < DP_N_CLASS - 1 && T > classes[cls].cap) ++cls;
				if (use_regw && T > 1024 && !(j.flag & EZ_APPROX_MAX) && w <= DP_WIN_MAX_W && (!regw8 || ((j.qlen + 15) & ~15) + T <= MW_SEQ_MAX)) {
					g = DP_G_REGW;
					const size_t sb = (size_t)((j.qlen + 15) & ~15) + (size_t)T;
					if (sb > regw_seq) regw_seq = sb;
				}
				else {
					if (cls == DP_N_CLASS - 1) { j.st_off = (int64_t)st_tot; st_tot += T; }
					g = cls * 2 + ((j.flag & EZ_APPROX_MAX)? 0 : 1);
				}
			}
		}
		grp[i] = (uint8_t)g; ++n_grp[g];
		int lb = (j.qlen + j.tlen) >> 3; if (lb < 0) lb = 0; if (lb >= LB) lb = LB - 1;
		++lcnt[LB - 1 - lb];
	}
	if (const char *dump = getenv("MM355_DP_DUMP")) {   // diagnostics: (qlen, tlen, w, flag) of every job of this launch group, appended
		static std::mutex dm; std::lock_guard<std::mutex> lk(dm);
		if (FILE *fp = fopen(dump, "ab")) { for (size_t i = 0; i < n; ++i) { int32_t v[4] = { jobs[i].qlen, jobs[i].tlen, jobs[i].w, jobs[i].flag }; fwrite(v, 4, 4, fp); } fclose(fp); }
	}
	if (c->h_ids.ensure((2 * n + 64) * 4)) return MM355_ENOMEM;
	int32_t *h_ids = (int32_t*)c->h_ids.p, *h_ord = h_ids + n + 8;
	size_t grp_off[DP_N_GROUP + 1], n_half_right = 0;
	{
		size_t acc = 0;
		for (int g = 0; g < DP_N_GROUP; ++g) { grp_off[g] = acc; acc += n_grp[g]; }
		grp_off[DP_N_GROUP] = acc;
		size_t cur[DP_N_GROUP];
		for (int g = 0; g < DP_N_GROUP; ++g) cur[g] = grp_off[g];
		uint32_t a2 = 0;
		for (int k = 0; k <= LB; ++k) { uint32_t t = lcnt[k]; lcnt[k] = a2; a2 += t; }
		for (size_t i = 0; i < n; ++i) {
			int lb = (jobs[i].qlen + jobs[i].tlen) >> 3; if (lb < 0) lb = 0; if (lb >= LB) lb = LB - 1;
			h_ord[lcnt[LB - 1 - lb]++] = (int32_t)i;
		}
		// launch lists in the same order (most anti-diagonals first): a wave is one alignment, so the longest sweeps of a class start at
		// t = 0 and the short ones fill in behind them instead of the launch ending on a late-started long one
		for (size_t k = 0; k < n; ++k) { const int32_t i = h_ord[k]; h_ids[cur[grp[i]]++] = i; }
		// k_ksw_band2 pairs neighbours of its list and both share KSW_EZ_RIGHT: the list is cut in two by that flag (order kept)
		n_half_right = (size_t)(std::stable_partition(h_ids + grp_off[DP_G_BANDH], h_ids + grp_off[DP_G_BANDH] + n_grp[DP_G_BANDH], [&](int32_t i) { return (jobs[i].flag & EZ_RIGHT) != 0; }) - (h_ids + grp_off[DP_G_BANDH]));
	}
	if (c->dp_jobs.ensure(n * sizeof(DpJobDev)) || c->dp_res.ensure(n * sizeof(mm355_dpres_t)) || c->dp_bt.ensure(p_tot + 64, 4) ||
	    c->dp_work.ensure((off_tot + 16) * 4 + (2 * n + 32) * 4) || c->dp_cig.ensure((cig_tot + 16) * 4) || c->dp_dense.ensure((cig_tot + 16) * 4) ||
	    c->dp_H.ensure((st_tot + 16) * 12)) return MM355_ENOMEM;
	// One extension round saturates the GPU.  Rounds of different contexts take turns (per device): run side by side they would all
	// finish late together and the contexts would march in lock-step -- GPU idle while every host tail runs, hosts idle while every
	// round runs.  Taking turns lets the first finisher start its host tail while the next round has the whole GPU.
	struct Turn {   // counting semaphore: MM355_DP_TURNS rounds at a time per device (default 1, 0 = unlimited)
		std::mutex m; std::condition_variable cv; int busy = 0;
		void lock(int cap) { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return busy < cap; }); ++busy; }
		void unlock() { { std::lock_guard<std::mutex> lk(m); --busy; } cv.notify_one(); }
	};
	static Turn dp_turn[16];
	static const int turn_cap = [] { const char *e = getenv("MM355_DP_TURNS"); return e? atoi(e) : 1; }();
	const bool take_turns = turn_cap > 0;
	struct TurnGuard { Turn *t = 0; void lock(Turn *x, int cap) { x->lock(cap); t = x; } void unlock() { if (t) { t->unlock(); t = 0; } } ~TurnGuard() { unlock(); } } turn;
	Turn *my_turn = &dp_turn[c->dev & 15];
	HIPCHK(hipMemcpyAsync(c->dp_jobs.p, jobs, n * sizeof(DpJobDev), hipMemcpyHostToDevice, c->st));
	int32_t *d_off = c->dp_work.as<int32_t>();
	int32_t *d_ids = d_off + off_tot + 16;
	uint64_t *d_S = c->dp_H.as<uint64_t>();
	int32_t *d_H = (int32_t*)(d_S + st_tot + 8);
	unsigned long long *d_cells = c->counters.as<unsigned long long>() + 4, *d_dense = c->counters.as<unsigned long long>() + 5;
	double t_turn0 = 0;
	bool redo_timed = false;
	HIPCHK(hipMemsetAsync(d_dense, 0, 8, c->st));
	unsigned long long *d_gcells = c->counters.as<unsigned long long>() + CTR_GCELLS_OFF;   // cells per group [CTR_GROUPS][DP_CTR_SPREAD]
	HIPCHK(hipMemsetAsync(d_cells, 0, 8, c->st));
	HIPCHK(hipMemsetAsync(d_gcells, 0, CTR_GCELLS_WORDS * 8, c->st));
	// band kernels: [0] = problems whose proof failed, then their ids; behind it the launch lists of their second run
	const size_t n_band = n_grp[DP_G_BAND1] + n_grp[DP_G_BAND2] + n_grp[DP_G_BAND4] + n_grp[DP_G_BANDH];
	if (c->dp_fail.ensure((2 * n_band + 64) * 4 + (n_band + 8) * sizeof(DpJobDev)) || c->h_fail.ensure((2 * n_band + 64) * 4 + (n_band + 8) * sizeof(DpJobDev))) return MM355_ENOMEM;
	int32_t *d_fail = c->dp_fail.as<int32_t>();
	if (n_band) HIPCHK(hipMemsetAsync(d_fail, 0, 4, c->st));
	// every group gets its own HIP stream: the few long alignments of the big classes run concurrently with the thousands of
	// short ones instead of holding the GPU alone (same-stream launches would serialise the classes)
	HIPCHK(hipMemcpyAsync(d_ids, h_ids, (2 * n + 8) * 4, hipMemcpyHostToDevice, c->st));   // launch lists + backtrack order (pinned source)
	if (c->dp_up_ev == 0) HIPCHK(hipEventCreateWithFlags(&c->dp_up_ev, hipEventDisableTiming));
	HIPCHK(hipEventRecord(c->dp_up_ev, c->st));   // the group streams start after the uploads
	(void)hipFuncSetAttribute((const void*)k_ksw_extd2<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 12288 * 12);
	// Streams (created back to back at context creation, so they sit on different hardware queues): 0 approx targets <= 256 (the wide
	// grids), 1 every exact register class, 2 approx 1024, 3 approx 512, 4 the eight-wave kernel.  The long-target classes (4096 /
	// 12288 LDS state, HBM state; approx and exact) are ONE launch: a few dozen latency-bound alignments that must not queue behind
	// one another, nor in front of the register classes on a shared hardware queue (they are not part of the turn).
	static const bool legacy_groups = [] { const char *e = getenv("MM355_DP_SPLIT_LONG"); return e && atoi(e) != 0; }();
	const DpJobDev *dj = c->dp_jobs.as<DpJobDev>();
	mm355_dpres_t *dres = c->dp_res.as<mm355_dpres_t>();
	auto group_stream = [&](int sidx, hipStream_t *out) -> int { return mm355_dp_stream(c, sidx, out); };
	auto group_begin = [&](int g, hipStream_t gst) -> int {
		if (c->dp_ev[g] == 0) HIPCHK(hipEventCreateWithFlags(&c->dp_ev[g], hipEventDisableTiming));
		HIPCHK(hipStreamWaitEvent(gst, c->dp_up_ev, 0));
		if (c->dp_ev0[g] == 0) HIPCHK(hipEventCreate(&c->dp_ev0[g]));
		if (c->dp_ev1[g] == 0) HIPCHK(hipEventCreate(&c->dp_ev1[g]));
		HIPCHK(hipEventRecord(c->dp_ev0[g], gst));
		return 0;
	};
	auto group_end = [&](int g, hipStream_t gst, bool in_turn) -> int {
		HIPCHK(hipEventRecord(c->dp_ev1[g], gst));
		HIPCHK(hipEventRecord(c->dp_ev[g], gst));
		if (in_turn) HIPCHK(hipStreamWaitEvent(c->st, c->dp_ev[g], 0));   // the eight-wave kernels are joined after the turn (below)
		return 0;
	};
	// The long-target classes are two launches: targets <= 4096 (49 KB of LDS state: three alignments per CU) on stream 4, and longer ones
	// (147 KB: one per CU; state in HBM beyond 12288 positions) on stream 5; approx and exact alignments share a launch.
	// (MM355_DP_LONG_NT=1024: sixteen waves per alignment -- measured slower, 1019 vs 880 ms/step: the wider barrier costs more than
	// the saved chunk rounds.)
	static const bool nt512 = [] { const char *e = getenv("MM355_DP_LONG_NT"); return !(e && atoi(e) == 1024); }();
	struct LongLaunch { int g0, g1, cap, sidx; };
	static const LongLaunch long_launch[2] = { { 10, 14, 12288, 5 }, { 8, 10, 4096, 4 } };   // (groups 14, 15 are the row kernels)   // the longest sweeps first
	bool long_used[2] = { false, false };
	if (!legacy && !legacy_groups) for (int li = 0; li < 2; ++li) {
		const LongLaunch &ll = long_launch[li];
		const size_t nl = grp_off[ll.g1] - grp_off[ll.g0];
		if (nl == 0) continue;
		long_used[li] = true;
		hipStream_t gst; int rc2;
		if ((rc2 = group_stream(ll.sidx, &gst))) return rc2;
		if ((rc2 = group_begin(ll.g0, gst))) return rc2;
		if (nt512)
			hipLaunchKernelGGL(k_ksw_extd2<512>, dim3((unsigned)nl), dim3(512), (size_t)ll.cap * 12, gst, dc, dj, d_ids + grp_off[ll.g0], (int)nl, d_q, d_t, c->dp_bt.as<uint8_t>(), d_off,
			                   c->dp_cig.as<uint32_t>(), d_S, d_H, dres, ll.cap, d_gcells, c->dp_dense.as<uint32_t>(), d_dense, 4);
		else {
			(void)hipFuncSetAttribute((const void*)k_ksw_extd2<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 12288 * 12);
			hipLaunchKernelGGL(k_ksw_extd2<1024>, dim3((unsigned)nl), dim3(1024), (size_t)ll.cap * 12, gst, dc, dj, d_ids + grp_off[ll.g0], (int)nl, d_q, d_t, c->dp_bt.as<uint8_t>(), d_off,
			                   c->dp_cig.as<uint32_t>(), d_S, d_H, dres, ll.cap, d_gcells, c->dp_dense.as<uint32_t>(), d_dense, 4);
		}
		if ((rc2 = group_end(ll.g0, gst, false))) return rc2;
	}
	if (n_grp[DP_G_REGW]) {   // the long narrow-band extensions: one wave each, thousands of anti-diagonals -- the longest latency chain of a round
		hipStream_t gst; int rc2;
		if ((rc2 = group_stream(7, &gst))) return rc2;
		if ((rc2 = group_begin(DP_G_REGW, gst))) return rc2;
		// eight waves per alignment (mm355_dpmw.h), the sequences of a block in (dynamic) LDS
		// (per launch, not once per process: the attribute belongs to the function on the CURRENT device, and one process may drive several)
		if (regw8 && regw_seq + 64 > 32768 && hipFuncSetAttribute((const void*)k_ksw_regw8, hipFuncAttributeMaxDynamicSharedMemorySize, MW_SEQ_MAX + 64) != hipSuccess) return MM355_EHIP;
		if (regw8) hipLaunchKernelGGL(k_ksw_regw8, dim3((unsigned)n_grp[DP_G_REGW]), dim3(MW_THREADS), regw_seq + 64, gst, dc, dj, d_ids + grp_off[DP_G_REGW], (int)n_grp[DP_G_REGW], d_q, d_t,
		                              c->dp_bt.as<uint8_t>(), dres, d_gcells + DP_CTR_SPREAD * DP_G_REGW);
		else hipLaunchKernelGGL(k_ksw_regw, dim3((unsigned)n_grp[DP_G_REGW]), dim3(64), 0, gst, dc, dj, d_ids + grp_off[DP_G_REGW], (int)n_grp[DP_G_REGW], d_q, d_t,
		                        c->dp_bt.as<uint8_t>(), dres, d_gcells + DP_CTR_SPREAD * DP_G_REGW);
		if ((rc2 = group_end(DP_G_REGW, gst, false))) return rc2;
	}
	if (n_grp[DP_G_ROWL]) {   // the eight-wave row sweep of the long full-band fills: a few hundred blocks at most, outside the turn as well
		hipStream_t gst; int rc2;
		if ((rc2 = group_stream(6, &gst))) return rc2;
		if ((rc2 = group_begin(DP_G_ROWL, gst))) return rc2;
		hipLaunchKernelGGL(k_ksw_rowl, dim3((unsigned)n_grp[DP_G_ROWL]), dim3(64 * ROWL_WAVES), 0, gst, dc, dj, d_ids + grp_off[DP_G_ROWL], (int)n_grp[DP_G_ROWL], d_q, d_t,
		                   c->dp_bt.as<uint8_t>(), dres, d_gcells + DP_CTR_SPREAD * DP_G_ROWL);
		if ((rc2 = group_end(DP_G_ROWL, gst, false))) return rc2;
	}
	// (the launches above -- the long-target kernels -- are not part of the turn: a few dozen latency-bound alignments that leave the GPU
	// almost empty; started BEFORE the turn is taken they run while this context waits for it)
	// everything this round depends on (code-string gather, descriptor uploads) is finished BEFORE the turn is taken: the turn then holds
	// nothing but extension kernels
	if (take_turns) { HIPCHK(mm355_wait_stream(c->st)); turn.lock(my_turn, turn_cap); }
	t_turn0 = mm355_now_ms();
	{
		EvTimer2 tm(c, &c->stats.ms_dp);
		for (int g = DP_N_GROUP - 1; g >= 0; --g) {   // big problems first
			if (n_grp[g] == 0 || g == DP_G_ROWL || g == DP_G_REGW) continue;
			if (g >= DP_G_ROW2) {   // the row sweep of the full-band approximate fills: the wide throughput grids of a round
				hipStream_t gst; int rc2;
				if ((rc2 = group_stream(g == DP_G_ROW2 || g == DP_G_BAND1 || g == DP_G_BANDH? 0 : g == DP_G_ROW4 || g == DP_G_BAND2? 3 : 2, &gst))) return rc2;
				if ((rc2 = group_begin(g, gst))) return rc2;
				if (g == DP_G_BANDH) {
					const size_t nr = n_half_right, nl = n_grp[g] - nr;
					if (nr) hipLaunchKernelGGL(k_ksw_band2, dim3((unsigned)((nr + 1) / 2)), dim3(64), 0, gst, dc, dj, d_ids + grp_off[g], (int)nr, d_q, d_t, c->dp_bt.as<uint8_t>(), dres, d_gcells + DP_CTR_SPREAD * g, d_fail);
					if (nl) hipLaunchKernelGGL(k_ksw_band2, dim3((unsigned)((nl + 1) / 2)), dim3(64), 0, gst, dc, dj, d_ids + grp_off[g] + nr, (int)nl, d_q, d_t, c->dp_bt.as<uint8_t>(), dres, d_gcells + DP_CTR_SPREAD * g, d_fail);
				}
				else if (g == DP_G_BAND1) hipLaunchKernelGGL(k_ksw_band<1>, dim3((unsigned)n_grp[g]), dim3(64), 0, gst, dc, dj, d_ids + grp_off[g], (int)n_grp[g], d_q, d_t, c->dp_bt.as<uint8_t>(), dres, d_gcells + DP_CTR_SPREAD * g, d_fail);
				else if (g == DP_G_BAND2) hipLaunchKernelGGL(k_ksw_band<2>, dim3((unsigned)n_grp[g]), dim3(64), 0, gst, dc, dj, d_ids + grp_off[g], (int)n_grp[g], d_q, d_t, c->dp_bt.as<uint8_t>(), dres, d_gcells + DP_CTR_SPREAD * g, d_fail);
				else if (g == DP_G_BAND4) hipLaunchKernelGGL(k_ksw_band<4>, dim3((unsigned)n_grp[g]), dim3(64), 0, gst, dc, dj, d_ids + grp_off[g], (int)n_grp[g], d_q, d_t, c->dp_bt.as<uint8_t>(), dres, d_gcells + DP_CTR_SPREAD * g, d_fail);
				else if (g == DP_G_ROW2) hipLaunchKernelGGL(k_ksw_row<2>, dim3((unsigned)n_grp[g]), dim3(64), 0, gst, dc, dj, d_ids + grp_off[g], (int)n_grp[g], d_q, d_t, c->dp_bt.as<uint8_t>(), dres, d_gcells + DP_CTR_SPREAD * g);
				else if (g == DP_G_ROW4) hipLaunchKernelGGL(k_ksw_row<4>, dim3((unsigned)n_grp[g]), dim3(64), 0, gst, dc, dj, d_ids + grp_off[g], (int)n_grp[g], d_q, d_t, c->dp_bt.as<uint8_t>(), dres, d_gcells + DP_CTR_SPREAD * g);
				else hipLaunchKernelGGL(k_ksw_row<8>, dim3((unsigned)n_grp[g]), dim3(64), 0, gst, dc, dj, d_ids + grp_off[g], (int)n_grp[g], d_q, d_t, c->dp_bt.as<uint8_t>(), dres, d_gcells + DP_CTR_SPREAD * g);
				if ((rc2 = group_end(g, gst, true))) return rc2;
				continue;
			}
			const DpClass &k = classes[g >> 1];
			if (k.kind == 2 && !legacy && !legacy_groups) continue;   // launched above
			const int sidx = k.kind != 0? 4 + (g - 8) : (g & 1)? 1 : g >= 6? 2 : g >= 4? 3 : 0;
			hipStream_t gst; int rc2;
			if ((rc2 = group_stream(sidx, &gst))) return rc2;
			if ((rc2 = group_begin(g, gst))) return rc2;
			unsigned long long *gc = d_gcells + DP_CTR_SPREAD * g;
			const unsigned nj = (unsigned)n_grp[g];
			const int32_t *gid = d_ids + grp_off[g];
			if (k.kind == 0) {
				const bool ex = g & 1;
				if (k.np == 1) launch_reg<1>(ex, nj, gst, dc, dj, gid, d_q, d_t, c->dp_bt.as<uint8_t>(), dres, gc);
				else if (k.np == 2) launch_reg<2>(ex, nj, gst, dc, dj, gid, d_q, d_t, c->dp_bt.as<uint8_t>(), dres, gc);
				else if (k.np == 4) launch_reg<4>(ex, nj, gst, dc, dj, gid, d_q, d_t, c->dp_bt.as<uint8_t>(), dres, gc);
				else launch_reg<8>(ex, nj, gst, dc, dj, gid, d_q, d_t, c->dp_bt.as<uint8_t>(), dres, gc);
			} else if (k.kind == 1)
				hipLaunchKernelGGL(k_ksw_extd2<64>, dim3(nj), dim3(64), (size_t)k.cap * 12, gst, dc, dj, gid, (int)nj, d_q, d_t, c->dp_bt.as<uint8_t>(), d_off,
				                   c->dp_cig.as<uint32_t>(), d_S, d_H, dres, k.cap, gc, c->dp_dense.as<uint32_t>(), d_dense, -1);
			else
				hipLaunchKernelGGL(k_ksw_extd2<512>, dim3(nj), dim3(512), (size_t)k.cap * 12, gst, dc, dj, gid, (int)nj, d_q, d_t, c->dp_bt.as<uint8_t>(), d_off,
				                   c->dp_cig.as<uint32_t>(), d_S, d_H, dres, k.cap, gc, c->dp_dense.as<uint32_t>(), d_dense, -1);
			if ((rc2 = group_end(g, gst, k.kind == 0 && !(g & 1)))) return rc2;   // in the turn: the approximate register classes; the exact ones are small latency-bound grids
		}
		// band kernels whose proof failed (a divergent stretch, a long indel: the optimal path may leave the band): the same problems again, on
		// the full-matrix row sweep, inside the same turn; their direction matrices go to a buffer of their own
		bool rowl_redo = false;
		if (n_band) {
			int32_t *h_fail = (int32_t*)c->h_fail.p;
			HIPCHK(hipMemcpyAsync(h_fail, d_fail, (1 + n_band) * 4, hipMemcpyDeviceToHost, c->st));   // (c->st has joined every band group: group_end)
			HIPCHK(mm355_wait_stream(c->st));
			const size_t n_fail = (size_t)h_fail[0];
			c->stats.n_dp_band += (int64_t)n_band; c->stats.n_dp_band_redo += (int64_t)n_fail;
			if (n_fail) {
				std::sort(h_fail + 1, h_fail + 1 + n_fail);                  // (the device appended them in no particular order)
				int32_t *h_rid = h_fail + 1 + n_band;                        // launch lists of the second run: row<8>, row<4>, row<2>, rowl
				DpJobDev *h_rj = (DpJobDev*)((char*)c->h_fail.p + (2 * n_band + 64) * 4);
				size_t p2 = 0, cnt[4] = {0, 0, 0, 0}, pos[4];
				auto cls_of = [&](const DpJobDev &j) { return j.tlen > ROW_MAX_T? 3 : j.tlen <= 256? 2 : j.tlen <= 512? 1 : 0; };
				for (size_t k = 0; k < n_fail; ++k) ++cnt[cls_of(jobs[h_fail[1 + k]])];
				pos[0] = 0; pos[1] = cnt[0]; pos[2] = pos[1] + cnt[1]; pos[3] = pos[2] + cnt[2];
				for (size_t k = 0; k < n_fail; ++k) {
					DpJobDev &j = jobs[h_fail[1 + k]];
					j.pad = 1 | 4; j.p_off = (int64_t)p2; j.bw = 0;
					p2 += (row_matrix_bytes(j.qlen, (j.tlen + 15) / 16 * 16) + 63) & ~(size_t)63;
					h_rj[k] = j;
					h_rid[pos[cls_of(j)]++] = h_fail[1 + k];
				}
				if (c->dp_bt2.ensure(p2 + 64, 4)) return MM355_ENOMEM;
				int32_t *d_rid = d_fail + 1 + n_band; DpJobDev *d_rj = (DpJobDev*)((char*)c->dp_fail.p + (2 * n_band + 64) * 4);
				HIPCHK(hipMemcpyAsync(d_fail + 1, h_fail + 1, n_fail * 4, hipMemcpyHostToDevice, c->st));
				HIPCHK(hipMemcpyAsync(d_rid, h_rid, n_fail * 4, hipMemcpyHostToDevice, c->st));
				HIPCHK(hipMemcpyAsync(d_rj, h_rj, n_fail * sizeof(DpJobDev), hipMemcpyHostToDevice, c->st));
				hipLaunchKernelGGL(k_dp_patch, dim3((unsigned)((n_fail + 255) / 256)), dim3(256), 0, c->st, c->dp_jobs.as<DpJobDev>(), d_fail + 1, d_rj, (int)n_fail);
				const int g = DP_G_REDO;
				if (c->dp_ev0[g] == 0) HIPCHK(hipEventCreate(&c->dp_ev0[g]));
				if (c->dp_ev1[g] == 0) HIPCHK(hipEventCreate(&c->dp_ev1[g]));
				HIPCHK(hipEventRecord(c->dp_ev0[g], c->st));
				unsigned long long *gc = d_gcells + DP_CTR_SPREAD * g;
				size_t o0 = 0;
				if (cnt[0]) hipLaunchKernelGGL(k_ksw_row<8>, dim3((unsigned)cnt[0]), dim3(64), 0, c->st, dc, dj, d_rid + o0, (int)cnt[0], d_q, d_t, c->dp_bt2.as<uint8_t>(), dres, gc);
				o0 += cnt[0];
				if (cnt[1]) hipLaunchKernelGGL(k_ksw_row<4>, dim3((unsigned)cnt[1]), dim3(64), 0, c->st, dc, dj, d_rid + o0, (int)cnt[1], d_q, d_t, c->dp_bt2.as<uint8_t>(), dres, gc);
				o0 += cnt[1];
				if (cnt[2]) hipLaunchKernelGGL(k_ksw_row<2>, dim3((unsigned)cnt[2]), dim3(64), 0, c->st, dc, dj, d_rid + o0, (int)cnt[2], d_q, d_t, c->dp_bt2.as<uint8_t>(), dres, gc);
				o0 += cnt[2];
				if (cnt[3]) { hipLaunchKernelGGL(k_ksw_rowl, dim3((unsigned)cnt[3]), dim3(64 * ROWL_WAVES), 0, c->st, dc, dj, d_rid + o0, (int)cnt[3], d_q, d_t, c->dp_bt2.as<uint8_t>(), dres, gc); rowl_redo = true; }
				HIPCHK(hipEventRecord(c->dp_ev1[g], c->st));
				redo_timed = true;
			}
		}
		(void)rowl_redo;
		// the turn ends when the extension kernels are done: the backtrack below is a latency-bound pointer walk and, like the result
		// copies, overlaps the next context's round
		// (only the wide register-kernel grids count: the few long alignments of the eight-wave classes are latency chains that
		// leave the GPU almost empty; they keep running while the next context's round starts)
		if (take_turns) { const double tl1 = mm355_now_ms(); HIPCHK(mm355_wait_stream(c->st)); turn.unlock(); const double tl2 = mm355_now_ms(); mm355_trace_add(c, "dpk", t_turn0, tl2); mm355_trace_add(c, "dpk_launch", t_turn0, tl1); }
		if (n_grp[DP_G_ROWL]) HIPCHK(hipStreamWaitEvent(c->st, c->dp_ev[DP_G_ROWL], 0));
		if (n_grp[DP_G_REGW]) HIPCHK(hipStreamWaitEvent(c->st, c->dp_ev[DP_G_REGW], 0));
		if (!legacy && !legacy_groups) { for (int li = 0; li < 2; ++li) if (long_used[li]) HIPCHK(hipStreamWaitEvent(c->st, c->dp_ev[long_launch[li].g0], 0)); }
		else { for (int g = 0; g < 2 * DP_N_CLASS; ++g) if (n_grp[g] && classes[g >> 1].kind != 0) HIPCHK(hipStreamWaitEvent(c->st, c->dp_ev[g], 0)); }
		for (int g = 1; g < 8; g += 2) if (n_grp[g] && classes[g >> 1].kind == 0) HIPCHK(hipStreamWaitEvent(c->st, c->dp_ev[g], 0));   // the exact register classes
		// backtrack: all jobs, longest first so that the lanes of a wave walk paths of similar length
		KtScope ks_bt(c, KT_DP_BACKTRACK, c->st);
		hipLaunchKernelGGL(k_ksw_backtrack, dim3((unsigned)((n + WAVE - 1) / WAVE)), dim3(WAVE), 0, c->st, c->dp_jobs.as<DpJobDev>(), d_ids + n + 8, (int)n,
		                   c->dp_bt.as<uint8_t>(), c->dp_bt2.as<uint8_t>(), c->dp_cig.as<uint32_t>(), c->dp_res.as<mm355_dpres_t>(), c->dp_dense.as<uint32_t>(), d_dense);
	}
	HIPCHK(hipGetLastError());
	if (c->h_res.ensure(n * sizeof(mm355_dpres_t) + 64 + CTR_GCELLS_WORDS * 8)) return MM355_ENOMEM;
	unsigned long long *ctr = (unsigned long long*)((char*)c->h_res.p + n * sizeof(mm355_dpres_t));   // pinned landing zone of the counters
	HIPCHK(hipMemcpyAsync(ctr, d_cells, 16, hipMemcpyDeviceToHost, c->st));
	HIPCHK(hipMemcpyAsync(ctr + 2, d_gcells, CTR_GCELLS_WORDS * 8, hipMemcpyDeviceToHost, c->st));
	HIPCHK(hipMemcpyAsync(c->h_res.p, c->dp_res.p, n * sizeof(mm355_dpres_t), hipMemcpyDeviceToHost, c->st));
	HIPCHK(mm355_wait_stream(c->st));
	if (const char *dump = getenv("MM355_DP_DUMP_BT")) {   // diagnostics: the direction matrices of this launch group, raw
		std::vector<uint8_t> hb(p_tot + 64);
		if (hipMemcpy(hb.data(), c->dp_bt.p, p_tot, hipMemcpyDeviceToHost) == hipSuccess) if (FILE *fp = fopen(dump, "wb")) { fwrite(hb.data(), 1, p_tot, fp); fclose(fp); }
	}
	const size_t n_dense = (size_t)ctr[1];
	if (arena->ensure((n_dense + 16) * 4)) return MM355_ENOMEM;
	if (n_dense) HIPCHK(hipMemcpyAsync(arena->p, c->dp_dense.p, n_dense * 4, hipMemcpyDeviceToHost, c->st));
	HIPCHK(mm355_wait_stream(c->st));
	{
		int64_t tot = 0;
		const bool merged_long = !legacy && getenv("MM355_DP_SPLIT_LONG") == 0;
		for (int g = 0; g < DP_N_GROUP; ++g) {
			// the merged long-target launches are timed as groups 8 (targets <= 4096) and 10 (longer)
			const bool timed_here = merged_long && g >= 8 && g < 14? ((g == 8 && grp_off[10] > grp_off[8]) || (g == 10 && grp_off[14] > grp_off[10])) : n_grp[g] != 0;
			float ms = 0.f;
			if (timed_here && hipEventElapsedTime(&ms, c->dp_ev0[g], c->dp_ev1[g]) == hipSuccess) c->stats.ms_dp_group[g] += ms;
			if (n_grp[g] == 0) continue;
			int64_t gc = 0;
			for (int k = 0; k < DP_CTR_SPREAD; ++k) gc += (int64_t)ctr[2 + DP_CTR_SPREAD * g + k];
			c->stats.dp_cells_group[g] += gc; ++c->stats.n_launch_group[g];
			tot += gc;
		}
		if (redo_timed) {   // the second run of the band problems whose proof failed (its cells are counted again: they were computed twice)
			float ms = 0.f;
			if (hipEventElapsedTime(&ms, c->dp_ev0[DP_G_REDO], c->dp_ev1[DP_G_REDO]) == hipSuccess) c->stats.ms_dp_group[DP_G_REDO] += ms;
			int64_t gc = 0;
			for (int k = 0; k < DP_CTR_SPREAD; ++k) gc += (int64_t)ctr[2 + DP_CTR_SPREAD * DP_G_REDO + k];
			c->stats.dp_cells_group[DP_G_REDO] += gc; ++c->stats.n_launch_group[DP_G_REDO];
		}
		c->stats.dp_cells += tot; c->stats.n_dp_jobs += (int64_t)n; ++c->stats.n_launch_dp;
	}
	*res_out = (const mm355_dpres_t*)c->h_res.p; *cigar_out = (const uint32_t*)arena->p;
	return 0;
}

// a target base for k_extra's forward walk: the 2-bit image, and code 4 inside the N run the pointer has reached
__device__ __forceinline__ uint32_t ex_target(const DevIndex &ix, const uint64_t o, uint32_t &nri)
{
	uint32_t c = ix.S2[o >> 4] >> ((o & 15) << 1) & 3u;
	while (nri < ix.n_nr && ix.nr[2 * nri + 1] <= o) ++nri;
	if (nri < ix.n_nr && ix.nr[2 * nri] <= o) c = 4;
	return c;
}

// ------------------------------------------------------------------ row f2: mm_update_extra's walk, cs and MD on the device
// U:align.c::mm_update_extra (after mm_fix_cigar, which stays on the host: it edits the CIGAR) walks every aligned column of a region for
// mlen / blen / n_ambi and the local-maximum score dp_max; U:format.c::write_cs_core (short form) and write_MD_core walk the same columns
// again for the cs / MD strings.  On the host that was ~17 us of CPU per GRCh38-scale read.  Here: all regions of a batch in one launch after
// the last extension round, one LANE per SEGMENT of at most 64 CIGAR operations and 2048 columns -- a lane per region was a 45 ms tail on a
// 100 kb read, every column two dependent-latency loads; ~150 000 equal segments hide that latency behind one another.  Long match
// operations (HiFi) are cut between segments.  What makes the cut exact (mm355_extra.h):
//   * counts add up;
//   * the score walk s <- max(0, s + d) with running maximum is a max-plus map: a segment leaves (A, m, C, P) and k_extra_compose replays
//     the segments of a region in order.  s is a double as in the reference; every partial sum of integer match scores and float-valued log
//     gap costs is exact in double (< 2^20 in magnitude, fractions of 2^-23), so regrouping the additions changes nothing;
//   * a run of matches that crosses a cut is printed once: a segment leaves the first number it would print to the composer (its `lead` plus
//     the `tail` carried over from the segments before it) and never prints what is still pending at its end.
// The string bytes go to worst-case sized slots and are compacted afterwards (k_extra_scan / k_extra_compact): only real bytes cross PCIe.
__global__ __launch_bounds__(WAVE) void k_extra(DevIndex ix, const uint8_t *rq, const Mm355ExtraJob *segs, int n_segs, const uint32_t *cig, Mm355ExtraScore sc,
                                                 char *cs, Mm355ExtraSegOut *out, int want)
{
	const int k = blockIdx.x * WAVE + threadIdx.x;
	if (k >= n_segs) return;
	const Mm355ExtraJob jb = segs[k];
	const uint8_t *q = rq + jb.q_src;
	const uint64_t tb = ix.seq_off[jb.rid] + (uint64_t)jb.t_st;
	const uint32_t *cg = cig + jb.cig_off;
	char *o = cs + jb.cs_off, *om = cs + jb.md_off;
	const bool want_cs = want & 1, want_md = want & 2;
	int32_t n_out = 0, n_md = 0, qoff = 0, toff = 0, mlen = 0, blen = 0, n_ambi_tot = 0;
	int32_t cs_lead = 0, md_lead = 0, flushed = 0, cs_pre = 0;
	unsigned run = 0, l_md = 0;      // matches pending for cs (inside the current match operation) and for MD (across operations)
	double A = 0.0, m = 1e300, C = -1e300, P = -1e300;
	const char *nt = "acgtn", *NT = "ACGTN";
	// the walk moves forward through the target, so the N run that can contain the current base only ever moves forward too
	uint32_t nri = 0;
	if (ix.n_nr) { uint32_t c_; ref_window(ix, tb, tb + 1, nri, c_); }
#define EX_T(off) ex_target(ix, tb + (uint64_t)(off), nri)
#define EX_PUT(buf, pos, lead, v) do { char bf_[12]; int nb_ = 0; unsigned v_ = (v); do { bf_[nb_++] = (char)('0' + v_ % 10); v_ /= 10; } while (v_); if (lead) buf[pos++] = (lead); while (nb_ > 0) buf[pos++] = bf_[--nb_]; } while (0)
#define EX_FLUSH_CS() do { if (!(flushed & 1)) { cs_lead = (int32_t)run; cs_pre = n_out; flushed |= 1; } else if (run) EX_PUT(o, n_out, ':', run); run = 0; } while (0)
#define EX_FLUSH_MD() do { if (!(flushed & 2)) { md_lead = (int32_t)l_md; flushed |= 2; } else EX_PUT(om, n_md, 0, l_md); l_md = 0; } while (0)
#define EX_STEP(d) do { A += (d); m = A < m? A : m; C = A > C? A : C; const double am_ = A - m; P = am_ > P? am_ : P; } while (0)
	for (int c = 0; c < jb.n_cigar; ++c) {
		const uint32_t op = cg[c] & 0xf, full = cg[c] >> 4;
		const uint32_t l0 = c == 0? (uint32_t)jb.skip0 : 0u, l1 = c == jb.n_cigar - 1 && jb.end_last > 0? (uint32_t)jb.end_last : full;
		const uint32_t len = l1 - l0;
		if (op == 0 || op == 7 || op == 8) {
			int n_ambi = 0, n_diff = 0;
			for (uint32_t l = 0; l < len; ++l) {
				const uint32_t cq = q[qoff + l], ct = EX_T(toff + l);
				if (ct > 3 || cq > 3) ++n_ambi;
				else if (ct != cq) ++n_diff;
				EX_STEP((double)sc.mat[ct * 5 + cq]);
				if (cq == ct) { ++run; ++l_md; }
				else {
					if (want_cs) { EX_FLUSH_CS(); o[n_out++] = '*'; o[n_out++] = nt[ct]; o[n_out++] = nt[cq]; }
					if (want_md) { EX_FLUSH_MD(); om[n_md++] = NT[ct]; }
				}
			}
			if (want_cs && l1 == full) EX_FLUSH_CS();          // the end of a match operation ends its run (a cut does not)
			blen += (int32_t)len - n_ambi; mlen += (int32_t)len - (n_ambi + n_diff); n_ambi_tot += n_ambi;
			toff += (int32_t)len; qoff += (int32_t)len;
		} else if (op == 1) {
			int n_ambi = 0;
			if (want_cs) o[n_out++] = '+';
			for (uint32_t l = 0; l < len; ++l) { const uint32_t cq = q[qoff + l]; if (cq > 3) ++n_ambi; if (want_cs) o[n_out++] = nt[cq]; }
			blen += (int32_t)len - n_ambi; n_ambi_tot += n_ambi;
			EX_STEP(-((double)sc.q + (double)sc.e * (double)mm_log2f_approx((float)(1.0 + (double)len))));
			qoff += (int32_t)len;
		} else if (op == 2) {
			int n_ambi = 0;
			if (want_cs) o[n_out++] = '-';
			if (want_md) { EX_FLUSH_MD(); om[n_md++] = '^'; }
			for (uint32_t l = 0; l < len; ++l) { const uint32_t ct = EX_T(toff + l); if (ct > 3) ++n_ambi; if (want_cs) o[n_out++] = nt[ct]; if (want_md) om[n_md++] = NT[ct]; }
			blen += (int32_t)len - n_ambi; n_ambi_tot += n_ambi;
			EX_STEP(-((double)sc.q + (double)sc.e * (double)mm_log2f_approx((float)(1.0 + (double)len))));
			toff += (int32_t)len;
		} else if (op == 3) toff += (int32_t)len;
	}
#undef EX_T
#undef EX_PUT
#undef EX_FLUSH_CS
#undef EX_FLUSH_MD
#undef EX_STEP
	Mm355ExtraSegOut r;
	r.A = A; r.m = m; r.C = C; r.P = P; r.mlen = mlen; r.blen = blen; r.n_ambi = n_ambi_tot;
	r.cs_len = n_out; r.md_len = n_md; r.flushed = flushed;
	r.cs_lead = (flushed & 1)? cs_lead : (int32_t)run; r.cs_tail = (flushed & 1)? (int32_t)run : 0;      // never flushed: everything it counted joins the carry
	r.md_lead = (flushed & 2)? md_lead : (int32_t)l_md; r.md_tail = (flushed & 2)? (int32_t)l_md : 0;
	r.cs_num = r.md_num = -1; r.cs_dense = r.md_dense = 0; r.cs_pre = (flushed & 1)? cs_pre : 0; r.pad = 0;
	out[k] = r;
}

__device__ inline int ex_digits(unsigned v) { int n = 1; while (v >= 10) { v /= 10; ++n; } return n; }

// exclusive prefix sum of the regions' string lengths (one block of 256 threads: a batch has a few thousand regions; the per-segment offsets
// inside a region come from k_extra_compose -- a scan over the ~10^5 segments in one 1024-thread block waited milliseconds for a CU under load)
__global__ __launch_bounds__(256) void k_extra_scan(Mm355ExtraOut *out, int n)
{
	__shared__ long long part[256];
	const int t = threadIdx.x, per = (n + 255) / 256, lo = t * per < n? t * per : n, hi = lo + per < n? lo + per : n;
	long long sum = 0;
	for (int i = lo; i < hi; ++i) sum += out[i].cs_len + out[i].md_len;
	part[t] = sum;
	__syncthreads();
	if (t == 0) { long long acc = 0; for (int i = 0; i < 256; ++i) { const long long v = part[i]; part[i] = acc; acc += v; } }
	__syncthreads();
	long long acc = part[t];
	for (int i = lo; i < hi; ++i) { out[i].cs_dense = acc; acc += out[i].cs_len + out[i].md_len; }
}

// one lane per region: its segments in order (U:align.c::mm_update_extra's `s` and `max`, `dp_max = (int32_t)(max + .499)`; the numbers that
// stand between the segments' string pieces)
__global__ __launch_bounds__(256) void k_extra_compose(Mm355ExtraSegOut *seg, const int64_t *seg_first, int n_regions, Mm355ExtraOut *out, int want)
{
	const int k = blockIdx.x * 256 + threadIdx.x;
	if (k >= n_regions) return;
	double s = 0.0, mx = 0.0;
	Mm355ExtraOut r; r.mlen = r.blen = r.n_ambi = 0; r.cs_len = r.md_len = 0; r.cs_dense = 0; r.md_end_num = -1; r.pad = 0;
	unsigned cs_carry = 0, md_carry = 0;
	const int64_t g0 = seg_first[k], g1 = seg_first[k + 1];
	for (int64_t g = g0; g < g1; ++g) {
		const Mm355ExtraSegOut e = seg[g];
		r.mlen += e.mlen; r.blen += e.blen; r.n_ambi += e.n_ambi;
		int32_t cs_num = -1, md_num = -1;
		if (want & 1) {
			if (e.flushed & 1) { const unsigned nn = cs_carry + (unsigned)e.cs_lead; if (nn) cs_num = (int32_t)nn; cs_carry = (unsigned)e.cs_tail; }
			else cs_carry += (unsigned)e.cs_lead;
			seg[g].cs_dense = r.cs_len;   // offset of [number][body] inside the region's cs string
			r.cs_len += (cs_num >= 0? 1 + ex_digits((unsigned)cs_num) : 0) + e.cs_len;
		}
		if (want & 2) {
			if (e.flushed & 2) { md_num = (int32_t)(md_carry + (unsigned)e.md_lead); md_carry = (unsigned)e.md_tail; }
			else md_carry += (unsigned)e.md_lead;
			seg[g].md_dense = r.md_len;
			r.md_len += (md_num >= 0? ex_digits((unsigned)md_num) : 0) + e.md_len;
		}
		seg[g].cs_num = cs_num; seg[g].md_num = md_num;
		if (e.C > -1e299) {   // the segment had score steps
			const double a = s + e.C, hi = a > e.P? a : e.P;
			mx = hi > mx? hi : mx;
			const double nm = -e.m;
			s = e.A + (s > nm? s : nm);
		}
	}
	if ((want & 2) && md_carry > 0) { r.md_end_num = (int32_t)md_carry; r.md_len += ex_digits(md_carry); }   // write_MD_core: the run still open at the end
	r.dp_max = (int32_t)(mx + .499);
	out[k] = r;
}

__device__ inline int ex_put_num(char *dst, char lead, unsigned v)
{
	char bf[12]; int nb = 0, n = 0;
	do { bf[nb++] = (char)('0' + v % 10); v /= 10; } while (v);
	if (lead) dst[n++] = lead;
	while (nb > 0) dst[n++] = bf[--nb];
	return n;
}
__global__ __launch_bounds__(256) void k_extra_compact(const Mm355ExtraJob *segs, const Mm355ExtraSegOut *out, const Mm355ExtraOut *reg, const int64_t *seg_first, int n, const char *cs, char *dense, int want)
{
	const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // one wave per segment
	if (k >= n) return;
	const Mm355ExtraJob jb = segs[k];
	const Mm355ExtraSegOut so = out[k];
	const Mm355ExtraOut ro = reg[jb.region];
	if (want & 1) {
		char *dst = dense + ro.cs_dense + so.cs_dense;
		int nn = 0;
		if (so.cs_num >= 0) { nn = 1 + ex_digits((unsigned)so.cs_num); if (lane == 0) ex_put_num(dst + so.cs_pre, ':', (unsigned)so.cs_num); }   // behind the text of leading gaps
		const char *src = cs + jb.cs_off;
		for (int i = lane; i < so.cs_len; i += 64) dst[i < so.cs_pre? i : nn + i] = src[i];
	}
	if (want & 2) {
		char *dst = dense + ro.cs_dense + ro.cs_len + so.md_dense;
		int nn = 0;
		if (so.md_num >= 0) { nn = ex_digits((unsigned)so.md_num); if (lane == 0) ex_put_num(dst, 0, (unsigned)so.md_num); }
		const char *src = cs + jb.md_off;
		for (int i = lane; i < so.md_len; i += 64) dst[nn + i] = src[i];
		if (ro.md_end_num >= 0 && (int64_t)k == seg_first[jb.region + 1] - 1 && lane == 0) ex_put_num(dst + nn + so.md_len, 0, (unsigned)ro.md_end_num);
	}
}

int mm355_extra_run(mm355_ctx *c, const mm355_mapopt_t *mo, const Mm355ExtraJob *segs, size_t n_segs, const int64_t *seg_first, size_t n_regions,
                    const uint32_t *cig, size_t n_cig, size_t cs_cap, int want, const Mm355ExtraOut **out, const char **cs)
{
	*out = 0; *cs = 0;
	if (n_regions == 0) return 0;
	const size_t seg_b = (n_segs * sizeof(Mm355ExtraJob) + 63) & ~(size_t)63, first_b = (n_regions + 1) * 8;
	if (c->x_jobs.ensure(seg_b + first_b + 64) || c->x_cig.ensure((n_cig + 16) * 4) || c->x_cs.ensure(cs_cap + 64) ||
	    c->x_out.ensure((n_segs + 1) * sizeof(Mm355ExtraSegOut) + n_regions * sizeof(Mm355ExtraOut)) || c->h_xout.ensure(n_regions * sizeof(Mm355ExtraOut) + 64)) return MM355_ENOMEM;
	Mm355ExtraScore sc;
	{
		int a = mo->a < 0? -mo->a : mo->a, b = mo->b > 0? -mo->b : mo->b, amb = mo->sc_ambi > 0? -mo->sc_ambi : mo->sc_ambi;   // U:ksw2.h::ksw_gen_simple_mat
		for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) sc.mat[i * 5 + j] = (int8_t)(i == j? a : b); sc.mat[i * 5 + 4] = (int8_t)amb; }
		for (int j = 0; j < 5; ++j) sc.mat[4 * 5 + j] = (int8_t)amb;
		sc.q = (int8_t)mo->q; sc.e = (int8_t)mo->e;
	}
	Mm355ExtraJob *d_segs = c->x_jobs.as<Mm355ExtraJob>();
	const double tx0 = mm355_now_ms();
	int64_t *d_first = (int64_t*)((char*)c->x_jobs.p + seg_b);
	Mm355ExtraSegOut *d_so = c->x_out.as<Mm355ExtraSegOut>();
	Mm355ExtraOut *d_ro = (Mm355ExtraOut*)(d_so + n_segs + 1);
	if (n_segs) HIPCHK(hipMemcpyAsync(d_segs, segs, n_segs * sizeof(Mm355ExtraJob), hipMemcpyHostToDevice, c->st));
	HIPCHK(hipMemcpyAsync(d_first, seg_first, first_b, hipMemcpyHostToDevice, c->st));
	if (n_cig) HIPCHK(hipMemcpyAsync(c->x_cig.p, cig, n_cig * 4, hipMemcpyHostToDevice, c->st));
	mm355_kt(c, KT_EXTRA, 0, c->st);
	if (n_segs) {
		hipLaunchKernelGGL(k_extra, dim3((unsigned)((n_segs + WAVE - 1) / WAVE)), dim3(WAVE), 0, c->st, c->dix, c->rq.as<uint8_t>(), d_segs, (int)n_segs,
		                   c->x_cig.as<uint32_t>(), sc, c->x_cs.as<char>(), d_so, want);
	}
	hipLaunchKernelGGL(k_extra_compose, dim3((unsigned)((n_regions + 255) / 256)), dim3(256), 0, c->st, d_so, d_first, (int)n_regions, d_ro, want);
	if (want) hipLaunchKernelGGL(k_extra_scan, dim3(1), dim3(256), 0, c->st, d_ro, (int)n_regions);
	mm355_kt(c, KT_EXTRA, 1, c->st);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(c->h_xout.p, d_ro, n_regions * sizeof(Mm355ExtraOut), hipMemcpyDeviceToHost, c->st));
	HIPCHK(mm355_wait_stream(c->st));
	const double tx1 = mm355_now_ms();
	mm355_trace_add(c, "x:walk", tx0, tx1);
	const Mm355ExtraOut *ho = (const Mm355ExtraOut*)c->h_xout.p;
	if (want) {
		size_t tot = 0;
		for (size_t k = 0; k < n_regions; ++k) tot += (size_t)ho[k].cs_len + (size_t)ho[k].md_len;
		if (c->x_dense.ensure(tot + 64) || c->h_xcs.ensure(tot + 64)) return MM355_ENOMEM;
		if (n_segs) hipLaunchKernelGGL(k_extra_compact, dim3((unsigned)((n_segs + 3) / 4)), dim3(256), 0, c->st, d_segs, d_so, d_ro, d_first, (int)n_segs, c->x_cs.as<char>(), c->x_dense.as<char>(), want);
		HIPCHK(hipGetLastError());
		if (tot) HIPCHK(hipMemcpyAsync(c->h_xcs.p, c->x_dense.p, tot, hipMemcpyDeviceToHost, c->st));
		HIPCHK(mm355_wait_stream(c->st));
		mm355_trace_add(c, "x:strings", tx1, mm355_now_ms());
		*cs = (const char*)c->h_xcs.p;
	}
	*out = ho;
	return 0;
}

// gather descriptors g[0..n) live in pinned host memory of the context until the next call
int mm355_dp_gather(mm355_ctx *c, const DpGather *g, size_t n, size_t q_tot, size_t t_tot)
{
	if (n == 0) return 0;
	if (c->dp_q.ensure(q_tot + 64) || c->dp_t.ensure(t_tot + 64) || c->dp_gather.ensure(n * sizeof(DpGather))) return MM355_ENOMEM;
	HIPCHK(hipMemcpyAsync(c->dp_gather.p, g, n * sizeof(DpGather), hipMemcpyHostToDevice, c->st));
	KtScope ks(c, KT_DP_GATHER, c->st);
	hipLaunchKernelGGL(k_dp_gather, dim3((unsigned)n), dim3(256), 0, c->st, c->dix, c->dp_gather.as<DpGather>(), (int)n,
	                   c->rq.as<uint8_t>(), c->dp_q.as<uint8_t>(), c->dp_t.as<uint8_t>());
	HIPCHK(hipGetLastError());
	return 0;
}

int mm355_run_read_codes(mm355_ctx *c)
{
	if (c->hb.n_reads == 0) return 0;
	if (c->rq.ensure((size_t)c->hb.n_bytes * 2 + 64)) return MM355_ENOMEM;
	DevBatch b; b.n_reads = (int32_t)c->hb.n_reads; b.seq = c->seq.as<uint8_t>(); b.roff = c->roff.as<int64_t>(); b.rlen = c->rlen.as<int32_t>(); b.order = c->order.as<int32_t>(); b.prof = 0;
	KtScope ks(c, KT_CODES, c->st);
	hipLaunchKernelGGL(k_read_codes, dim3((unsigned)c->hb.n_reads), dim3(256), 0, c->st, b, c->rq.as<uint8_t>());
	HIPCHK(hipGetLastError());
	return 0;
}

// C-ABI: one batch of extension problems on caller-provided code strings (parity tests, kernel bench)
extern "C" int mm355_stage_dp(mm355_ctx_t *c, const mm355_mapopt_t *mo, int64_t n_jobs, const mm355_dpjob_t *jobs,
                              const uint8_t *qcodes, int64_t n_q, const uint8_t *tcodes, int64_t n_t,
                              mm355_dpres_t *res, uint32_t *cigar, int64_t cigar_cap)
{
	if (c == 0 || mo == 0) return MM355_EINVAL;
	HIPCHK(hipSetDevice(c->dev));
	c->n_tpend = 0; memset(&c->stats, 0, sizeof(c->stats));
	HIPCHK(hipMemsetAsync(c->counters.p, 0, CTR_BYTES, c->st));
	if (c->dp_q.ensure((size_t)n_q + 64) || c->dp_t.ensure((size_t)n_t + 64)) return MM355_ENOMEM;
	if (n_q) HIPCHK(hipMemcpyAsync(c->dp_q.p, qcodes, n_q, hipMemcpyHostToDevice, c->st));
	if (n_t) HIPCHK(hipMemcpyAsync(c->dp_t.p, tcodes, n_t, hipMemcpyHostToDevice, c->st));
	std::vector<DpJobDev> dj(n_jobs);
	for (int64_t i = 0; i < n_jobs; ++i) {
		memset(&dj[i], 0, sizeof(DpJobDev));
		dj[i].qlen = jobs[i].qlen; dj[i].tlen = jobs[i].tlen; dj[i].qoff = jobs[i].qoff; dj[i].toff = jobs[i].toff;
		dj[i].w = jobs[i].w; dj[i].zdrop = jobs[i].zdrop; dj[i].end_bonus = jobs[i].end_bonus; dj[i].flag = jobs[i].flag;
	}
	const mm355_dpres_t *r = 0; const uint32_t *cg = 0;
	int rc = mm355_dp_run(c, mo, dj.data(), dj.size(), c->dp_q.as<uint8_t>(), c->dp_t.as<uint8_t>(), &c->h_cig, &r, &cg);
	if (rc) return rc;
	int64_t tot = 0;
	for (int64_t i = 0; i < n_jobs; ++i) {
		res[i] = r[i];
		if (tot + r[i].n_cigar > cigar_cap) return MM355_ENOMEM;
		if (r[i].n_cigar) memcpy(cigar + tot, cg + r[i].cigar_off, (size_t)r[i].n_cigar * 4);
		res[i].cigar_off = tot; tot += r[i].n_cigar;
	}
	return 0;
}

// C-ABI: the device form of mm_update_extra's walk + cs / MD on caller-provided regions (parity tests).  want_cs: bit 0 cs, bit 1 MD
extern "C" int mm355_stage_extra(mm355_ctx_t *c, const mm355_mapopt_t *mo, int64_t n_regions, const mm355_extrajob_t *jobs,
                                 const uint8_t *qcodes, int64_t n_q, const uint32_t *cigar, int64_t n_cigar, int want_cs,
                                 mm355_extrares_t *res, char *cs, int64_t cs_cap)
{
	if (c == 0 || mo == 0 || n_regions < 0) return MM355_EINVAL;
	if (n_regions == 0) return 0;
	HIPCHK(hipSetDevice(c->dev));
	if (c->rq.ensure((size_t)n_q + 64)) return MM355_ENOMEM;
	if (n_q) HIPCHK(hipMemcpyAsync(c->rq.p, qcodes, (size_t)n_q, hipMemcpyHostToDevice, c->st));
	size_t n_segs = 0;
	for (int64_t k = 0; k < n_regions; ++k) {
		if (jobs[k].n_cigar < 0 || jobs[k].cigar_off < 0 || jobs[k].cigar_off + jobs[k].n_cigar > n_cigar || (uint32_t)jobs[k].rid >= c->mi->n_seq) return MM355_EINVAL;
		// '=' / 'X' CIGARs never reach the device walk in the mapping path (MM_F_EQX keeps the host walk, which rewrites the CIGAR): U:align.c::mm_update_extra
		// counts match columns on op 0 only, so a stage call with ops 7 / 8 would disagree with its own oracle -- refused
		for (int32_t x = 0; x < jobs[k].n_cigar; ++x) { const uint32_t op = cigar[jobs[k].cigar_off + x] & 0xf; if (op == 7 || op == 8) return MM355_EINVAL; }
		n_segs += (size_t)mm355_extra_split(cigar + jobs[k].cigar_off, jobs[k].n_cigar, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
	}
	const size_t seg_b = (n_segs * sizeof(Mm355ExtraJob) + 63) & ~(size_t)63;
	if (c->h_xjobs.ensure(seg_b + ((size_t)n_regions + 1) * 8 + 64) || c->h_xcig.ensure(((size_t)n_cigar + 16) * 4)) return MM355_ENOMEM;
	Mm355ExtraJob *segs = (Mm355ExtraJob*)c->h_xjobs.p; int64_t *first = (int64_t*)((char*)c->h_xjobs.p + seg_b);
	if (n_cigar) memcpy(c->h_xcig.p, cigar, (size_t)n_cigar * 4);
	size_t g = 0; int64_t slot = 0;
	for (int64_t k = 0; k < n_regions; ++k) {
		first[k] = (int64_t)g;
		int64_t cc = 0, mc = 0;
		mm355_extra_split(cigar + jobs[k].cigar_off, jobs[k].n_cigar, 0, 0, 0, 0, 0, 0, 0, 0, &cc, &mc);
		g += (size_t)mm355_extra_split(cigar + jobs[k].cigar_off, jobs[k].n_cigar, jobs[k].q_off, (uint32_t)jobs[k].rid, jobs[k].t_st, jobs[k].cigar_off, slot, slot + cc, (int32_t)k, segs + g, 0, 0);
		slot += cc + mc;
	}
	first[n_regions] = (int64_t)g;
	const Mm355ExtraOut *xo = 0; const char *xcs = 0;
	int rc = mm355_extra_run(c, mo, segs, n_segs, first, (size_t)n_regions, (const uint32_t*)c->h_xcig.p, (size_t)n_cigar, (size_t)slot, want_cs & 3, &xo, &xcs);
	if (rc) return rc;
	for (int64_t k = 0; k < n_regions; ++k) {
		mm355_extrares_t o; o.mlen = xo[k].mlen; o.blen = xo[k].blen; o.n_ambi = xo[k].n_ambi; o.dp_max = xo[k].dp_max; o.cs_off = xo[k].cs_dense; o.cs_len = (want_cs & 1)? xo[k].cs_len : 0; o.pad = 0;
		o.md_off = xo[k].cs_dense + xo[k].cs_len; o.md_len = (want_cs & 2)? xo[k].md_len : 0; o.pad2 = 0;
		if (want_cs & 3) { if (o.cs_off + xo[k].cs_len + xo[k].md_len > cs_cap) return MM355_ENOMEM; memcpy(cs + o.cs_off, xcs + o.cs_off, (size_t)xo[k].cs_len + (size_t)xo[k].md_len); }
		res[k] = o;
	}
	return 0;
}
