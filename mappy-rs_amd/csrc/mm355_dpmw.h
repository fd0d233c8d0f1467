// mm355_dpmw.h -- eight waves per alignment for the long narrow-band exact sweeps (row a12; the extensions of a read's ends: band w = 751
// with the ONT preset, thousands of anti-diagonals).  Included by mm355_dp.hip after mm355_dpreg.h; same results, bit for bit, as
// k_ksw_regw / k_ksw_extd2 (U:ksw2_extd2_sse.c::ksw_extd2_sse with exact score tracking and z-drop).
//
// k_ksw_regw gave such an alignment ONE wave with eight 128-cell blocks in registers: ~1000 instructions per anti-diagonal on a wave that,
// alone on its SIMD, issues one packed 16-bit instruction every ~9 cycles (profiles/r03_valubench.txt) -- 10 000 anti-diagonals are 30 ms.
// Here every block has its own wave (same registers, same per-block code: dp_score / dp_core_f / dp_block_h of mm355_dpreg.h):
//   * wave w owns the target blocks b = w (mod 8); the live window is the eight blocks from blow = (st - 1) >> 7 upwards (the band, its
//     score spill and the cell below st fit: DP_WIN_MAX_W), so when the band has left a block its wave starts afresh eight blocks further
//     up -- no state ever moves between waves;
//   * what a block needs from the block below it on the previous anti-diagonal -- x, v, x2 (and H) of that block's last cell pair -- goes
//     through an LDS mailbox, double-buffered by the parity of the anti-diagonal; the query enters every block from memory (64 bases per
//     64 anti-diagonals), as it enters block 0 of the single-wave kernel;
//   * exact score tracking: every wave reduces its own block, the per-block maxima and the two border values (H at st0 and at en0) meet in
//     LDS, ONE workgroup barrier per anti-diagonal (s_barrier behind an lgkmcnt wait only: the direction bytes of the anti-diagonal are still on
//     their way to HBM);
//   * a NINTH wave keeps the books: behind the barrier of anti-diagonal r it takes the maximum of the eight block maxima and replays the scalar
//     z-drop / mqe / mte bookkeeping of r while the eight block waves are already on r + 1.  That replay is a chain of LDS round trips and
//     ~15 scalar branches -- 0.76 of the 1.7 us an anti-diagonal took when every wave did it after its own block -- and nothing in a block's
//     recurrence depends on it, except for the decision to stop: the bookkeeper raises a flag in LDS, the block waves read it with their
//     mailbox and leave one or two anti-diagonals late (the direction bytes they wrote past the end are never read: the backtrack starts at
//     the bookkeeper's maximum).
#pragma once

#define MW_WAVES 8
#define MW_SEQ_MAX 98304             // query + target bytes staged in LDS (padded to 16 each); longer problems keep the other kernels
// The two sequences live in LDS for the whole sweep: a global load inside the loop -- even one behind a branch that is taken once in 64
// anti-diagonals -- makes the compiler wait for vmcnt(0) at the join, i.e. for the direction bytes of the previous anti-diagonal to be
// acknowledged by the L2: a memory round trip on every step of the dependency chain.
extern __shared__ uint8_t mw_seq[];   // [0, qpad): query, [qpad, qpad + tpad): target
struct MwBox { uint32_t X, V, X2; int32_t Hh; };
struct MwLds {
	MwBox box[2][MW_WAVES];          // lane 63 of every wave's block after the anti-diagonal of that parity
	long long best[2][MW_WAVES];     // per-block (H, priority) maximum, priority relative to the window start
	int32_t hen[2], hst0[2];
	int any_n;
	int stop;                        // raised by the bookkeeper: z-drop
};
#define MW_THREADS (64 * (MW_WAVES + 1))

__device__ __forceinline__ int mw_cell8(uint32_t reg, int t_rel) { return (int)(int8_t)(rdlane(reg, (t_rel >> 1) & 63) >> ((t_rel & 1)? 24 : 8)); }
__device__ __forceinline__ int32_t mw_hat(int32_t Hl, int32_t Hh, int t_rel) { return (int32_t)rdlane((uint32_t)((t_rel & 1)? Hh : Hl), (t_rel >> 1) & 63); }

// the ninth wave: anti-diagonal r's maximum and the z-drop / mqe / mte bookkeeping, one barrier behind the block waves
__device__ __forceinline__ void mw_books(DpRun &R, MwLds *L, const int lane)
{
	EzState &ez = R.ez;
	for (;;) {
		const int r = R.r, st = R.st, en = R.en, st0 = R.st0, en0 = R.en0;
		const int blow = st > 0? (st - 1) >> 7 : 0;
		const int par = r & 1;
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the block waves have left anti-diagonal r
		R.cells += (unsigned long long)(en0 - st0 + 1);
		{
			long long bb = lane < MW_WAVES? L->best[par][lane] : INT64_MIN;
			const int32_t bh = (int32_t)(bb >> 32);
			const int32_t mh = (int32_t)rdlane((uint32_t)dp_wave_max_i32(bh), 63);
			const uint32_t bl = bh == mh? (uint32_t)bb : 0u;
			const uint32_t ml = rdlane(dp_wave_max_u32(bl), 63);
			bb = (long long)(((unsigned long long)(uint32_t)mh << 32) | ml);
			const int32_t hen = __builtin_amdgcn_readfirstlane(L->hen[par]), Hst0 = __builtin_amdgcn_readfirstlane(L->hst0[par]);
			int32_t max_H = hen, max_t = en0;
			if (r > 0 && bb != INT64_MIN) {
				const int32_t ch = (int32_t)(bb >> 32);
				if (ch > hen) { max_H = ch; max_t = (int)(~(uint32_t)bb & 0xffffu) + (blow << 7); }
			}
			if (r == 0) max_t = 0;
			const int32_t Hen0 = hen;
			if (en0 == R.tlen - 1 && Hen0 > ez.mte) ez.mte = Hen0, ez.mte_q = r - en;
			if (r - st0 == R.qlen - 1 && Hst0 > ez.mqe) ez.mqe = Hst0, ez.mqe_t = st0;
			if (apply_zdrop(ez, max_H, r, max_t, R.zdrop, R.e2)) {
				if (lane == 0) L->stop = 1;
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				return;
			}
			if (r == R.qlen + R.tlen - 2 && en0 == R.tlen - 1) ez.score = Hen0;
		}
		R.last_st = R.st; R.last_en = R.en;
		if (++R.r >= R.r_total) return;
		if (!dp_bounds(R)) { ez.zdropped = 1; return; }
	}
}

template <bool RIGHT>
__device__ __forceinline__ void mw_sweep(DpRun &R, const DpK &K, MwLds *L, const int lane, const int wv, const int toff)
{
	if (R.r_total <= 0) return;
	if (!dp_bounds(R)) { R.ez.zdropped = 1; return; }                  // (only the bookkeeper's ez is ever read)
	if (wv == MW_WAVES) { mw_books(R, L, lane); return; }
	uint32_t U, V, X, Y, X2, Y2, SC, TQ, QQ; int32_t Hl, Hh;
	int bcur = -1;
	uint32_t qv = 0;
	for (;;) {
		const int r = R.r, st = R.st, en = R.en, st0 = R.st0, en0 = R.en0;
		const int blow = st > 0? (st - 1) >> 7 : 0;
		const int b = blow + ((wv - blow) & (MW_WAVES - 1));
		const int base = b << 7;
		if (b != bcur) {   // a fresh block: what the SSE kernel's arrays hold for positions the band has not reached (the registers stand at anti-diagonal r - 1)
			const int t = base + 2 * lane;
			U = V = X = Y = K.nqe; X2 = Y2 = K.nq2e2; SC = 0; Hl = Hh = KSW_NEG_INF;
			TQ = (t < R.tlen? (uint32_t)mw_seq[toff + t] : 0u) | (t + 1 < R.tlen? (uint32_t)mw_seq[toff + t + 1] : 0u) << 16;
			const int qi = r - 1 - t;
			QQ = (qi >= 0 && qi < R.qlen? (uint32_t)mw_seq[qi] : 0u) | (qi - 1 >= 0 && qi - 1 < R.qlen? (uint32_t)mw_seq[qi - 1] : 0u) << 16;
			bcur = b;
		}
		const int q0 = (r & ~63) - base;                       // the 64 query bases that enter cell 0 of this block during this 64-diagonal period
		if ((r & 63) == 0 || R.base != base) { qv = q0 + lane >= 0 && q0 + lane < R.qlen? mw_seq[q0 + lane] : 0; R.base = base; }
		const int st0_r = st0 - base, en0_r = en0 - base;
		DpDiag g;
		g.r = r; g.st = st - base; g.en = en - base; g.st0 = st0_r; g.any_n = R.any_n;
		g.use_def = st == 0 || !(st - 1 >= R.last_st && st - 1 <= R.last_en);
		const int edge_u = r == 0? -R.q - R.e : r < R.long_thres? -R.e : r == R.long_thres? R.long_diff : -R.e2;
		g.edge_u8 = pk8(edge_u);
		g.dv1 = st > 0? K.nqe & 0xffff0000u : g.edge_u8 & 0xffff0000u;
		g.edge = en >= r;
		int sce = st0 + ((en0 - st0) / 16 + 1) * 16;
		if (sce > R.T) sce = R.T;
		g.sclen = (uint32_t)(sce - st0);
		g.qc_hi = rdlane(qv, r & 63) << 16;
		g.jq = 0; g.lane_st = g.lane_r = -1;
		g.pos_st = g.use_def? st - base : -1; g.pos_r = g.edge && r >= base? (r - base) & ~1 : -1;   // (r below this block: no top-row cell here)
		{
			const uint64_t rp = (uint64_t)(R.p + ((size_t)r * R.n_col - (size_t)st)) + (uint64_t)(int64_t)base;
			g.rowp = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rp) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rp >> 32)) << 32;
		}
		const int par = r & 1;
		const MwBox nb = L->box[par ^ 1][(wv - 1) & (MW_WAVES - 1)];   // the block below, after the previous anti-diagonal
		if (*(volatile int*)&L->stop) return;                          // (one LDS round trip with the mailbox)
		// 1. the query moves one cell to the right
		dp_slide(QQ, g.qc_hi);
		// 2. scores of [st0, sce)
		if (base < sce && base + 127 >= st0) dp_score<0>(g, K, lane, SC, TQ, QQ);
		// 3. the recurrence on [st, en]
		const bool touches = base <= en && base + 127 >= st;
		if (touches) dp_core_f<0, true, true, RIGHT, true, true>(g, K, lane, R.p, U, V, X, Y, X2, Y2, SC, nb.X, nb.V, nb.X2);
		// 4. exact score tracking of this block
		long long best = INT64_MIN;
		if (r > 0) {
			int32_t hen = 0;
			const bool own_en = (en0 >> 7) == b;
			if (own_en) {
				if (en0 > 0) {
					const int32_t hp = en0_r > 0? mw_hat(Hl, Hh, en0_r - 1) : (int32_t)__builtin_amdgcn_readfirstlane(nb.Hh);
					hen = hp + mw_cell8(U, en0_r);
				} else hen = mw_hat(Hl, Hh, en0_r) + mw_cell8(V, en0_r);
			}
			if (base <= en0 && base + 127 >= st0) {
				const int en1_r = st0_r + (en0_r - st0_r) / 4 * 4;
				dp_block_h<0>(lane, st0_r, en0_r, en1_r, hen, V, Hl, Hh, best);
				const int32_t bh = (int32_t)(best >> 32);
				const int32_t mh = (int32_t)rdlane((uint32_t)dp_wave_max_i32(bh), 63);
				const uint32_t bl = bh == mh? (uint32_t)best : 0u;
				const uint32_t ml = rdlane(dp_wave_max_u32(bl), 63);
				best = (long long)(((unsigned long long)(uint32_t)mh << 32) | ml);
				if (best != INT64_MIN) {   // the position inside the priority word becomes relative to the window start (comparable between waves)
					const uint32_t prio = ~(uint32_t)best;
					const uint32_t p2 = (prio & 0xffff0000u) | ((prio & 0xffffu) + (uint32_t)((b - blow) << 7));
					best = (long long)(((unsigned long long)(uint32_t)mh << 32) | (uint32_t)~p2);
				}
			}
			if (lane == 0) {
				L->best[par][wv] = best;
				if (own_en) L->hen[par] = hen;
			}
			if ((st0 >> 7) == b) {
				const int32_t hs = st0 == en0? hen : mw_hat(Hl, Hh, st0_r);
				if (lane == 0) L->hst0[par] = hs;
			}
		} else {
			if (b == 0) {
				const int32_t h0 = mw_cell8(V, 0) - R.qe;
				Hl = lane == 0? h0 : Hl;
				if (lane == 0) { L->hen[par] = h0; L->hst0[par] = h0; }
			}
			if (lane == 0) L->best[par][wv] = INT64_MIN;
		}
		if (lane == 63) { MwBox o; o.X = X; o.V = V; o.X2 = X2; o.Hh = Hh; L->box[par][wv] = o; }
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
		R.last_st = R.st; R.last_en = R.en;
		if (++R.r >= R.r_total) return;
		if (!dp_bounds(R)) return;
	}
}

__global__ __launch_bounds__(MW_THREADS) void k_ksw_regw8(DpConst dc, const DpJobDev *jobs, const int32_t *job_ids, int n_jobs,
                                                            const uint8_t *qbase, const uint8_t *tbase, uint8_t *pbase, mm355_dpres_t *res, unsigned long long *cells_ctr)
{
	__shared__ MwLds L;
	const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform, and the compiler must know it:
	if ((int)blockIdx.x >= n_jobs) return;                                                            //  the block index and all that follows from it stay scalar)
	__builtin_amdgcn_s_setprio(3);
	const int jid = job_ids[blockIdx.x];
	const DpJobDev jb = jobs[jid];
	DpRun R;
	R.qlen = jb.qlen; R.tlen = jb.tlen; R.flag = jb.flag; R.zdrop = jb.zdrop; R.end_bonus = jb.end_bonus;
	EzState &ez = R.ez;
	ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1;
	ez.max = 0; ez.score = ez.mqe = ez.mte = KSW_NEG_INF; ez.zdropped = 0; ez.reach_end = 0;
	if (R.qlen <= 0 || R.tlen <= 0 || jb.skip) {   // skip: tlen*qlen > max_sw_mat => treated as z-dropped by mm_align_pair
		if (threadIdx.x == 0) {
			mm355_dpres_t o; o.max = 0; o.zdropped = jb.skip? 1 : 0; o.max_q = o.max_t = o.mqe_t = o.mte_q = -1;
			o.mqe = o.mte = o.score = KSW_NEG_INF; o.reach_end = 0; o.n_cigar = -1; o.cigar_off = -1;
			res[jid] = o;
		}
		return;
	}
	const uint8_t *target = tbase + jb.toff;
	R.query = qbase + jb.qoff; R.target = target; R.base = -1;
	R.q = dc.q; R.e = dc.e; R.q2 = dc.q2; R.e2 = dc.e2; R.qe = dc.qe_preswap; R.long_thres = dc.long_thres; R.long_diff = dc.long_diff;
	const int qlen = R.qlen, tlen = R.tlen;
	R.w = jb.w < 0? (tlen > qlen? tlen : qlen) : jb.w;
	R.T = (tlen + 15) / 16 * 16;
	int n_col_ = qlen < tlen? qlen : tlen;
	n_col_ = ((n_col_ < R.w + 1? n_col_ : R.w + 1) + 15) / 16 + 1;
	R.n_col = n_col_ * 16;
	R.p = pbase + jb.p_off;
	R.r_total = qlen + tlen - 1;
	R.full = false;
	R.r = 0; R.last_st = R.last_en = -1; R.H0 = 0; R.last_H0_t = 0; R.cells = 0; R.qv = 0;
	DpK K;
	K.nqe = pk8(-R.q - R.e); K.nq2e2 = pk8(-R.q2 - R.e2); K.q = pk8(R.q); K.q2 = pk8(R.q2); K.qe = pk8(R.q + R.e); K.q2e2 = pk8(R.q2 + R.e2);
	K.mch = pk8(dc.sc_mch); K.dmis_v = vreg_const(pk8(dc.sc_mis - dc.sc_mch)); K.N = pk8(dc.sc_N); K.one = 0x00010001u; K.c256 = 0x01000100u; K.m256 = 0xff00ff00u;
	K.two = 0x00020002u; K.three = 0x00030003u; K.four = 0x00040004u; K.f8 = 0x00080008u; K.f16 = 0x00100010u; K.f32 = 0x00200020u; K.f64 = 0x00400040u;
	K.dx1 = K.nqe & 0xffff0000u; K.dx21 = K.nq2e2 & 0xffff0000u;
	const int toff = (qlen + 15) & ~15;
	{   // the sequences move into LDS; ambiguous bases anywhere?
		bool n = false;
		for (int i = threadIdx.x; i < tlen; i += MW_THREADS) { const uint8_t c = target[i]; mw_seq[toff + i] = c; n |= c > 3; }
		for (int i = threadIdx.x; i < qlen; i += MW_THREADS) { const uint8_t c = R.query[i]; mw_seq[i] = c; n |= c > 3; }
		R.any_n = __syncthreads_or(n) != 0;
	}
	if (threadIdx.x == 0) L.stop = 0;
	if (threadIdx.x < 2 * MW_WAVES) { MwBox z; z.X = z.V = K.nqe; z.X2 = K.nq2e2; z.Hh = KSW_NEG_INF; L.box[threadIdx.x >> 3][threadIdx.x & 7] = z; }
	__syncthreads();
	if (R.flag & EZ_RIGHT) mw_sweep<true>(R, K, &L, lane, wv, toff);
	else mw_sweep<false>(R, K, &L, lane, wv, toff);
	if (threadIdx.x == 64 * MW_WAVES) {   // the bookkeeper holds ez
		const int flag = R.flag;
		int i0 = -1, j0 = -1;
		if (!ez.zdropped && !(flag & EZ_EXTZ_ONLY)) { i0 = tlen - 1; j0 = qlen - 1; }
		else if (!ez.zdropped && (flag & EZ_EXTZ_ONLY) && ez.mqe + R.end_bonus > ez.max) { ez.reach_end = 1; i0 = ez.mqe_t; j0 = qlen - 1; }
		else if (ez.max_t >= 0 && ez.max_q >= 0) { i0 = ez.max_t; j0 = ez.max_q; }
		mm355_dpres_t o;
		o.max = ez.max; o.zdropped = ez.zdropped; o.max_q = ez.max_q; o.max_t = ez.max_t; o.mqe = ez.mqe; o.mqe_t = ez.mqe_t;
		o.mte = ez.mte; o.mte_q = ez.mte_q; o.score = ez.score; o.reach_end = ez.reach_end;
		o.n_cigar = i0; o.cigar_off = j0;   // start cell for k_ksw_backtrack
		res[jid] = o;
		if (R.cells) atomicAdd(cells_ctr + (blockIdx.x & (DP_CTR_SPREAD - 1)), R.cells);
	}
}
