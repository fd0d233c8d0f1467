// mm355_dprow.h -- row-sweep form of the banded extension kernel for the problems that dominate a batch: gap fills
// (KSW_EZ_APPROX_MAX without KSW_EZ_APPROX_DROP) whose band never binds (w >= qlen + tlen), targets up to 1024 bases.
// Included by mm355_dp.hip; same results, bit for bit, as k_ksw_reg / U:ksw2_extd2_sse.c::ksw_extd2_sse for these problems.
//
// Why another form.  The anti-diagonal sweep of k_ksw_reg pays one 128-cell block evaluation per anti-diagonal and block it touches,
// whatever the number of cells of the matrix on it: on a 212 x 213 fill (the typical one) 55 % of the evaluated lanes are inside the
// matrix, and every block needs its neighbours' boundary cells, the band masks and the systolic query.  With a band that never binds
// none of the SSE kernel's order-dependent behaviour is left (no stale band-edge cells can feed a cell of the matrix, no z-drop on the
// approximate score): every cell holds the true H / E / F / E2 / F2 of the two-piece affine recurrence, and the direction byte of a cell
// is a function of those values alone --
//     d & 7  = position of the first (KSW_EZ_RIGHT: last) maximum among (H(t-1,q-1) + s, E, F, E2, F2),
//     0x08   = E  - H + q  > 0 (RIGHT: >= 0),   0x10 = F  - H + q  > 0,   0x20 = E2 - H + q2 > 0,   0x40 = F2 - H + q2 > 0
// (the difference recurrences of the SSE kernel are these comparisons shifted by H(t-1,q-1); int8 never wraps on true cells).
// So the matrix can be filled in any order.  Here: one wave per alignment, lane l of register set k owns target cells 128 k + 2 l, + 1
// (two int16 halves per VGPR), and the sweep goes ROW by row of the query -- every lane of the target is useful on every step:
//     F, F2 (gaps that consume query)  come from the row above in the same lane;
//     M needs H of the row above one cell to the left: one DPP wave_shr + v_alignbit per set;
//     E, E2 (gaps that consume target) run ALONG the row: E(t) = max_{k<t} (G(k) + k e) - q - e - (t - 1) e with G = max(M, F, F2)
//     (opening a gap from an H that is itself an E never beats extending that E), i.e. an exclusive prefix maximum over the lanes:
//     six v_max_i32 DPP steps per set and gap type, the carry between sets through one v_readlane.
// About 74 VALU per 128 cells of a row, all of them cells of the matrix: ~0.6 VALU per cell against ~1.6 for the anti-diagonal form.
// The direction bytes are written in TILES of 4 query rows x 16 target cells (64 B, one HBM fetch granule): a lane keeps the bytes of its
// two cells over four rows in two registers and stores them as ONE 8-byte word per set (512 B per wave and set, every line written whole,
// once) -- byte (q, t) of a job's matrix is at row_cell_off(q, t, tstride) = (q >> 2) * 4 tstride + (t >> 1) * 8 + (q & 3) * 2 + (t & 1).
// k_ksw_backtrack's walk moves at most one row and one column per step: along a diagonal it enters a new tile every 3.2 steps (1/4 + 1/16
// per step) instead of a new row -- a new line -- on every step of a row-major matrix (PMC: 53 B fetched per CIGAR column before).  The job
// descriptor names the layout (DpJobDev::pad = 1) and k_ksw_backtrack reads it that way.  ez: only `score` is defined for these problems
// (max = 0, max_t = max_q = -1, not z-dropped), as in the approximate full-band path of k_ksw_reg.
#pragma once

#define ROW_NEG (-16384)
#define ROW_TILE_ROWS 4
// bytes of the tiled direction matrix of a qlen x tlen problem (T = tlen rounded up to 16; the matrix starts on a 64-byte boundary)
__host__ __device__ __forceinline__ size_t row_matrix_bytes(int qlen, int T) { return (size_t)((qlen + ROW_TILE_ROWS - 1) & ~(ROW_TILE_ROWS - 1)) * ((size_t)T + 16) + 64; }
__host__ __device__ __forceinline__ size_t row_cell_off(int q, int t, int tstride) { return (size_t)(q >> 2) * (size_t)(4 * tstride) + (size_t)(t >> 1) * 8 + (size_t)((q & 3) * 2 + (t & 1)); }
#define ROW_MAX_T 1024              // eight register sets
#define ROW_MAX_QT 6000             // qlen + tlen: keeps every value inside int16

__device__ __forceinline__ uint32_t pk8w(int v) { const uint32_t h = (uint32_t)(uint16_t)(int16_t)v; return h | h << 16; }   // plain int16 in both halves
__device__ __forceinline__ uint32_t pk_sign16(uint32_t a) { uint32_t r; asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a)); return r; }   // 0xffff where the half is negative
__device__ __forceinline__ uint32_t pk_rsubsat_s(uint32_t c, uint32_t a) { uint32_t r; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "s"(c), "v"(a)); return r; }   // max(c - a, 0), unsigned
__device__ __forceinline__ uint32_t pk_mad_vsv(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c)); return r; }
__device__ __forceinline__ int32_t wave_incl_scan_max32(int32_t x)   // inclusive prefix maximum over lanes 0..lane (DPP row shifts + row broadcasts)
{
	int32_t y;
	y = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x111, 0xf, 0xf, false); x = x > y? x : y;
	y = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x112, 0xf, 0xf, false); x = x > y? x : y;
	y = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x114, 0xf, 0xf, false); x = x > y? x : y;
	y = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x118, 0xf, 0xf, false); x = x > y? x : y;
	y = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x142, 0xa, 0xf, false); x = x > y? x : y;
	y = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x143, 0xc, 0xf, false); x = x > y? x : y;
	return x;
}

struct RowK {                       // wave-uniform constants (SGPRs)
	uint32_t qe1, e1, qe2, e2, q1, q2, mch, N, one, two, three, four, f8, f16, f32, f64;
	uint32_t q1p, q2p;              // q + 1, q2 + 1 (the ">= 0" continuation test of KSW_EZ_RIGHT)
	int32_t q1i, q2i;
};

// boundary H(t, -1) = H(-1, t): the sum of the first t + 1 boundary differences of U:ksw2_extd2_sse.c (-q-e, then -e while the first
// gap piece is the cheaper one, long_diff at the crossover, -e2 after it)
__device__ __forceinline__ int row_hb(int t, const DpConst &dc)
{
	if (t < 0) return 0;
	const int lt = dc.long_thres;
	int n1 = lt - 1 < t? lt - 1 : t; if (n1 < 0) n1 = 0;
	const int has = (lt >= 1 && lt <= t)? 1 : 0;
	const int n2 = t - n1 - has;
	return -(dc.q + dc.e) - n1 * dc.e + has * dc.long_diff - n2 * dc.e2;
}

__device__ __forceinline__ uint32_t pk2(int lo, int hi) { return (uint32_t)(uint16_t)(int16_t)lo | (uint32_t)(uint16_t)(int16_t)hi << 16; }
__device__ __forceinline__ uint32_t pk_max_swap(uint32_t a) { uint32_t r; asm("v_pk_max_i16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a)); return r; }   // both halves = max(lo, hi)

// exclusive prefix maximum of A over the cells of one set, with carry-in C (includes all earlier sets); returns the packed prefix (lo: the
// lane's first cell, hi: its second, which also sees the first) and the carry for the next set in C.  The scan runs on "doubled" words --
// the 16-bit value in BOTH halves, which as an int32 orders like the value itself -- so that neither the lane maximum has to be sign-extended
// nor the result re-packed: v_pk_max (swap), six v_max_i32 dpp, the shifted carry, one v_pk_max against {-inf, A.lo}.  C is doubled too (row_dbl).
// (The caller subtracts q + t e in ONE packed operation: the constant is folded into its per-set register.)
#define ROW_PKNEG 0x80008000u
__device__ __forceinline__ int32_t row_dbl(int v) { return (int32_t)((uint32_t)(v & 0xffff) * 0x10001u); }
__device__ __forceinline__ uint32_t row_scan(const uint32_t A, int32_t &C)
{
	const int32_t incl = wave_incl_scan_max32((int32_t)pk_max_swap(A));
	int32_t ex = __builtin_amdgcn_update_dpp(C, incl, 0x138, 0xf, 0xf, false);   // wave_shr:1, lane 0 <- carry
	ex = ex > C? ex : C;
	const int32_t tot = __builtin_amdgcn_readlane(incl, 63);
	C = C > tot? C : tot;
	return pk_max((uint32_t)ex, __builtin_amdgcn_perm(A, ROW_PKNEG, 0x05040100));   // {ex, max(ex, A.lo)}
}

// one register set (128 target cells) of one row: in: the row above (Hp, Fp, F2p), the cell to the left of the set's first cell in the row
// above (hi half of carry_h), the prefix maxima of everything to the left in this row (C1, C2); out: this row's H / F / F2 in their place,
// carries for the next set, the direction bytes of the two cells of this lane in accw (rows 0 / 2 of a tile: the low half, the high half
// cleared; rows 1 / 3 -- `odd`, a constant after unrolling -- the high half)
template <bool RIGHT>
__device__ __forceinline__ void row_set(const RowK &K, const uint32_t dmis, const bool any_n, const uint32_t qc2, uint32_t &Hp, uint32_t &Fp, uint32_t &F2p,
                                        const uint32_t TQ, const uint32_t KE1, const uint32_t KE2, const uint32_t KQ1, const uint32_t KQ2, uint32_t &carry_h, int32_t &C1, int32_t &C2, const bool odd, uint32_t &accw)
{
	// H(t-1, q-1): the row above, shifted one cell to the right
	const uint32_t sh = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_h, (int)Hp, 0x138, 0xf, 0xf, false);
	const uint32_t Hd = __builtin_amdgcn_alignbit(Hp, sh, 16);
	carry_h = (uint32_t)__builtin_amdgcn_readlane((int)Hp, 63);
	// substitution score
	uint32_t s = pk_mad_vvs(pk_minu_s(TQ ^ qc2, K.one), dmis, K.mch);
	if (any_n) s = pk_mad(pk_shr2(TQ | qc2), pk_rsub_s(K.N, s), s);   // either base ambiguous (code 4): sc_N
	const uint32_t M = pk_add(Hd, s);
	const uint32_t F = pk_max(pk_sub_s(Hp, K.qe1), pk_sub_s(Fp, K.e1));
	const uint32_t F2 = pk_max(pk_sub_s(Hp, K.qe2), pk_sub_s(F2p, K.e2));
	const uint32_t G = pk_max(pk_max(M, F), F2);
	// E(t) = max_{k<t} (G(k) + k e) - (q + e) - (t - 1) e = [prefix - q] - t e
	const uint32_t E = pk_sub(row_scan(pk_add(G, KE1), C1), KQ1);        // KQ = KE + q: [prefix - q] - t e in one subtraction
	const uint32_t E2 = pk_sub(row_scan(pk_add(G, KE2), C2), KQ2);
	const uint32_t H = pk_max(pk_max(G, E), E2);
	// direction byte
	const uint32_t xE = pk_sub(H, E), xF = pk_sub(H, F), xE2 = pk_sub(H, E2), xF2 = pk_sub(H, F2);   // >= 0: H is the maximum
	const uint32_t n1 = pk_minu_s(xE, K.one), n2 = pk_minu_s(xF, K.one), n3 = pk_minu_s(xE2, K.one);
	uint32_t d;
	if (!RIGHT) {   // first maximum
		const uint32_t n0 = pk_minu_s(pk_sub(H, M), K.one);
		d = pk_mad_vss(n3, K.one);
		d = pk_mad_vvs(n2, d, K.one);
		d = pk_mad_vvs(n1, d, K.one);
		d = pk_mul(n0, d);
	} else {        // last maximum
		const uint32_t n4 = pk_minu_s(xF2, K.one);
		d = pk_rsub_s(K.one, n1);
		d = pk_mad_vvs(n2, pk_sub_s(d, K.two), K.two);
		d = pk_mad_vvs(n3, pk_sub_s(d, K.three), K.three);
		d = pk_mad_vvs(n4, pk_sub_s(d, K.four), K.four);
	}
	// continuation flags: X - H + q > 0 (RIGHT: >= 0)  <=>  H - X < q (RIGHT: < q + 1) -- a saturating q - (H - X) is non-zero exactly then;
	// its 0 / 1 image times the flag bit is accumulated into d (the bits are disjoint: + is |)
	const uint32_t c1 = RIGHT? K.q1p : K.q1, c2 = RIGHT? K.q2p : K.q2;
	d = pk_mad_vsv(pk_minu_s(pk_rsubsat_s(c1, xE), K.one), K.f8, d);
	d = pk_mad_vsv(pk_minu_s(pk_rsubsat_s(c1, xF), K.one), K.f16, d);
	d = pk_mad_vsv(pk_minu_s(pk_rsubsat_s(c2, xE2), K.one), K.f32, d);
	d = pk_mad_vsv(pk_minu_s(pk_rsubsat_s(c2, xF2), K.one), K.f64, d);
	if (!odd) accw = __builtin_amdgcn_perm(0, d, 0x0c0c0200);          // { d.lo, d.hi, 0, 0 }
	else accw = __builtin_amdgcn_perm(d, accw, 0x06040100);            // { accw.b0, accw.b1, d.lo, d.hi }
	Hp = H; Fp = F; F2p = F2;
}

template <int NS, bool RIGHT>
__device__ __forceinline__ void row_sweep(const DpConst &dc, const RowK &K, const DpJobDev &jb, const uint8_t *query, const uint8_t *target, uint8_t *p, const int tstride,
                                          const bool any_n, const int lane, int32_t &score_out)
{
	const int qlen = jb.qlen, tlen = jb.tlen;
	uint32_t Hp[NS], Fp[NS], F2p[NS], TQ[NS], KE1[NS], KE2[NS], KQ1[NS], KQ2[NS];
#pragma unroll
	for (int k = 0; k < NS; ++k) {
		const int t0 = 128 * k + 2 * lane;
		Hp[k] = pk2(row_hb(t0, dc), row_hb(t0 + 1, dc));
		Fp[k] = F2p[k] = pk2(ROW_NEG, ROW_NEG);
		TQ[k] = (t0 < tlen? (uint32_t)target[t0] : 0u) | (t0 + 1 < tlen? (uint32_t)target[t0 + 1] : 0u) << 16;
		KE1[k] = pk2(t0 * dc.e, (t0 + 1) * dc.e);
		KE2[k] = pk2(t0 * dc.e2, (t0 + 1) * dc.e2);
		KQ1[k] = pk2(t0 * dc.e + dc.q, (t0 + 1) * dc.e + dc.q);
		KQ2[k] = pk2(t0 * dc.e2 + dc.q2, (t0 + 1) * dc.e2 + dc.q2);
	}
	const uint32_t dmis = vreg_const(pk8w(dc.sc_mis - dc.sc_mch));
	int32_t hl_prev = 0, hl = row_hb(0, dc);          // H(-1, q - 1), H(-1, q)
	uint32_t qv = 0;
	uint32_t A0[NS], A1[NS];                           // direction bytes of rows (0, 1) and (2, 3) of the current tile row
#pragma unroll
	for (int k = 0; k < NS; ++k) A0[k] = A1[k] = 0;
	uint8_t *ptile = p + 8 * lane;                     // the lane's 8-byte word of set 0 in the current tile row
	for (int q0 = 0; q0 < qlen; q0 += ROW_TILE_ROWS) {
#pragma unroll
		for (int u = 0; u < ROW_TILE_ROWS; ++u) {
			const int q = q0 + u;
			if (q >= qlen) break;                      // (wave-uniform) the last tile row may be short: its other bytes are never read
			if ((q & 63) == 0) qv = q + lane < qlen? query[q + lane] : 0;
			const uint32_t qc = (uint32_t)__builtin_amdgcn_readlane((int)qv, q & 63);
			const uint32_t qc2 = qc | qc << 16;
			int32_t C1 = row_dbl(hl - dc.e), C2 = row_dbl(hl - dc.e2);   // the k = -1 term of both prefix maxima: a gap opened at the left border
			uint32_t carry_h = pk2(0, hl_prev);        // (hi half) H of the row above, one cell to the left of this set's first cell
#pragma unroll
			for (int k = 0; k < NS; ++k) {
				if (128 * k >= tlen) break;            // (wave-uniform) sets beyond the target
				row_set<RIGHT>(K, dmis, any_n, qc2, Hp[k], Fp[k], F2p[k], TQ[k], KE1[k], KE2[k], KQ1[k], KQ2[k], carry_h, C1, C2, (u & 1) != 0, u < 2? A0[k] : A1[k]);
			}
			hl_prev = hl;
			hl = row_hb(q + 1, dc);
		}
#pragma unroll
		for (int k = 0; k < NS; ++k)
			if (128 * k + 2 * lane < tlen) *(uint2*)(ptile + 512 * k) = make_uint2(A0[k], A1[k]);
		ptile += 4 * tstride;
	}
	// H(tlen - 1, qlen - 1)
	const int tl = tlen - 1;
	uint32_t hv = 0;
#pragma unroll
	for (int k = 0; k < NS; ++k) if ((tl >> 7) == k) hv = (uint32_t)__builtin_amdgcn_readlane((int)Hp[k], (tl >> 1) & 63);
	// U:ksw2_extd2_sse.c anchors the absolute score with `H0 = v[0] - qe` where qe was taken BEFORE it ordered the two gap pieces: when
	// it swaps them every absolute score carries the constant (q + e)_ordered - (q + e)_given
	score_out = (int32_t)(int16_t)(tl & 1? hv >> 16 : hv & 0xffff) + (dc.q + dc.e) - dc.qe_preswap;
}

template <int NS>
__global__ __launch_bounds__(64) void k_ksw_row(DpConst dc, const DpJobDev *jobs, const int32_t *job_ids, int n_jobs,
                                                 const uint8_t *qbase, const uint8_t *tbase, uint8_t *pbase, mm355_dpres_t *res, unsigned long long *cells_ctr)
{
	const int lane = threadIdx.x;
	if ((int)blockIdx.x >= n_jobs) return;
	const int jid = job_ids[blockIdx.x];
	const DpJobDev jb = jobs[jid];
	const uint8_t *target = tbase + jb.toff, *query = qbase + jb.qoff;
	RowK K;
	K.qe1 = pk8w(dc.q + dc.e); K.e1 = pk8w(dc.e); K.qe2 = pk8w(dc.q2 + dc.e2); K.e2 = pk8w(dc.e2); K.q1 = pk8w(dc.q); K.q2 = pk8w(dc.q2);
	K.mch = pk8w(dc.sc_mch); K.N = pk8w(dc.sc_N); K.one = 0x00010001u; K.two = 0x00020002u; K.three = 0x00030003u; K.four = 0x00040004u;
	K.f8 = 0x00080008u; K.f16 = 0x00100010u; K.f32 = 0x00200020u; K.f64 = 0x00400040u; K.q1i = dc.q; K.q2i = dc.q2; K.q1p = pk8w(dc.q + 1); K.q2p = pk8w(dc.q2 + 1);
	bool n = false;
	for (int i = lane; i < jb.tlen; i += 64) n |= target[i] > 3;
	for (int i = lane; i < jb.qlen; i += 64) n |= query[i] > 3;
	const bool any_n = __ballot(n) != 0;
	const int tstride = (jb.tlen + 15) / 16 * 16 + 16;
	int32_t score = KSW_NEG_INF;
	if (jb.flag & EZ_RIGHT) row_sweep<NS, true>(dc, K, jb, query, target, pbase + jb.p_off, tstride, any_n, lane, score);
	else row_sweep<NS, false>(dc, K, jb, query, target, pbase + jb.p_off, tstride, any_n, lane, score);
	if (lane == 0) {
		mm355_dpres_t o;
		o.max = 0; o.zdropped = 0; o.max_q = o.max_t = o.mqe_t = o.mte_q = -1; o.mqe = o.mte = KSW_NEG_INF; o.score = score; o.reach_end = 0;
		o.n_cigar = jb.tlen - 1; o.cigar_off = jb.qlen - 1;   // start cell for k_ksw_backtrack (not z-dropped, not KSW_EZ_EXTZ_ONLY)
		res[jid] = o;
		atomicAdd(cells_ctr + (blockIdx.x & (DP_CTR_SPREAD - 1)), (unsigned long long)jb.qlen * (unsigned long long)jb.tlen);
	}
}

// ------------------------------------------------------------------ long targets: eight waves, one 512-column panel each
// The same row sweep for full-band approximate fills with targets of 1025..8192 bases (on GRCh38-scale ONT batches: the fills between
// distant anchors, ~3000 x 3000, that used to be the latency tail of every extension round on the eight-wave anti-diagonal kernel).  Wave w
// owns columns [512 w, 512 w + 512) and sweeps the rows like k_ksw_row<4>; what a row needs from the panels to its left is three numbers --
// the two running prefix maxima (E, E2) at the panel edge and H of the panel's last column (for the diagonal of the next row) -- which
// the left neighbour leaves in LDS (one slot per row, rewritten in place by every panel in turn: a panel reads row q before it writes it,
// and its right neighbour reads it only after that).  Panels run 64 rows apart: `done[p]` = rows finished of panel p, published every 64
// rows; panel p waits for done[p - 1] before it loads the next 64 slots.  No barrier after the first one, every wave ends after qlen rows.
#define ROWL_NS 4
#define ROWL_PANEL (128 * ROWL_NS)
#define ROWL_WAVES 8
#define ROWL_MAX_PANELS 16           // targets beyond 4096: wave w takes panels w and w + 8 one after the other (the fills of ~5000 x 5000 between
#define ROWL_MAX_T (ROWL_PANEL * ROWL_MAX_PANELS)   // anchors a max_gap apart were the last single-block jobs of the eight-wave LDS kernel, 41 ms each)
#define ROWL_MAX_Q 5120

template <bool RIGHT>
__device__ __forceinline__ void rowl_panel(const DpConst &dc, const RowK &K, const DpJobDev &jb, const uint8_t *query, const uint8_t *target, uint8_t *p, const int tstride,
                                           const bool any_n, const int lane, const int pw, int32_t *colC1, int32_t *colC2, int32_t *colH, int *done, int32_t &score_out)
{
	const int qlen = jb.qlen, tlen = jb.tlen, tb = pw * ROWL_PANEL;
	uint32_t Hp[ROWL_NS], Fp[ROWL_NS], F2p[ROWL_NS], TQ[ROWL_NS], KE1[ROWL_NS], KE2[ROWL_NS], KQ1[ROWL_NS], KQ2[ROWL_NS];
#pragma unroll
	for (int k = 0; k < ROWL_NS; ++k) {
		const int t0 = tb + 128 * k + 2 * lane;
		Hp[k] = pk2(row_hb(t0, dc), row_hb(t0 + 1, dc));
		Fp[k] = F2p[k] = pk2(ROW_NEG, ROW_NEG);
		TQ[k] = (t0 < tlen? (uint32_t)target[t0] : 0u) | (t0 + 1 < tlen? (uint32_t)target[t0 + 1] : 0u) << 16;
		KE1[k] = pk2(t0 * dc.e, (t0 + 1) * dc.e);
		KE2[k] = pk2(t0 * dc.e2, (t0 + 1) * dc.e2);
		KQ1[k] = pk2(t0 * dc.e + dc.q, (t0 + 1) * dc.e + dc.q);
		KQ2[k] = pk2(t0 * dc.e2 + dc.q2, (t0 + 1) * dc.e2 + dc.q2);
	}
	const uint32_t dmis = vreg_const(pk8w(dc.sc_mis - dc.sc_mch));
	const bool last = tb + ROWL_PANEL >= tlen;         // nobody reads this panel's edge
	int32_t hl_prev = row_hb(tb - 1, dc);              // H(tb - 1, q - 1): the boundary row above the first row
	int32_t vC1 = 0, vC2 = 0, vH = 0;                  // lane l: the left neighbour's edge values of row (q & ~63) + l
	uint32_t qv = 0;
	uint32_t A0[ROWL_NS], A1[ROWL_NS];
#pragma unroll
	for (int k = 0; k < ROWL_NS; ++k) A0[k] = A1[k] = 0;
	uint8_t *ptile = p + 4 * tb + 8 * lane;            // (tb >> 1) * 8: the panel's first cell pair
	for (int q0 = 0; q0 < qlen; q0 += ROW_TILE_ROWS) {
#pragma unroll
		for (int u = 0; u < ROW_TILE_ROWS; ++u) {
			const int q = q0 + u;
			if (q >= qlen) break;
			if ((q & 63) == 0) {
				qv = q + lane < qlen? query[q + lane] : 0;
				if (pw > 0) {
					const int need = q + 64 < qlen? q + 64 : qlen;
					while (__hip_atomic_load(&done[pw - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
					if (q + lane < qlen) { vC1 = colC1[q + lane]; vC2 = colC2[q + lane]; vH = colH[q + lane]; }
				}
			}
			const uint32_t qc = (uint32_t)__builtin_amdgcn_readlane((int)qv, q & 63);
			const uint32_t qc2 = qc | qc << 16;
			int32_t C1, C2, hl_cur;
			if (pw == 0) { hl_cur = row_hb(q, dc); C1 = row_dbl(hl_cur - dc.e); C2 = row_dbl(hl_cur - dc.e2); }      // the k = -1 term: a gap opened at the left border (carries travel doubled, also through LDS)
			else { C1 = __builtin_amdgcn_readlane(vC1, q & 63); C2 = __builtin_amdgcn_readlane(vC2, q & 63); hl_cur = __builtin_amdgcn_readlane(vH, q & 63); }
			uint32_t carry_h = pk2(0, hl_prev);
#pragma unroll
			for (int k = 0; k < ROWL_NS; ++k) {
				if (tb + 128 * k >= tlen) break;            // (wave-uniform) sets beyond the target
				row_set<RIGHT>(K, dmis, any_n, qc2, Hp[k], Fp[k], F2p[k], TQ[k], KE1[k], KE2[k], KQ1[k], KQ2[k], carry_h, C1, C2, (u & 1) != 0, u < 2? A0[k] : A1[k]);
			}
			if (!last) {
				const int32_t he = (int32_t)__builtin_amdgcn_readlane((int)Hp[ROWL_NS - 1], 63) >> 16;     // H(tb + 511, q)
				if (lane == 0) { colC1[q] = C1; colC2[q] = C2; colH[q] = he; }
				if ((q & 63) == 63 || q == qlen - 1) __hip_atomic_store(&done[pw], q + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
			hl_prev = hl_cur;
		}
#pragma unroll
		for (int k = 0; k < ROWL_NS; ++k)
			if (tb + 128 * k + 2 * lane < tlen) *(uint2*)(ptile + 512 * k) = make_uint2(A0[k], A1[k]);
		ptile += 4 * tstride;
	}
	if (last) {
		const int tl = tlen - 1 - tb;
		uint32_t hv = 0;
#pragma unroll
		for (int k = 0; k < ROWL_NS; ++k) if ((tl >> 7) == k) hv = (uint32_t)__builtin_amdgcn_readlane((int)Hp[k], (tl >> 1) & 63);
		score_out = (int32_t)(int16_t)(tl & 1? hv >> 16 : hv & 0xffff) + (dc.q + dc.e) - dc.qe_preswap;
	}
}

__global__ __launch_bounds__(64 * ROWL_WAVES) void k_ksw_rowl(DpConst dc, const DpJobDev *jobs, const int32_t *job_ids, int n_jobs,
                                                              const uint8_t *qbase, const uint8_t *tbase, uint8_t *pbase, mm355_dpres_t *res, unsigned long long *cells_ctr)
{
	__shared__ int32_t colC1[ROWL_MAX_Q], colC2[ROWL_MAX_Q], colH[ROWL_MAX_Q];
	__shared__ int done[ROWL_MAX_PANELS], s_n;
	const int lane = threadIdx.x & 63, pw = threadIdx.x >> 6;
	if ((int)blockIdx.x >= n_jobs) return;
	__builtin_amdgcn_s_setprio(3);                     // a few hundred long sweeps beside the wide grids of the round
	const int jid = job_ids[blockIdx.x];
	const DpJobDev jb = jobs[jid];
	const uint8_t *target = tbase + jb.toff, *query = qbase + jb.qoff;
	if (threadIdx.x < ROWL_MAX_PANELS) done[threadIdx.x] = 0;
	if (threadIdx.x == 0) s_n = 0;
	__syncthreads();
	bool n = false;
	for (int i = threadIdx.x; i < jb.tlen; i += 64 * ROWL_WAVES) n |= target[i] > 3;
	for (int i = threadIdx.x; i < jb.qlen; i += 64 * ROWL_WAVES) n |= query[i] > 3;
	if (n) s_n = 1;
	__syncthreads();                                   // the only barriers: every wave is still here
	const bool any_n = s_n != 0;
	if (pw * ROWL_PANEL >= jb.tlen) return;            // no panel for this wave
	RowK K;
	K.qe1 = pk8w(dc.q + dc.e); K.e1 = pk8w(dc.e); K.qe2 = pk8w(dc.q2 + dc.e2); K.e2 = pk8w(dc.e2); K.q1 = pk8w(dc.q); K.q2 = pk8w(dc.q2);
	K.mch = pk8w(dc.sc_mch); K.N = pk8w(dc.sc_N); K.one = 0x00010001u; K.two = 0x00020002u; K.three = 0x00030003u; K.four = 0x00040004u;
	K.f8 = 0x00080008u; K.f16 = 0x00100010u; K.f32 = 0x00200020u; K.f64 = 0x00400040u; K.q1i = dc.q; K.q2i = dc.q2; K.q1p = pk8w(dc.q + 1); K.q2p = pk8w(dc.q2 + 1);
	const int tstride = (jb.tlen + 15) / 16 * 16 + 16;
	int32_t score = KSW_NEG_INF;
	bool last_mine = false;
	for (int pn = pw; pn * ROWL_PANEL < jb.tlen; pn += ROWL_WAVES) {   // (a second panel only for targets beyond 4096)
		if (jb.flag & EZ_RIGHT) rowl_panel<true>(dc, K, jb, query, target, pbase + jb.p_off, tstride, any_n, lane, pn, colC1, colC2, colH, done, score);
		else rowl_panel<false>(dc, K, jb, query, target, pbase + jb.p_off, tstride, any_n, lane, pn, colC1, colC2, colH, done, score);
		last_mine = (pn + 1) * ROWL_PANEL >= jb.tlen;
	}
	if (lane == 0 && last_mine) {   // the wave of the last panel
		mm355_dpres_t o;
		o.max = 0; o.zdropped = 0; o.max_q = o.max_t = o.mqe_t = o.mte_q = -1; o.mqe = o.mte = KSW_NEG_INF; o.score = score; o.reach_end = 0;
		o.n_cigar = jb.tlen - 1; o.cigar_off = jb.qlen - 1;   // start cell for k_ksw_backtrack (not z-dropped, not KSW_EZ_EXTZ_ONLY)
		res[jid] = o;
		atomicAdd(cells_ctr + (blockIdx.x & (DP_CTR_SPREAD - 1)), (unsigned long long)jb.qlen * (unsigned long long)jb.tlen);
	}
}
