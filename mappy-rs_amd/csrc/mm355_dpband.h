// mm355_dpband.h -- the row sweep of mm355_dprow.h on a DIAGONAL BAND of the matrix, with a proof per problem that the band is enough.
// Included by mm355_dp.hip.  Same problems as k_ksw_row / k_ksw_rowl: full-band approximate gap fills (KSW_EZ_APPROX_MAX without
// KSW_EZ_APPROX_DROP, w >= qlen + tlen, regular two-piece cost) -- whose results are the score H(tlen - 1, qlen - 1) and the CIGAR of the
// backtrack from that cell (U:ksw2_extd2_sse.c, reached from /root/reference/src/lib.rs:587 through mm_align1's gap fills).
//
// Why a band gives the SAME result.  The reference fills the whole qlen x tlen matrix; a gap fill between two chained anchors is a pair of
// nearly equal sequences, and its optimal path stays within a few dozen cells of the straight line from (0, 0) to the end cell.  Take the
// cells on the diagonals d = t - q in [dlo, dlo + W) -- a band that holds diagonal 0 and the end cell's diagonal D0 = tlen - qlen -- and
// run the recurrence with everything outside at -infinity.  Then
//   (1) every value computed inside is <= its true value (options were only removed), and the end cell's score L is the score of a real
//       alignment: L <= OPT;
//   (2) any path through a cell of diagonal d outside [min(0, D0), max(0, D0)] contains at least |d| gap columns of one kind before that cell
//       and |D0 - d| of the other kind after it, in separate gap runs, and at most (qlen + tlen - |d| - |D0 - d|) / 2 match columns:
//       its score is at most U(d) = a (qlen + tlen - |d| - |D0 - d|) / 2 - cost(|d|) - cost(|D0 - d|), cost(g) = min(q + g e, q2 + g e2),
//       a = the largest substitution score; U falls monotonically away from the band;
//   (3) if U(dlo - 1) < L and U(dlo + W) < L, no path through a cell outside the band reaches L <= OPT: every optimal path -- every
//       path the backtrack can follow, whatever the tie rule, because a tie at a cell of the walk is between two optimal continuations --
//       lies inside the band, where a cell ON an optimal path has its true value by (1) (its best prefix is itself inside).  The direction
//       byte of such a cell compares true winners with losers that are at most their true (losing) value: same byte decisions as the full
//       matrix wherever the walk looks, L = OPT, same CIGAR.
// The host computes lmin = max(U(dlo - 1), U(dlo + W)) + 1 per problem; the kernel compares and, when the proof fails (a divergent stretch,
// a long indel), appends the problem to a list that mm355_dp_run runs again on the full-matrix kernels in the same round.
//
// Layout: one wave per alignment, lane l of register set k owns diagonals dlo + 128 k + 2 l, + 1 (two int16 halves per VGPR).  In diagonal
// coordinates M comes from the SAME lane of the row above (no shift); F / F2 come from diagonal d + 1 of the row above: the next-row
// candidates max(H - q - e, F - e) are formed in place at the end of a row and shifted one cell down the lanes (DPP wave_shl + v_alignbit);
// the target base of a lane changes every row (t = q + d): the code register shifts the same way, the new base enters at the top; E / E2 run
// along the row as in the row sweep (the q e terms of the prefix cancel: the per-lane constants are d e).  The left border column and the
// top border row are ordinary cells of the band: H(t, -1) = boundary(t), H(-1, -1) = 0, everything further out ROW_NEG, and the recurrence
// reproduces H(-1, q) = boundary(q) by itself (a vertical gap from the corner).  Cells beyond the target only feed cells beyond the target.
// Direction bytes: the tiles of the row sweep (4 rows x 16 cells = 64 B) over a W-byte row: byte (q, t) at row_cell_off(q, t - q - dlo, W).
#pragma once

#define BAND_MIN_MARGIN 8

__host__ __device__ __forceinline__ size_t band_matrix_bytes(int qlen, int W) { return (size_t)((qlen + ROW_TILE_ROWS - 1) & ~(ROW_TILE_ROWS - 1)) * (size_t)W + 64; }

// wave_shl:1 with a carry word for lane 63, then the 16-bit funnel: { cur.hi, next.lo } = the cells one diagonal up
__device__ __forceinline__ uint32_t band_shl(uint32_t cur, uint32_t carry)
{
	const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)cur, 0x130, 0xf, 0xf, false);
	return __builtin_amdgcn_alignbit(nx, cur, 16);
}

template <bool RIGHT>
__device__ __forceinline__ void band_set(const RowK &K, const uint32_t dmis, const bool any_n, const uint32_t qc2, uint32_t &Hp, uint32_t &Fn, uint32_t &F2n, const uint32_t cF, const uint32_t cF2,
                                         const uint32_t TQ, const uint32_t KE1, const uint32_t KE2, const uint32_t KQ1, const uint32_t KQ2, int32_t &C1, int32_t &C2, const bool odd, uint32_t &accw)
{
	const uint32_t F = band_shl(Fn, cF), F2 = band_shl(F2n, cF2);       // gaps that consume query: from (t, q - 1), one diagonal up
	uint32_t s = pk_mad_vvs(pk_minu_s(TQ ^ qc2, K.one), dmis, K.mch);
	if (any_n) s = pk_mad(pk_shr2(TQ | qc2), pk_rsub_s(K.N, s), s);
	const uint32_t M = pk_add(Hp, s);                                   // H(t - 1, q - 1): the same diagonal, the row above
	const uint32_t G = pk_max(pk_max(M, F), F2);
	const uint32_t E = pk_sub(row_scan(pk_add(G, KE1), C1), KQ1);
	const uint32_t E2 = pk_sub(row_scan(pk_add(G, KE2), C2), KQ2);
	const uint32_t H = pk_max(pk_max(G, E), E2);
	const uint32_t xE = pk_sub(H, E), xF = pk_sub(H, F), xE2 = pk_sub(H, E2), xF2 = pk_sub(H, F2);
	const uint32_t n1 = pk_minu_s(xE, K.one), n2 = pk_minu_s(xF, K.one), n3 = pk_minu_s(xE2, K.one);
	uint32_t d;
	if (!RIGHT) {
		const uint32_t n0 = pk_minu_s(pk_sub(H, M), K.one);
		d = pk_mad_vss(n3, K.one);
		d = pk_mad_vvs(n2, d, K.one);
		d = pk_mad_vvs(n1, d, K.one);
		d = pk_mul(n0, d);
	} else {
		const uint32_t n4 = pk_minu_s(xF2, K.one);
		d = pk_rsub_s(K.one, n1);
		d = pk_mad_vvs(n2, pk_sub_s(d, K.two), K.two);
		d = pk_mad_vvs(n3, pk_sub_s(d, K.three), K.three);
		d = pk_mad_vvs(n4, pk_sub_s(d, K.four), K.four);
	}
	const uint32_t c1 = RIGHT? K.q1p : K.q1, c2 = RIGHT? K.q2p : K.q2;
	d = pk_mad_vsv(pk_minu_s(pk_rsubsat_s(c1, xE), K.one), K.f8, d);
	d = pk_mad_vsv(pk_minu_s(pk_rsubsat_s(c1, xF), K.one), K.f16, d);
	d = pk_mad_vsv(pk_minu_s(pk_rsubsat_s(c2, xE2), K.one), K.f32, d);
	d = pk_mad_vsv(pk_minu_s(pk_rsubsat_s(c2, xF2), K.one), K.f64, d);
	if (!odd) accw = __builtin_amdgcn_perm(0, d, 0x0c0c0200);
	else accw = __builtin_amdgcn_perm(d, accw, 0x06040100);
	Hp = H;
	Fn = pk_max(pk_sub_s(H, K.qe1), pk_sub_s(F, K.e1));                 // what the cell below will see as F / F2
	F2n = pk_max(pk_sub_s(H, K.qe2), pk_sub_s(F2, K.e2));
}

template <int NSB, bool RIGHT>
__device__ __forceinline__ void band_sweep(const DpConst &dc, const RowK &K, const DpJobDev &jb, const uint8_t *query, const uint8_t *target, uint8_t *p,
                                           const bool any_n, const int lane, int32_t &h_end)
{
	const int qlen = jb.qlen, tlen = jb.tlen, dlo = jb.dlo, W = 128 * NSB;
	uint32_t Hp[NSB], Fn[NSB], F2n[NSB], TQ[NSB], KE1[NSB], KE2[NSB], KQ1[NSB], KQ2[NSB];
	const uint32_t negw = pk2(ROW_NEG, ROW_NEG);
#pragma unroll
	for (int k = 0; k < NSB; ++k) {
		const int d0 = dlo + 128 * k + 2 * lane;
		int h[2];
#pragma unroll
		for (int x = 0; x < 2; ++x) { const int t = d0 + x - 1; h[x] = t >= 0? row_hb(t, dc) : t == -1? 0 : ROW_NEG; }   // the row above row 0: H(t, -1) at diagonal t + 1
		Hp[k] = pk2(h[0], h[1]);
		Fn[k] = pk_max(pk_sub_s(Hp[k], K.qe1), negw);
		F2n[k] = pk_max(pk_sub_s(Hp[k], K.qe2), negw);
		TQ[k] = ((d0 >= 0 && d0 < tlen)? (uint32_t)target[d0] : 0u) | ((d0 + 1 >= 0 && d0 + 1 < tlen)? (uint32_t)target[d0 + 1] : 0u) << 16;   // row 0: t = d
		KE1[k] = pk2(d0 * dc.e, (d0 + 1) * dc.e);
		KE2[k] = pk2(d0 * dc.e2, (d0 + 1) * dc.e2);
		KQ1[k] = pk2(d0 * dc.e + dc.q, (d0 + 1) * dc.e + dc.q);
		KQ2[k] = pk2(d0 * dc.e2 + dc.q2, (d0 + 1) * dc.e2 + dc.q2);
	}
	const uint32_t dmis = vreg_const(pk8w(dc.sc_mis - dc.sc_mch));
	const int dhi = dlo + W - 1;
	uint32_t qv = 0, tv = 0;                                  // 64 query bases / 64 incoming target bases at a time
	uint32_t A0[NSB], A1[NSB];
#pragma unroll
	for (int k = 0; k < NSB; ++k) A0[k] = A1[k] = 0;
	uint8_t *ptile = p + 8 * lane;
	const int32_t cneg = row_dbl(ROW_NEG);
	for (int q0 = 0; q0 < qlen; q0 += ROW_TILE_ROWS) {
#pragma unroll
		for (int u = 0; u < ROW_TILE_ROWS; ++u) {
			const int q = q0 + u;
			if (q >= qlen) break;
			if ((q & 63) == 0) {
				qv = q + lane < qlen? query[q + lane] : 0;
				const int ti = q + 1 + dhi + lane;                // the base that enters the band's top cell after row q + lane
				tv = (ti >= 0 && ti < tlen)? target[ti] : 0;
			}
			const uint32_t qc = (uint32_t)__builtin_amdgcn_readlane((int)qv, q & 63);
			const uint32_t qc2 = qc | qc << 16;
			const uint32_t tin = (uint32_t)__builtin_amdgcn_readlane((int)tv, q & 63);
			int32_t C1 = cneg, C2 = cneg;                      // nothing to the left of the band
#pragma unroll
			for (int k = 0; k < NSB; ++k) {
				// lane 63's upper cell looks one diagonal up: lane 0 of the next set (still the row above: that set is updated after this one)
				const uint32_t cF = k + 1 < NSB? (uint32_t)__builtin_amdgcn_readlane((int)Fn[k + 1 < NSB? k + 1 : k], 0) : negw;
				const uint32_t cF2 = k + 1 < NSB? (uint32_t)__builtin_amdgcn_readlane((int)F2n[k + 1 < NSB? k + 1 : k], 0) : negw;
				const uint32_t cT = k + 1 < NSB? (uint32_t)__builtin_amdgcn_readlane((int)TQ[k + 1 < NSB? k + 1 : k], 0) : tin;
				band_set<RIGHT>(K, dmis, any_n, qc2, Hp[k], Fn[k], F2n[k], cF, cF2, TQ[k], KE1[k], KE2[k], KQ1[k], KQ2[k], C1, C2, (u & 1) != 0, u < 2? A0[k] : A1[k]);
				TQ[k] = band_shl(TQ[k], cT);                   // the next row's bases: t = q + 1 + d
			}
		}
#pragma unroll
		for (int k = 0; k < NSB; ++k) *(uint2*)(ptile + 512 * k) = make_uint2(A0[k], A1[k]);
		ptile += 4 * W;
	}
	// H(tlen - 1, qlen - 1): diagonal D0 = tlen - qlen of the last row
	const int idx = tlen - qlen - dlo;
	uint32_t hv = 0;
#pragma unroll
	for (int k = 0; k < NSB; ++k) if ((idx >> 7) == k) hv = (uint32_t)__builtin_amdgcn_readlane((int)Hp[k], (idx >> 1) & 63);
	h_end = (int32_t)(int16_t)(idx & 1? hv >> 16 : hv & 0xffff);
}

// fail: [0] = number of problems whose band proof failed, then their job ids
template <int NSB>
__global__ __launch_bounds__(64) void k_ksw_band(DpConst dc, const DpJobDev *jobs, const int32_t *job_ids, int n_jobs,
                                                  const uint8_t *qbase, const uint8_t *tbase, uint8_t *pbase, mm355_dpres_t *res, unsigned long long *cells_ctr, int32_t *fail)
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
	int32_t h_end = KSW_NEG_INF;
	if (jb.flag & EZ_RIGHT) band_sweep<NSB, true>(dc, K, jb, query, target, pbase + jb.p_off, any_n, lane, h_end);
	else band_sweep<NSB, false>(dc, K, jb, query, target, pbase + jb.p_off, any_n, lane, h_end);
	if (lane == 0) {
		const bool ok = h_end >= jb.lmin;
		mm355_dpres_t o;
		o.max = 0; o.zdropped = 0; o.max_q = o.max_t = o.mqe_t = o.mte_q = -1; o.mqe = o.mte = KSW_NEG_INF; o.reach_end = 0;
		o.score = h_end + (dc.q + dc.e) - dc.qe_preswap;        // (the anchor of the absolute score: see row_sweep)
		o.n_cigar = ok? jb.tlen - 1 : -1; o.cigar_off = ok? jb.qlen - 1 : -1;   // start cell for k_ksw_backtrack; none when the problem is run again
		res[jid] = o;
		if (!ok) fail[1 + atomicAdd(&fail[0], 1)] = jid;
		atomicAdd(cells_ctr + (blockIdx.x & (DP_CTR_SPREAD - 1)), (unsigned long long)jb.qlen * (unsigned long long)jb.tlen);
	}
}

// ------------------------------------------------------------------ two problems per wave: bands of 64 diagonals
// The typical gap fill of an ONT read is ~210 x 210 with |tlen - qlen| of a few bases: a band of 64 diagonals proves itself there
// (lmin = a n - 4 B - 2 q2 - ... with B ~ 27: a score of 0.63 of all-matches suffices at n = 212).  Lanes 0..31 own problem A, lanes 32..63
// problem B (neighbours of the launch list: similar lengths); every per-problem quantity is a per-lane value, the row loop runs to the longer
// query, and the three things that cross lanes -- the one-diagonal shifts, the prefix scan, the row's query / incoming target base -- stop at
// the half boundary (a select on lane 31 / 32, one DPP step less in the scan, two v_readlane + a select).  Half the instructions per cell of
// k_ksw_band<1>, half the direction bytes.
__device__ __forceinline__ int32_t half_incl_scan_max32(int32_t x)   // inclusive prefix maximum over the lanes of the own half (0..31 / 32..63)
{
	int32_t y;
	y = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x111, 0xf, 0xf, false); x = x > y? x : y;
	y = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x112, 0xf, 0xf, false); x = x > y? x : y;
	y = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x114, 0xf, 0xf, false); x = x > y? x : y;
	y = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x118, 0xf, 0xf, false); x = x > y? x : y;
	y = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x142, 0xa, 0xf, false); x = x > y? x : y;
	return x;
}
__device__ __forceinline__ uint32_t half_shl(uint32_t cur, uint32_t carry, bool top)   // band_shl inside a half: the half's last lane takes the carry
{
	uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)cur, 0x130, 0xf, 0xf, false);
	nx = top? carry : nx;
	return __builtin_amdgcn_alignbit(nx, cur, 16);
}
__device__ __forceinline__ uint32_t half_scan(const uint32_t A, const int32_t cneg, bool first)
{
	const int32_t incl = half_incl_scan_max32((int32_t)pk_max_swap(A));
	int32_t ex = __builtin_amdgcn_update_dpp(cneg, incl, 0x138, 0xf, 0xf, false);
	ex = first? cneg : ex;
	return pk_max((uint32_t)ex, __builtin_amdgcn_perm(A, ROW_PKNEG, 0x05040100));
}

template <bool RIGHT>
__device__ __forceinline__ void band2_sweep(const DpConst &dc, const RowK &K, const int qlen, const int tlen, const int dlo, const int qmax, const uint8_t *query, const uint8_t *target,
                                            uint8_t *p, const bool any_n, const int lane, int32_t &h_end)
{
	const int hl = lane & 31;                                 // lane inside the half
	const bool top = hl == 31, first = hl == 0, hi = lane >= 32;
	const uint32_t negw = pk2(ROW_NEG, ROW_NEG);
	const int d0 = dlo + 2 * hl;
	int h0, h1;
	{ const int t = d0 - 1; h0 = t >= 0? row_hb(t, dc) : t == -1? 0 : ROW_NEG; }
	{ const int t = d0; h1 = t >= 0? row_hb(t, dc) : t == -1? 0 : ROW_NEG; }
	uint32_t Hp = pk2(h0, h1);
	uint32_t Fn = pk_max(pk_sub_s(Hp, K.qe1), negw), F2n = pk_max(pk_sub_s(Hp, K.qe2), negw);
	uint32_t TQ = ((d0 >= 0 && d0 < tlen)? (uint32_t)target[d0] : 0u) | ((d0 + 1 >= 0 && d0 + 1 < tlen)? (uint32_t)target[d0 + 1] : 0u) << 16;
	const uint32_t KE1 = pk2(d0 * dc.e, (d0 + 1) * dc.e), KE2 = pk2(d0 * dc.e2, (d0 + 1) * dc.e2);
	const uint32_t KQ1 = pk2(d0 * dc.e + dc.q, (d0 + 1) * dc.e + dc.q), KQ2 = pk2(d0 * dc.e2 + dc.q2, (d0 + 1) * dc.e2 + dc.q2);
	const uint32_t dmis = vreg_const(pk8w(dc.sc_mis - dc.sc_mch));
	const int dhi = dlo + 63;
	uint32_t qv = 0, tv = 0, A0 = 0, A1 = 0, Hend = 0;
	uint8_t *ptile = p + 8 * hl;
	const int32_t cneg = row_dbl(ROW_NEG);
	for (int q0 = 0; q0 < qmax; q0 += ROW_TILE_ROWS) {
#pragma unroll
		for (int u = 0; u < ROW_TILE_ROWS; ++u) {
			const int q = q0 + u;
			if (q >= qmax) break;                              // (wave-uniform)
			if ((q & 31) == 0) {
				qv = q + hl < qlen? query[q + hl] : 0;
				const int ti = q + 1 + dhi + hl;
				tv = (ti >= 0 && ti < tlen)? target[ti] : 0;
			}
			const uint32_t qa = (uint32_t)__builtin_amdgcn_readlane((int)qv, q & 31), qb = (uint32_t)__builtin_amdgcn_readlane((int)qv, 32 + (q & 31));
			const uint32_t ta = (uint32_t)__builtin_amdgcn_readlane((int)tv, q & 31), tb = (uint32_t)__builtin_amdgcn_readlane((int)tv, 32 + (q & 31));
			const uint32_t qc2 = hi? (qb | qb << 16) : (qa | qa << 16), tin = hi? tb : ta;
			// one register set, as band_set
			const uint32_t F = half_shl(Fn, negw, top), F2 = half_shl(F2n, negw, top);
			uint32_t s = pk_mad_vvs(pk_minu_s(TQ ^ qc2, K.one), dmis, K.mch);
			if (any_n) s = pk_mad(pk_shr2(TQ | qc2), pk_rsub_s(K.N, s), s);
			const uint32_t M = pk_add(Hp, s);
			const uint32_t G = pk_max(pk_max(M, F), F2);
			const uint32_t E = pk_sub(half_scan(pk_add(G, KE1), cneg, first), KQ1);
			const uint32_t E2 = pk_sub(half_scan(pk_add(G, KE2), cneg, first), KQ2);
			const uint32_t H = pk_max(pk_max(G, E), E2);
			const uint32_t xE = pk_sub(H, E), xF = pk_sub(H, F), xE2 = pk_sub(H, E2), xF2 = pk_sub(H, F2);
			const uint32_t n1 = pk_minu_s(xE, K.one), n2 = pk_minu_s(xF, K.one), n3 = pk_minu_s(xE2, K.one);
			uint32_t d;
			if (!RIGHT) {
				const uint32_t n0 = pk_minu_s(pk_sub(H, M), K.one);
				d = pk_mad_vss(n3, K.one);
				d = pk_mad_vvs(n2, d, K.one);
				d = pk_mad_vvs(n1, d, K.one);
				d = pk_mul(n0, d);
			} else {
				const uint32_t n4 = pk_minu_s(xF2, K.one);
				d = pk_rsub_s(K.one, n1);
				d = pk_mad_vvs(n2, pk_sub_s(d, K.two), K.two);
				d = pk_mad_vvs(n3, pk_sub_s(d, K.three), K.three);
				d = pk_mad_vvs(n4, pk_sub_s(d, K.four), K.four);
			}
			const uint32_t c1 = RIGHT? K.q1p : K.q1, c2 = RIGHT? K.q2p : K.q2;
			d = pk_mad_vsv(pk_minu_s(pk_rsubsat_s(c1, xE), K.one), K.f8, d);
			d = pk_mad_vsv(pk_minu_s(pk_rsubsat_s(c1, xF), K.one), K.f16, d);
			d = pk_mad_vsv(pk_minu_s(pk_rsubsat_s(c2, xE2), K.one), K.f32, d);
			d = pk_mad_vsv(pk_minu_s(pk_rsubsat_s(c2, xF2), K.one), K.f64, d);
			uint32_t &accw = u < 2? A0 : A1;
			if (!(u & 1)) accw = __builtin_amdgcn_perm(0, d, 0x0c0c0200);
			else accw = __builtin_amdgcn_perm(d, accw, 0x06040100);
			Hp = H;
			Fn = pk_max(pk_sub_s(H, K.qe1), pk_sub_s(F, K.e1));
			F2n = pk_max(pk_sub_s(H, K.qe2), pk_sub_s(F2, K.e2));
			TQ = half_shl(TQ, tin, top);
			if (q == qlen - 1) Hend = Hp;                      // (per lane: the halves end on different rows)
		}
		if (q0 < qlen) *(uint2*)ptile = make_uint2(A0, A1);     // rows of the own matrix only
		ptile += 4 * 64;
	}
	const int idx = tlen - qlen - dlo;
	const uint32_t hv = (uint32_t)__shfl((int)Hend, (lane & 32) | ((idx >> 1) & 31));
	h_end = (int32_t)(int16_t)(idx & 1? hv >> 16 : hv & 0xffff);
}

__global__ __launch_bounds__(64) void k_ksw_band2(DpConst dc, const DpJobDev *jobs, const int32_t *job_ids, int n_jobs,
                                                   const uint8_t *qbase, const uint8_t *tbase, uint8_t *pbase, mm355_dpres_t *res, unsigned long long *cells_ctr, int32_t *fail)
{
	const int lane = threadIdx.x;
	const int ja = 2 * (int)blockIdx.x, jb_ = ja + 1;
	if (ja >= n_jobs) return;
	const bool has_b = jb_ < n_jobs, hi = lane >= 32;
	const int jid = job_ids[hi && has_b? jb_ : ja];          // (an odd list: the upper half repeats problem A and writes nothing)
	const DpJobDev jb = jobs[jid];
	const bool live = !hi || has_b;
	const uint8_t *target = tbase + jb.toff, *query = qbase + jb.qoff;
	RowK K;
	K.qe1 = pk8w(dc.q + dc.e); K.e1 = pk8w(dc.e); K.qe2 = pk8w(dc.q2 + dc.e2); K.e2 = pk8w(dc.e2); K.q1 = pk8w(dc.q); K.q2 = pk8w(dc.q2);
	K.mch = pk8w(dc.sc_mch); K.N = pk8w(dc.sc_N); K.one = 0x00010001u; K.two = 0x00020002u; K.three = 0x00030003u; K.four = 0x00040004u;
	K.f8 = 0x00080008u; K.f16 = 0x00100010u; K.f32 = 0x00200020u; K.f64 = 0x00400040u; K.q1i = dc.q; K.q2i = dc.q2; K.q1p = pk8w(dc.q + 1); K.q2p = pk8w(dc.q2 + 1);
	bool n = false;
	for (int i = lane & 31; i < jb.tlen; i += 32) n |= target[i] > 3;
	for (int i = lane & 31; i < jb.qlen; i += 32) n |= query[i] > 3;
	const bool any_n = __ballot(n) != 0;                      // (either problem: the ambiguity term is exact for both)
	const int qlen = live? jb.qlen : 0;                       // the dead half stores nothing (its rows are "beyond its matrix")
	int qmax = jb.qlen;
	{ const int o = __shfl(qmax, lane ^ 32); qmax = qmax > o? qmax : o; }
	qmax = __builtin_amdgcn_readfirstlane(qmax);
	int32_t h_end = KSW_NEG_INF;
	const int rflag = __builtin_amdgcn_readfirstlane(jb.flag & EZ_RIGHT);   // (the two problems of a wave share KSW_EZ_RIGHT: the launch list is split by it)
	if (rflag) band2_sweep<true>(dc, K, qlen, jb.tlen, jb.dlo, qmax, query, target, pbase + jb.p_off, any_n, lane, h_end);
	else band2_sweep<false>(dc, K, qlen, jb.tlen, jb.dlo, qmax, query, target, pbase + jb.p_off, any_n, lane, h_end);
	if ((lane & 31) == 0 && live) {
		const bool ok = h_end >= jb.lmin;
		mm355_dpres_t o;
		o.max = 0; o.zdropped = 0; o.max_q = o.max_t = o.mqe_t = o.mte_q = -1; o.mqe = o.mte = KSW_NEG_INF; o.reach_end = 0;
		o.score = h_end + (dc.q + dc.e) - dc.qe_preswap;
		o.n_cigar = ok? jb.tlen - 1 : -1; o.cigar_off = ok? jb.qlen - 1 : -1;
		res[jid] = o;
		if (!ok) fail[1 + atomicAdd(&fail[0], 1)] = jid;
		atomicAdd(cells_ctr + ((blockIdx.x + (hi? 1 : 0)) & (DP_CTR_SPREAD - 1)), (unsigned long long)jb.qlen * (unsigned long long)jb.tlen);
	}
}
