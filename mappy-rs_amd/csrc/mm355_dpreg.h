// mm355_dpreg.h -- register-resident form of the banded extension kernel (row a12, targets up to 1024 bases).
// Included by mm355_dp.hip; same results, bit for bit, as k_ksw_extd2 (U:ksw2_extd2_sse.c::ksw_extd2_sse).
//
// One wave = one alignment.  Lane l of block j OWNS the two target positions t = 128 j + 2 l, t + 1 for the whole
// sweep: their state (u, v, x, y, x2, y2, s) sits in VGPRs as two 16-bit halves per register, each holding the int8
// value shifted left by 8.  In that "8.8" form the packed 16-bit VALU operations (v_pk_add_u16, v_pk_max_i16, ...)
// reproduce the int8 wrap-around of the SSE kernel exactly -- a carry out of bit 15 is the int8 wrap, signed order is
// preserved -- so one instruction advances two cells and no value is ever unpacked.  Nothing goes through LDS:
//   * x[t-1], v[t-1], x2[t-1] of the previous anti-diagonal come from the lower half of the own register and, for the
//     even cell, from the upper half of lane l-1 (one DPP wave_shr:1 move + v_alignbit);
//   * the query is a systolic register: every anti-diagonal it moves one cell to the right and query[r] enters at t = 0,
//     so cell t always holds query[r - t];
//   * only the blocks j that intersect the band [st, en] of the anti-diagonal are evaluated (cells outside keep their
//     stale state, as the 16-lane SSE blocks do), so the cost follows the band width, not the target length.  The sweep
//     is cut into segments of anti-diagonals with the same first/last active block (both only ever move up) and every
//     (first, last) pair has its own straight-line loop: a per-block "skip" branch would make the compiler copy the six
//     state registers of the block at the merge point, which costs more than the skipped work;
//   * the direction bytes of an anti-diagonal leave as one 2-byte store per lane (128 B per block, coalesced).
// The direction code is the argmax position among (s, a, b, a2, b2) -- first maximum for left-aligned gaps, last maximum
// with KSW_EZ_RIGHT -- computed without compares: n_i = min_u16(zmax - v_i, 1) is 0 exactly where v_i attains the maximum.
#pragma once

// The packed operations are emitted as written: left to itself the optimiser turns min_u16(x, 1) and the multiplications by
// 0/1 flags back into per-half compares and selects (SDWA v_cmp + v_cndmask + v_perm), several instructions per half.
// v = VGPR operand, s = wave-uniform constant read straight from an SGPR (one constant-bus operand per instruction)
#define DP_PK2(name, ins) __device__ __forceinline__ uint32_t name(uint32_t a, uint32_t b) { uint32_t r; asm(ins " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; } \
	__device__ __forceinline__ uint32_t name##_s(uint32_t a, uint32_t b) { uint32_t r; asm(ins " %0, %1, %2" : "=v"(r) : "v"(a), "s"(b)); return r; } \
	__device__ __forceinline__ void name##_to(uint32_t &dst, uint32_t a, uint32_t b) { asm(ins " %0, %1, %2" : "+v"(dst) : "v"(a), "v"(b)); } \
	__device__ __forceinline__ void name##_s_to(uint32_t &dst, uint32_t a, uint32_t b) { asm(ins " %0, %1, %2" : "+v"(dst) : "v"(a), "s"(b)); }
DP_PK2(pk_add, "v_pk_add_u16")
DP_PK2(pk_sub, "v_pk_sub_u16")
DP_PK2(pk_max, "v_pk_max_i16")
DP_PK2(pk_min, "v_pk_min_i16")
DP_PK2(pk_minu, "v_pk_min_u16")
__device__ __forceinline__ uint32_t pk_rsub_s(uint32_t c, uint32_t a) { uint32_t r; asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "s"(c), "v"(a)); return r; }   // c - a
__device__ __forceinline__ uint32_t pk_max0(uint32_t a) { uint32_t r; asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(a)); return r; }
__device__ __forceinline__ uint32_t pk_mul(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ uint32_t pk_mad_vvs(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c)); return r; }
__device__ __forceinline__ uint32_t pk_mad_vss(uint32_t a, uint32_t c) { uint32_t r; asm("v_pk_mad_u16 %0, %1, %2, %2" : "=v"(r) : "v"(a), "s"(c)); return r; }   // a * c + c
__device__ __forceinline__ uint32_t pk_shr2(uint32_t a) { uint32_t r; asm("v_pk_lshrrev_b16 %0, 2, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a)); return r; }
__device__ __forceinline__ uint32_t vreg_const(uint32_t c) { uint32_t r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(c)); return r; }   // constant pinned in a VGPR (not rematerialised)
__device__ __forceinline__ uint32_t pk8(int v) { const uint32_t h = (uint32_t)(uint8_t)v << 8; return h | h << 16; }   // int8 -> both halves, 8.8
__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }           // m ? a : b (v_bfi_b32)
// value of lane-1 (lane 0: `first`)
__device__ __forceinline__ uint32_t lane_shr1(uint32_t first, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t lane_shr1_z(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xf, 0xf, true); }   // lane 0 gets 0
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// Per-lane state: individually named members (arrays indexed by the unrolled block number get promoted to vector values by
// the optimiser, and every update of one element then copies the whole tuple).
#define DP_MEMB(k) uint32_t U##k, V##k, X##k, Y##k, X2##k, Y2##k, SC##k, TQ##k, QQ##k; int32_t Hl##k, Hh##k;
struct DpSt { DP_MEMB(0) DP_MEMB(1) DP_MEMB(2) DP_MEMB(3) DP_MEMB(4) DP_MEMB(5) DP_MEMB(6) DP_MEMB(7) };
struct DpK { uint32_t nqe, nq2e2, q, q2, qe, q2e2, mch, dmis_v, N, one, c256, m256, two, three, four, f8, f16, f32, f64, dx1, dx21; };
struct DpRun {             // wave-uniform state of the sweep
	int qlen, tlen, w, T, n_col, flag, zdrop, end_bonus, q, e, q2, e2, qe, long_thres, long_diff, r_total;
	const uint8_t *query, *target; uint8_t *p;
	int r, st, en, st0, en0, last_st, last_en;
	int base;              // WIN: target position of cell 0 of block 0 (a multiple of 128); 0 otherwise
	int32_t H0, last_H0_t;
	unsigned long long cells;
	uint32_t qv;
	bool any_n;
	bool full;             // the band never binds (w >= qlen + tlen) and no KSW_EZ_APPROX_DROP: see dp_bounds / the approximate score tracking
	EzState ez;
};

__device__ __forceinline__ bool dp_bounds(DpRun &R)   // U:ksw2_extd2_sse.c: st/en of anti-diagonal R.r; false: the band has left the matrix
{
	const int r = R.r;
	int st = 0, en = R.tlen - 1;
	if (R.full) {   // (r - w + 1) >> 1 <= 0 and (r + w) >> 1 >= r: only the matrix borders bind, and st <= en always
		if (st < r - R.qlen + 1) st = r - R.qlen + 1;
		if (en > r) en = r;
		R.st0 = st; R.en0 = en;
		R.st = st & ~15; R.en = (en + 16) / 16 * 16 - 1;
		return true;
	}
	if (st < r - R.qlen + 1) st = r - R.qlen + 1;
	if (en > r) en = r;
	if (st < (r - R.w + 1) >> 1) st = (r - R.w + 1) >> 1;
	if (en > (r + R.w) >> 1) en = (r + R.w) >> 1;
	if (st > en) return false;
	R.st0 = st; R.en0 = en;
	R.st = st / 16 * 16; R.en = (en + 16) / 16 * 16 - 1;
	return true;
}

struct DpDiag {            // wave-uniform description of one anti-diagonal
	int r, st, en, st0, jq, lane_st, lane_r, pos_st, pos_r;   // jq: block of the top-row cell t = r; pos_st / pos_r: (even) cell of st / of t = r, or -1 (catch-all instance)
	uint32_t sclen, dv1, edge_u8, qc_hi;
	bool use_def, edge, any_n;
	uint64_t rowp;         // p + r * n_col - st: where cell 0 of block 0 would be stored (wave-uniform, made so explicitly: an "s" asm operand the
	                       // compiler believes divergent is silently given VGPRs)
};

__device__ __forceinline__ void dp_slide(uint32_t &QQ, const uint32_t first_hi)   // the query moves one cell to the right
{
	QQ = __builtin_amdgcn_alignbit(QQ, lane_shr1(first_hi, QQ), 16);
}

template <int J>
__device__ __forceinline__ void dp_score(const DpDiag &g, const DpK &k, const int lane, uint32_t &SC, const uint32_t TQ, const uint32_t QQ)
{
	uint32_t s = pk_mad_vvs(pk_minu_s(TQ ^ QQ, k.one), k.dmis_v, k.mch);
	// either base ambiguous (code 4): sc_N.  Unconditional -- (TQ | QQ) >> 2 is 0 for plain bases and the multiply-add then returns s; under
	// `if (any_n)` this was a wave-uniform branch in the middle of every block of every anti-diagonal, i.e. a basic-block boundary the
	// scheduler cannot move the neighbouring blocks' instructions across (three instructions saved, the interleaving of eight chains lost)
	s = pk_mad(pk_shr2(TQ | QQ), pk_rsub_s(k.N, s), s);
	const uint32_t tl = (uint32_t)(128 * J + 2 * lane - g.st0);
	const uint32_t m = (tl < g.sclen? 0xffffu : 0u) | (tl + 1 < g.sclen? 0xffff0000u : 0u);
	SC = bfi(m, s, SC);
}

// one block of one anti-diagonal; pX/pV/pX2 = registers of block J-1 (previous anti-diagonal)
// HASF: fx / fv / fx2 = the registers of the cell pair just below this block (lane 63 of block J-1 on the previous anti-diagonal); the
// multi-wave kernel (mm355_dpmw.h) passes them in from its neighbour wave's mailbox
template <int J, bool IS_LO, bool IS_HI, bool RIGHT, bool ANYBLK, bool HASF>
__device__ __forceinline__ void dp_core_f(const DpDiag &g, const DpK &k, const int lane, uint8_t *p,
                                          uint32_t &U, uint32_t &V, uint32_t &X, uint32_t &Y, uint32_t &X2, uint32_t &Y2, const uint32_t SC,
                                          const uint32_t fx, const uint32_t fv, const uint32_t fx2);
template <int J, bool IS_LO, bool IS_HI, bool RIGHT, bool ANYBLK>
__device__ __forceinline__ void dp_core(const DpDiag &g, const DpK &k, const int lane, uint8_t *p,
                                        uint32_t &U, uint32_t &V, uint32_t &X, uint32_t &Y, uint32_t &X2, uint32_t &Y2, const uint32_t SC,
                                        const uint32_t pX, const uint32_t pV, const uint32_t pX2)
{
	uint32_t fx = 0, fv = 0, fx2 = 0;
	if (J > 0) { fx = rdlane(pX, 63); fv = rdlane(pV, 63); fx2 = rdlane(pX2, 63); }
	dp_core_f<J, IS_LO, IS_HI, RIGHT, ANYBLK, (J > 0)>(g, k, lane, p, U, V, X, Y, X2, Y2, SC, fx, fv, fx2);
}
template <int J, bool IS_LO, bool IS_HI, bool RIGHT, bool ANYBLK, bool HASF>
__device__ __forceinline__ void dp_core_f(const DpDiag &g, const DpK &k, const int lane, uint8_t *p,
                                          uint32_t &U, uint32_t &V, uint32_t &X, uint32_t &Y, uint32_t &X2, uint32_t &Y2, const uint32_t SC,
                                          const uint32_t fx, const uint32_t fv, const uint32_t fx2)
{
	uint32_t nx_, nv_, nx2_;
	if (HASF) { nx_ = lane_shr1(fx, X); nv_ = lane_shr1(fv, V); nx2_ = lane_shr1(fx2, X2); }
	else { nx_ = lane_shr1_z(X); nv_ = lane_shr1_z(V); nx2_ = lane_shr1_z(X2); }   // lane 0 of block 0: t-1 = -1, always a boundary value (IS_LO below)
	// (ANYBLK: the catch-all instance, whose first / last ACTIVE block is only known at run time -- every block compares its cells'
	// position with that of st / of the top-row cell: selects, no branch, so that the eight blocks of an anti-diagonal stay ONE
	// basic block and the scheduler can interleave their dependency chains -- a single wave has nothing else to hide latency with)
	const int tl = 128 * J + 2 * lane;
	if (IS_LO) {   // x[st-1], v[st-1], x2[st-1] are boundary values when that cell was outside the previous anti-diagonal
		const bool at = ANYBLK? tl == g.pos_st : lane == g.lane_st;   // (lane_st / pos_st = -1 when the neighbour's state is to be used)
		nx_ = at? k.dx1 : nx_; nv_ = at? g.dv1 : nv_; nx2_ = at? k.dx21 : nx2_;
	}
	const uint32_t XT = __builtin_amdgcn_alignbit(X, nx_, 16), VT = __builtin_amdgcn_alignbit(V, nv_, 16), X2T = __builtin_amdgcn_alignbit(X2, nx2_, 16);
	uint32_t yi = Y, y2i = Y2, ui = U;
	if (IS_HI) {   // top row (query position 0, only ever in the last active block): y, y2 and u are the boundary values
		const uint32_t hm = (g.r & 1)? 0xffff0000u : 0xffffu;
		const uint32_t m = (ANYBLK? tl == g.pos_r : lane == (J == g.jq? g.lane_r : -1))? hm : 0u;   // lane_r / pos_r = -1 without edge
		yi = bfi(m, k.nqe, yi); y2i = bfi(m, k.nq2e2, y2i); ui = bfi(m, g.edge_u8, ui);
	}
	// every lane computes; the six state registers are then overwritten under an EXEC mask (lanes outside [st, en] keep theirs).
	// The masked writes are one asm block: written as C++ under `if (act)`, or as selects, the compiler copies/selects all six.
	const bool act = (uint32_t)(tl - g.st) <= (uint32_t)(g.en - g.st);   // st <= tl <= en in one compare
	const uint32_t z0 = SC;
	uint32_t a = pk_add(XT, VT), b = pk_add(yi, ui), a2 = pk_add(X2T, VT), b2 = pk_add(y2i, ui);
	const uint32_t zm = pk_max(pk_max(pk_max(pk_max(z0, a), b), a2), b2);
	const uint32_t n1 = pk_minu_s(pk_sub(zm, a), k.one), n2 = pk_minu_s(pk_sub(zm, b), k.one), n3 = pk_minu_s(pk_sub(zm, a2), k.one);
	uint32_t d;
	if (!RIGHT) {   // first maximum
		const uint32_t n0 = pk_minu_s(pk_sub(zm, z0), k.one);
		d = pk_mad_vss(n3, k.one);             // 1 + n3
		d = pk_mad_vvs(n2, d, k.one);          // 1 + n2 (1 + n3)
		d = pk_mad_vvs(n1, d, k.one);
		d = pk_mul(n0, d);
	} else {        // last maximum
		const uint32_t n4 = pk_minu_s(pk_sub(zm, b2), k.one);
		d = pk_rsub_s(k.one, n1);                              // a attains the maximum -> 1
		d = pk_mad_vvs(n2, pk_sub_s(d, k.two), k.two);         // b attains it -> 2, else keep
		d = pk_mad_vvs(n3, pk_sub_s(d, k.three), k.three);
		d = pk_mad_vvs(n4, pk_sub_s(d, k.four), k.four);
	}
	const uint32_t z = pk_min_s(zm, k.mch);
	uint32_t tmp = pk_sub_s(z, k.q);
	a = pk_sub(a, tmp); b = pk_sub(b, tmp);
	tmp = pk_sub_s(z, k.q2);
	a2 = pk_sub(a2, tmp); b2 = pk_sub(b2, tmp);
	const uint32_t pa = pk_max0(a), pb = pk_max0(b), pa2 = pk_max0(a2), pb2 = pk_max0(b2);
	uint32_t fa, fb, fa2, fb2;   // continuation flags already at their bit positions (0x08, 0x10, 0x20, 0x40)
	if (!RIGHT) { fa = pk_minu_s(pa, k.f8); fb = pk_minu_s(pb, k.f16); fa2 = pk_minu_s(pa2, k.f32); fb2 = pk_minu_s(pb2, k.f64); }   // > 0
	else {          // >= 0
		fa = pk_minu_s(pk_add_s(pk_max_s(a, k.m256), k.c256), k.f8); fb = pk_minu_s(pk_add_s(pk_max_s(b, k.m256), k.c256), k.f16);
		fa2 = pk_minu_s(pk_add_s(pk_max_s(a2, k.m256), k.c256), k.f32); fb2 = pk_minu_s(pk_add_s(pk_max_s(b2, k.m256), k.c256), k.f64);
	}
	d = d | fa | fb; d = d | fa2 | fb2;
	{   // ... and the two direction bytes of the lane leave in the same EXEC region (written as C++ under `if (act)` the compiler opens its own
		// EXEC region with a branch around it: a scheduling boundary after every block)
		const uint64_t actm = __builtin_amdgcn_ballot_w64(act);
		const uint32_t dd = __builtin_amdgcn_perm(0, d, 0x0c0c0200);
		uint64_t saved;
		asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
		             "v_pk_sub_u16 %[X], %[pa], %[qe]\n\t"
		             "v_pk_sub_u16 %[Y], %[pb], %[qe]\n\t"
		             "v_pk_sub_u16 %[X2], %[pa2], %[q2e2]\n\t"
		             "v_pk_sub_u16 %[Y2], %[pb2], %[q2e2]\n\t"
		             "v_pk_sub_u16 %[U], %[z], %[VT]\n\t"
		             "v_pk_sub_u16 %[V], %[z], %[ui]\n\t"
		             "global_store_short %[off], %[dd], %[rowp]\n\t"
		             "s_mov_b64 exec, %[sv]"
		             : [X] "+&v"(X), [Y] "+&v"(Y), [X2] "+&v"(X2), [Y2] "+&v"(Y2), [U] "+&v"(U), [V] "+&v"(V), [sv] "=&s"(saved)
		             : [m] "s"(actm), [pa] "v"(pa), [pb] "v"(pb), [pa2] "v"(pa2), [pb2] "v"(pb2), [z] "v"(z), [VT] "v"(VT), [ui] "v"(ui),
		               [qe] "s"(k.qe), [q2e2] "s"(k.q2e2), [off] "v"(tl), [dd] "v"(dd), [rowp] "s"(g.rowp)
		             : "scc");
	}
}

// lane 63 <- maximum over the wave (inclusive scan by row shifts and row broadcasts; the other lanes hold prefix maxima)
__device__ __forceinline__ int32_t dp_wave_max_i32(int32_t x)
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
__device__ __forceinline__ uint32_t dp_wave_max_u32(uint32_t x)
{
	uint32_t y;
	y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false); x = x > y? x : y;
	y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false); x = x > y? x : y;
	y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false); x = x > y? x : y;
	y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false); x = x > y? x : y;
	y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); x = x > y? x : y;
	y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); x = x > y? x : y;
	return x;
}

// exact score tracking of one block: H[t] += v[t] on [st0, en0), H[en0] = hen, best (H, priority) candidate of the lane
template <int J>
__device__ __forceinline__ void dp_block_h(const int lane, const int st0, const int en0, const int en1, const int32_t hen, const uint32_t V, int32_t &Hl, int32_t &Hh, long long &best)
{
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		int32_t &Hc = h? Hh : Hl;
		const int t = 128 * J + 2 * lane + h;
		int32_t hv = Hc + (int32_t)(int8_t)(V >> (h? 24 : 8));
		const bool in = t >= st0 && t < en0;
		const uint32_t prio = (uint32_t)(t < en1? (t - st0) & 3 : 4) << 16 | (uint32_t)t;
		const long long key = in? (long long)(((unsigned long long)(uint32_t)hv << 32) | (uint32_t)~prio) : INT64_MIN;
		best = key > best? key : best;
		hv = in? hv : Hc;
		Hc = t == en0? hen : hv;
	}
}

// value of register R of block j (wave-uniform j) at lane l: the block's register is chosen with selects (one v_cndmask per block on a
// scalar condition), then ONE v_readlane -- as nested `j == k? readlane(Rk)` the compiler built a branch tree per pick, ~100 tiny basic
// blocks per anti-diagonal in the exact kernels
template <int NP>
__device__ __forceinline__ uint32_t dp_sel8(const int j, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t r4, uint32_t r5, uint32_t r6, uint32_t r7)
{
	uint32_t x = r0;
	if (NP > 1) x = j == 1? r1 : x;
	if (NP > 2) x = j == 2? r2 : x;
	if (NP > 3) x = j == 3? r3 : x;
	if (NP > 4) x = j == 4? r4 : x;
	if (NP > 5) x = j == 5? r5 : x;
	if (NP > 6) x = j == 6? r6 : x;
	if (NP > 7) x = j == 7? r7 : x;
	return x;
}
#define DP_PICK(R, j, l) rdlane(dp_sel8<NP>((j), (uint32_t)S.R##0, (uint32_t)S.R##1, (uint32_t)S.R##2, (uint32_t)S.R##3, (uint32_t)S.R##4, (uint32_t)S.R##5, (uint32_t)S.R##6, (uint32_t)S.R##7), l)
#define DP_CELL8(R, t) ((int)(int8_t)(DP_PICK(R, (t) >> 7, ((t) >> 1) & 63) >> (((t) & 1)? 24 : 8)))
#define DP_SEL(R, j) dp_sel8<NP>((j), (uint32_t)S.R##0, (uint32_t)S.R##1, (uint32_t)S.R##2, (uint32_t)S.R##3, (uint32_t)S.R##4, (uint32_t)S.R##5, (uint32_t)S.R##6, (uint32_t)S.R##7)
#define DP_HAT(t) (int32_t)rdlane(((t) & 1)? DP_SEL(Hh, (t) >> 7) : DP_SEL(Hl, (t) >> 7), ((t) >> 1) & 63)
#define DP_IN(k) (NP > k && k >= JLO && k <= JHI)

// one anti-diagonal with active blocks JLO..JHI (a superset of the blocks that intersect [st, en] is fine: lanes outside are
// masked); returns false when the sweep ends here (z-drop)
template <int NP, bool EXACT, bool RIGHT, int JLO, int JHI, bool WIN, bool WIDE>
__device__ __forceinline__ bool dp_diag(DpRun &R, DpSt &S, const DpK &K, const int lane)
{
	const int r = R.r, st = R.st, en = R.en, st0 = R.st0, en0 = R.en0;
	const int base = WIN? R.base : 0;                  // the registers hold target positions [base, base + 128 NP): `_r` = relative to base
	const int st0_r = st0 - base, en0_r = en0 - base;
	DpDiag g;
	g.r = r; g.st = st - base; g.en = en - base; g.st0 = st0_r; g.any_n = R.any_n;
	g.use_def = st == 0 || !(st - 1 >= R.last_st && st - 1 <= R.last_en);
	const int edge_u = r == 0? -R.q - R.e : r < R.long_thres? -R.e : r == R.long_thres? R.long_diff : -R.e2;
	g.edge_u8 = pk8(edge_u);
	g.dv1 = st > 0? K.nqe & 0xffff0000u : g.edge_u8 & 0xffff0000u;   // v1 at st == 0 follows the same schedule as the edge u
	g.edge = en >= r;
	int sce = st0 + ((en0 - st0) / 16 + 1) * 16;   // scores are (re)written for t in [st0, sce), clipped to the padded target
	if (sce > R.T) sce = R.T;
	g.sclen = (uint32_t)(sce - st0);
	if ((r & 63) == 0) R.qv = r - base + lane < R.qlen? R.query[r - base + lane] : 0;   // query[r - base] enters at cell 0
	g.qc_hi = rdlane(R.qv, r & 63) << 16;
	g.jq = (r - base) >> 7;
	g.lane_st = g.use_def? (st & 127) >> 1 : -1; g.lane_r = g.edge? (r & 127) >> 1 : -1;
	g.pos_st = g.use_def? st - base : -1; g.pos_r = g.edge? (r - base) & ~1 : -1;
	{
		const uint64_t rp = (uint64_t)(R.p + ((size_t)r * R.n_col - (size_t)(st - base)));
		g.rowp = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rp) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rp >> 32)) << 32;
	}
	const int jsh = (sce - 1 - base) >> 7;
	// 1. the query moves (descending: block J takes the last cell of block J-1 before that one moves); blocks beyond t = r hold zeros
#define DP_SLIDE(k, m) if constexpr (NP > k) { if (k <= JHI || k <= g.jq) dp_slide(S.QQ##k, k > 0? rdlane(S.QQ##m, 63) : g.qc_hi); }
	DP_SLIDE(7, 6) DP_SLIDE(6, 5) DP_SLIDE(5, 4) DP_SLIDE(4, 3) DP_SLIDE(3, 2) DP_SLIDE(2, 1) DP_SLIDE(1, 0) DP_SLIDE(0, 0)
	// 2. scores of [st0, sce): the active blocks, plus the next one when the last 16-cell chunk spills over
#define DP_SCORE(k) if constexpr (NP > k) { if (DP_IN(k) || (k == JHI + 1 && jsh > JHI)) dp_score<k>(g, K, lane, S.SC##k, S.TQ##k, S.QQ##k); }
	DP_SCORE(0) DP_SCORE(1) DP_SCORE(2) DP_SCORE(3) DP_SCORE(4) DP_SCORE(5) DP_SCORE(6) DP_SCORE(7)
	// 3. the recurrence, descending for the same reason
#define DP_CORE(k, m) if constexpr (DP_IN(k)) dp_core<k, WIDE || k == JLO, WIDE || k == JHI, RIGHT, WIDE>(g, K, lane, R.p, S.U##k, S.V##k, S.X##k, S.Y##k, S.X2##k, S.Y2##k, S.SC##k, S.X##m, S.V##m, S.X2##m);
	DP_CORE(7, 6) DP_CORE(6, 5) DP_CORE(5, 4) DP_CORE(4, 3) DP_CORE(3, 2) DP_CORE(2, 1) DP_CORE(1, 0) DP_CORE(0, 0)
	R.cells += (unsigned long long)(en0 - st0 + 1);
	EzState &ez = R.ez;
	if constexpr (EXACT) {
		int32_t max_H, max_t, Hen0, Hst0;
		if (r > 0) {
			const int32_t hen = en0 > 0? DP_HAT(en0_r - 1) + DP_CELL8(U, en0_r) : DP_HAT(en0_r) + DP_CELL8(V, en0_r);
			const int en1_r = st0_r + (en0_r - st0_r) / 4 * 4;
			long long best = INT64_MIN;
#define DP_STEP_H(k) if constexpr (DP_IN(k)) dp_block_h<k>(lane, st0_r, en0_r, en1_r, hen, S.V##k, S.Hl##k, S.Hh##k, best);
			DP_STEP_H(0) DP_STEP_H(1) DP_STEP_H(2) DP_STEP_H(3) DP_STEP_H(4) DP_STEP_H(5) DP_STEP_H(6) DP_STEP_H(7)
			// wave maximum of the 64-bit keys in two 32-bit DPP reductions (H first, then the priority word among the lanes that hold that H):
			// six row-shift / row-broadcast steps each, instead of six dependent 64-bit ds_bpermute exchanges (~1000 cycles per anti-diagonal)
			{
				const int32_t bh = (int32_t)(best >> 32);
				const int32_t mh = (int32_t)rdlane((uint32_t)dp_wave_max_i32(bh), 63);
				const uint32_t bl = bh == mh? (uint32_t)best : 0u;
				const uint32_t ml = rdlane(dp_wave_max_u32(bl), 63);
				best = (long long)(((unsigned long long)(uint32_t)mh << 32) | ml);
			}
			max_H = hen; max_t = en0;
			if (best != INT64_MIN) {
				const int32_t ch = (int32_t)(best >> 32);
				if (ch > hen) { max_H = ch; max_t = (int)(~(uint32_t)best & 0xffffu) + base; }
			}
			Hen0 = hen; Hst0 = st0 == en0? hen : DP_HAT(st0_r);
		} else {
			const int32_t h0 = DP_CELL8(V, 0) - R.qe;
			S.Hl0 = lane == 0? h0 : S.Hl0;
			max_H = h0; max_t = 0; Hen0 = Hst0 = h0;
		}
		if (en0 == R.tlen - 1 && Hen0 > ez.mte) ez.mte = Hen0, ez.mte_q = r - en;
		if (r - st0 == R.qlen - 1 && Hst0 > ez.mqe) ez.mqe = Hst0, ez.mqe_t = st0;
		if (apply_zdrop(ez, max_H, r, max_t, R.zdrop, R.e2)) return false;
		if (r == R.qlen + R.tlen - 2 && en0 == R.tlen - 1) ez.score = Hen0;   // en0 == tlen - 1: H[tlen-1] is H[en0]
	} else {
		if (R.full) {
			// Full band and no z-drop on the approximate score: H0 only feeds ez.score at the last anti-diagonal, and with every cell of
			// the matrix exact any monotone path sums to the same H(tlen-1, qlen-1).  Take the top row (u of the edge cell t = r) and then
			// the last column (v of t = tlen-1) instead of replaying the reference's v-or-u walk: one register read per anti-diagonal.
			if (r == 0) R.H0 = DP_CELL8(V, 0) - R.qe;
			else if (r < R.tlen) R.H0 += DP_CELL8(U, r);
			else R.H0 += DP_CELL8(V, R.tlen - 1);
			if (r == R.qlen + R.tlen - 2) ez.score = R.H0;   // en0 == tlen - 1 there
			return true;
		}
		if (r > 0) {
			const int lt = R.last_H0_t - base;
			const bool in0 = lt >= st0_r && lt <= en0_r, in1 = lt + 1 >= st0_r && lt + 1 <= en0_r;
			if (in0 && in1) {
				const int32_t d0 = DP_CELL8(V, lt), d1 = DP_CELL8(U, lt + 1);
				if (d0 > d1) R.H0 += d0;
				else R.H0 += d1, ++R.last_H0_t;
			} else if (in0) R.H0 += DP_CELL8(V, lt);
			else { ++R.last_H0_t; R.H0 += DP_CELL8(U, lt + 1); }
		} else R.H0 = DP_CELL8(V, 0) - R.qe, R.last_H0_t = 0;
		if ((R.flag & EZ_APPROX_DROP) && apply_zdrop(ez, R.H0, r, R.last_H0_t, R.zdrop, R.e2)) return false;
		if (r == R.qlen + R.tlen - 2 && en0 == R.tlen - 1) ez.score = R.H0;
	}
	return true;
}

// WIN (NP = 8): the register window moves up by one block -- block k takes over the state of block k + 1, block 7 starts fresh on the
// next 128 target positions.  Called between two anti-diagonals (R.r is the next one) once the band has left block 0 for good (st and en
// never decrease): cells below st - 1 are never read again.  A fresh cell is what the SSE kernel's arrays hold for a position the band
// has not reached yet: the initial u/v/x/y, score 0, H = -inf, and the query base the systolic register would have carried there.
__device__ __forceinline__ void dp_rebase(DpRun &R, DpSt &S, const DpK &K, const int lane)
{
#define DP_SHIFT(k, m) S.U##k = S.U##m; S.V##k = S.V##m; S.X##k = S.X##m; S.Y##k = S.Y##m; S.X2##k = S.X2##m; S.Y2##k = S.Y2##m; S.SC##k = S.SC##m; \
		S.TQ##k = S.TQ##m; S.QQ##k = S.QQ##m; S.Hl##k = S.Hl##m; S.Hh##k = S.Hh##m;
	DP_SHIFT(0, 1) DP_SHIFT(1, 2) DP_SHIFT(2, 3) DP_SHIFT(3, 4) DP_SHIFT(4, 5) DP_SHIFT(5, 6) DP_SHIFT(6, 7)
	R.base += 128;
	const int t = R.base + 896 + 2 * lane;
	S.U7 = S.V7 = S.X7 = S.Y7 = K.nqe; S.X27 = S.Y27 = K.nq2e2; S.SC7 = 0; S.Hl7 = S.Hh7 = KSW_NEG_INF;
	S.TQ7 = (t < R.tlen? (uint32_t)R.target[t] : 0u) | (t + 1 < R.tlen? (uint32_t)R.target[t + 1] : 0u) << 16;
	const int qi = R.r - 1 - t;                        // the registers stand at anti-diagonal r - 1: cell t holds query[r - 1 - t]
	S.QQ7 = (qi >= 0 && qi < R.qlen? (uint32_t)R.query[qi] : 0u) | (qi - 1 >= 0 && qi - 1 < R.qlen? (uint32_t)R.query[qi - 1] : 0u) << 16;
	const int q0 = (R.r & ~63) - R.base;               // the 64 query bases that enter at cell 0 during this 64-diagonal period
	R.qv = q0 + lane >= 0 && q0 + lane < R.qlen? R.query[q0 + lane] : 0;
}

// all anti-diagonals whose active blocks are JLO..JHI (WIDE: the catch-all instance that covers every block, used while the
// band spans more than DP_MAX_TIGHT blocks); returns true when the sweep is finished
#define DP_MAX_TIGHT 4
template <int NP, bool EXACT, bool RIGHT, int JLO, int JHI, bool WIDE, bool WIN>
__device__ __forceinline__ bool dp_segment(DpRun &R, DpSt &S, const DpK &K, const int lane)
{
	for (;;) {
		if (!dp_diag<NP, EXACT, RIGHT, JLO, JHI, WIN, WIDE>(R, S, K, lane)) return true;
		R.last_st = R.st; R.last_en = R.en;
		if (++R.r >= R.r_total) return true;
		if (!dp_bounds(R)) { R.ez.zdropped = 1; return true; }
		if constexpr (WIN) { while (R.st - 1 - R.base >= 128) dp_rebase(R, S, K, lane); }
		const int base = WIN? R.base : 0;
		const int jlo = (R.st - base) >> 7, jhi = (R.en - base) >> 7;
		// (WIN runs on catch-all instances only: blocks 0..6 or 1..7 -- a 752-cell band never touches both end blocks of the window at once --
		// and all eight for bands close to the limit; an instance is left when the band moves out of its blocks or fits a narrower one)
		if (WIN) { if (jlo < JLO || jhi > JHI || (JHI - JLO == NP - 1 && (jhi < NP - 1 || jlo > 0))) return false; }
		else if (WIDE? jhi - jlo < DP_MAX_TIGHT : (jlo != JLO || jhi != JHI)) return false;
	}
}

template <int NP, bool EXACT, bool RIGHT, int JLO, int JHI, bool WIN>
__device__ __forceinline__ bool dp_dispatch(const int jlo, const int jhi, DpRun &R, DpSt &S, const DpK &K, const int lane)
{
	if (jlo == JLO && jhi == JHI) return dp_segment<NP, EXACT, RIGHT, JLO, JHI, false, WIN>(R, S, K, lane);
	if constexpr (JHI + 1 < NP && JHI + 1 - JLO < DP_MAX_TIGHT) return dp_dispatch<NP, EXACT, RIGHT, JLO, JHI + 1, WIN>(jlo, jhi, R, S, K, lane);
	else if constexpr (JLO + 1 < NP) return dp_dispatch<NP, EXACT, RIGHT, JLO + 1, JLO + 1, WIN>(jlo, jhi, R, S, K, lane);
	else return true;   // not reached
}

template <int NP, bool EXACT, bool RIGHT, bool WIN>
__device__ __forceinline__ void dp_sweep(DpRun &R, DpSt &S, const DpK &K, const int lane)
{
	if (R.r_total <= 0) return;
	if (!dp_bounds(R)) { R.ez.zdropped = 1; return; }
	for (;;) {
		const int base = WIN? R.base : 0;
		const int jlo = (R.st - base) >> 7, jhi = (R.en - base) >> 7;
		bool done;
		if constexpr (WIN) {
			if (jhi < NP - 1) done = dp_segment<NP, EXACT, RIGHT, 0, NP - 2, true, true>(R, S, K, lane);
			else if (jlo > 0) done = dp_segment<NP, EXACT, RIGHT, 1, NP - 1, true, true>(R, S, K, lane);
			else done = dp_segment<NP, EXACT, RIGHT, 0, NP - 1, true, true>(R, S, K, lane);
		}
		else if (NP > DP_MAX_TIGHT && jhi - jlo >= DP_MAX_TIGHT) done = dp_segment<NP, EXACT, RIGHT, 0, NP - 1, true, false>(R, S, K, lane);
		else done = dp_dispatch<NP, EXACT, RIGHT, 0, 0, false>(jlo, jhi, R, S, K, lane);
		if (done) break;
	}
}

template <int NP, bool EXACT, bool WIN>
__device__ __forceinline__ void dp_reg_body(const DpConst &dc, const DpJobDev *jobs, const int32_t *job_ids, int n_jobs,
                                            const uint8_t *qbase, const uint8_t *tbase, uint8_t *pbase, mm355_dpres_t *res, unsigned long long *cells_ctr)
{
	const int lane = threadIdx.x;
	if ((int)blockIdx.x >= n_jobs) return;
	// the exact classes are small grids of long dependent sweeps next to the wide row-sweep grids of the round: issue priority over them
	if constexpr (EXACT) __builtin_amdgcn_s_setprio(3);
	const int jid = job_ids[blockIdx.x];
	const DpJobDev jb = jobs[jid];
	DpRun R;
	R.qlen = jb.qlen; R.tlen = jb.tlen; R.flag = jb.flag; R.zdrop = jb.zdrop; R.end_bonus = jb.end_bonus;
	EzState &ez = R.ez;
	ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1;
	ez.max = 0; ez.score = ez.mqe = ez.mte = KSW_NEG_INF; ez.zdropped = 0; ez.reach_end = 0;
	if (R.qlen <= 0 || R.tlen <= 0 || jb.skip) {   // skip: tlen*qlen > max_sw_mat => treated as z-dropped by mm_align_pair
		if (lane == 0) {
			mm355_dpres_t o; o.max = 0; o.zdropped = jb.skip? 1 : 0; o.max_q = o.max_t = o.mqe_t = o.mte_q = -1;
			o.mqe = o.mte = o.score = KSW_NEG_INF; o.reach_end = 0; o.n_cigar = -1; o.cigar_off = -1;
			res[jid] = o;
		}
		return;
	}
	const uint8_t *target = tbase + jb.toff;
	R.query = qbase + jb.qoff; R.target = target; R.base = 0;
	R.q = dc.q; R.e = dc.e; R.q2 = dc.q2; R.e2 = dc.e2; R.qe = dc.qe_preswap; R.long_thres = dc.long_thres; R.long_diff = dc.long_diff;
	const int qlen = R.qlen, tlen = R.tlen;
	R.w = jb.w < 0? (tlen > qlen? tlen : qlen) : jb.w;
	R.T = (tlen + 15) / 16 * 16;
	int n_col_ = qlen < tlen? qlen : tlen;
	n_col_ = ((n_col_ < R.w + 1? n_col_ : R.w + 1) + 15) / 16 + 1;
	R.n_col = n_col_ * 16;
	R.p = pbase + jb.p_off;
	R.r_total = qlen + tlen - 1;
	R.full = !EXACT && !(R.flag & EZ_APPROX_DROP) && R.w >= qlen + tlen;
	R.r = 0; R.last_st = R.last_en = -1; R.H0 = 0; R.last_H0_t = 0; R.cells = 0; R.qv = 0;
	DpK K;
	K.nqe = pk8(-R.q - R.e); K.nq2e2 = pk8(-R.q2 - R.e2); K.q = pk8(R.q); K.q2 = pk8(R.q2); K.qe = pk8(R.q + R.e); K.q2e2 = pk8(R.q2 + R.e2);
	K.mch = pk8(dc.sc_mch); K.dmis_v = vreg_const(pk8(dc.sc_mis - dc.sc_mch)); K.N = pk8(dc.sc_N); K.one = 0x00010001u; K.c256 = 0x01000100u; K.m256 = 0xff00ff00u;
	K.two = 0x00020002u; K.three = 0x00030003u; K.four = 0x00040004u; K.f8 = 0x00080008u; K.f16 = 0x00100010u; K.f32 = 0x00200020u; K.f64 = 0x00400040u;
	K.dx1 = K.nqe & 0xffff0000u; K.dx21 = K.nq2e2 & 0xffff0000u;
	DpSt S;
#define DP_INIT(k) { const int t = 128 * k + 2 * lane; \
		S.U##k = S.V##k = S.X##k = S.Y##k = K.nqe; S.X2##k = S.Y2##k = K.nq2e2; S.SC##k = 0; S.QQ##k = 0; S.Hl##k = S.Hh##k = KSW_NEG_INF; \
		S.TQ##k = (NP > k && t < tlen? (uint32_t)target[t] : 0u) | (NP > k && t + 1 < tlen? (uint32_t)target[t + 1] : 0u) << 16; }
	DP_INIT(0) DP_INIT(1) DP_INIT(2) DP_INIT(3) DP_INIT(4) DP_INIT(5) DP_INIT(6) DP_INIT(7)
	{   // ambiguous bases anywhere?  (the common case has none and skips the sc_N selection)
		bool n = false;
		for (int i = lane; i < tlen; i += 64) n |= target[i] > 3;
		for (int i = lane; i < qlen; i += 64) n |= R.query[i] > 3;
		R.any_n = __ballot(n) != 0;
	}
	if (R.flag & EZ_RIGHT) dp_sweep<NP, EXACT, true, WIN>(R, S, K, lane);
	else dp_sweep<NP, EXACT, false, WIN>(R, S, K, lane);
	if (lane == 0) {
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

template <int NP, bool EXACT>
__global__ __launch_bounds__(64) void k_ksw_reg(DpConst dc, const DpJobDev *jobs, const int32_t *job_ids, int n_jobs,
                                                 const uint8_t *qbase, const uint8_t *tbase, uint8_t *pbase, mm355_dpres_t *res, unsigned long long *cells_ctr)
{
	dp_reg_body<NP, EXACT, false>(dc, jobs, job_ids, n_jobs, qbase, tbase, pbase, res, cells_ctr);
}

// Targets of any length whose band is narrow (w <= DP_WIN_MAX_W: the extensions of a read's ends, w = 751 with the ONT preset): the same
// sweep with the eight blocks as a WINDOW of 1024 target positions that follows the band (dp_rebase).  Exact score tracking only.
#define DP_WIN_MAX_W 832                // live positions per anti-diagonal: st - 1 - base < 128, en + 16 (score spill) <= st + 15 + w + 31 -> < 1024
__global__ __launch_bounds__(64) void k_ksw_regw(DpConst dc, const DpJobDev *jobs, const int32_t *job_ids, int n_jobs,
                                                  const uint8_t *qbase, const uint8_t *tbase, uint8_t *pbase, mm355_dpres_t *res, unsigned long long *cells_ctr)
{
	dp_reg_body<8, true, true>(dc, jobs, job_ids, n_jobs, qbase, tbase, pbase, res, cells_ctr);
}
