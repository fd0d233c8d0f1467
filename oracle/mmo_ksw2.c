/* ORACLE (test infrastructure).  Scalar restatement of minimap2 2.26's
 * U:ksw2_extd2_sse.c::ksw_extd2_sse (SSE4.1 code path, 16 int8 lanes) together
 * with U:ksw2.h::ksw_backtrack / ksw_apply_zdrop / ksw_reset_extz, and of
 * U:ksw2_ll_sse.c::ksw_ll_i16 (score + end cell; used by the inversion test).
 * Reached from R:src/lib.rs:482 / :587 via mm_map -> align_regs -> mm_align1 ->
 * mm_align_pair.  map-ont (q=4,e=2,q2=24,e2=1) and map-hifi (6,2,26,1) both take
 * the two-piece kernel; q==q2&&e==e2 (4-tuple scoring=, R:src/lib.rs:375-376)
 * upstream dispatches to ksw_extz2_sse, which computes the same recurrence; here
 * it runs through this function too (documented deviation: band-edge cells only).
 *
 * The SIMD kernel computes whole 16-lane blocks [st/16*16, (en+16)/16*16-1] on
 * every anti-diagonal, so cells outside the band are evaluated on stale inputs
 * and can feed in-band cells at the band edge.  To stay bit-identical this
 * restatement keeps the same memory image (u,v,x,y,x2,y2,s,sf,qr contiguous and
 * zero-initialised like kcalloc) and evaluates the same cell set in the same
 * order with int8 wrap-around arithmetic.
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "mmo.h"

void mmo_ksw_reset_extz(mmo_extz_t *ez)
{
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->max = 0, ez->score = ez->mqe = ez->mte = KSW_NEG_INF;
	ez->n_cigar = 0, ez->zdropped = 0, ez->reach_end = 0;
}

static inline int ksw_apply_zdrop(mmo_extz_t *ez, int is_rot, int32_t H, int a, int b, int zdrop, int8_t e)
{
	int r, t;
	if (is_rot) r = a, t = b;
	else r = a + b, t = a;
	if (H > (int32_t)ez->max) {
		ez->max = H, ez->max_t = t, ez->max_q = r - t;
	} else if (t >= ez->max_t && r - t >= ez->max_q) {
		int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l;
		l = tl > ql? tl - ql : ql - tl;
		if (zdrop >= 0 && (int32_t)ez->max - H > zdrop + l * e) {
			ez->zdropped = 1;
			return 1;
		}
	}
	return 0;
}

static inline uint32_t *ksw_push_cigar(int *n_cigar, int *m_cigar, uint32_t *cigar, uint32_t op, int len)
{
	if (*n_cigar == 0 || op != (cigar[(*n_cigar) - 1]&0xf)) {
		if (*n_cigar == *m_cigar) {
			*m_cigar = *m_cigar? (*m_cigar)<<1 : 4;
			cigar = (uint32_t*)realloc(cigar, (*m_cigar) << 2);
		}
		cigar[(*n_cigar)++] = len<<4 | op;
	} else cigar[(*n_cigar)-1] += len<<4;
	return cigar;
}

/* U:ksw2.h::ksw_backtrack with is_rot=1, min_intron_len=0 */
static void ksw_backtrack(int is_rev, const uint8_t *p, const int *off, const int *off_end, int n_col, int i0, int j0,
                          int *m_cigar_, int *n_cigar_, uint32_t **cigar_)
{
	int n_cigar = 0, m_cigar = *m_cigar_, i = i0, j = j0, r, state = 0;
	uint32_t *cigar = *cigar_, tmp;
	while (i >= 0 && j >= 0) { /* at the beginning of the loop, _state_ tells us which state to check */
		int force_state = -1;
		r = i + j;
		if (i < off[r]) force_state = 2;
		if (off_end && i > off_end[r]) force_state = 1;
		tmp = force_state < 0? p[(size_t)r * n_col + i - off[r]] : 0;
		if (state == 0) state = tmp & 7;
		else if (!(tmp >> (state + 2) & 1)) state = 0;
		if (state == 0) state = tmp & 7;
		if (force_state >= 0) state = force_state;
		if (state == 0) cigar = ksw_push_cigar(&n_cigar, &m_cigar, cigar, 0, 1), --i, --j; /* match */
		else if (state == 1 || state == 3) cigar = ksw_push_cigar(&n_cigar, &m_cigar, cigar, 2, 1), --i; /* deletion */
		else cigar = ksw_push_cigar(&n_cigar, &m_cigar, cigar, 1, 1), --j; /* insertion */
	}
	if (i >= 0) cigar = ksw_push_cigar(&n_cigar, &m_cigar, cigar, 2, i + 1); /* first deletion */
	if (j >= 0) cigar = ksw_push_cigar(&n_cigar, &m_cigar, cigar, 1, j + 1); /* first insertion */
	if (!is_rev)
		for (i = 0; i < n_cigar>>1; ++i) /* reverse CIGAR */
			tmp = cigar[i], cigar[i] = cigar[n_cigar-1-i], cigar[n_cigar-1-i] = tmp;
	*m_cigar_ = m_cigar, *n_cigar_ = n_cigar, *cigar_ = cigar;
}

void mmo_ksw_gen_simple_mat(int m, int8_t *mat, int8_t a, int8_t b, int8_t sc_ambi)
{
	int i, j;
	a = a < 0? -a : a;
	b = b > 0? -b : b;
	sc_ambi = sc_ambi > 0? -sc_ambi : sc_ambi;
	for (i = 0; i < m - 1; ++i) {
		for (j = 0; j < m - 1; ++j)
			mat[i * m + j] = i == j? a : b;
		mat[i * m + m - 1] = sc_ambi;
	}
	for (j = 0; j < m; ++j)
		mat[(m - 1) * m + j] = sc_ambi;
}

#define I8(x) ((int8_t)(x))

void mmo_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, mmo_extz_t *ez)
{
	int r, t, qe = q + e, n_col_, *off = 0, *off_end = 0, tlen_, qlen_, last_st, last_en, wl, wr, max_sc, min_sc, long_thres, long_diff;
	int with_cigar = !(flag&KSW_EZ_SCORE_ONLY), approx_max = !!(flag&KSW_EZ_APPROX_MAX);
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	uint8_t *qr, *sf, *mem, *p = 0;
	int8_t *u, *v, *x, *y, *x2, *y2, *s;
	int8_t sc_mch, sc_mis, sc_N, m1, qe8, qe28;
	size_t T;

	mmo_ksw_reset_extz(ez);
	if (m <= 1 || qlen <= 0 || tlen <= 0) return;

	if (q2 + e2 < q + e) t = q, q = q2, q2 = t, t = e, e = e2, e2 = t; /* make sure q+e no larger than q2+e2 */
	/* NB: upstream initialises `qe` before the swap and never refreshes it (only matters if q+e > q2+e2) */
	qe8 = I8(q + e), qe28 = I8(q2 + e2);
	sc_mch = mat[0], sc_mis = mat[1];
	sc_N = mat[m*m-1] == 0? I8(-e2) : mat[m*m-1];
	m1 = m - 1;

	if (w < 0) w = tlen > qlen? tlen : qlen;
	wl = wr = w;
	tlen_ = (tlen + 15) / 16;
	n_col_ = qlen < tlen? qlen : tlen;
	n_col_ = ((n_col_ < w + 1? n_col_ : w + 1) + 15) / 16 + 1;
	qlen_ = (qlen + 15) / 16;
	for (t = 1, max_sc = mat[0], min_sc = mat[1]; t < m * m; ++t) {
		max_sc = max_sc > mat[t]? max_sc : mat[t];
		min_sc = min_sc < mat[t]? min_sc : mat[t];
	}
	if (-min_sc > 2 * (q + e)) return; /* otherwise, we won't see any mismatches */

	long_thres = e != e2? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e)
		++long_thres;
	long_diff = long_thres * (e - e2) - (q2 - q) - e2;

	T = (size_t)tlen_ * 16;
	mem = (uint8_t*)calloc((size_t)tlen_ * 8 + qlen_ + 1 + 1, 16);
	u = (int8_t*)mem; v = u + T, x = v + T, y = x + T, x2 = y + T, y2 = x2 + T;
	s = y2 + T, sf = (uint8_t*)(s + T), qr = sf + T;
	memset(u,  -q  - e,  T);
	memset(v,  -q  - e,  T);
	memset(x,  -q  - e,  T);
	memset(y,  -q  - e,  T);
	memset(x2, -q2 - e2, T);
	memset(y2, -q2 - e2, T);
	if (!approx_max) {
		H = (int32_t*)malloc(T * 4);
		for (t = 0; t < (int)T; ++t) H[t] = KSW_NEG_INF;
	}
	if (with_cigar) {
		p = (uint8_t*)malloc(((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16);
		off = (int*)malloc((qlen + tlen - 1) * sizeof(int) * 2);
		off_end = off + qlen + tlen - 1;
	}

	for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
	memcpy(sf, target, tlen);

	for (r = 0, last_st = last_en = -1; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1, st0, en0;
		int8_t x1, x21, v1;
		uint8_t *qrr = qr + (qlen - 1 - r);
		/* find the boundaries */
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r-wr+1)>>1) st = (r-wr+1)>>1; /* take the ceil */
		if (en > (r+wl)>>1) en = (r+wl)>>1; /* take the floor */
		if (st > en) {
			ez->zdropped = 1;
			break;
		}
		st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		/* set boundary conditions */
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) {
				x1 = x[st - 1], x21 = x2[st - 1], v1 = v[st - 1]; /* (r-1,s-1) calculated in the last round */
			} else {
				x1 = -q - e, x21 = -q2 - e2;
				v1 = -q - e;
			}
		} else {
			x1 = -q - e, x21 = -q2 - e2;
			v1 = r == 0? -q - e : r < long_thres? -e : r == long_thres? long_diff : -e2;
		}
		if (en >= r) {
			y[r] = -q - e, y2[r] = -q2 - e2;
			u[r] = r == 0? -q - e : r < long_thres? -e : r == long_thres? long_diff : -e2;
		}
		/* loop fission: set scores first (non-GENERIC_SC path; 16-byte unaligned blocks from st0) */
		for (t = st0; t <= en0; t += 16) {
			int l;
			for (l = 0; l < 16; ++l) {
				uint8_t sq = sf[t + l], stq = qrr[t + l];
				int8_t tmp = sq == stq? sc_mch : sc_mis;
				if (sq == (uint8_t)m1 || stq == (uint8_t)m1) tmp = sc_N;
				s[t + l] = tmp; /* may spill past s[] into sf[0..14]: same as the SIMD store, never re-read */
			}
		}
		/* core loop */
		assert(en / 16 - st / 16 + 1 <= n_col_);
		if (with_cigar) {
			uint8_t *pr = p + (size_t)r * n_col_ * 16 - st;
			int8_t xp = x1, x2p = x21, vp = v1;
			off[r] = st, off_end[r] = en;
			for (t = st; t <= en; ++t) {
				int8_t z, a, b, a2, b2, xt1, x2t1, vt1, ut, tmp;
				uint8_t d;
				z = s[t];
				xt1 = xp, xp = x[t];
				vt1 = vp, vp = v[t];
				x2t1 = x2p, x2p = x2[t];
				a = I8(xt1 + vt1);
				ut = u[t];
				b = I8(y[t] + ut);
				a2 = I8(x2t1 + vt1);
				b2 = I8(y2[t] + ut);
				if (!(flag & KSW_EZ_RIGHT)) { /* gap left-alignment */
					d = a > z? 1 : 0;
					z = z > a? z : a;
					d = b > z? 2 : d;
					z = z > b? z : b;
					d = a2 > z? 3 : d;
					z = z > a2? z : a2;
					d = b2 > z? 4 : d;
					z = z > b2? z : b2;
					z = z < sc_mch? z : sc_mch;
					u[t] = I8(z - vt1);
					v[t] = I8(z - ut);
					tmp = I8(z - q);
					a = I8(a - tmp), b = I8(b - tmp);
					tmp = I8(z - q2);
					a2 = I8(a2 - tmp), b2 = I8(b2 - tmp);
					x[t]  = I8((a  > 0? a  : 0) - qe8);  if (a  > 0) d |= 0x08;
					y[t]  = I8((b  > 0? b  : 0) - qe8);  if (b  > 0) d |= 0x10;
					x2[t] = I8((a2 > 0? a2 : 0) - qe28); if (a2 > 0) d |= 0x20;
					y2[t] = I8((b2 > 0? b2 : 0) - qe28); if (b2 > 0) d |= 0x40;
				} else { /* gap right-alignment */
					d = z > a? 0 : 1;
					z = z > a? z : a;
					d = z > b? d : 2;
					z = z > b? z : b;
					d = z > a2? d : 3;
					z = z > a2? z : a2;
					d = z > b2? d : 4;
					z = z > b2? z : b2;
					z = z < sc_mch? z : sc_mch;
					u[t] = I8(z - vt1);
					v[t] = I8(z - ut);
					tmp = I8(z - q);
					a = I8(a - tmp), b = I8(b - tmp);
					tmp = I8(z - q2);
					a2 = I8(a2 - tmp), b2 = I8(b2 - tmp);
					x[t]  = I8((0 > a?  0 : a)  - qe8);  if (!(0 > a))  d |= 0x08;
					y[t]  = I8((0 > b?  0 : b)  - qe8);  if (!(0 > b))  d |= 0x10;
					x2[t] = I8((0 > a2? 0 : a2) - qe28); if (!(0 > a2)) d |= 0x20;
					y2[t] = I8((0 > b2? 0 : b2) - qe28); if (!(0 > b2)) d |= 0x40;
				}
				pr[t] = d;
			}
		} else { /* score only */
			int8_t xp = x1, x2p = x21, vp = v1;
			for (t = st; t <= en; ++t) {
				int8_t z, a, b, a2, b2, xt1, x2t1, vt1, ut, tmp;
				z = s[t];
				xt1 = xp, xp = x[t];
				vt1 = vp, vp = v[t];
				x2t1 = x2p, x2p = x2[t];
				a = I8(xt1 + vt1);
				ut = u[t];
				b = I8(y[t] + ut);
				a2 = I8(x2t1 + vt1);
				b2 = I8(y2[t] + ut);
				z = z > a? z : a;
				z = z > b? z : b;
				z = z > a2? z : a2;
				z = z > b2? z : b2;
				z = z < sc_mch? z : sc_mch;
				u[t] = I8(z - vt1);
				v[t] = I8(z - ut);
				tmp = I8(z - q);
				a = I8(a - tmp), b = I8(b - tmp);
				tmp = I8(z - q2);
				a2 = I8(a2 - tmp), b2 = I8(b2 - tmp);
				x[t]  = I8((a  > 0? a  : 0) - qe8);
				y[t]  = I8((b  > 0? b  : 0) - qe8);
				x2[t] = I8((a2 > 0? a2 : 0) - qe28);
				y2[t] = I8((b2 > 0? b2 : 0) - qe28);
			}
		}
		mmo_stats.dp_cells += en0 - st0 + 1;
		if (!approx_max) { /* find the exact max with a 32-bit score array */
			int32_t max_H, max_t;
			if (r > 0) {
				int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4, i;
				max_H = H[en0] = en0 > 0? H[en0-1] + u[en0] : H[en0] + v[en0]; /* special casing the last element */
				max_t = en0;
				for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
				for (t = st0; t < en1; t += 4) { /* 4 int32 lanes, strict > */
					for (i = 0; i < 4; ++i) {
						H[t + i] += (int32_t)v[t + i];
						if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t;
					}
				}
				for (i = 0; i < 4; ++i)
					if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
				for (; t < en0; ++t) { /* for the rest of values that haven't been computed with SSE */
					H[t] += (int32_t)v[t];
					if (H[t] > max_H)
						max_H = H[t], max_t = t;
				}
			} else H[0] = v[0] - qe, max_H = H[0], max_t = 0; /* special casing r==0 */
			/* update ez */
			if (en0 == tlen - 1 && H[en0] > ez->mte)
				ez->mte = H[en0], ez->mte_q = r - en;
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe)
				ez->mqe = H[st0], ez->mqe_t = st0;
			if (ksw_apply_zdrop(ez, 1, max_H, r, max_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1)
				ez->score = H[tlen - 1];
		} else { /* find approximate max; Z-drop might be inaccurate, too. */
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = v[last_H0_t];
					int32_t d1 = u[last_H0_t + 1];
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += v[last_H0_t];
				} else {
					++last_H0_t, H0 += u[last_H0_t];
				}
			} else H0 = v[0] - qe, last_H0_t = 0;
			if ((flag & KSW_EZ_APPROX_DROP) && ksw_apply_zdrop(ez, 1, H0, r, last_H0_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1)
				ez->score = H0;
		}
		last_st = st, last_en = en;
	}
	free(mem);
	if (!approx_max) free(H);
	if (with_cigar) { /* backtrack */
		int rev_cigar = !!(flag & KSW_EZ_REV_CIGAR);
		if (!ez->zdropped && !(flag&KSW_EZ_EXTZ_ONLY)) {
			ksw_backtrack(rev_cigar, p, off, off_end, n_col_*16, tlen-1, qlen-1, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
		} else if (!ez->zdropped && (flag&KSW_EZ_EXTZ_ONLY) && ez->mqe + end_bonus > (int)ez->max) {
			ez->reach_end = 1;
			ksw_backtrack(rev_cigar, p, off, off_end, n_col_*16, ez->mqe_t, qlen-1, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
		} else if (ez->max_t >= 0 && ez->max_q >= 0) {
			ksw_backtrack(rev_cigar, p, off, off_end, n_col_*16, ez->max_t, ez->max_q, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
		}
		free(p); free(off);
	}
	++mmo_stats.n_dp_calls;
}

/* U:ksw2_ll_sse.c::ksw_ll_qinit(size=2) + ksw_ll_i16: local SW (affine) over the
 * query padded to a multiple of 8 with zero-score columns; returns the best score,
 * *te = LAST target row whose row maximum is >= the running best, *qe = the query
 * index mapped from the LAST striped slot (slot = seg*8+lane, pos = seg+lane*slen)
 * of that row holding the best score.  Status: recalled, unverified;
 * only the score feeds mm_test_zdrop, positions feed mm_align1_inv. */
int mmo_ksw_ll(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat, int gapo, int gape, int *qe, int *te)
{
	int slen = (qlen + 7) / 8, qlen8 = slen * 8, i, j, gmax = 0;
	int gapoe = gapo + gape;
	int32_t *H0, *H1, *E, *Hmax, *tmpp;
	*qe = *te = -1;
	if (qlen <= 0) return 0;
	H0 = (int32_t*)calloc(qlen8 + 1, 4); H1 = (int32_t*)calloc(qlen8 + 1, 4);
	E = (int32_t*)calloc(qlen8 + 1, 4); Hmax = (int32_t*)calloc(qlen8 + 1, 4);
	for (i = 0; i < tlen; ++i) {
		const int8_t *ma = mat + target[i] * m;
		int32_t f = 0, imax = 0, hdiag = 0;
		for (j = 0; j < qlen8; ++j) {
			int32_t sc = j < qlen? ma[query[j]] : 0;
			int32_t h = hdiag + sc, e = E[j], t;
			hdiag = H0[j];
			h = h > e? h : e;
			h = h > f? h : f;
			if (h < 0) h = 0;
			if (h > 32767) h = 32767;
			H1[j] = h;
			imax = imax > h? imax : h;
			t = h - gapoe; if (t < 0) t = 0;
			e -= gape; if (e < 0) e = 0;
			E[j] = e > t? e : t;
			f -= gape; if (f < 0) f = 0;
			f = f > t? f : t;
		}
		if (imax >= gmax) {
			gmax = imax; *te = i;
			memcpy(Hmax, H1, qlen8 * 4);
		}
		tmpp = H1; H1 = H0; H0 = tmpp;
	}
	for (i = 0; i < qlen8; ++i) { /* striped memory order: slot i <-> query pos i/8 + i%8*slen */
		int pos = i / 8 + i % 8 * slen;
		if (Hmax[pos] == gmax) *qe = pos;
	}
	free(H0); free(H1); free(E); free(Hmax);
	return gmax;
}
