/* ORACLE (test infrastructure).  Restates U:align.c of minimap2 2.26 for the
 * long-read genomic path: mm_align_skeleton, mm_align1, mm_align_pair,
 * mm_fix_bad_ends, mm_filter_bad_seeds(_alt), mm_adjust_minier, mm_test_zdrop,
 * mm_align1_inv, mm_update_extra, mm_fix_cigar, mm_append_cigar, mm_insert_reg.
 * Reached from R:src/lib.rs:482 / :587 via mm_map -> align_regs (MM_F_CIGAR is
 * forced on at R:src/lib.rs:339).  Splice / short-read / qstrand branches
 * are out of scope and omitted; the HPC branch (map-pb / ava-pb)
 * is mm_adjust_minier's only.
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "mmo.h"
#define mm_seq4_get(s, i)    ((s)[(i)>>3] >> (((i)&7)<<2) & 0xf)

#define kroundup32(x) (--(x), (x)|=(x)>>1, (x)|=(x)>>2, (x)|=(x)>>4, (x)|=(x)>>8, (x)|=(x)>>16, ++(x))

static inline float mg_log2(float x)
{
	union { float f; uint32_t i; } z = { x };
	float log_2 = ((z.i >> 23) & 255) - 128;
	z.i &= ~(255 << 23);
	z.i += 127 << 23;
	log_2 += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return log_2;
}

static inline void mm_seq_rev(uint32_t len, uint8_t *seq)
{
	uint32_t i;
	uint8_t t;
	for (i = 0; i < len>>1; ++i)
		t = seq[i], seq[i] = seq[len - 1 - i], seq[len - 1 - i] = t;
}

static inline void update_max_zdrop(int32_t score, int i, int j, int32_t *max, int *max_i, int *max_j, int e, int *max_zdrop, int pos[2][2])
{
	if (score < *max) {
		int li = i - *max_i;
		int lj = j - *max_j;
		int diff = li > lj? li - lj : lj - li;
		int z = *max - score - diff * e;
		if (z > *max_zdrop) {
			*max_zdrop = z;
			pos[0][0] = *max_i, pos[0][1] = i;
			pos[1][0] = *max_j, pos[1][1] = j;
		}
	} else *max = score, *max_i = i, *max_j = j;
}

static int mm_test_zdrop(const mmo_mapopt_t *opt, const uint8_t *qseq, const uint8_t *tseq, uint32_t n_cigar, uint32_t *cigar, const int8_t *mat)
{
	uint32_t k;
	int32_t score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
	int pos[2][2] = {{-1, -1}, {-1, -1}}, q_len, t_len;

	/* find the score and the region where score drops most along diagonal */
	for (k = 0, score = 0; k < n_cigar; ++k) {
		uint32_t l, op = cigar[k]&0xf, len = cigar[k]>>4;
		if (op == MM_CIGAR_MATCH) {
			for (l = 0; l < len; ++l) {
				score += mat[tseq[i + l] * 5 + qseq[j + l]];
				update_max_zdrop(score, i+l, j+l, &max, &max_i, &max_j, opt->e, &max_zdrop, pos);
			}
			i += len, j += len;
		} else if (op == MM_CIGAR_INS || op == MM_CIGAR_DEL || op == MM_CIGAR_N_SKIP) {
			score -= opt->q + opt->e * len;
			if (op == MM_CIGAR_INS) j += len;
			else i += len;
			update_max_zdrop(score, i, j, &max, &max_i, &max_j, opt->e, &max_zdrop, pos);
		}
	}

	/* test if there is an inversion in the most dropped region */
	q_len = pos[1][1] - pos[1][0], t_len = pos[0][1] - pos[0][0];
	if (!(opt->flag&(MM_F_SPLICE|MM_F_SR|MM_F_FOR_ONLY|MM_F_REV_ONLY)) && max_zdrop > opt->zdrop_inv && q_len < opt->max_gap && t_len < opt->max_gap) {
		uint8_t *qseq2;
		int q_off, t_off;
		qseq2 = (uint8_t*)malloc(q_len > 0? q_len : 1);
		for (i = 0; i < q_len; ++i) {
			int c = qseq[pos[1][1] - i - 1];
			qseq2[i] = c >= 4? 4 : 3 - c;
		}
		score = mmo_ksw_ll(q_len, qseq2, t_len, tseq + pos[0][0], 5, mat, opt->q, opt->e, &q_off, &t_off);
		free(qseq2);
		if (score >= opt->min_chain_score * opt->a && score >= opt->min_dp_max)
			return 2; /* there is a potential inversion */
	}
	return max_zdrop > opt->zdrop? 1 : 0;
}

/* test hook: mm_test_zdrop on its own (tests/test_zdrop_bound_model.py) */
int mmo_test_zdrop(const mmo_mapopt_t *opt, const uint8_t *qseq, const uint8_t *tseq, uint32_t n_cigar, uint32_t *cigar, const int8_t *mat)
{
	return mm_test_zdrop(opt, qseq, tseq, n_cigar, cigar, mat);
}

static void mm_fix_cigar(mmo_reg1_t *r, const uint8_t *qseq, const uint8_t *tseq, int *qshift, int *tshift)
{
	mmo_extra_t *p = r->p;
	int32_t toff = 0, qoff = 0, to_shrink = 0;
	uint32_t k;
	*qshift = *tshift = 0;
	if (p->n_cigar <= 1) return;
	for (k = 0; k < p->n_cigar; ++k) { /* indel left alignment */
		uint32_t op = p->cigar[k]&0xf, len = p->cigar[k]>>4;
		if (len == 0) to_shrink = 1;
		if (op == MM_CIGAR_MATCH) {
			toff += len, qoff += len;
		} else if (op == MM_CIGAR_INS || op == MM_CIGAR_DEL) {
			if (k > 0 && k < p->n_cigar - 1 && (p->cigar[k-1]&0xf) == 0 && (p->cigar[k+1]&0xf) == 0) {
				int l, prev_len = p->cigar[k-1] >> 4;
				if (op == MM_CIGAR_INS) {
					for (l = 0; l < prev_len; ++l)
						if (qseq[qoff - 1 - l] != qseq[qoff + len - 1 - l])
							break;
				} else {
					for (l = 0; l < prev_len; ++l)
						if (tseq[toff - 1 - l] != tseq[toff + len - 1 - l])
							break;
				}
				if (l > 0)
					p->cigar[k-1] -= l<<4, p->cigar[k+1] += l<<4, qoff -= l, toff -= l;
				if (l == prev_len) to_shrink = 1;
			}
			if (op == MM_CIGAR_INS) qoff += len;
			else toff += len;
		} else if (op == MM_CIGAR_N_SKIP) {
			toff += len;
		}
	}
	assert(qoff == r->qe - r->qs && toff == r->re - r->rs);
	for (k = 0; k + 2 < p->n_cigar; ++k) { /* fix CIGAR like 5I6D7I */
		if ((p->cigar[k]&0xf) > 0 && (p->cigar[k]&0xf) + (p->cigar[k+1]&0xf) == 3) {
			uint32_t l, s[3] = {0,0,0};
			for (l = k; l < p->n_cigar; ++l) { /* count number of adjacent I and D */
				uint32_t op = p->cigar[l]&0xf;
				if (op == MM_CIGAR_INS || op == MM_CIGAR_DEL || p->cigar[l]>>4 == 0)
					s[op] += p->cigar[l] >> 4;
				else break;
			}
			if (s[1] > 0 && s[2] > 0 && l - k > 2) { /* turn to a single I and a single D */
				p->cigar[k]   = s[1]<<4|MM_CIGAR_INS;
				p->cigar[k+1] = s[2]<<4|MM_CIGAR_DEL;
				for (k += 2; k < l; ++k)
					p->cigar[k] &= 0xf;
				to_shrink = 1;
			}
			k = l;
		}
	}
	if (to_shrink) { /* squeeze out zero-length operations */
		int32_t l = 0;
		for (k = 0; k < p->n_cigar; ++k)
			if (p->cigar[k]>>4 != 0)
				p->cigar[l++] = p->cigar[k];
		p->n_cigar = l;
		for (k = l = 0; k < p->n_cigar; ++k) /* merge two adjacent operations if they are the same */
			if (k == p->n_cigar - 1 || (p->cigar[k]&0xf) != (p->cigar[k+1]&0xf))
				p->cigar[l++] = p->cigar[k];
			else p->cigar[k+1] += p->cigar[k]>>4<<4; /* add length to the next CIGAR operator */
		p->n_cigar = l;
	}
	if ((p->cigar[0]&0xf) == MM_CIGAR_INS || (p->cigar[0]&0xf) == MM_CIGAR_DEL) { /* get rid of leading I or D */
		int32_t l = p->cigar[0] >> 4;
		if ((p->cigar[0]&0xf) == MM_CIGAR_INS) {
			if (r->rev) r->qe -= l;
			else r->qs += l;
			*qshift = l;
		} else r->rs += l, *tshift = l;
		--p->n_cigar;
		memmove(p->cigar, p->cigar + 1, p->n_cigar * 4);
	}
}

/* U:align.c::mm_update_cigar_eqx (MM_F_EQX): every M becomes alternating runs of '=' (7) and 'X' (8); N vs N counts as '='.
 * Upstream rewrites in place when every M is a single '=' run and rebuilds the array otherwise -- the resulting CIGAR is the same. */
static void mm_update_cigar_eqx(mmo_reg1_t *r, const uint8_t *qseq, const uint8_t *tseq)
{
	uint32_t k, l, n_new = 0, toff = 0, qoff = 0, *nc;
	mmo_extra_t *p = r->p, *np;
	uint32_t capacity;
	if (p == 0) return;
	nc = (uint32_t*)malloc(((size_t)(r->qe - r->qs) + p->n_cigar + 1) * 4); /* at most one op per query base + the non-M ops */
	for (k = 0; k < p->n_cigar; ++k) {
		uint32_t op = p->cigar[k]&0xf, len = p->cigar[k]>>4;
		if (op == MM_CIGAR_MATCH) {
			while (len > 0) {
				for (l = 0; l < len && qseq[qoff + l] == tseq[toff + l]; ++l) {}
				if (l > 0) { nc[n_new++] = l << 4 | 7; len -= l; toff += l; qoff += l; }
				for (l = 0; l < len && qseq[qoff + l] != tseq[toff + l]; ++l) {}
				if (l > 0) { nc[n_new++] = l << 4 | 8; len -= l; toff += l; qoff += l; }
			}
		} else {
			if (op == MM_CIGAR_INS) qoff += len;
			else if (op == MM_CIGAR_DEL || op == MM_CIGAR_N_SKIP) toff += len;
			nc[n_new++] = p->cigar[k];
		}
	}
	capacity = n_new + sizeof(mmo_extra_t)/4;
	kroundup32(capacity);
	np = (mmo_extra_t*)calloc(capacity, 4);
	memcpy(np, p, sizeof(mmo_extra_t));
	np->capacity = capacity; np->n_cigar = n_new;
	memcpy(np->cigar, nc, (size_t)n_new * 4);
	free(nc); free(p);
	r->p = np;
}

/* the per-base walk of U:align.c::mm_update_extra (after mm_fix_cigar): mlen, blen, n_ambi and dp_max of a CIGAR over given code strings.
 * Exported: mm_update_extra below runs it, and the parity test of the device walk (k_extra) calls it as its stage oracle. */
void mmo_extra_walk(const uint32_t *cigar, int n_cigar, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int8_t q, int8_t e, int log_gap,
                    int32_t *mlen_, int32_t *blen_, int32_t *n_ambi_, int32_t *dp_max_, int32_t *q_len, int32_t *t_len)
{
	int32_t k, toff = 0, qoff = 0, mlen = 0, blen = 0, n_ambi_tot = 0;
	uint32_t l;
	double s = 0.0, max = 0.0;
	for (k = 0; k < n_cigar; ++k) {
		uint32_t op = cigar[k]&0xf, len = cigar[k]>>4;
		if (op == MM_CIGAR_MATCH) {
			int n_ambi = 0, n_diff = 0;
			for (l = 0; l < len; ++l) {
				int cq = qseq[qoff + l], ct = tseq[toff + l];
				if (ct > 3 || cq > 3) ++n_ambi;
				else if (ct != cq) ++n_diff;
				s += mat[ct * 5 + cq];
				if (s < 0) s = 0;
				else max = max > s? max : s;
			}
			blen += len - n_ambi, mlen += len - (n_ambi + n_diff), n_ambi_tot += n_ambi;
			toff += len, qoff += len;
		} else if (op == MM_CIGAR_INS) {
			int n_ambi = 0;
			for (l = 0; l < len; ++l)
				if (qseq[qoff + l] > 3) ++n_ambi;
			blen += len - n_ambi, n_ambi_tot += n_ambi;
			if (log_gap) s -= q + (double)e * mg_log2(1.0 + len);
			else s -= q + e;
			if (s < 0) s = 0;
			qoff += len;
		} else if (op == MM_CIGAR_DEL) {
			int n_ambi = 0;
			for (l = 0; l < len; ++l)
				if (tseq[toff + l] > 3) ++n_ambi;
			blen += len - n_ambi, n_ambi_tot += n_ambi;
			if (log_gap) s -= q + (double)e * mg_log2(1.0 + len);
			else s -= q + e;
			if (s < 0) s = 0;
			toff += len;
		} else if (op == MM_CIGAR_N_SKIP) {
			toff += len;
		}
	}
	*mlen_ = mlen; *blen_ = blen; *n_ambi_ = n_ambi_tot; *dp_max_ = (int32_t)(max + .499);
	if (q_len) *q_len = qoff;
	if (t_len) *t_len = toff;
}

static void mm_update_extra(mmo_reg1_t *r, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int8_t q, int8_t e, int is_eqx, int log_gap)
{
	int32_t qshift, tshift, toff = 0, qoff = 0, n_ambi = 0;
	mmo_extra_t *p = r->p;
	if (p == 0) return;
	mm_fix_cigar(r, qseq, tseq, &qshift, &tshift);
	qseq += qshift, tseq += tshift; /* qseq and tseq may be shifted due to the removal of leading I/D */
	mmo_extra_walk(p->cigar, (int)p->n_cigar, qseq, tseq, mat, q, e, log_gap, &r->mlen, &r->blen, &n_ambi, &p->dp_max, &qoff, &toff);
	p->n_ambi += n_ambi;
	assert(qoff == r->qe - r->qs && toff == r->re - r->rs);
	if (is_eqx) mm_update_cigar_eqx(r, qseq, tseq); /* here: the shifts of qseq/tseq are local to this function */
}

static void mm_append_cigar(mmo_reg1_t *r, uint32_t n_cigar, uint32_t *cigar)
{
	mmo_extra_t *p;
	if (n_cigar == 0) return;
	if (r->p == 0) {
		uint32_t capacity = n_cigar + sizeof(mmo_extra_t)/4;
		kroundup32(capacity);
		r->p = (mmo_extra_t*)calloc(capacity, 4);
		r->p->capacity = capacity;
	} else if (r->p->n_cigar + n_cigar + sizeof(mmo_extra_t)/4 > r->p->capacity) {
		r->p->capacity = r->p->n_cigar + n_cigar + sizeof(mmo_extra_t)/4;
		kroundup32(r->p->capacity);
		r->p = (mmo_extra_t*)realloc(r->p, r->p->capacity * 4);
	}
	p = r->p;
	if (p->n_cigar > 0 && (p->cigar[p->n_cigar-1]&0xf) == (cigar[0]&0xf)) { /* same CIGAR op at the boundary */
		p->cigar[p->n_cigar-1] += (cigar[0]>>4) << 4;
		if (n_cigar > 1) memcpy(p->cigar + p->n_cigar, cigar + 1, (n_cigar - 1) * 4);
		p->n_cigar += n_cigar - 1;
	} else {
		memcpy(p->cigar + p->n_cigar, cigar, n_cigar * 4);
		p->n_cigar += n_cigar;
	}
}

static void mm_align_pair(const mmo_mapopt_t *opt, int qlen, const uint8_t *qseq, int tlen, const uint8_t *tseq, const int8_t *mat, int w, int end_bonus, int zdrop, int flag, mmo_extz_t *ez)
{
	if (opt->max_sw_mat > 0 && (int64_t)tlen * qlen > opt->max_sw_mat) {
		mmo_ksw_reset_extz(ez);
		ez->zdropped = 1;
	} else /* q==q2&&e==e2 -> ksw_extz2_sse upstream; same recurrence, see mmo_ksw2.c header */
		mmo_ksw_extd2(qlen, qseq, tlen, tseq, 5, mat, opt->q, opt->e, opt->q2, opt->e2, w, zdrop, end_bonus, flag, ez);
}

/* U:align.c::mm_get_hplen_back: length of the homopolymer run of the reference that ends at x (inside contig rid) */
static int mm_get_hplen_back(const mmo_idx_t *mi, uint32_t rid, uint32_t x)
{
	int64_t i, off0 = mi->seq[rid].offset, off = off0 + x;
	int c = mm_seq4_get(mi->S, off);
	for (i = off - 1; i >= off0; --i)
		if (mm_seq4_get(mi->S, i) != c) break;
	return (int)(off - i);
}

/* U:align.c::mm_adjust_minier.  HPC index (map-pb / ava-pb): a seed ends on the LAST base of a homopolymer run, on both
 * sequences; the alignment is cut at the FIRST base of that run instead of the middle of the k-mer */
static inline void mm_adjust_minier(const mmo_idx_t *mi, uint8_t *const qseq0[2], mm128_t *a, int32_t *r, int32_t *q)
{
	if (mi->flag & MM_I_HPC) {
		const uint8_t *qseq = qseq0[a->x>>63];
		int i, c;
		*q = (int32_t)a->y;
		for (i = *q - 1, c = qseq[*q]; i > 0; --i)
			if (qseq[i] != c) break;
		*q = i + 1;
		c = mm_get_hplen_back(mi, a->x<<1>>33, (int32_t)a->x);
		*r = (int32_t)a->x + 1 - c;
	} else {
		*r = (int32_t)a->x - (mi->k>>1);
		*q = (int32_t)a->y - (mi->k>>1);
	}
}

static int *collect_long_gaps(int as1, int cnt1, mm128_t *a, int min_gap, int *n_)
{
	int i, n, *K;
	*n_ = 0;
	for (i = 1, n = 0; i < cnt1; ++i) { /* count the number of gaps longer than min_gap */
		int gap = ((int32_t)a[as1 + i].y - a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - a[as1 + i - 1].x);
		if (gap < -min_gap || gap > min_gap) ++n;
	}
	if (n <= 1) return 0;
	K = (int*)malloc(n * sizeof(int));
	for (i = 1, n = 0; i < cnt1; ++i) { /* store the positions of long gaps */
		int gap = ((int32_t)a[as1 + i].y - a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - a[as1 + i - 1].x);
		if (gap < -min_gap || gap > min_gap)
			K[n++] = i;
	}
	*n_ = n;
	return K;
}

static void mm_filter_bad_seeds(int as1, int cnt1, mm128_t *a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt)
{
	int max_st, max_en, n, i, k, max, *K;
	K = collect_long_gaps(as1, cnt1, a, min_gap, &n);
	if (K == 0) return;
	max = 0, max_st = max_en = -1;
	for (k = 0;; ++k) { /* traverse long gaps */
		int gap, l, n_ins = 0, n_del = 0, qs, rs, max_diff = 0, max_diff_l = -1;
		if (k == n || k >= max_en) {
			if (max_en > 0)
				for (i = K[max_st]; i < K[max_en]; ++i)
					a[as1 + i].y |= MM_SEED_IGNORE;
			max = 0, max_st = max_en = -1;
			if (k == n) break;
		}
		i = K[k];
		gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - (int32_t)(a[as1 + i].x - a[as1 + i - 1].x);
		if (gap > 0) n_ins += gap;
		else n_del += -gap;
		qs = (int32_t)a[as1 + i - 1].y;
		rs = (int32_t)a[as1 + i - 1].x;
		for (l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
			int j = K[l], diff;
			if ((int32_t)a[as1 + j].y - qs > max_ext_len || (int32_t)a[as1 + j].x - rs > max_ext_len) break;
			gap = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
			if (gap > 0) n_ins += gap;
			else n_del += -gap;
			diff = n_ins + n_del - abs(n_ins - n_del);
			if (max_diff < diff)
				max_diff = diff, max_diff_l = l;
		}
		if (max_diff > diff_thres && max_diff > max)
			max = max_diff, max_st = k, max_en = max_diff_l;
	}
	free(K);
}

static void mm_filter_bad_seeds_alt(int as1, int cnt1, mm128_t *a, int min_gap, int max_ext)
{
	int n, k, *K;
	K = collect_long_gaps(as1, cnt1, a, min_gap, &n);
	if (K == 0) return;
	for (k = 0; k < n;) {
		int i = K[k], l;
		int gap1 = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - (int32_t)a[as1 + i - 1].x);
		int re1 = (int32_t)a[as1 + i].x;
		int qe1 = (int32_t)a[as1 + i].y;
		gap1 = gap1 > 0? gap1 : -gap1;
		for (l = k + 1; l < n; ++l) {
			int j = K[l], gap2, q_span_pre, rs2, qs2, m;
			if ((int32_t)a[as1 + j].y - qe1 > max_ext || (int32_t)a[as1 + j].x - re1 > max_ext) break;
			gap2 = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
			q_span_pre = a[as1 + j - 1].y >> 32 & 0xff;
			rs2 = (int32_t)a[as1 + j - 1].x + q_span_pre;
			qs2 = (int32_t)a[as1 + j - 1].y + q_span_pre;
			m = rs2 - re1 < qs2 - qe1? rs2 - re1 : qs2 - qe1;
			gap2 = gap2 > 0? gap2 : -gap2;
			if (m > gap1 + gap2) break;
			re1 = (int32_t)a[as1 + j].x;
			qe1 = (int32_t)a[as1 + j].y;
			gap1 = gap2;
		}
		if (l > k + 1) {
			int j, end = K[l - 1];
			for (j = K[k]; j < end; ++j)
				a[as1 + j].y |= MM_SEED_IGNORE;
			a[as1 + end].y |= MM_SEED_LONG_JOIN;
		}
		k = l;
	}
	free(K);
}

static void mm_fix_bad_ends(const mmo_reg1_t *r, const mm128_t *a, int bw, int min_match, int32_t *as, int32_t *cnt)
{
	int32_t i, l, m;
	*as = r->as, *cnt = r->cnt;
	if (r->cnt < 3) return;
	m = l = a[r->as].y >> 32 & 0xff;
	for (i = r->as + 1; i < r->as + r->cnt - 1; ++i) {
		int32_t lq, lr, min, max;
		int32_t q_span = a[i].y >> 32 & 0xff;
		if (a[i].y & MM_SEED_LONG_JOIN) break;
		lr = (int32_t)a[i].x - (int32_t)a[i-1].x;
		lq = (int32_t)a[i].y - (int32_t)a[i-1].y;
		min = lr < lq? lr : lq;
		max = lr > lq? lr : lq;
		if (max - min > l >> 1) *as = i;
		l += min;
		m += min < q_span? min : q_span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->mlen >> 1) break;
	}
	*cnt = r->as + r->cnt - *as;
	m = l = a[r->as + r->cnt - 1].y >> 32 & 0xff;
	for (i = r->as + r->cnt - 2; i > *as; --i) {
		int32_t lq, lr, min, max;
		int32_t q_span = a[i+1].y >> 32 & 0xff;
		if (a[i+1].y & MM_SEED_LONG_JOIN) break;
		lr = (int32_t)a[i+1].x - (int32_t)a[i].x;
		lq = (int32_t)a[i+1].y - (int32_t)a[i].y;
		min = lr < lq? lr : lq;
		max = lr > lq? lr : lq;
		if (max - min > l >> 1) *cnt = i + 1 - *as;
		l += min;
		m += min < q_span? min : q_span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->mlen >> 1) break;
	}
}

static void mm_align1(const mmo_mapopt_t *opt, const mmo_idx_t *mi, int qlen, uint8_t *qseq0[2], mmo_reg1_t *r, mmo_reg1_t *r2, int n_a, mm128_t *a, mmo_extz_t *ez)
{
	int32_t rid = a[r->as].x<<1>>33, rev = a[r->as].x>>63, as1, cnt1;
	uint8_t *tseq, *qseq;
	int32_t i, l, bw, bw_long, dropped = 0, extra_flag = 0, rs0, re0, qs0, qe0;
	int32_t rs, re, qs, qe;
	int32_t rs1, qs1, re1, qe1;
	int8_t mat[25];

	r2->cnt = 0;
	if (r->cnt == 0) return;
	mmo_ksw_gen_simple_mat(5, mat, opt->a, opt->b, opt->sc_ambi);
	bw = (int)(opt->bw * 1.5 + 1.);
	bw_long = (int)(opt->bw_long * 1.5 + 1.);
	if (bw_long < bw) bw_long = bw;

	if (!(opt->flag & MM_F_NO_END_FLT))
		mm_fix_bad_ends(r, a, opt->bw, opt->min_chain_score * 2, &as1, &cnt1);
	else as1 = r->as, cnt1 = r->cnt;
	mm_filter_bad_seeds(as1, cnt1, a, 10, 40, opt->max_gap>>1, 10);
	mm_filter_bad_seeds_alt(as1, cnt1, a, 30, opt->max_gap>>1);
	mm_adjust_minier(mi, qseq0, &a[as1], &rs, &qs);
	mm_adjust_minier(mi, qseq0, &a[as1 + cnt1 - 1], &re, &qe);
	assert(cnt1 > 0);

	/* compute rs0 and qs0 */
	rs0 = (int32_t)a[r->as].x + 1 - (int32_t)(a[r->as].y>>32&0xff);
	qs0 = (int32_t)a[r->as].y + 1 - (int32_t)(a[r->as].y>>32&0xff);
	if (rs0 < 0) rs0 = 0;
	assert(qs0 >= 0);
	rs1 = qs1 = 0;
	for (i = r->as - 1, l = 0; i >= 0 && a[i].x>>32 == a[r->as].x>>32; --i) { /* inspect nearby seeds */
		int32_t x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y>>32&0xff);
		int32_t y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y>>32&0xff);
		if (x < rs0 && y < qs0) {
			if (++l > opt->min_cnt) {
				l = rs0 - x > qs0 - y? rs0 - x : qs0 - y;
				rs1 = rs0 - l, qs1 = qs0 - l;
				if (rs1 < 0) rs1 = 0;
				break;
			}
		}
	}
	if (qs > 0 && rs > 0) {
		l = qs < opt->max_gap? qs : opt->max_gap;
		qs1 = qs1 > qs - l? qs1 : qs - l;
		qs0 = qs0 < qs1? qs0 : qs1; /* at least include qs0 */
		l += l * opt->a > opt->q? (l * opt->a - opt->q) / opt->e : 0;
		l = l < opt->max_gap? l : opt->max_gap;
		l = l < rs? l : rs;
		rs1 = rs1 > rs - l? rs1 : rs - l;
		rs0 = rs0 < rs1? rs0 : rs1;
		rs0 = rs0 < rs? rs0 : rs;
	} else rs0 = rs, qs0 = qs;
	/* compute re0 and qe0 */
	re0 = (int32_t)a[r->as + r->cnt - 1].x + 1;
	qe0 = (int32_t)a[r->as + r->cnt - 1].y + 1;
	re1 = mi->seq[rid].len, qe1 = qlen;
	for (i = r->as + r->cnt, l = 0; i < n_a && a[i].x>>32 == a[r->as].x>>32; ++i) { /* inspect nearby seeds */
		int32_t x = (int32_t)a[i].x + 1;
		int32_t y = (int32_t)a[i].y + 1;
		if (x > re0 && y > qe0) {
			if (++l > opt->min_cnt) {
				l = x - re0 > y - qe0? x - re0 : y - qe0;
				re1 = re0 + l, qe1 = qe0 + l;
				break;
			}
		}
	}
	if (qe < qlen && re < (int32_t)mi->seq[rid].len) {
		l = qlen - qe < opt->max_gap? qlen - qe : opt->max_gap;
		qe1 = qe1 < qe + l? qe1 : qe + l;
		qe0 = qe0 > qe1? qe0 : qe1; /* at least include qe0 */
		l += l * opt->a > opt->q? (l * opt->a - opt->q) / opt->e : 0;
		l = l < opt->max_gap? l : opt->max_gap;
		l = l < (int32_t)mi->seq[rid].len - re? l : (int32_t)mi->seq[rid].len - re;
		re1 = re1 < re + l? re1 : re + l;
		re0 = re0 > re1? re0 : re1;
	} else re0 = re, qe0 = qe;
	/* MM_SEED_SELF is never set (qname == NULL through this reference) */

	assert(re0 > rs0);
	tseq = (uint8_t*)malloc(re0 - rs0);

	if (qs > 0 && rs > 0) { /* left extension */
		qseq = &qseq0[rev][qs0];
		mmo_idx_getseq(mi, rid, rs0, rs, tseq);
		mm_seq_rev(qs - qs0, qseq);
		mm_seq_rev(rs - rs0, tseq);
		mm_align_pair(opt, qs - qs0, qseq, rs - rs0, tseq, mat, bw, opt->end_bonus, r->split_inv? opt->zdrop_inv : opt->zdrop, extra_flag|KSW_EZ_EXTZ_ONLY|KSW_EZ_RIGHT|KSW_EZ_REV_CIGAR, ez);
		if (ez->n_cigar > 0) {
			mm_append_cigar(r, ez->n_cigar, ez->cigar);
			r->p->dp_score += ez->max;
		}
		rs1 = rs - (ez->reach_end? ez->mqe_t + 1 : ez->max_t + 1);
		qs1 = qs - (ez->reach_end? qs - qs0 : ez->max_q + 1);
		mm_seq_rev(qs - qs0, qseq);
	} else rs1 = rs, qs1 = qs;
	re1 = rs, qe1 = qs;
	assert(qs1 >= 0 && rs1 >= 0);

	for (i = 1; i < cnt1; ++i) { /* gap filling */
		if ((a[as1+i].y & (MM_SEED_IGNORE|MM_SEED_TANDEM)) && i != cnt1 - 1) continue;
		mm_adjust_minier(mi, qseq0, &a[as1 + i], &re, &qe);
		re1 = re, qe1 = qe;
		if (i == cnt1 - 1 || (a[as1+i].y&MM_SEED_LONG_JOIN) || (qe - qs >= opt->min_ksw_len && re - rs >= opt->min_ksw_len)) {
			int j, bw1 = bw_long, zdrop_code;
			if (a[as1+i].y & MM_SEED_LONG_JOIN)
				bw1 = qe - qs > re - rs? qe - qs : re - rs;
			/* perform normal gapped alignment */
			qseq = &qseq0[rev][qs];
			mmo_idx_getseq(mi, rid, rs, re, tseq);
			mm_align_pair(opt, qe - qs, qseq, re - rs, tseq, mat, bw1, -1, opt->zdrop, extra_flag|KSW_EZ_APPROX_MAX, ez); /* first pass: with approximate Z-drop */
			/* test Z-drop and inversion Z-drop */
			if ((zdrop_code = mm_test_zdrop(opt, qseq, tseq, ez->n_cigar, ez->cigar, mat)) != 0)
				mm_align_pair(opt, qe - qs, qseq, re - rs, tseq, mat, bw1, -1, zdrop_code == 2? opt->zdrop_inv : opt->zdrop, extra_flag, ez); /* second pass: lift approximate */
			/* update CIGAR */
			if (ez->n_cigar > 0)
				mm_append_cigar(r, ez->n_cigar, ez->cigar);
			if (ez->zdropped) { /* truncated by Z-drop */
				if (!r->p) {
					uint32_t capacity = sizeof(mmo_extra_t)/4;
					assert(ez->n_cigar == 0);
					kroundup32(capacity);
					r->p = (mmo_extra_t*)calloc(capacity, 4);
					r->p->capacity = capacity;
				}
				for (j = i - 1; j >= 0; --j)
					if ((int32_t)a[as1 + j].x <= rs + ez->max_t)
						break;
				dropped = 1;
				if (j < 0) j = 0;
				r->p->dp_score += ez->max;
				re1 = rs + (ez->max_t + 1);
				qe1 = qs + (ez->max_q + 1);
				if (cnt1 - (j + 1) >= opt->min_cnt) {
					mmo_split_reg(r, r2, as1 + j + 1 - r->as, qlen, a, !!(opt->flag&MM_F_QSTRAND));
					if (zdrop_code == 2) r2->split_inv = 1;
				}
				break;
			} else r->p->dp_score += ez->score;
			rs = re, qs = qe;
		}
	}

	if (!dropped && qe < qe0 && re < re0) { /* right extension */
		qseq = &qseq0[rev][qe];
		mmo_idx_getseq(mi, rid, re, re0, tseq);
		mm_align_pair(opt, qe0 - qe, qseq, re0 - re, tseq, mat, bw, opt->end_bonus, opt->zdrop, extra_flag|KSW_EZ_EXTZ_ONLY, ez);
		if (ez->n_cigar > 0) {
			mm_append_cigar(r, ez->n_cigar, ez->cigar);
			r->p->dp_score += ez->max;
		}
		re1 = re + (ez->reach_end? ez->mqe_t + 1 : ez->max_t + 1);
		qe1 = qe + (ez->reach_end? qe0 - qe : ez->max_q + 1);
	}
	assert(qe1 <= qlen);

	r->rs = rs1, r->re = re1;
	if (rev) r->qs = qlen - qe1, r->qe = qlen - qs1;
	else r->qs = qs1, r->qe = qe1;

	assert(re1 - rs1 <= re0 - rs0);
	if (r->p) {
		mmo_idx_getseq(mi, rid, rs1, re1, tseq);
		mm_update_extra(r, &qseq0[r->rev][qs1], tseq, mat, opt->q, opt->e, !!(opt->flag & MM_F_EQX), !(opt->flag & MM_F_SR));
	}

	free(tseq);
}

static int mm_align1_inv(const mmo_mapopt_t *opt, const mmo_idx_t *mi, int qlen, uint8_t *qseq0[2], const mmo_reg1_t *r1, const mmo_reg1_t *r2, mmo_reg1_t *r_inv, mmo_extz_t *ez)
{
	int tl, ql, score, ret = 0, q_off, t_off;
	uint8_t *tseq, *qseq;
	int8_t mat[25];

	memset(r_inv, 0, sizeof(mmo_reg1_t));
	if (!(r1->split&1) || !(r2->split&2)) return 0;
	if (r1->id != r1->parent && r1->parent != MM_PARENT_TMP_PRI) return 0;
	if (r2->id != r2->parent && r2->parent != MM_PARENT_TMP_PRI) return 0;
	if (r1->rid != r2->rid || r1->rev != r2->rev) return 0;
	ql = r1->rev? r1->qs - r2->qe : r2->qs - r1->qe;
	tl = r2->rs - r1->re;
	if (ql < opt->min_chain_score || ql > opt->max_gap) return 0;
	if (tl < opt->min_chain_score || tl > opt->max_gap) return 0;

	mmo_ksw_gen_simple_mat(5, mat, opt->a, opt->b, opt->sc_ambi);
	tseq = (uint8_t*)malloc(tl);
	mmo_idx_getseq(mi, r1->rid, r1->re, r2->rs, tseq);
	qseq = r1->rev? &qseq0[0][r2->qe] : &qseq0[1][qlen - r2->qs];

	mm_seq_rev(ql, qseq);
	mm_seq_rev(tl, tseq);
	score = mmo_ksw_ll(ql, qseq, tl, tseq, 5, mat, opt->q, opt->e, &q_off, &t_off);
	mm_seq_rev(ql, qseq);
	mm_seq_rev(tl, tseq);
	if (score < opt->min_dp_max) goto end_align1_inv;
	q_off = ql - (q_off + 1), t_off = tl - (t_off + 1);
	if (q_off < 0 || t_off < 0) goto end_align1_inv; /* guard: upstream would read out of bounds here */
	mm_align_pair(opt, ql - q_off, qseq + q_off, tl - t_off, tseq + t_off, mat, (int)(opt->bw * 1.5), -1, opt->zdrop, KSW_EZ_EXTZ_ONLY, ez);
	if (ez->n_cigar == 0) goto end_align1_inv; /* should never be here */
	mm_append_cigar(r_inv, ez->n_cigar, ez->cigar);
	r_inv->p->dp_score = ez->max;
	r_inv->id = -1;
	r_inv->parent = MM_PARENT_UNSET;
	r_inv->inv = 1;
	r_inv->rev = !r1->rev;
	r_inv->rid = r1->rid;
	r_inv->div = -1.0f;
	if (r_inv->rev == 0) {
		r_inv->qs = r2->qe + q_off;
		r_inv->qe = r_inv->qs + ez->max_q + 1;
	} else {
		r_inv->qe = r2->qs - q_off;
		r_inv->qs = r_inv->qe - (ez->max_q + 1);
	}
	r_inv->rs = r1->re + t_off;
	r_inv->re = r_inv->rs + ez->max_t + 1;
	mm_update_extra(r_inv, &qseq[q_off], &tseq[t_off], mat, opt->q, opt->e, !!(opt->flag & MM_F_EQX), !(opt->flag & MM_F_SR));
	ret = 1;
end_align1_inv:
	free(tseq);
	return ret;
}

static mmo_reg1_t *mm_insert_reg(const mmo_reg1_t *r, int i, int *n_regs, mmo_reg1_t *regs)
{
	regs = (mmo_reg1_t*)realloc(regs, (*n_regs + 1) * sizeof(mmo_reg1_t));
	if (i + 1 != *n_regs)
		memmove(&regs[i + 2], &regs[i + 1], sizeof(mmo_reg1_t) * (*n_regs - i - 1));
	regs[i + 1] = *r;
	++*n_regs;
	return regs;
}

mmo_reg1_t *mmo_align_skeleton(const mmo_mapopt_t *opt, const mmo_idx_t *mi, int qlen, const char *qstr, int *n_regs_, mmo_reg1_t *regs, mm128_t *a)
{
	int32_t i, n_regs = *n_regs_, n_a;
	uint8_t *qseq0[2];
	mmo_extz_t ez;

	/* encode the query sequence */
	qseq0[0] = (uint8_t*)malloc(qlen * 2);
	qseq0[1] = qseq0[0] + qlen;
	for (i = 0; i < qlen; ++i) {
		qseq0[0][i] = mmo_seq_nt4_table[(uint8_t)qstr[i]];
		qseq0[1][qlen - 1 - i] = qseq0[0][i] < 4? 3 - qseq0[0][i] : 4;
	}

	/* align through seed hits */
	n_a = mmo_squeeze_a(n_regs, regs, a);
	memset(&ez, 0, sizeof(mmo_extz_t));
	for (i = 0; i < n_regs; ++i) {
		mmo_reg1_t r2;
		mm_align1(opt, mi, qlen, qseq0, &regs[i], &r2, n_a, a, &ez);
		if (r2.cnt > 0) regs = mm_insert_reg(&r2, i, &n_regs, regs);
		if (i > 0 && regs[i].split_inv && !(opt->flag & MM_F_NO_INV)) {
			if (mm_align1_inv(opt, mi, qlen, qseq0, &regs[i-1], &regs[i], &r2, &ez)) {
				regs = mm_insert_reg(&r2, i, &n_regs, regs);
				++i; /* skip the inserted INV alignment */
			}
		}
	}
	*n_regs_ = n_regs;
	free(qseq0[0]);
	free(ez.cigar);
	mmo_filter_regs(opt, qlen, n_regs_, regs);
	mmo_hit_sort(n_regs_, regs, opt->alt_drop);
	return regs;
}
