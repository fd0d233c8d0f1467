/* ORACLE (test infrastructure).  Restates, for minimap2 2.26:
 *   U:seed.c::mm_seed_mz_flt, mm_seed_collect_all, mm_seed_select, mm_collect_matches
 *   U:map.c::collect_minimizers, collect_seed_hits, chain_post, align_regs, mm_map_frag, mm_map
 * and the L2 crate's Aligner::map record conversion (is_primary = parent==id,
 * NM = blen - mlen + n_ambi, CIGAR unpack, cs/MD) as consumed at
 * R:src/lib.rs:489-511 and :594-616.  Reference call sites of mm_map:
 * R:src/lib.rs:482-488 (single read) and :587-593 (batch worker).
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "mmo.h"

__thread mmo_stats_t mmo_stats;

/* ---------------- U:seed.c ---------------- */

void mmo_seed_mz_flt(mm128_v *mv, int32_t q_occ_max, float q_occ_frac)
{
	mm128_t *a;
	size_t i, j, st;
	if (mv->n <= (size_t)q_occ_max || q_occ_frac <= 0.0f || q_occ_max <= 0) return;
	a = (mm128_t*)malloc(mv->n * sizeof(mm128_t));
	for (i = 0; i < mv->n; ++i)
		a[i].x = mv->a[i].x, a[i].y = i;
	mmo_radix_sort_128x(a, a + mv->n);
	for (st = 0, i = 1; i <= mv->n; ++i) {
		if (i == mv->n || a[i].x != a[st].x) {
			int32_t cnt = i - st;
			if (cnt > q_occ_max && cnt > mv->n * q_occ_frac)
				for (j = st; j < i; ++j)
					mv->a[a[j].y].x = 0;
			st = i;
		}
	}
	free(a);
	for (i = j = 0; i < mv->n; ++i)
		if (mv->a[i].x != 0)
			mv->a[j++] = mv->a[i];
	mv->n = j;
}

static mmo_seed_t *seed_collect_all(const mmo_idx_t *mi, const mm128_v *mv, int32_t *n_m_)
{
	mmo_seed_t *m;
	size_t i;
	int32_t k;
	m = (mmo_seed_t*)malloc((mv->n? mv->n : 1) * sizeof(mmo_seed_t));
	for (i = k = 0; i < mv->n; ++i) {
		const uint64_t *cr;
		mmo_seed_t *q;
		mm128_t *p = &mv->a[i];
		uint32_t q_pos = (uint32_t)p->y, q_span = p->x & 0xff;
		int t;
		cr = mmo_idx_get(mi, p->x>>8, &t);
		if (t == 0) continue;
		q = &m[k++];
		q->q_pos = q_pos, q->q_span = q_span, q->cr = cr, q->n = t, q->seg_id = p->y >> 32;
		q->is_tandem = q->flt = 0;
		if (i > 0 && p->x>>8 == mv->a[i - 1].x>>8) q->is_tandem = 1;
		if (i < mv->n - 1 && p->x>>8 == mv->a[i + 1].x>>8) q->is_tandem = 1;
	}
	*n_m_ = k;
	return m;
}

/* U:ksort.h heap on uint64_t (max-heap) as used by mm_seed_select */
static void heapdown_u64(size_t i, size_t n, uint64_t l[])
{
	size_t k = i;
	uint64_t tmp = l[i];
	while ((k = (k << 1) + 1) < n) {
		if (k != n - 1 && l[k] < l[k+1]) ++k;
		if (l[k] < tmp) break;
		l[i] = l[k]; i = k;
	}
	l[i] = tmp;
}
static void heapmake_u64(size_t lsize, uint64_t l[])
{
	size_t i;
	for (i = (lsize >> 1) - 1; i != (size_t)(-1); --i)
		heapdown_u64(i, lsize, l);
}

#define MAX_MAX_HIGH_OCC 128

static void seed_select(int32_t n, mmo_seed_t *a, int len, int max_occ, int max_max_occ, int dist)
{
	int32_t i, last0, m;
	uint64_t b[MAX_MAX_HIGH_OCC];

	if (n == 0 || n == 1) return;
	for (i = m = 0; i < n; ++i)
		if (a[i].n > (uint32_t)max_occ) ++m;
	if (m == 0) return;
	for (i = 0, last0 = -1; i <= n; ++i) {
		if (i == n || a[i].n <= (uint32_t)max_occ) {
			if (i - last0 > 1) {
				int32_t ps = last0 < 0? 0 : (uint32_t)a[last0].q_pos>>1;
				int32_t pe = i == n? len : (uint32_t)a[i].q_pos>>1;
				int32_t j, k, st = last0 + 1, en = i;
				int32_t max_high_occ = (int32_t)((double)(pe - ps) / dist + .499);
				if (max_high_occ > 0) {
					if (max_high_occ > MAX_MAX_HIGH_OCC)
						max_high_occ = MAX_MAX_HIGH_OCC;
					for (j = st, k = 0; j < en && k < max_high_occ; ++j, ++k)
						b[k] = (uint64_t)a[j].n<<32 | j;
					heapmake_u64(k, b);
					for (; j < en; ++j) {
						if ((int32_t)a[j].n < (int32_t)(b[0]>>32)) {
							b[0] = (uint64_t)a[j].n<<32 | j;
							heapdown_u64(0, k, b);
						}
					}
					for (j = 0; j < k; ++j) a[(uint32_t)b[j]].flt = 1;
				}
				for (j = st; j < en; ++j) a[j].flt ^= 1;
				for (j = st; j < en; ++j)
					if (a[j].n > (uint32_t)max_max_occ)
						a[j].flt = 1;
			}
			last0 = i;
		}
	}
}

static mmo_seed_t *collect_matches(int *_n_m, int qlen, int max_occ, int max_max_occ, int dist, const mmo_idx_t *mi, const mm128_v *mv,
                                   int64_t *n_a, int *rep_len, int *n_mini_pos, uint64_t **mini_pos)
{
	int rep_st = 0, rep_en = 0, n_m, n_m0;
	size_t i;
	mmo_seed_t *m;
	*n_mini_pos = 0;
	*mini_pos = (uint64_t*)malloc((mv->n? mv->n : 1) * sizeof(uint64_t));
	m = seed_collect_all(mi, mv, &n_m0);
	mmo_stats.n_hit += n_m0;
	if (dist > 0 && max_max_occ > max_occ) {
		seed_select(n_m0, m, qlen, max_occ, max_max_occ, dist);
	} else {
		for (i = 0; i < (size_t)n_m0; ++i)
			if (m[i].n > (uint32_t)max_occ)
				m[i].flt = 1;
	}
	for (i = 0, n_m = 0, *rep_len = 0, *n_a = 0; i < (size_t)n_m0; ++i) {
		mmo_seed_t *q = &m[i];
		if (q->flt) {
			int en = (q->q_pos >> 1) + 1, st = en - q->q_span;
			if (st > rep_en) {
				*rep_len += rep_en - rep_st;
				rep_st = st, rep_en = en;
			} else rep_en = en;
		} else {
			*n_a += q->n;
			(*mini_pos)[(*n_mini_pos)++] = (uint64_t)q->q_span<<32 | q->q_pos>>1;
			m[n_m++] = *q;
		}
	}
	*rep_len += rep_en - rep_st;
	*_n_m = n_m;
	return m;
}

/* ---------------- U:map.c ---------------- */

static inline int skip_seed(int64_t flag, uint64_t r, const mmo_seed_t *q)
{
	/* qname is always NULL through the reference (L2 passes null): NO_DIAG/NO_DUAL branch is dead */
	if (flag & (MM_F_FOR_ONLY|MM_F_REV_ONLY)) {
		if ((r&1) == (q->q_pos&1)) { /* forward strand */
			if (flag & MM_F_REV_ONLY) return 1;
		} else {
			if (flag & MM_F_FOR_ONLY) return 1;
		}
	}
	return 0;
}

/* U:map.c::collect_seed_hits.  `sorted`=0 returns anchors in generation order (for kernel parity tests). */
mm128_t *mmo_collect_seed_hits(const mmo_mapopt_t *opt, int max_occ, const mmo_idx_t *mi, const mm128_v *mv, int qlen,
                               int64_t *n_a, int *rep_len, int *n_mini_pos, uint64_t **mini_pos, int sorted)
{
	int i, n_m;
	mmo_seed_t *m;
	mm128_t *a;
	m = collect_matches(&n_m, qlen, max_occ, opt->max_max_occ, opt->occ_dist, mi, mv, n_a, rep_len, n_mini_pos, mini_pos);
	a = (mm128_t*)malloc((*n_a > 0? *n_a : 1) * sizeof(mm128_t));
	for (i = 0, *n_a = 0; i < n_m; ++i) {
		mmo_seed_t *q = &m[i];
		const uint64_t *r = q->cr;
		uint32_t k;
		if (q->n > 1) mmo_stats.n_a_multi += q->n;
		for (k = 0; k < q->n; ++k) {
			int32_t rpos = (uint32_t)r[k] >> 1;
			mm128_t *p;
			if (skip_seed(opt->flag, r[k], q)) continue;
			p = &a[(*n_a)++];
			if ((r[k]&1) == (q->q_pos&1)) { /* forward strand */
				p->x = (r[k]&0xffffffff00000000ULL) | rpos;
				p->y = (uint64_t)q->q_span << 32 | q->q_pos >> 1;
			} else { /* reverse strand (query-strand mode is out of scope) */
				p->x = 1ULL<<63 | (r[k]&0xffffffff00000000ULL) | rpos;
				p->y = (uint64_t)q->q_span << 32 | (qlen - ((q->q_pos>>1) + 1 - q->q_span) - 1);
			}
			p->y |= (uint64_t)q->seg_id << MM_SEED_SEG_SHIFT;
			if (q->is_tandem) p->y |= MM_SEED_TANDEM;
		}
	}
	free(m);
	if (sorted) mmo_radix_sort_128x(a, a + (*n_a));
	return a;
}

static inline uint32_t wang_hash32(uint32_t key)
{
	key += ~(key << 15);
	key ^=  (key >> 10);
	key +=  (key << 3);
	key ^=  (key >> 6);
	key += ~(key << 11);
	key ^=  (key >> 16);
	return key;
}

static inline uint32_t x31_hash_string(const char *s)
{
	uint32_t h = (uint32_t)*s;
	if (h) for (++s ; *s; ++s) h = (h << 5) - h + (uint32_t)*s;
	return h;
}

static void chain_post(const mmo_mapopt_t *opt, const mmo_idx_t *mi, int *n_regs, mmo_reg1_t *regs)
{
	if (!(opt->flag & MM_F_ALL_CHAINS)) {
		mmo_set_parent(opt->mask_level, opt->mask_len, *n_regs, regs, opt->a * 2 + opt->b, opt->flag&MM_F_HARD_MLEVEL, opt->alt_drop);
		mmo_select_sub(opt->pri_ratio, mi->k*2, opt->best_n, 1, opt->max_gap * 0.8, n_regs, regs);
	}
}

static mmo_reg1_t *align_regs(const mmo_mapopt_t *opt, const mmo_idx_t *mi, int qlen, const char *seq, int *n_regs, mmo_reg1_t *regs, mm128_t *a)
{
	if (!(opt->flag & MM_F_CIGAR)) return regs;
	regs = mmo_align_skeleton(opt, mi, qlen, seq, n_regs, regs, a);
	if (!(opt->flag & MM_F_ALL_CHAINS)) {
		mmo_set_parent(opt->mask_level, opt->mask_len, *n_regs, regs, opt->a * 2 + opt->b, opt->flag&MM_F_HARD_MLEVEL, opt->alt_drop);
		mmo_select_sub(opt->pri_ratio, mi->k*2, opt->best_n, 0, opt->max_gap * 0.8, n_regs, regs);
		mmo_set_sam_pri(*n_regs, regs);
	}
	return regs;
}

/* U:map.c::mm_map_frag with n_segs == 1 (the only way mm_map calls it) */
mmo_reg1_t *mmo_map(const mmo_idx_t *mi, int qlen, const char *seq, int *n_regs, const mmo_mapopt_t *opt, const char *qname)
{
	int rep_len, n_regs0, n_mini_pos;
	int max_chain_gap_qry, max_chain_gap_ref, is_splice = !!(opt->flag & MM_F_SPLICE), is_sr = !!(opt->flag & MM_F_SR);
	uint32_t hash;
	int64_t n_a;
	uint64_t *u, *mini_pos;
	mm128_t *a;
	mm128_v mv = {0,0,0};
	mmo_reg1_t *regs0;
	float chn_pen_gap, chn_pen_skip;

	memset(&mmo_stats, 0, sizeof(mmo_stats));
	*n_regs = 0;
	if (qlen == 0) return 0;
	if (opt->max_qlen > 0 && qlen > opt->max_qlen) return 0;
	if (is_splice || is_sr) return 0; /* out of scope */

	hash  = qname && !(opt->flag & MM_F_NO_HASH_NAME)? x31_hash_string(qname) : 0;
	hash ^= wang_hash32(qlen) + wang_hash32(opt->seed);
	hash  = wang_hash32(hash);

	/* collect_minimizers (n_segs==1: the y += sum<<1 shift is a no-op; sdust_thres==0) */
	mmo_sketch(seq, qlen, mi->w, mi->k, 0, mi->flag&MM_I_HPC, &mv);
	if (opt->q_occ_frac > 0.0f) mmo_seed_mz_flt(&mv, opt->mid_occ, opt->q_occ_frac);
	mmo_stats.n_mz = mv.n;
	a = mmo_collect_seed_hits(opt, opt->mid_occ, mi, &mv, qlen, &n_a, &rep_len, &n_mini_pos, &mini_pos, 1);
	mmo_stats.n_a = n_a; mmo_stats.rep_len = rep_len;

	max_chain_gap_qry = opt->max_gap;
	if (opt->max_gap_ref > 0) {
		max_chain_gap_ref = opt->max_gap_ref;
	} else if (opt->max_frag_len > 0) {
		max_chain_gap_ref = opt->max_frag_len - qlen;
		if (max_chain_gap_ref < opt->max_gap) max_chain_gap_ref = opt->max_gap;
	} else max_chain_gap_ref = opt->max_gap;

	chn_pen_gap  = opt->chain_gap_scale * 0.01 * mi->k;
	chn_pen_skip = opt->chain_skip_scale * 0.01 * mi->k;
	if (opt->flag & MM_F_RMQ) {
		a = mmo_lchain_rmq(opt->max_gap, opt->rmq_inner_dist, opt->bw, opt->max_chain_skip, opt->rmq_size_cap, opt->min_cnt, opt->min_chain_score,
		                   chn_pen_gap, chn_pen_skip, n_a, a, &n_regs0, &u);
	} else {
		a = mmo_lchain_dp(max_chain_gap_ref, max_chain_gap_qry, opt->bw, opt->max_chain_skip, opt->max_chain_iter, opt->min_cnt, opt->min_chain_score,
		                  chn_pen_gap, chn_pen_skip, is_splice, 1, n_a, a, &n_regs0, &u);
	}
	mmo_stats.n_chain0 = n_regs0;

	if (opt->bw_long > opt->bw && (opt->flag & (MM_F_SPLICE|MM_F_SR|MM_F_NO_LJOIN)) == 0 && n_regs0 > 1) { /* re-chain/long-join */
		int32_t st = (int32_t)a[0].y, en = (int32_t)a[(int32_t)u[0] - 1].y;
		if (qlen - (en - st) > opt->rmq_rescue_size || en - st > qlen * opt->rmq_rescue_ratio) {
			int32_t i;
			for (i = 0, n_a = 0; i < n_regs0; ++i) n_a += (int32_t)u[i];
			free(u);
			mmo_radix_sort_128x(a, a + n_a);
			a = mmo_lchain_rmq(opt->max_gap, opt->rmq_inner_dist, opt->bw_long, opt->max_chain_skip, opt->rmq_size_cap, opt->min_cnt, opt->min_chain_score,
			                   chn_pen_gap, chn_pen_skip, n_a, a, &n_regs0, &u);
			mmo_stats.did_rmq = 1;
		}
	}
	/* (the short-read "max_occ > mid_occ" re-chain branch is unreachable for long-read presets: max_occ == 0) */
	mmo_stats.n_chain1 = n_regs0;

	regs0 = mmo_gen_regs(hash, qlen, n_regs0, u, a, !!(opt->flag&MM_F_QSTRAND));
	/* mi->n_alt == 0: no ALT marking through this reference */

	chain_post(opt, mi, &n_regs0, regs0);
	if (!is_sr && !(opt->flag&MM_F_QSTRAND)) {
		mmo_est_err(mi, qlen, n_regs0, regs0, a, n_mini_pos, mini_pos);
		n_regs0 = mmo_filter_strand_retained(n_regs0, regs0);
	}

	regs0 = align_regs(opt, mi, qlen, seq, &n_regs0, regs0, a);
	mmo_set_mapq(n_regs0, regs0, opt->min_chain_score, opt->a, rep_len, is_sr);
	*n_regs = n_regs0;

	free(mv.a); free(a); free(u); free(mini_pos);
	return regs0;
}

/* ---------------- L2 crate conversion: minimap2::Aligner::map ---------------- */

void mmo_result_free(mmo_result_t *res)
{
	free(res->hits); free(res->cigar); free(res->str);
	memset(res, 0, sizeof(*res));
}

static int64_t res_push_str(mmo_result_t *res, const char *s, size_t l)
{
	int64_t off = res->n_str;
	if (res->n_str + l + 1 > res->m_str) {
		res->m_str = (res->n_str + l + 1) * 2;
		res->str = (char*)realloc(res->str, res->m_str);
	}
	memcpy(res->str + res->n_str, s, l);
	res->str[res->n_str + l] = 0;
	res->n_str += l + 1;
	return off;
}

int mmo_map_flat(const mmo_idx_t *mi, const mmo_mapopt_t *opt, const char *seq, int len, int with_cs, int with_md, mmo_result_t *res)
{
	int n_regs = 0, i;
	mmo_reg1_t *regs;
	memset(res, 0, sizeof(*res));
	if (mi == 0) return -1;     /* "No index" */
	if (len == 0) return -2;    /* "Sequence is empty" */
	regs = mmo_map(mi, len, seq, &n_regs, opt, 0);
	res->n_hits = n_regs;
	res->hits = (mmo_hit_t*)calloc(n_regs > 0? n_regs : 1, sizeof(mmo_hit_t));
	for (i = 0; i < n_regs; ++i) {
		mmo_reg1_t *r = &regs[i];
		mmo_hit_t *h = &res->hits[i];
		h->query_start = r->qs, h->query_end = r->qe;
		h->strand = r->rev? -1 : 1;
		h->rid = r->rid;
		h->target_len = mi->seq[r->rid].len;
		h->target_start = r->rs, h->target_end = r->re;
		h->match_len = r->mlen, h->block_len = r->blen;
		h->mapq = r->mapq;
		h->is_primary = (r->parent == r->id);
		h->cs_len = h->md_len = -1;
		h->score0 = r->score0, h->cnt = r->cnt, h->n_sub = r->n_sub, h->subsc = r->subsc, h->sam_pri = r->sam_pri;
		if (r->p) {
			uint32_t k;
			h->NM = r->blen - r->mlen + r->p->n_ambi;
			h->n_cigar = r->p->n_cigar;
			h->dp_max = r->p->dp_max, h->dp_max2 = r->p->dp_max2, h->dp_score = r->p->dp_score;
			if (res->n_cigar + r->p->n_cigar > res->m_cigar) {
				res->m_cigar = (res->n_cigar + r->p->n_cigar) * 2;
				res->cigar = (uint32_t*)realloc(res->cigar, res->m_cigar * 4);
			}
			h->cigar_off = res->n_cigar;
			for (k = 0; k < r->p->n_cigar; ++k) res->cigar[res->n_cigar++] = r->p->cigar[k];
			if (with_cs) {
				char *s = mmo_gen_cs(mi, r, seq, 1);
				h->cs_len = strlen(s);
				h->cs_off = res_push_str(res, s, h->cs_len);
				free(s);
			}
			if (with_md) {
				char *s = mmo_gen_MD(mi, r, seq);
				h->md_len = strlen(s);
				h->md_off = res_push_str(res, s, h->md_len);
				free(s);
			}
		}
		free(r->p);
	}
	free(regs);
	return 0;
}
