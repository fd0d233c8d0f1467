/* ORACLE (test infrastructure).  Restates U:lchain.c of minimap2 2.26:
 * mg_log2 (U:mmpriv.h), comput_sc, mg_lchain_dp, mg_chain_bk_end,
 * mg_chain_backtrack, compact_a, comput_sc_simple, mg_lchain_rmq, and the
 * AVL tree with range-min of U:krmq.h that mg_lchain_rmq depends on (tree shape
 * decides equal-priority ties, so it is restated literally).
 * Reference call site: mm_map at R:src/lib.rs:482 / :587.
 * All float32 arithmetic is compiled with -ffp-contract=off (see Makefile).
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "mmo.h"

static inline float mg_log2(float x) /* NB: this doesn't work when x<2 */
{
	union { float f; uint32_t i; } z = { x };
	float log_2 = ((z.i >> 23) & 255) - 128;
	z.i &= ~(255 << 23);
	z.i += 127 << 23;
	log_2 += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return log_2;
}

static int64_t mg_chain_bk_end(int32_t max_drop, const mm128_t *z, const int32_t *f, const int64_t *p, int32_t *t, int64_t k)
{
	int64_t i = z[k].y, end_i = -1, max_i = i;
	int32_t max_s = 0;
	if (i < 0 || t[i] != 0) return i;
	do {
		int32_t s;
		t[i] = 2;
		end_i = i = p[i];
		s = i < 0? z[k].x : (int32_t)z[k].x - f[i];
		if (s > max_s) max_s = s, max_i = i;
		else if (max_s - s > max_drop) break;
	} while (i >= 0 && t[i] == 0);
	for (i = z[k].y; i >= 0 && i != end_i; i = p[i]) /* reset modified t[] */
		t[i] = 0;
	return max_i;
}

static uint64_t *mg_chain_backtrack(int64_t n, const int32_t *f, const int64_t *p, int32_t *v, int32_t *t, int32_t min_cnt, int32_t min_sc, int32_t max_drop, int32_t *n_u_, int32_t *n_v_)
{
	mm128_t *z;
	uint64_t *u;
	int64_t i, k, n_z, n_v;
	int32_t n_u;

	*n_u_ = *n_v_ = 0;
	for (i = 0, n_z = 0; i < n; ++i)
		if (f[i] >= min_sc) ++n_z;
	if (n_z == 0) return 0;
	z = (mm128_t*)malloc(n_z * sizeof(mm128_t));
	for (i = 0, k = 0; i < n; ++i)
		if (f[i] >= min_sc) z[k].x = f[i], z[k++].y = i;
	mmo_radix_sort_128x(z, z + n_z);

	memset(t, 0, n * 4);
	for (k = n_z - 1, n_v = n_u = 0; k >= 0; --k) { /* precompute n_u */
		if (t[z[k].y] == 0) {
			int64_t n_v0 = n_v, end_i;
			int32_t sc;
			end_i = mg_chain_bk_end(max_drop, z, f, p, t, k);
			for (i = z[k].y; i != end_i; i = p[i])
				++n_v, t[i] = 1;
			sc = i < 0? z[k].x : (int32_t)z[k].x - f[i];
			if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt)
				++n_u;
			else n_v = n_v0;
		}
	}
	u = (uint64_t*)malloc((n_u > 0? n_u : 1) * 8);
	memset(t, 0, n * 4);
	for (k = n_z - 1, n_v = n_u = 0; k >= 0; --k) { /* populate u[] */
		if (t[z[k].y] == 0) {
			int64_t n_v0 = n_v, end_i;
			int32_t sc;
			end_i = mg_chain_bk_end(max_drop, z, f, p, t, k);
			for (i = z[k].y; i != end_i; i = p[i])
				v[n_v++] = i, t[i] = 1;
			sc = i < 0? z[k].x : (int32_t)z[k].x - f[i];
			if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt)
				u[n_u++] = (uint64_t)sc << 32 | (n_v - n_v0);
			else n_v = n_v0;
		}
	}
	free(z);
	*n_u_ = n_u, *n_v_ = n_v;
	return u;
}

static mm128_t *compact_a(int32_t n_u, uint64_t *u, int32_t n_v, int32_t *v, mm128_t *a)
{
	mm128_t *b, *w;
	uint64_t *u2;
	int64_t i, j, k;

	b = (mm128_t*)malloc((n_v > 0? n_v : 1) * sizeof(mm128_t));
	for (i = 0, k = 0; i < n_u; ++i) {
		int32_t k0 = k, ni = (int32_t)u[i];
		for (j = 0; j < ni; ++j)
			b[k++] = a[v[k0 + (ni - j - 1)]];
	}
	free(v);

	/* sort u[] and a[] by the target position, such that adjacent chains may be joined */
	w = (mm128_t*)malloc(n_u * sizeof(mm128_t));
	for (i = k = 0; i < n_u; ++i) {
		w[i].x = b[k].x, w[i].y = (uint64_t)k<<32|i;
		k += (int32_t)u[i];
	}
	mmo_radix_sort_128x(w, w + n_u);
	u2 = (uint64_t*)malloc(n_u * 8);
	for (i = k = 0; i < n_u; ++i) {
		int32_t j = (uint32_t)w[i].y, n = (uint32_t)u[j];
		u2[i] = u[j];
		memcpy(&a[k], &b[w[i].y>>32], n * sizeof(mm128_t));
		k += n;
	}
	memcpy(u, u2, n_u * 8);
	memcpy(b, a, k * sizeof(mm128_t));
	free(a); free(w); free(u2);
	return b;
}

static inline int32_t comput_sc(const mm128_t *ai, const mm128_t *aj, int32_t max_dist_x, int32_t max_dist_y, int32_t bw, float chn_pen_gap, float chn_pen_skip, int is_cdna, int n_seg)
{
	int32_t dq = (int32_t)ai->y - (int32_t)aj->y, dr, dd, dg, q_span, sc;
	int32_t sidi = (ai->y & MM_SEED_SEG_MASK) >> MM_SEED_SEG_SHIFT;
	int32_t sidj = (aj->y & MM_SEED_SEG_MASK) >> MM_SEED_SEG_SHIFT;
	if (dq <= 0 || dq > max_dist_x) return INT32_MIN;
	dr = (int32_t)(ai->x - aj->x);
	if (sidi == sidj && (dr == 0 || dq > max_dist_y)) return INT32_MIN;
	dd = dr > dq? dr - dq : dq - dr;
	if (sidi == sidj && dd > bw) return INT32_MIN;
	if (n_seg > 1 && !is_cdna && sidi == sidj && dr > max_dist_y) return INT32_MIN;
	dg = dr < dq? dr : dq;
	q_span = aj->y>>32&0xff;
	sc = q_span < dg? q_span : dg;
	if (dd || dg > q_span) {
		float lin_pen, log_pen;
		lin_pen = chn_pen_gap * (float)dd + chn_pen_skip * (float)dg;
		log_pen = dd >= 1? mg_log2(dd + 1) : 0.0f; /* mg_log2() only works for dd>=2 */
		/* is_cdna / multi-segment branches are out of scope (long-read genomic only) */
		sc -= (int)(lin_pen + .5f * log_pen);
	}
	return sc;
}

/* the DP fill of U:lchain.c::mg_lchain_dp, exposed for kernel parity */
void mmo_lchain_dp_fill(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, float chn_pen_gap, float chn_pen_skip,
                        int64_t n, const mm128_t *a, int32_t *f, int64_t *p, int32_t *v, int32_t *t)
{
	int64_t i, j, max_ii, st = 0;
	int32_t mmax_f = 0;
	if (max_dist_x < bw) max_dist_x = bw;
	if (max_dist_y < bw) max_dist_y = bw;
	memset(t, 0, n * 4);
	for (i = 0, max_ii = -1; i < n; ++i) {
		int64_t max_j = -1, end_j;
		int32_t max_f = a[i].y>>32&0xff, n_skip = 0;
		while (st < i && (a[i].x>>32 != a[st].x>>32 || a[i].x > a[st].x + max_dist_x)) ++st;
		if (i - st > max_iter) st = i - max_iter;
		for (j = i - 1; j >= st; --j) {
			int32_t sc;
			sc = comput_sc(&a[i], &a[j], max_dist_x, max_dist_y, bw, chn_pen_gap, chn_pen_skip, 0, 1);
			++mmo_stats.chain_pairs;
			if (sc == INT32_MIN) continue;
			sc += f[j];
			if (sc > max_f) {
				max_f = sc, max_j = j;
				if (n_skip > 0) --n_skip;
			} else if (t[j] == (int32_t)i) {
				if (++n_skip > max_skip)
					break;
			}
			if (p[j] >= 0) t[p[j]] = i;
		}
		end_j = j;
		if (max_ii < 0 || a[i].x - a[max_ii].x > (uint64_t)(int64_t)max_dist_x) {
			int32_t max = INT32_MIN;
			max_ii = -1;
			for (j = i - 1; j >= st; --j)
				if (max < f[j]) max = f[j], max_ii = j;
		}
		if (max_ii >= 0 && max_ii < end_j) {
			int32_t tmp;
			tmp = comput_sc(&a[i], &a[max_ii], max_dist_x, max_dist_y, bw, chn_pen_gap, chn_pen_skip, 0, 1);
			if (tmp != INT32_MIN && max_f < tmp + f[max_ii])
				max_f = tmp + f[max_ii], max_j = max_ii;
		}
		f[i] = max_f, p[i] = max_j;
		v[i] = max_j >= 0 && v[max_j] > max_f? v[max_j] : max_f;
		if (max_ii < 0 || (a[i].x - a[max_ii].x <= (uint64_t)(int64_t)max_dist_x && f[max_ii] < f[i]))
			max_ii = i;
		if (mmax_f < max_f) mmax_f = max_f;
	}
}

mm128_t *mmo_lchain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc, float chn_pen_gap, float chn_pen_skip,
                       int is_cdna, int n_seg, int64_t n, mm128_t *a, int *n_u_, uint64_t **_u)
{
	int32_t *f, *t, *v, n_u, n_v, max_drop = bw;
	int64_t *p;
	uint64_t *u;
	(void)is_cdna; (void)n_seg;

	if (_u) *_u = 0, *n_u_ = 0;
	if (n == 0 || a == 0) {
		free(a);
		return 0;
	}
	p = (int64_t*)malloc(n * 8);
	f = (int32_t*)malloc(n * 4);
	v = (int32_t*)malloc(n * 4);
	t = (int32_t*)calloc(n, 4);
	mmo_lchain_dp_fill(max_dist_x, max_dist_y, bw, max_skip, max_iter, chn_pen_gap, chn_pen_skip, n, a, f, p, v, t);
	u = mg_chain_backtrack(n, f, p, v, t, min_cnt, min_sc, max_drop, &n_u, &n_v);
	*n_u_ = n_u, *_u = u;
	free(p); free(f); free(t);
	if (n_u == 0) {
		free(a); free(v);
		return 0;
	}
	return compact_a(n_u, u, n_v, v, a);
}

/* ---------------- U:krmq.h (AVL tree + range-min), literal ---------------- */

#define KRMQ_MAX_DEPTH 64

typedef struct lc_elem_s {
	int32_t y;
	int64_t i;
	double pri;
	struct { struct lc_elem_s *p[2], *s; signed char balance; unsigned size; } head;
} lc_elem_t;

#define lc_elem_cmp(a, b) ((a)->y < (b)->y? -1 : (a)->y > (b)->y? 1 : ((a)->i > (b)->i) - ((a)->i < (b)->i))
#define lc_elem_lt2(a, b) ((a)->pri < (b)->pri)
#define krmq_size(p) ((p)? (p)->head.size : 0)
#define krmq_size_child(q, i) ((q)->head.p[(i)]? (q)->head.p[(i)]->head.size : 0)

static lc_elem_t *krmq_find(const lc_elem_t *root, const lc_elem_t *x)
{
	const lc_elem_t *p = root;
	while (p != 0) {
		int cmp = lc_elem_cmp(x, p);
		if (cmp < 0) p = p->head.p[0];
		else if (cmp > 0) p = p->head.p[1];
		else break;
	}
	return (lc_elem_t*)p;
}

static lc_elem_t *krmq_interval(const lc_elem_t *root, const lc_elem_t *x, lc_elem_t **lower, lc_elem_t **upper)
{
	const lc_elem_t *p = root, *l = 0, *u = 0;
	while (p != 0) {
		int cmp = lc_elem_cmp(x, p);
		if (cmp < 0) u = p, p = p->head.p[0];
		else if (cmp > 0) l = p, p = p->head.p[1];
		else { l = u = p; break; }
	}
	if (lower) *lower = (lc_elem_t*)l;
	if (upper) *upper = (lc_elem_t*)u;
	return (lc_elem_t*)p;
}

static lc_elem_t *krmq_rmq(const lc_elem_t *root, const lc_elem_t *lo, const lc_elem_t *up) /* CLOSED interval */
{
	const lc_elem_t *p = root, *path[2][KRMQ_MAX_DEPTH], *min;
	int plen[2] = {0, 0}, pcmp[2][KRMQ_MAX_DEPTH], i, cmp, lca;
	if (root == 0) return 0;
	while (p) {
		cmp = lc_elem_cmp(lo, p);
		path[0][plen[0]] = p, pcmp[0][plen[0]++] = cmp;
		if (cmp < 0) p = p->head.p[0];
		else if (cmp > 0) p = p->head.p[1];
		else break;
	}
	p = root;
	while (p) {
		cmp = lc_elem_cmp(up, p);
		path[1][plen[1]] = p, pcmp[1][plen[1]++] = cmp;
		if (cmp < 0) p = p->head.p[0];
		else if (cmp > 0) p = p->head.p[1];
		else break;
	}
	for (i = 0; i < plen[0] && i < plen[1]; ++i) /* find the LCA */
		if (path[0][i] == path[1][i] && pcmp[0][i] <= 0 && pcmp[1][i] >= 0)
			break;
	if (i == plen[0] || i == plen[1]) return 0; /* no elements in the closed interval */
	lca = i, min = path[0][lca];
	for (i = lca + 1; i < plen[0]; ++i) {
		if (pcmp[0][i] <= 0) {
			if (lc_elem_lt2(path[0][i], min)) min = path[0][i];
			if (path[0][i]->head.p[1] && lc_elem_lt2(path[0][i]->head.p[1]->head.s, min))
				min = path[0][i]->head.p[1]->head.s;
		}
	}
	for (i = lca + 1; i < plen[1]; ++i) {
		if (pcmp[1][i] >= 0) {
			if (lc_elem_lt2(path[1][i], min)) min = path[1][i];
			if (path[1][i]->head.p[0] && lc_elem_lt2(path[1][i]->head.p[0]->head.s, min))
				min = path[1][i]->head.p[0]->head.s;
		}
	}
	return (lc_elem_t*)min;
}

static inline void krmq_update_min(lc_elem_t *p, const lc_elem_t *q, const lc_elem_t *r)
{
	p->head.s = !q || lc_elem_lt2(p, q->head.s)? p : q->head.s;
	p->head.s = !r || lc_elem_lt2(p->head.s, r->head.s)? p->head.s : r->head.s;
}

/* one rotation: (a,(b,c)q)p => ((a,b)p,c)q */
static inline lc_elem_t *krmq_rotate1(lc_elem_t *p, int dir) /* dir=0 to left; dir=1 to right */
{
	int opp = 1 - dir;
	lc_elem_t *q = p->head.p[opp], *s = p->head.s;
	unsigned size_p = p->head.size;
	p->head.size -= q->head.size - krmq_size_child(q, dir);
	q->head.size = size_p;
	krmq_update_min(p, p->head.p[dir], q->head.p[dir]);
	q->head.s = s;
	p->head.p[opp] = q->head.p[dir];
	q->head.p[dir] = p;
	return q;
}

/* two consecutive rotations: (a,((b,c)r,d)q)p => ((a,b)p,(c,d)q)r */
static inline lc_elem_t *krmq_rotate2(lc_elem_t *p, int dir)
{
	int b1, opp = 1 - dir;
	lc_elem_t *q = p->head.p[opp], *r = q->head.p[dir], *s = p->head.s;
	unsigned size_x_dir = krmq_size_child(r, dir);
	r->head.size = p->head.size;
	p->head.size -= q->head.size - size_x_dir;
	q->head.size -= size_x_dir + 1;
	krmq_update_min(p, p->head.p[dir], r->head.p[dir]);
	krmq_update_min(q, q->head.p[opp], r->head.p[opp]);
	r->head.s = s;
	p->head.p[opp] = r->head.p[dir];
	r->head.p[dir] = p;
	q->head.p[dir] = r->head.p[opp];
	r->head.p[opp] = q;
	b1 = dir == 0? +1 : -1;
	if (r->head.balance == b1) q->head.balance = 0, p->head.balance = -b1;
	else if (r->head.balance == 0) q->head.balance = p->head.balance = 0;
	else q->head.balance = b1, p->head.balance = 0;
	r->head.balance = 0;
	return r;
}

static lc_elem_t *krmq_insert(lc_elem_t **root_, lc_elem_t *x)
{
	unsigned char stack[KRMQ_MAX_DEPTH];
	lc_elem_t *path[KRMQ_MAX_DEPTH];
	lc_elem_t *bp, *bq;
	lc_elem_t *p, *q, *r = 0; /* _r_ is potentially the new root */
	int i, which = 0, top, b1, path_len;
	bp = *root_, bq = 0;
	/* find the insertion location */
	for (p = bp, q = bq, top = path_len = 0; p; q = p, p = p->head.p[which]) {
		int cmp = lc_elem_cmp(x, p);
		if (cmp == 0) return p;
		if (p->head.balance != 0)
			bq = q, bp = p, top = 0;
		stack[top++] = which = (cmp > 0);
		path[path_len++] = p;
	}
	x->head.balance = 0, x->head.size = 1, x->head.p[0] = x->head.p[1] = 0, x->head.s = x;
	if (q == 0) *root_ = x;
	else q->head.p[which] = x;
	if (bp == 0) return x;
	for (i = 0; i < path_len; ++i) ++path[i]->head.size;
	for (i = path_len - 1; i >= 0; --i) {
		krmq_update_min(path[i], path[i]->head.p[0], path[i]->head.p[1]);
		if (path[i]->head.s != x) break;
	}
	for (p = bp, top = 0; p != x; p = p->head.p[stack[top]], ++top) /* update balance factors */
		if (stack[top] == 0) --p->head.balance;
		else ++p->head.balance;
	if (bp->head.balance > -2 && bp->head.balance < 2) return x; /* no re-balance needed */
	/* re-balance */
	which = (bp->head.balance < 0);
	b1 = which == 0? +1 : -1;
	q = bp->head.p[1 - which];
	if (q->head.balance == b1) {
		r = krmq_rotate1(bp, which);
		q->head.balance = bp->head.balance = 0;
	} else r = krmq_rotate2(bp, which);
	if (bq == 0) *root_ = r;
	else bq->head.p[bp != bq->head.p[0]] = r;
	return x;
}

static lc_elem_t *krmq_erase(lc_elem_t **root_, const lc_elem_t *x)
{
	lc_elem_t *p, *path[KRMQ_MAX_DEPTH], fake;
	unsigned char dir[KRMQ_MAX_DEPTH];
	int i, d = 0, cmp;
	fake = **root_, fake.head.p[0] = *root_, fake.head.p[1] = 0;
	if (x) {
		for (cmp = -1, p = &fake; cmp; cmp = lc_elem_cmp(x, p)) {
			int which = (cmp > 0);
			dir[d] = which;
			path[d++] = p;
			p = p->head.p[which];
			if (p == 0) return 0;
		}
	} else {
		for (p = &fake; p; p = p->head.p[0])
			dir[d] = 0, path[d++] = p;
		p = path[--d];
	}
	for (i = 1; i < d; ++i) --path[i]->head.size;
	if (p->head.p[1] == 0) { /* ((1,.)2,3)4 => (1,3)4; p=2 */
		path[d-1]->head.p[dir[d-1]] = p->head.p[0];
	} else {
		lc_elem_t *q = p->head.p[1];
		if (q->head.p[0] == 0) { /* ((1,2)3,4)5 => ((1)2,4)5; p=3 */
			q->head.p[0] = p->head.p[0];
			q->head.balance = p->head.balance;
			path[d-1]->head.p[dir[d-1]] = q;
			path[d] = q, dir[d++] = 1;
			q->head.size = p->head.size - 1;
		} else { /* ((1,((.,2)3,4)5)6,7)8 => ((1,(2,4)5)3,7)8; p=6 */
			lc_elem_t *r;
			int e = d++; /* backup _d_ */
			for (;;) {
				dir[d] = 0;
				path[d++] = q;
				r = q->head.p[0];
				if (r->head.p[0] == 0) break;
				q = r;
			}
			r->head.p[0] = p->head.p[0];
			q->head.p[0] = r->head.p[1];
			r->head.p[1] = p->head.p[1];
			r->head.balance = p->head.balance;
			path[e-1]->head.p[dir[e-1]] = r;
			path[e] = r, dir[e] = 1;
			for (i = e + 1; i < d; ++i) --path[i]->head.size;
			r->head.size = p->head.size - 1;
		}
	}
	for (i = d - 1; i >= 0; --i)
		krmq_update_min(path[i], path[i]->head.p[0], path[i]->head.p[1]);
	while (--d > 0) {
		lc_elem_t *q = path[d];
		int which, other, b1 = 1, b2 = 2;
		which = dir[d], other = 1 - which;
		if (which) b1 = -b1, b2 = -b2;
		q->head.balance += b1;
		if (q->head.balance == b1) break;
		else if (q->head.balance == b2) {
			lc_elem_t *r = q->head.p[other];
			if (r->head.balance == -b1) {
				path[d-1]->head.p[dir[d-1]] = krmq_rotate2(q, which);
			} else {
				path[d-1]->head.p[dir[d-1]] = krmq_rotate1(q, which);
				if (r->head.balance == 0) {
					r->head.balance = -b1;
					q->head.balance = b1;
					break;
				} else r->head.balance = q->head.balance = 0;
			}
		}
	}
	*root_ = fake.head.p[0];
	return p;
}

typedef struct { const lc_elem_t *stack[KRMQ_MAX_DEPTH], **top; } krmq_itr_t;

static int krmq_itr_find(const lc_elem_t *root, const lc_elem_t *x, krmq_itr_t *itr)
{
	const lc_elem_t *p = root;
	itr->top = itr->stack - 1;
	while (p != 0) {
		int cmp;
		*++itr->top = p;
		cmp = lc_elem_cmp(x, p);
		if (cmp < 0) p = p->head.p[0];
		else if (cmp > 0) p = p->head.p[1];
		else break;
	}
	return p? 1 : 0;
}

static int krmq_itr_next_bidir(krmq_itr_t *itr, int dir)
{
	const lc_elem_t *p;
	if (itr->top < itr->stack) return 0;
	dir = !!dir;
	p = (*itr->top)->head.p[dir];
	if (p) { /* go down */
		for (; p; p = p->head.p[!dir])
			*++itr->top = p;
		return 1;
	} else { /* go up */
		do {
			p = *itr->top--;
		} while (itr->top >= itr->stack && p == (*itr->top)->head.p[dir]);
		return itr->top < itr->stack? 0 : 1;
	}
}
#define krmq_itr_prev(itr) krmq_itr_next_bidir(itr, 0)
#define krmq_at(itr) ((itr)->top < (itr)->stack? 0 : *(itr)->top)

/* simple free-list pool standing in for U:kalloc.h KALLOC_POOL (allocation order is unobservable) */
typedef struct { lc_elem_t **blocks; int n_blocks, m_blocks; lc_elem_t *free_list; int used_in_block; } pool_t;
#define POOL_BLOCK 4096
static lc_elem_t *pool_alloc(pool_t *mp)
{
	lc_elem_t *q;
	if (mp->free_list) { q = mp->free_list; mp->free_list = q->head.p[0]; return q; }
	if (mp->n_blocks == 0 || mp->used_in_block == POOL_BLOCK) {
		if (mp->n_blocks == mp->m_blocks) {
			mp->m_blocks = mp->m_blocks? mp->m_blocks<<1 : 8;
			mp->blocks = (lc_elem_t**)realloc(mp->blocks, mp->m_blocks * sizeof(lc_elem_t*));
		}
		mp->blocks[mp->n_blocks++] = (lc_elem_t*)malloc(POOL_BLOCK * sizeof(lc_elem_t));
		mp->used_in_block = 0;
	}
	return &mp->blocks[mp->n_blocks-1][mp->used_in_block++];
}
static void pool_free(pool_t *mp, lc_elem_t *q) { q->head.p[0] = mp->free_list; mp->free_list = q; }
static void pool_destroy(pool_t *mp) { int i; for (i = 0; i < mp->n_blocks; ++i) free(mp->blocks[i]); free(mp->blocks); }

static inline int32_t comput_sc_simple(const mm128_t *ai, const mm128_t *aj, float chn_pen_gap, float chn_pen_skip, int32_t *exact, int32_t *width)
{
	int32_t dq = (int32_t)ai->y - (int32_t)aj->y, dr, dd, dg, q_span, sc;
	dr = (int32_t)(ai->x - aj->x);
	*width = dd = dr > dq? dr - dq : dq - dr;
	dg = dr < dq? dr : dq;
	q_span = aj->y>>32&0xff;
	sc = q_span < dg? q_span : dg;
	if (exact) *exact = (dd == 0 && dg <= q_span);
	if (dd || dq > q_span) {
		float lin_pen, log_pen;
		lin_pen = chn_pen_gap * (float)dd + chn_pen_skip * (float)dg;
		log_pen = dd >= 1? mg_log2(dd + 1) : 0.0f; /* mg_log2() only works for dd>=2 */
		sc -= (int)(lin_pen + .5f * log_pen);
	}
	return sc;
}

mm128_t *mmo_lchain_rmq(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, int min_cnt, int min_sc, float chn_pen_gap, float chn_pen_skip,
                        int64_t n, mm128_t *a, int *n_u_, uint64_t **_u)
{
	int32_t *f,*t, *v, n_u, n_v, mmax_f = 0, max_drop = bw;
	int64_t *p, i, i0, st = 0, st_inner = 0;
	uint64_t *u;
	lc_elem_t *root = 0, *root_inner = 0;
	pool_t mp;

	if (_u) *_u = 0, *n_u_ = 0;
	if (n == 0 || a == 0) {
		free(a);
		return 0;
	}
	if (max_dist < bw) max_dist = bw;
	if (max_dist_inner < 0) max_dist_inner = 0;
	if (max_dist_inner > max_dist) max_dist_inner = max_dist;
	p = (int64_t*)malloc(n * 8);
	f = (int32_t*)malloc(n * 4);
	t = (int32_t*)calloc(n, 4);
	v = (int32_t*)malloc(n * 4);
	memset(&mp, 0, sizeof(mp));

	for (i = i0 = 0; i < n; ++i) {
		int64_t max_j = -1;
		int32_t q_span = a[i].y>>32&0xff, max_f = q_span;
		lc_elem_t s, *q, *r, lo, hi;
		/* add in-range anchors */
		if (i0 < i && a[i0].x != a[i].x) {
			int64_t j;
			for (j = i0; j < i; ++j) {
				q = pool_alloc(&mp);
				q->y = (int32_t)a[j].y, q->i = j, q->pri = -(f[j] + 0.5 * chn_pen_gap * ((int32_t)a[j].x + (int32_t)a[j].y));
				krmq_insert(&root, q);
				if (max_dist_inner > 0) {
					r = pool_alloc(&mp);
					*r = *q;
					krmq_insert(&root_inner, r);
				}
			}
			i0 = i;
		}
		/* get rid of active chains out of range */
		while (st < i && (a[i].x>>32 != a[st].x>>32 || a[i].x > a[st].x + max_dist || krmq_size(root) > (unsigned)cap_rmq_size)) {
			s.y = (int32_t)a[st].y, s.i = st;
			if ((q = krmq_find(root, &s)) != 0) {
				q = krmq_erase(&root, q);
				pool_free(&mp, q);
			}
			++st;
		}
		if (max_dist_inner > 0)  { /* similar to the block above, but applied to the inner tree */
			while (st_inner < i && (a[i].x>>32 != a[st_inner].x>>32 || a[i].x > a[st_inner].x + max_dist_inner || krmq_size(root_inner) > (unsigned)cap_rmq_size)) {
				s.y = (int32_t)a[st_inner].y, s.i = st_inner;
				if ((q = krmq_find(root_inner, &s)) != 0) {
					q = krmq_erase(&root_inner, q);
					pool_free(&mp, q);
				}
				++st_inner;
			}
		}
		/* RMQ */
		lo.i = INT32_MAX, lo.y = (int32_t)a[i].y - max_dist;
		hi.i = 0, hi.y = (int32_t)a[i].y;
		if ((q = krmq_rmq(root, &lo, &hi)) != 0) {
			int32_t sc, exact, width, n_skip = 0;
			int64_t j = q->i;
			assert(q->y >= lo.y && q->y <= hi.y);
			sc = f[j] + comput_sc_simple(&a[i], &a[j], chn_pen_gap, chn_pen_skip, &exact, &width);
			if (width <= bw && sc > max_f) max_f = sc, max_j = j;
			if (!exact && root_inner && (int32_t)a[i].y > 0) {
				lc_elem_t *lo2, *hi2;
				s.y = (int32_t)a[i].y - 1, s.i = n;
				krmq_interval(root_inner, &s, &lo2, &hi2);
				if (lo2) {
					const lc_elem_t *q2;
					int32_t width2;
					krmq_itr_t itr;
					krmq_itr_find(root_inner, lo2, &itr);
					while ((q2 = krmq_at(&itr)) != 0) {
						if (q2->y < (int32_t)a[i].y - max_dist_inner) break;
						j = q2->i;
						sc = f[j] + comput_sc_simple(&a[i], &a[j], chn_pen_gap, chn_pen_skip, 0, &width2);
						if (width2 <= bw) {
							if (sc > max_f) {
								max_f = sc, max_j = j;
								if (n_skip > 0) --n_skip;
							} else if (t[j] == (int32_t)i) {
								if (++n_skip > max_chn_skip)
									break;
							}
							if (p[j] >= 0) t[p[j]] = i;
						}
						if (!krmq_itr_prev(&itr)) break;
					}
				}
			}
		}
		/* set max */
		assert(max_j < 0 || (a[max_j].x < a[i].x && (int32_t)a[max_j].y < (int32_t)a[i].y));
		f[i] = max_f, p[i] = max_j;
		v[i] = max_j >= 0 && v[max_j] > max_f? v[max_j] : max_f;
		if (mmax_f < max_f) mmax_f = max_f;
	}
	pool_destroy(&mp);

	u = mg_chain_backtrack(n, f, p, v, t, min_cnt, min_sc, max_drop, &n_u, &n_v);
	*n_u_ = n_u, *_u = u;
	free(p); free(f); free(t);
	if (n_u == 0) {
		free(a); free(v);
		return 0;
	}
	return compact_a(n_u, u, n_v, v, a);
}
