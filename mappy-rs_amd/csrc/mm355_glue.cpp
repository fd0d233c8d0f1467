// mm355_glue.cpp -- host-resident tail of the MI355X mapping path (SURVEY.md 8a rows a9, a10, a11, a13 and the
// cs/MD strings of 8f-2).  These steps are O(#chains) or O(#chained anchors) per read, strictly sequential and
// tie-order sensitive, so round 1 keeps them on host cores next to the GPU; every heavy loop (hashing, index
// gather, anchor sort, chaining DP, backtrack, banded extension) runs in the HIP kernels.
// minimap2 2.26 units whose observable behaviour is reproduced (reference call site: mm_map at
// /root/reference/src/lib.rs:482 and :587):
//   U:lchain.c::mg_lchain_rmq (+U:krmq.h), mg_chain_backtrack, compact_a      -> rechain_rmq()
//   U:hit.c::mm_gen_regs .. mm_set_mapq, U:esterr.c::mm_est_err               -> regs_*()
//   U:align.c::mm_align_skeleton / mm_align1 / mm_test_zdrop / mm_update_extra / mm_align1_inv -> align_*()
//   U:format.c::mm_gen_cs / mm_gen_MD                                          -> gen_cs() / gen_md()
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include <algorithm>
#include "mm355_glue.h"
#include "mm355_prof.h"
#include <atomic>
#include <chrono>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#define PARENT_UNSET   (-1)
#define PARENT_TMP_PRI (-2)
#define EZ_RIGHT       0x02
#define EZ_APPROX_MAX  0x08
#define EZ_EXTZ_ONLY   0x40
#define EZ_REV_CIGAR   0x80

// ================================================================== chaining tail on the host (re-chain only)
static int64_t bk_end(int32_t max_drop, const mm128 *z, const int32_t *f, const int64_t *p, int32_t *t, int64_t k)
{
	int64_t i = (int64_t)z[k].y, end_i = -1, max_i = i;
	int32_t max_s = 0;
	if (i < 0 || t[i] != 0) return i;
	do {
		int32_t s;
		t[i] = 2;
		end_i = i = p[i];
		s = i < 0? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
		if (s > max_s) max_s = s, max_i = i;
		else if (max_s - s > max_drop) break;
	} while (i >= 0 && t[i] == 0);
	for (i = (int64_t)z[k].y; i >= 0 && i != end_i; i = p[i]) t[i] = 0;
	return max_i;
}

static void chain_backtrack(int64_t n, const int32_t *f, const int64_t *p, std::vector<int32_t> &v, int32_t *t, int32_t min_cnt, int32_t min_sc,
                            int32_t max_drop, std::vector<uint64_t> &u)
{
	static thread_local std::vector<mm128> z;   // (scratch kept per host thread: this runs once per read)
	z.clear();
	u.clear(); v.clear();
	for (int64_t i = 0; i < n; ++i) if (f[i] >= min_sc) { mm128 e; e.x = (uint64_t)f[i]; e.y = (uint64_t)i; z.push_back(e); }
	if (z.empty()) return;
	mm_radix_sort(z.data(), z.data() + z.size(), mm_key_x());
	memset(t, 0, n * 4);
	for (int64_t k = (int64_t)z.size() - 1; k >= 0; --k) {
		if (t[z[k].y] == 0) {
			size_t n_v0 = v.size();
			int64_t end_i = bk_end(max_drop, z.data(), f, p, t, k), i;
			for (i = (int64_t)z[k].y; i != end_i; i = p[i]) v.push_back((int32_t)i), t[i] = 1;
			int32_t sc = i < 0? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
			if (sc >= min_sc && v.size() > n_v0 && (int32_t)(v.size() - n_v0) >= min_cnt)
				u.push_back((uint64_t)(uint32_t)sc << 32 | (uint32_t)(v.size() - n_v0));
			else v.resize(n_v0);
		}
	}
}

static void compact_chains(std::vector<uint64_t> &u, const std::vector<int32_t> &v, std::vector<mm128> &a)
{
	const int n_u = (int)u.size();
	static thread_local std::vector<mm128> b, w, out;   // (scratch kept per host thread; `out` / `u2` trade storage with the caller's vectors)
	static thread_local std::vector<uint64_t> u2;
	b.resize(v.size()); w.resize(n_u); out.resize(v.size()); u2.resize(n_u);
	int64_t k = 0;
	for (int i = 0; i < n_u; ++i) {
		int32_t k0 = (int32_t)k, ni = (int32_t)u[i];
		for (int j = 0; j < ni; ++j) b[k++] = a[v[k0 + (ni - j - 1)]];
	}
	k = 0;
	for (int i = 0; i < n_u; ++i) { w[i].x = b[k].x; w[i].y = (uint64_t)k << 32 | (uint32_t)i; k += (int32_t)u[i]; }
	mm_radix_sort(w.data(), w.data() + n_u, mm_key_x());
	k = 0;
	for (int i = 0; i < n_u; ++i) {
		int32_t j = (int32_t)(uint32_t)w[i].y, n = (int32_t)(uint32_t)u[j];
		u2[i] = u[j];
		memcpy(&out[k], &b[w[i].y >> 32], (size_t)n * sizeof(mm128));
		k += n;
	}
	u.swap(u2);
	a.swap(out);
}

// ---- U:krmq.h: AVL tree keyed (y,i) with range-min on pri; tree shape decides equal-priority ties, so literal ----
#define KRMQ_MAX_DEPTH 64
struct LcElem {
	int32_t y; int64_t i; double pri;
	LcElem *c[2], *s; signed char balance; unsigned size;
};
static inline int lc_cmp(const LcElem *a, const LcElem *b) { return a->y < b->y? -1 : a->y > b->y? 1 : (a->i > b->i) - (a->i < b->i); }
static inline bool lc_lt2(const LcElem *a, const LcElem *b) { return a->pri < b->pri; }
static inline unsigned lc_size(const LcElem *p) { return p? p->size : 0; }
static inline unsigned lc_size_child(const LcElem *q, int i) { return q->c[i]? q->c[i]->size : 0; }

static LcElem *krmq_find(LcElem *root, const LcElem *x)
{
	LcElem *p = root;
	while (p) { int cmp = lc_cmp(x, p); if (cmp < 0) p = p->c[0]; else if (cmp > 0) p = p->c[1]; else break; }
	return p;
}
static void krmq_interval(LcElem *root, const LcElem *x, LcElem **lower, LcElem **upper)
{
	LcElem *p = root, *l = 0, *u = 0;
	while (p) {
		int cmp = lc_cmp(x, p);
		if (cmp < 0) u = p, p = p->c[0];
		else if (cmp > 0) l = p, p = p->c[1];
		else { l = u = p; break; }
	}
	*lower = l; *upper = u;
}
static LcElem *krmq_rmq(LcElem *root, const LcElem *lo, const LcElem *up)
{
	LcElem *p = root, *path[2][KRMQ_MAX_DEPTH], *min;
	int plen[2] = {0, 0}, pcmp[2][KRMQ_MAX_DEPTH], i, cmp, lca;
	if (root == 0) return 0;
	while (p) {
		cmp = lc_cmp(lo, p);
		path[0][plen[0]] = p, pcmp[0][plen[0]++] = cmp;
		if (cmp < 0) p = p->c[0]; else if (cmp > 0) p = p->c[1]; else break;
	}
	p = root;
	while (p) {
		cmp = lc_cmp(up, p);
		path[1][plen[1]] = p, pcmp[1][plen[1]++] = cmp;
		if (cmp < 0) p = p->c[0]; else if (cmp > 0) p = p->c[1]; else break;
	}
	for (i = 0; i < plen[0] && i < plen[1]; ++i)
		if (path[0][i] == path[1][i] && pcmp[0][i] <= 0 && pcmp[1][i] >= 0) break;
	if (i == plen[0] || i == plen[1]) return 0;
	lca = i, min = path[0][lca];
	for (i = lca + 1; i < plen[0]; ++i) {
		if (pcmp[0][i] <= 0) {
			if (lc_lt2(path[0][i], min)) min = path[0][i];
			if (path[0][i]->c[1] && lc_lt2(path[0][i]->c[1]->s, min)) min = path[0][i]->c[1]->s;
		}
	}
	for (i = lca + 1; i < plen[1]; ++i) {
		if (pcmp[1][i] >= 0) {
			if (lc_lt2(path[1][i], min)) min = path[1][i];
			if (path[1][i]->c[0] && lc_lt2(path[1][i]->c[0]->s, min)) min = path[1][i]->c[0]->s;
		}
	}
	return min;
}
static inline void krmq_update_min(LcElem *p, const LcElem *q, const LcElem *r)
{
	p->s = !q || lc_lt2(p, q->s)? p : q->s;
	p->s = !r || lc_lt2(p->s, r->s)? p->s : r->s;
}
static inline LcElem *krmq_rotate1(LcElem *p, int dir)
{
	int opp = 1 - dir;
	LcElem *q = p->c[opp], *s = p->s;
	unsigned size_p = p->size;
	p->size -= q->size - lc_size_child(q, dir);
	q->size = size_p;
	krmq_update_min(p, p->c[dir], q->c[dir]);
	q->s = s;
	p->c[opp] = q->c[dir];
	q->c[dir] = p;
	return q;
}
static inline LcElem *krmq_rotate2(LcElem *p, int dir)
{
	int b1, opp = 1 - dir;
	LcElem *q = p->c[opp], *r = q->c[dir], *s = p->s;
	unsigned size_x_dir = lc_size_child(r, dir);
	r->size = p->size;
	p->size -= q->size - size_x_dir;
	q->size -= size_x_dir + 1;
	krmq_update_min(p, p->c[dir], r->c[dir]);
	krmq_update_min(q, q->c[opp], r->c[opp]);
	r->s = s;
	p->c[opp] = r->c[dir];
	r->c[dir] = p;
	q->c[dir] = r->c[opp];
	r->c[opp] = q;
	b1 = dir == 0? +1 : -1;
	if (r->balance == b1) q->balance = 0, p->balance = -b1;
	else if (r->balance == 0) q->balance = p->balance = 0;
	else q->balance = b1, p->balance = 0;
	r->balance = 0;
	return r;
}
static void krmq_insert(LcElem **root_, LcElem *x)
{
	unsigned char stack[KRMQ_MAX_DEPTH];
	LcElem *path[KRMQ_MAX_DEPTH];
	LcElem *bp, *bq, *p, *q, *r = 0;
	int i, which = 0, top, b1, path_len;
	bp = *root_, bq = 0;
	for (p = bp, q = bq, top = path_len = 0; p; q = p, p = p->c[which]) {
		int cmp = lc_cmp(x, p);
		if (cmp == 0) return;
		if (p->balance != 0) bq = q, bp = p, top = 0;
		stack[top++] = which = (cmp > 0);
		path[path_len++] = p;
	}
	x->balance = 0, x->size = 1, x->c[0] = x->c[1] = 0, x->s = x;
	if (q == 0) *root_ = x;
	else q->c[which] = x;
	if (bp == 0) return;
	for (i = 0; i < path_len; ++i) ++path[i]->size;
	for (i = path_len - 1; i >= 0; --i) {
		krmq_update_min(path[i], path[i]->c[0], path[i]->c[1]);
		if (path[i]->s != x) break;
	}
	for (p = bp, top = 0; p != x; p = p->c[stack[top]], ++top)
		if (stack[top] == 0) --p->balance; else ++p->balance;
	if (bp->balance > -2 && bp->balance < 2) return;
	which = (bp->balance < 0);
	b1 = which == 0? +1 : -1;
	q = bp->c[1 - which];
	if (q->balance == b1) { r = krmq_rotate1(bp, which); q->balance = bp->balance = 0; }
	else r = krmq_rotate2(bp, which);
	if (bq == 0) *root_ = r;
	else bq->c[bp != bq->c[0]] = r;
}
static LcElem *krmq_erase(LcElem **root_, const LcElem *x)
{
	LcElem *p, *path[KRMQ_MAX_DEPTH], fake;
	unsigned char dir[KRMQ_MAX_DEPTH];
	int i, d = 0, cmp;
	fake = **root_, fake.c[0] = *root_, fake.c[1] = 0;
	for (cmp = -1, p = &fake; cmp; cmp = lc_cmp(x, p)) {
		int which = (cmp > 0);
		dir[d] = which; path[d++] = p;
		p = p->c[which];
		if (p == 0) return 0;
	}
	for (i = 1; i < d; ++i) --path[i]->size;
	if (p->c[1] == 0) {
		path[d-1]->c[dir[d-1]] = p->c[0];
	} else {
		LcElem *q = p->c[1];
		if (q->c[0] == 0) {
			q->c[0] = p->c[0];
			q->balance = p->balance;
			path[d-1]->c[dir[d-1]] = q;
			path[d] = q, dir[d++] = 1;
			q->size = p->size - 1;
		} else {
			LcElem *r;
			int e = d++;
			for (;;) {
				dir[d] = 0; path[d++] = q;
				r = q->c[0];
				if (r->c[0] == 0) break;
				q = r;
			}
			r->c[0] = p->c[0];
			q->c[0] = r->c[1];
			r->c[1] = p->c[1];
			r->balance = p->balance;
			path[e-1]->c[dir[e-1]] = r;
			path[e] = r, dir[e] = 1;
			for (i = e + 1; i < d; ++i) --path[i]->size;
			r->size = p->size - 1;
		}
	}
	for (i = d - 1; i >= 0; --i) krmq_update_min(path[i], path[i]->c[0], path[i]->c[1]);
	while (--d > 0) {
		LcElem *q = path[d];
		int which, other, b1 = 1, b2 = 2;
		which = dir[d], other = 1 - which;
		if (which) b1 = -b1, b2 = -b2;
		q->balance += b1;
		if (q->balance == b1) break;
		else if (q->balance == b2) {
			LcElem *r = q->c[other];
			if (r->balance == -b1) {
				path[d-1]->c[dir[d-1]] = krmq_rotate2(q, which);
			} else {
				path[d-1]->c[dir[d-1]] = krmq_rotate1(q, which);
				if (r->balance == 0) { r->balance = -b1; q->balance = b1; break; }
				else r->balance = q->balance = 0;
			}
		}
	}
	*root_ = fake.c[0];
	return p;
}
struct KrmqItr { const LcElem *stack[KRMQ_MAX_DEPTH], **top; };
static void krmq_itr_find(const LcElem *root, const LcElem *x, KrmqItr *itr)
{
	const LcElem *p = root;
	itr->top = itr->stack - 1;
	while (p) {
		*++itr->top = p;
		int cmp = lc_cmp(x, p);
		if (cmp < 0) p = p->c[0]; else if (cmp > 0) p = p->c[1]; else break;
	}
}
static int krmq_itr_prev(KrmqItr *itr)
{
	const LcElem *p;
	if (itr->top < itr->stack) return 0;
	p = (*itr->top)->c[0];
	if (p) { for (; p; p = p->c[1]) *++itr->top = p; return 1; }
	do { p = *itr->top--; } while (itr->top >= itr->stack && p == (*itr->top)->c[0]);
	return itr->top < itr->stack? 0 : 1;
}

struct LcPool {   // one per host thread, blocks kept between reads (a fresh 4096-node block per read was a 200 KB malloc/free -- mmap, page faults -- per read)
	std::vector<LcElem*> blocks; LcElem *free_list = 0; int used = 4096; size_t cur = 0;
	void reset() { free_list = 0; cur = 0; used = blocks.empty()? 4096 : 0; }
	LcElem *alloc() {
		if (free_list) { LcElem *q = free_list; free_list = q->c[0]; return q; }
		if (used == 4096) {
			if (!blocks.empty() && cur + 1 < blocks.size()) ++cur;
			else { blocks.push_back((LcElem*)malloc(4096 * sizeof(LcElem))); cur = blocks.size() - 1; }
			used = 0;
		}
		return &blocks[cur][used++];
	}
	void release(LcElem *q) { q->c[0] = free_list; free_list = q; }
	~LcPool() { for (LcElem *b : blocks) free(b); }
};

static inline int32_t comput_sc_simple(const mm128 *ai, const mm128 *aj, float pen_gap, float pen_skip, int32_t *exact, int32_t *width)
{
	int32_t dq = (int32_t)ai->y - (int32_t)aj->y, dr, dd, dg, q_span, sc;
	dr = (int32_t)(ai->x - aj->x);
	*width = dd = dr > dq? dr - dq : dq - dr;
	dg = dr < dq? dr : dq;
	q_span = (int32_t)(aj->y >> 32 & 0xff);
	sc = q_span < dg? q_span : dg;
	if (exact) *exact = (dd == 0 && dg <= q_span);
	if (dd || dq > q_span) {
		float lin_pen, log_pen;
		lin_pen = pen_gap * (float)dd + pen_skip * (float)dg;
		log_pen = dd >= 1? mm_log2f_approx((float)(dd + 1)) : 0.0f;
		sc -= (int)(lin_pen + .5f * log_pen);
	}
	return sc;
}

// U:lchain.c::mg_lchain_rmq on rs.a (already sorted); replaces rs.a / rs.u with the re-chained result
static void rechain_rmq(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, int min_cnt, int min_sc,
                        float pen_gap, float pen_skip, std::vector<mm128> &av, std::vector<uint64_t> &u)
{
	const int64_t n = (int64_t)av.size();
	u.clear();
	if (n == 0) return;
	mm128 *a = av.data();
	int32_t max_drop = bw;
	int64_t i0, st = 0, st_inner = 0;
	LcElem *root = 0;
	static thread_local LcPool mp;
	mp.reset();
	if (max_dist < bw) max_dist = bw;
	if (max_dist_inner < 0) max_dist_inner = 0;
	if (max_dist_inner > max_dist) max_dist_inner = max_dist;
	static thread_local std::vector<int64_t> p; static thread_local std::vector<int32_t> f, t, vv;
	p.resize(n); f.resize(n); t.assign(n, 0); vv.clear();
	// The second tree of U:lchain.c (root_inner: the anchors within max_dist_inner) is only ever walked in key order -- krmq_interval for
	// the largest key <= (y - 1, n), then krmq_itr_prev -- and asked for its size: nothing of it depends on the tree's shape.  It is kept as
	// a sorted vector of keys (y << 32 | i; a few dozen entries): same members, same order, half of the AVL work of a read gone.
	static thread_local std::vector<int64_t> inn;
	inn.clear();
	auto ikey = [](int32_t y, int64_t j) { return (int64_t)((uint64_t)(int64_t)y << 32 | (uint64_t)(uint32_t)j); };
	i0 = 0;
	for (int64_t i = 0; i < n; ++i) {
		int64_t max_j = -1;
		int32_t q_span = (int32_t)(a[i].y >> 32 & 0xff), max_f = q_span;
		LcElem s, *q, lo, hi;
		if (i0 < i && a[i0].x != a[i].x) {
			for (int64_t j = i0; j < i; ++j) {
				q = mp.alloc();
				q->y = (int32_t)a[j].y, q->i = j, q->pri = -(f[j] + 0.5 * pen_gap * ((int32_t)a[j].x + (int32_t)a[j].y));
				krmq_insert(&root, q);
				if (max_dist_inner > 0) { const int64_t k = ikey((int32_t)a[j].y, j); inn.insert(std::lower_bound(inn.begin(), inn.end(), k), k); }
			}
			i0 = i;
		}
		while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + (uint64_t)max_dist || lc_size(root) > (unsigned)cap_rmq_size)) {
			s.y = (int32_t)a[st].y, s.i = st;
			if ((q = krmq_find(root, &s)) != 0) { q = krmq_erase(&root, q); mp.release(q); }
			++st;
		}
		if (max_dist_inner > 0) {
			while (st_inner < i && (a[i].x >> 32 != a[st_inner].x >> 32 || a[i].x > a[st_inner].x + (uint64_t)max_dist_inner || inn.size() > (size_t)(unsigned)cap_rmq_size)) {
				const int64_t k = ikey((int32_t)a[st_inner].y, st_inner);
				auto it = std::lower_bound(inn.begin(), inn.end(), k);
				if (it != inn.end() && *it == k) inn.erase(it);
				++st_inner;
			}
		}
		lo.i = INT32_MAX, lo.y = (int32_t)a[i].y - max_dist;
		hi.i = 0, hi.y = (int32_t)a[i].y;
		if ((q = krmq_rmq(root, &lo, &hi)) != 0) {
			int32_t sc, exact, width, n_skip = 0;
			int64_t j = q->i;
			sc = f[j] + comput_sc_simple(&a[i], &a[j], pen_gap, pen_skip, &exact, &width);
			if (width <= bw && sc > max_f) max_f = sc, max_j = j;
			if (!exact && !inn.empty() && (int32_t)a[i].y > 0) {
				// largest key <= (y - 1, n), then downwards
				size_t idx = (size_t)(std::upper_bound(inn.begin(), inn.end(), ikey((int32_t)a[i].y - 1, n)) - inn.begin());
				while (idx > 0) {
					--idx;
					const int32_t y2 = (int32_t)(inn[idx] >> 32);
					int32_t width2;
					if (y2 < (int32_t)a[i].y - max_dist_inner) break;
					j = (int64_t)(uint32_t)inn[idx];
					sc = f[j] + comput_sc_simple(&a[i], &a[j], pen_gap, pen_skip, 0, &width2);
					if (width2 <= bw) {
						if (sc > max_f) { max_f = sc, max_j = j; if (n_skip > 0) --n_skip; }
						else if (t[j] == (int32_t)i) { if (++n_skip > max_chn_skip) break; }
						if (p[j] >= 0) t[p[j]] = (int32_t)i;
					}
				}
			}
		}
		f[i] = max_f, p[i] = max_j;
	}
	chain_backtrack(n, f.data(), p.data(), vv, t.data(), min_cnt, min_sc, max_drop, u);
	if (u.empty()) { av.clear(); return; }
	compact_chains(u, vv, av);
}

// ================================================================== regions (U:hit.c, U:esterr.c)
static inline uint32_t wang32(uint32_t key)
{
	key += ~(key << 15); key ^= (key >> 10); key += (key << 3); key ^= (key >> 6); key += ~(key << 11); key ^= (key >> 16);
	return key;
}
static inline uint64_t hash64u(uint64_t key)
{
	key = (~key + (key << 21)); key = key ^ key >> 24; key = ((key + (key << 3)) + (key << 8)); key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)); key = key ^ key >> 28; key = (key + (key << 31));
	return key;
}

static void reg_fuzzy_len(Reg *r, const mm128 *a)
{
	r->mlen = r->blen = 0;
	if (r->cnt <= 0) return;
	r->mlen = r->blen = (int32_t)(a[r->as].y >> 32 & 0xff);
	for (int i = r->as + 1; i < r->as + r->cnt; ++i) {
		int span = (int)(a[i].y >> 32 & 0xff);
		int tl = (int32_t)a[i].x - (int32_t)a[i-1].x;
		int ql = (int32_t)a[i].y - (int32_t)a[i-1].y;
		r->blen += tl > ql? tl : ql;
		r->mlen += tl > span && ql > span? span : tl < ql? tl : ql;
	}
}

static void reg_set_coor(Reg *r, int32_t qlen, const mm128 *a)
{
	int32_t k = r->as, q_span = (int32_t)(a[k].y >> 32 & 0xff);
	r->rev = (uint32_t)(a[k].x >> 63);
	r->rid = (int32_t)(a[k].x << 1 >> 33);
	r->rs = (int32_t)a[k].x + 1 > q_span? (int32_t)a[k].x + 1 - q_span : 0;
	r->re = (int32_t)a[k + r->cnt - 1].x + 1;
	if (!r->rev) {
		r->qs = (int32_t)a[k].y + 1 - q_span;
		r->qe = (int32_t)a[k + r->cnt - 1].y + 1;
	} else {
		r->qs = qlen - ((int32_t)a[k + r->cnt - 1].y + 1);
		r->qe = qlen - ((int32_t)a[k].y + 1 - q_span);
	}
	reg_fuzzy_len(r, a);
}

static void gen_regs(uint32_t hash, int qlen, ReadState &rs)
{
	const int n_u = (int)rs.u.size();
	const mm128 *a = rs.a.data();
	rs.regs.clear();
	if (n_u == 0) return;
	std::vector<mm128> z(n_u);
	int k = 0;
	for (int i = 0; i < n_u; ++i) {
		uint32_t h = (uint32_t)hash64u((hash64u(a[k].x) + hash64u(a[k].y)) ^ hash);
		z[i].x = rs.u[i] ^ h;
		z[i].y = (uint64_t)k << 32 | (uint32_t)(int32_t)rs.u[i];
		k += (int32_t)rs.u[i];
	}
	mm_radix_sort(z.data(), z.data() + n_u, mm_key_x());
	for (int i = 0; i < n_u >> 1; ++i) std::swap(z[i], z[n_u - 1 - i]);
	rs.regs.resize(n_u);
	for (int i = 0; i < n_u; ++i) {
		Reg *ri = &rs.regs[i];
		*ri = Reg();
		ri->id = i;
		ri->parent = PARENT_UNSET;
		ri->score = ri->score0 = (int32_t)(z[i].x >> 32);
		ri->hash = (uint32_t)z[i].x;
		ri->cnt = (int32_t)z[i].y;
		ri->as = (int32_t)(z[i].y >> 32);
		ri->div = -1.0f;
		reg_set_coor(ri, qlen, a);
	}
}

static void split_reg(Reg *r, Reg *r2, int n, int qlen, const mm128 *a)
{
	if (n <= 0 || n >= r->cnt) return;
	*r2 = *r;
	r2->id = -1;
	r2->sam_pri = 0;
	r2->p = 0;
	r2->task = -1;
	r2->split_inv = 0;
	r2->cnt = r->cnt - n;
	r2->score = (int32_t)(r->score * ((float)r2->cnt / r->cnt) + .499);
	r2->as = r->as + n;
	if (r->parent == r->id) r2->parent = PARENT_TMP_PRI;
	reg_set_coor(r2, qlen, a);
	r->cnt -= r2->cnt;
	r->score -= r2->score;
	reg_set_coor(r, qlen, a);
	r->split |= 1, r2->split |= 2;
}

static void set_parent(float mask_level, int mask_len, int n, Reg *r, int sub_diff, int hard_mask_level)
{
	if (n <= 0) return;
	for (int i = 0; i < n; ++i) r[i].id = i;
	std::vector<uint64_t> cov(n);
	std::vector<int> w(n);
	int k = 1, j;
	w[0] = 0, r[0].parent = 0;
	for (int i = 1; i < n; ++i) {
		Reg *ri = &r[i];
		int si = ri->qs, ei = ri->qe, n_cov = 0, uncov_len = 0;
		if (hard_mask_level) goto skip_uncov;
		for (j = 0; j < k; ++j) {
			Reg *rp = &r[w[j]];
			int sj = rp->qs, ej = rp->qe;
			if (ej <= si || sj >= ei) continue;
			if (sj < si) sj = si;
			if (ej > ei) ej = ei;
			cov[n_cov++] = (uint64_t)sj << 32 | (uint32_t)ej;
		}
		if (n_cov == 0) {
			goto set_parent_test;
		} else {
			int x = si;
			mm_radix_sort(cov.data(), cov.data() + n_cov, mm_key_u64());
			for (int jj = 0; jj < n_cov; ++jj) {
				if ((int)(cov[jj] >> 32) > x) uncov_len += (int)(cov[jj] >> 32) - x;
				x = (int32_t)cov[jj] > x? (int32_t)cov[jj] : x;
			}
			if (ei > x) uncov_len += ei - x;
		}
skip_uncov:
		for (j = 0; j < k; ++j) {
			Reg *rp = &r[w[j]];
			int sj = rp->qs, ej = rp->qe, min, max, ol;
			if (ej <= si || sj >= ei) continue;
			min = ej - sj < ei - si? ej - sj : ei - si;
			max = ej - sj > ei - si? ej - sj : ei - si;
			ol = si < sj? (ei < sj? 0 : ei < ej? ei - sj : ej - sj) : (ej < si? 0 : ej < ei? ej - si : ei - si);
			if ((float)ol / min - (float)uncov_len / max > mask_level && uncov_len <= mask_len) {
				int cnt_sub = 0, sci = ri->score;
				ri->parent = rp->parent;
				rp->subsc = rp->subsc > sci? rp->subsc : sci;
				if (ri->cnt >= rp->cnt) cnt_sub = 1;
				if (rp->p && ri->p && (rp->rid != ri->rid || rp->rs != ri->rs || rp->re != ri->re || ol != min)) {
					sci = ri->p->dp_max;
					rp->p->dp_max2 = rp->p->dp_max2 > sci? rp->p->dp_max2 : sci;
					if (rp->p->dp_max - ri->p->dp_max <= sub_diff) cnt_sub = 1;
				}
				if (cnt_sub) ++rp->n_sub;
				break;
			}
		}
set_parent_test:
		if (j == k) w[k++] = i, ri->parent = i, ri->n_sub = 0;
	}
}

static int set_sam_pri(int n, Reg *r)
{
	int n_pri = 0;
	for (int i = 0; i < n; ++i)
		if (r[i].id == r[i].parent) { ++n_pri; r[i].sam_pri = (n_pri == 1); }
		else r[i].sam_pri = 0;
	return n_pri;
}

static void sync_regs(int n_regs, Reg *regs)
{
	int max_id = -1;
	if (n_regs <= 0) return;
	for (int i = 0; i < n_regs; ++i) max_id = max_id > regs[i].id? max_id : regs[i].id;
	int n_tmp = max_id + 1;
	std::vector<int> tmp(n_tmp > 0? n_tmp : 1, -1);
	for (int i = 0; i < n_regs; ++i) if (regs[i].id >= 0) tmp[regs[i].id] = i;
	for (int i = 0; i < n_regs; ++i) {
		Reg *r = &regs[i];
		r->id = i;
		if (r->parent == PARENT_TMP_PRI) r->parent = i;
		else if (r->parent >= 0 && r->parent < n_tmp && tmp[r->parent] >= 0) r->parent = tmp[r->parent];
		else r->parent = PARENT_UNSET;
	}
	set_sam_pri(n_regs, regs);
}

static void select_sub(float pri_ratio, int min_diff, int best_n, int check_strand, int min_strand_sc, int *n_, Reg *r)
{
	if (pri_ratio > 0.0f && *n_ > 0) {
		int k = 0, n = *n_, n_2nd = 0;
		for (int i = 0; i < n; ++i) {
			int p = r[i].parent;
			if (p == i || r[i].inv) {
				r[k++] = r[i];
			} else if ((r[i].score >= r[p].score * pri_ratio || r[i].score + min_diff >= r[p].score) && n_2nd < best_n) {
				if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs && r[i].re == r[p].re))
					r[k++] = r[i], ++n_2nd;
				else if (r[i].p) delete r[i].p;
			} else if (check_strand && n_2nd < best_n && r[i].score > min_strand_sc && r[p].rev != r[i].rev) {
				r[i].strand_retained = 1;
				r[k++] = r[i], ++n_2nd;
			} else if (r[i].p) delete r[i].p;
		}
		if (k != n) sync_regs(k, r);
		*n_ = k;
	}
}

static int filter_strand_retained(int n_regs, Reg *r)
{
	int k = 0;
	for (int i = 0; i < n_regs; ++i) {
		int p = r[i].parent;
		if (!r[i].strand_retained || r[i].div < r[p].div * 5.0f || r[i].div < 0.01f) {
			if (k < i) r[k++] = r[i]; else ++k;
		}
	}
	return k;
}

static void filter_regs(const mm355_mapopt_t *opt, int qlen, int *n_regs, Reg *regs)
{
	int k = 0;
	for (int i = 0; i < *n_regs; ++i) {
		Reg *r = &regs[i];
		int flt = 0;
		if (!r->inv && !r->seg_split && r->cnt < opt->min_cnt) flt = 1;
		if (r->p) {
			if (r->mlen < opt->min_chain_score) flt = 1;
			else if (r->p->dp_max < opt->min_dp_max) flt = 1;
			else if (r->qs > qlen * opt->max_clip_ratio && qlen - r->qe > qlen * opt->max_clip_ratio) flt = 1;
		}
		if (flt) { if (r->p) delete r->p; r->p = 0; }
		else { if (k < i) regs[k++] = regs[i]; else ++k; }
	}
	*n_regs = k;
}

static void hit_sort(int *n_regs, Reg *r)
{
	int n = *n_regs, n_aux = 0;
	if (n <= 1) return;
	std::vector<mm128> aux(n);
	std::vector<Reg> t(n);
	for (int i = 0; i < n; ++i) {
		if (r[i].inv || r[i].cnt > 0) {
			int score = r[i].p? r[i].p->dp_max : r[i].score;
			aux[n_aux].x = (uint64_t)(uint32_t)score << 32 | r[i].hash;
			aux[n_aux++].y = (uint64_t)i;
		} else if (r[i].p) { delete r[i].p; r[i].p = 0; }
	}
	mm_radix_sort(aux.data(), aux.data() + n_aux, mm_key_x());
	for (int i = n_aux - 1; i >= 0; --i) t[n_aux - 1 - i] = r[aux[i].y];
	for (int i = 0; i < n_aux; ++i) r[i] = t[i];
	*n_regs = n_aux;
}

static int squeeze_a(int n_regs, Reg *regs, mm128 *a)
{
	int as = 0;
	std::vector<uint64_t> aux(n_regs > 0? n_regs : 1);
	for (int i = 0; i < n_regs; ++i) aux[i] = (uint64_t)(uint32_t)regs[i].as << 32 | (uint32_t)i;
	std::sort(aux.begin(), aux.begin() + n_regs);   // keys are unique: any sort gives radix_sort_64's result
	for (int i = 0; i < n_regs; ++i) {
		Reg *r = &regs[(int32_t)(uint32_t)aux[i]];
		if (r->as != as) { memmove(&a[as], &a[r->as], (size_t)r->cnt * 16); r->as = as; }
		as += r->cnt;
	}
	return as;
}

static void set_inv_mapq(int n_regs, Reg *regs)
{
	int i, n_aux = 0;
	if (n_regs < 3) return;
	for (i = 0; i < n_regs; ++i) if (regs[i].inv) break;
	if (i == n_regs) return;
	std::vector<mm128> aux(n_regs);
	for (i = 0; i < n_regs; ++i)
		if (regs[i].parent == i || regs[i].parent < 0)
			aux[n_aux].y = (uint64_t)i, aux[n_aux++].x = (uint64_t)(uint32_t)regs[i].rid << 32 | (uint32_t)regs[i].rs;
	mm_radix_sort(aux.data(), aux.data() + n_aux, mm_key_x());
	for (i = 1; i < n_aux - 1; ++i) {
		Reg *inv = &regs[aux[i].y];
		if (inv->inv) {
			Reg *l = &regs[aux[i-1].y], *r = &regs[aux[i+1].y];
			inv->mapq = l->mapq < r->mapq? l->mapq : r->mapq;
		}
	}
}

static void set_mapq(int n_regs, Reg *regs, int min_chain_sc, int match_sc, int rep_len)
{
	static const float q_coef = 40.0f;
	int64_t sum_sc = 0;
	float uniq_ratio;
	if (n_regs == 0) return;
	for (int i = 0; i < n_regs; ++i) if (regs[i].parent == regs[i].id) sum_sc += regs[i].score;
	uniq_ratio = (float)sum_sc / (sum_sc + rep_len);
	for (int i = 0; i < n_regs; ++i) {
		Reg *r = &regs[i];
		if (r->inv) {
			r->mapq = 0;
		} else if (r->parent == r->id) {
			int mapq, subsc;
			float pen_s1 = (r->score > 100? 1.0f : 0.01f * r->score) * uniq_ratio;
			float pen_cm = r->cnt > 10? 1.0f : 0.1f * r->cnt;
			pen_cm = pen_s1 < pen_cm? pen_s1 : pen_cm;
			subsc = r->subsc > min_chain_sc? r->subsc : min_chain_sc;
			if (r->p && r->p->dp_max2 > 0 && r->p->dp_max > 0) {
				float identity = (float)r->mlen / r->blen;
				float x = (float)r->p->dp_max2 * subsc / r->p->dp_max / r->score0;
				mapq = (int)(identity * pen_cm * q_coef * (1.0f - x * x) * logf((float)r->p->dp_max / match_sc));
				int mapq_alt = (int)(6.02f * identity * identity * (r->p->dp_max - r->p->dp_max2) / match_sc + .499f);
				mapq = mapq < mapq_alt? mapq : mapq_alt;
			} else {
				float x = (float)subsc / r->score0;
				if (r->p) {
					float identity = (float)r->mlen / r->blen;
					mapq = (int)(identity * pen_cm * q_coef * (1.0f - x) * logf((float)r->p->dp_max / match_sc));
				} else mapq = (int)(pen_cm * q_coef * (1.0f - x) * logf(r->score));
			}
			mapq -= (int)(4.343f * logf(r->n_sub + 1) + .499f);
			mapq = mapq > 0? mapq : 0;
			r->mapq = mapq < 60? mapq : 60;
			if (r->p && r->p->dp_max > r->p->dp_max2 && r->mapq == 0) r->mapq = 1;
		} else r->mapq = 0;
	}
	set_inv_mapq(n_regs, regs);
}

static inline int32_t get_for_qpos(int32_t qlen, const mm128 *a)
{
	int32_t x = (int32_t)a->y, q_span = (int32_t)(a->y >> 32 & 0xff);
	if (a->x >> 63) x = qlen - 1 - (x + 1 - q_span);
	return x;
}

static void est_err(const mm355_index *mi, int qlen, int n_regs, Reg *regs, const mm128 *a, int32_t n, const uint64_t *mini_pos)
{
	uint64_t sum_k = 0;
	if (n == 0) return;
	for (int i = 0; i < n; ++i) sum_k += mini_pos[i] >> 32 & 0xff;
	float avg_k = (float)sum_k / n;
	for (int i = 0; i < n_regs; ++i) {
		Reg *r = &regs[i];
		int32_t st, en, j, k, n_match, n_tot, l_ref;
		r->div = -1.0f;
		if (r->cnt == 0) continue;
		{   // get_mini_idx
			int32_t x = get_for_qpos(qlen, r->rev? &a[r->as + r->cnt - 1] : &a[r->as]), L = 0, R = n - 1;
			st = -1;
			while (L <= R) {
				int32_t m = (int32_t)(((uint64_t)L + R) >> 1), y = (int32_t)mini_pos[m];
				if (y < x) L = m + 1; else if (y > x) R = m - 1; else { st = m; break; }
			}
		}
		en = st;
		if (st < 0) continue;
		l_ref = (int32_t)mi->seq_len[r->rid];
		for (k = 1, j = st + 1, n_match = 1; j < n && k < r->cnt; ++j) {
			int32_t x = get_for_qpos(qlen, r->rev? &a[r->as + r->cnt - 1 - k] : &a[r->as + k]);
			if (x == (int32_t)mini_pos[j]) ++k, en = j, ++n_match;
		}
		n_tot = en - st + 1;
		if (r->qs > avg_k && r->rs > avg_k) ++n_tot;
		if (qlen - r->qs > avg_k && l_ref - r->re > avg_k) ++n_tot;
		r->div = n_match >= n_tot? 0.0f : (float)(1.0 - pow((double)n_match / n_tot, 1.0 / avg_k));
	}
}

// ================================================================== stage 0 (MM_F_RMQ presets only: asm5/asm10/asm20)
// U:map.c::mm_map_frag with MM_F_RMQ: the *primary* chainer is mg_lchain_rmq over all sorted anchors of the read (not
// mg_lchain_dp).  Its AVL-tree walk is strictly sequential per read (tree shape decides ties), so it stays in this
// host-resident tail like the long-join re-chain; rs.a holds the read's sorted anchors on entry.
void mm355_glue_chain_rmq(const mm355_index *mi, const mm355_mapopt_t *opt, ReadState &rs)
{
	const float pen_gap = (float)(opt->chain_gap_scale * 0.01 * mi->k), pen_skip = (float)(opt->chain_skip_scale * 0.01 * mi->k);
	rechain_rmq(opt->max_gap, opt->rmq_inner_dist, opt->bw, opt->max_chain_skip, opt->rmq_size_cap, opt->min_cnt, opt->min_chain_score,
	            pen_gap, pen_skip, rs.a, rs.u);
}

// ================================================================== stage 1: after the chain kernels
void mm355_glue_pre_align(const mm355_index *mi, const mm355_mapopt_t *opt, ReadState &rs, int rmq_state)
{
	const int qlen = rs.qlen;
	uint32_t hash = 0;   // qname is NULL through the reference (the L2 crate passes null)
	hash ^= wang32((uint32_t)qlen) + wang32((uint32_t)opt->seed);
	hash = wang32(hash);
	const float pen_gap = (float)(opt->chain_gap_scale * 0.01 * mi->k), pen_skip = (float)(opt->chain_skip_scale * 0.01 * mi->k);
	int n_regs0 = (int)rs.u.size();
	{ ProfScope pf(PF_PRE_RMQ);
	if (rmq_state == 2) {   // MM355_RMQ_HOST: the device sorted the chained anchors and left the chaining to the literal code
		rechain_rmq(opt->max_gap, opt->rmq_inner_dist, opt->bw_long, opt->max_chain_skip, opt->rmq_size_cap, opt->min_cnt, opt->min_chain_score,
		            pen_gap, pen_skip, rs.a, rs.u);
	} else if (rmq_state < 0 && opt->bw_long > opt->bw && (opt->flag & (MMF_SPLICE | MMF_SR | MMF_NO_LJOIN)) == 0 && n_regs0 > 1) {
		int32_t st = (int32_t)rs.a[0].y, en = (int32_t)rs.a[(int32_t)rs.u[0] - 1].y;
		if (qlen - (en - st) > opt->rmq_rescue_size || en - st > qlen * opt->rmq_rescue_ratio) {
			int64_t n_a = 0;
			for (int i = 0; i < n_regs0; ++i) n_a += (int32_t)rs.u[i];
			rs.a.resize(n_a);
			mm_radix_sort(rs.a.data(), rs.a.data() + n_a, mm_key_x());
			static const bool dbg_rmq = getenv("MM355_PROF_RMQ") != 0;
			const auto t_dbg = std::chrono::steady_clock::now();
			rechain_rmq(opt->max_gap, opt->rmq_inner_dist, opt->bw_long, opt->max_chain_skip, opt->rmq_size_cap, opt->min_cnt, opt->min_chain_score,
			            pen_gap, pen_skip, rs.a, rs.u);
			if (dbg_rmq) fprintf(stderr, "[rmq] n_a %lld n_regs0 %d qlen %d: %.3f ms\n", (long long)n_a, n_regs0, qlen, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dbg).count());
		}
	}
	}
	int n;
	{ ProfScope pf(PF_PRE_REGS);
	gen_regs(hash, qlen, rs);
	n = (int)rs.regs.size();
	if (!(opt->flag & MMF_ALL_CHAINS) && n > 0) {
		set_parent(opt->mask_level, opt->mask_len, n, rs.regs.data(), opt->a * 2 + opt->b, (int)(opt->flag & MMF_HARD_MLEVEL));
		select_sub(opt->pri_ratio, mi->k * 2, opt->best_n, 1, (int)(opt->max_gap * 0.8), &n, rs.regs.data());
	}
	}
	{ ProfScope pf(PF_PRE_ESTERR);
	est_err(mi, qlen, n, rs.regs.data(), rs.a.data(), (int32_t)rs.mini_pos.size(), rs.mini_pos.data());
	}
	n = filter_strand_retained(n, rs.regs.data());
	rs.regs.resize(n);
	ProfScope pfc(PF_PRE_CODES);
	// U:align.c::mm_align_skeleton prologue: query codes and anchor squeeze
	static const struct Nt4Lut { uint8_t t[256]; Nt4Lut() { for (int c = 0; c < 256; ++c) t[c] = (uint8_t)mm_nt4((uint8_t)c); } } lut;
	rs.qc[0].resize(qlen);
	for (int i = 0; i < qlen; ++i) rs.qc[0][i] = lut.t[(uint8_t)rs.seq[i]];
	bool any_rev = false;
	for (int i = 0; i < n; ++i) any_rev |= rs.regs[i].rev != 0;
	rs.qc[1].clear();
	if (any_rev || n == 0) {   // the reverse-complement code string is only read by reverse-strand regions (and inversions, see ensure_rev)
		rs.qc[1].resize(qlen);
		for (int i = 0; i < qlen; ++i) { uint8_t c = rs.qc[0][i]; rs.qc[1][qlen - 1 - i] = c < 4? 3 - c : 4; }
	}
	rs.n_a = squeeze_a(n, rs.regs.data(), rs.a.data());
	rs.cursor = 0; rs.aligned = n == 0;
	rs.tasks.clear();
}

bool g_prof_on = getenv("MM355_PROF") != 0;
static std::atomic<ProfThread*> g_prof_threads(nullptr);
ProfThread *mm355_prof_thread()
{
	static thread_local ProfThread *me = nullptr;
	if (me == nullptr) {
		me = new ProfThread();   // leaked with the thread
		memset(me, 0, sizeof(*me));
		ProfThread *head = g_prof_threads.load();
		do { me->next = head; } while (!g_prof_threads.compare_exchange_weak(head, me));
	}
	return me;
}
static double tsc_per_us()
{
	static double v = [] { auto t0 = std::chrono::steady_clock::now(); uint64_t c0 = __rdtsc();
		while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 20000.0) {}
		uint64_t c1 = __rdtsc(); return (double)(c1 - c0) / std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }();
	return v;
}
void mm355_prof_dump(int64_t n_reads)   // call while no worker is inside a section
{
	static const char *nm[PF_N] = { "pre:copy", "pre:rmq", "pre:regs", "pre:est_err", "pre:codes", "pre:squeeze", "task_prepare", "getseq", "test_zdrop", "add_cigar",
	                                "update_extra", "task_run(total)", "align_step(total)", "finish", "dp:distribute", "dp:gather_build", "assemble", "x1:newExtra+getseq", "x2:fix_cigar", "x3:extra_loop", "x4:cs_md", "x5" };
	if (!g_prof_on) return;
	const double f = tsc_per_us(), nr = (double)(n_reads > 0? n_reads : 1);
	fprintf(stderr, "[mm355 prof] CPU microseconds per read (%lld reads):", (long long)n_reads);
	for (int i = 0; i < PF_N; ++i) {
		uint64_t t = 0, c = 0;
		for (ProfThread *p = g_prof_threads.load(); p; p = p->next) { t += p->tsc[i]; c += p->cnt[i]; p->tsc[i] = 0; p->cnt[i] = 0; }
		if (c) fprintf(stderr, " %s=%.1f(x%.1f)", nm[i], t / f / nr, c / nr);
	}
	fprintf(stderr, "\n");
}

// ================================================================== alignment driver (U:align.c)
static void gen_simple_mat(int8_t *mat, int a, int b, int sc_ambi)
{
	a = a < 0? -a : a; b = b > 0? -b : b; sc_ambi = sc_ambi > 0? -sc_ambi : sc_ambi;
	for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) mat[i * 5 + j] = i == j? a : b; mat[i * 5 + 4] = sc_ambi; }
	for (int j = 0; j < 5; ++j) mat[4 * 5 + j] = sc_ambi;
}

static void getseq(const mm355_index *mi, uint32_t rid, int32_t st, int32_t en, uint8_t *out)
{
	if (rid >= mi->n_seq || (uint32_t)st >= mi->seq_len[rid]) return;
	if ((uint32_t)en > mi->seq_len[rid]) en = (int32_t)mi->seq_len[rid];
	uint64_t o = mi->seq_off[rid] + (uint64_t)st, e = mi->seq_off[rid] + (uint64_t)en;
	const uint32_t *S = mi->S.data();
	for (; o < e && (o & 7); ++o) *out++ = (uint8_t)(S[o >> 3] >> ((o & 7) << 2) & 0xf);
	for (; o + 8 <= e; o += 8, out += 8) {   // one packed word = 8 bases
		uint32_t w = S[o >> 3];
		out[0] = w & 0xf; out[1] = w >> 4 & 0xf; out[2] = w >> 8 & 0xf; out[3] = w >> 12 & 0xf;
		out[4] = w >> 16 & 0xf; out[5] = w >> 20 & 0xf; out[6] = w >> 24 & 0xf; out[7] = w >> 28;
	}
	for (; o < e; ++o) *out++ = (uint8_t)(S[o >> 3] >> ((o & 7) << 2) & 0xf);
}

// local Smith-Waterman score/end as U:ksw2_ll_sse.c::ksw_ll_i16 reports them (query padded to a multiple of 8 with
// zero-score columns; te = last target row reaching the running best; qe = last striped slot holding the best)
static int ksw_ll(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int gapo, int gape, int *qe, int *te)
{
	int slen = (qlen + 7) / 8, qlen8 = slen * 8, gmax = 0, gapoe = gapo + gape;
	*qe = *te = -1;
	if (qlen <= 0) return 0;
	std::vector<int32_t> H0(qlen8 + 1, 0), H1(qlen8 + 1, 0), E(qlen8 + 1, 0), Hmax(qlen8 + 1, 0);
	for (int i = 0; i < tlen; ++i) {
		const int8_t *ma = mat + target[i] * 5;
		int32_t f = 0, imax = 0, hdiag = 0;
		for (int j = 0; j < qlen8; ++j) {
			int32_t sc = j < qlen? ma[query[j]] : 0, h = hdiag + sc, e = E[j], t;
			hdiag = H0[j];
			h = h > e? h : e; h = h > f? h : f;
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
		if (imax >= gmax) { gmax = imax; *te = i; Hmax = H1; }
		H1.swap(H0);
	}
	for (int i = 0; i < qlen8; ++i) { int pos = i / 8 + i % 8 * slen; if (Hmax[pos] == gmax) *qe = pos; }
	return gmax;
}

static inline void update_max_zdrop(int32_t score, int i, int j, int32_t *max, int *max_i, int *max_j, int e, int *max_zdrop, int pos[2][2])
{
	if (score < *max) {
		int li = i - *max_i, lj = j - *max_j;
		int diff = li > lj? li - lj : lj - li;
		int z = *max - score - diff * e;
		if (z > *max_zdrop) { *max_zdrop = z; pos[0][0] = *max_i, pos[0][1] = i; pos[1][0] = *max_j, pos[1][1] = j; }
	} else *max = score, *max_i = i, *max_j = j;
}

static int test_zdrop(const mm355_mapopt_t *opt, const uint8_t *qseq, const uint8_t *tseq, const uint32_t *cigar, int n_cigar, const int8_t *mat)
{
	int32_t score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
	int pos[2][2] = {{-1, -1}, {-1, -1}}, q_len, t_len;
	for (int k = 0; k < n_cigar; ++k) {
		uint32_t l, op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (op == 0) {
			// Runs of equal unambiguous bases are folded: along such a run the score grows by `a` per base, so (U:align.c::
			// mm_update_max_zdrop) only its first base can raise max_zdrop while score < max, and once score >= max every
			// further base just moves the maximum -- same final state as the base-by-base loop.
			const int32_t a_m = mat[0];
			l = 0;
			while (l < len) {
				uint32_t run = 0;
				if (a_m > 0) {
					while (l + run + 8 <= len) {
						uint64_t tw, qw; memcpy(&tw, tseq + i + l + run, 8); memcpy(&qw, qseq + j + l + run, 8);
						const uint64_t bad = (tw ^ qw) | ((tw | qw) & 0xfcfcfcfcfcfcfcfcULL);
						if (bad == 0) { run += 8; continue; }
						run += (uint32_t)(__builtin_ctzll(bad) >> 3);
						goto run_done;
					}
					while (l + run < len && tseq[i + l + run] == qseq[j + l + run] && tseq[i + l + run] < 4) ++run;
				}
			run_done:
				if (run > 0) {
					const int bi = i + (int)l, bj = j + (int)l;     // first base of the run
					int64_t k_star = 1;                              // first base (1-based) whose score reaches max
					if (max != INT32_MIN && (int64_t)max - score > a_m) k_star = ((int64_t)max - score + a_m - 1) / a_m;
					if (k_star > 1) update_max_zdrop(score + a_m, bi, bj, &max, &max_i, &max_j, opt->e, &max_zdrop, pos);   // score < max there
					score += a_m * (int32_t)run;
					if (k_star <= (int64_t)run) { max = score; max_i = bi + (int)run - 1; max_j = bj + (int)run - 1; }
					l += run;
				}
				if (l < len) {   // a mismatch or an ambiguous base
					score += mat[tseq[i + l] * 5 + qseq[j + l]];
					update_max_zdrop(score, i + l, j + l, &max, &max_i, &max_j, opt->e, &max_zdrop, pos);
					++l;
				}
			}
			i += len, j += len;
		} else if (op == 1 || op == 2 || op == 3) {
			score -= opt->q + opt->e * len;
			if (op == 1) j += len; else i += len;
			update_max_zdrop(score, i, j, &max, &max_i, &max_j, opt->e, &max_zdrop, pos);
		}
	}
	q_len = pos[1][1] - pos[1][0], t_len = pos[0][1] - pos[0][0];
	if (!(opt->flag & (MMF_SPLICE | MMF_SR | MMF_FOR_ONLY | MMF_REV_ONLY)) && max_zdrop > opt->zdrop_inv && q_len < opt->max_gap && t_len < opt->max_gap) {
		std::vector<uint8_t> qseq2(q_len > 0? q_len : 1);
		int q_off, t_off;
		for (i = 0; i < q_len; ++i) { int c = qseq[pos[1][1] - i - 1]; qseq2[i] = c >= 4? 4 : 3 - c; }
		score = ksw_ll(q_len, qseq2.data(), t_len, tseq + pos[0][0], mat, opt->q, opt->e, &q_off, &t_off);
		if (score >= opt->min_chain_score * opt->a && score >= opt->min_dp_max) return 2;
	}
	return max_zdrop > opt->zdrop? 1 : 0;
}

static void append_cigar(Reg *r, const uint32_t *cigar, int n)
{
	if (n <= 0) return;
	if (r->p == 0) r->p = new Extra();
	std::vector<uint32_t> &c = r->p->cigar;
	if (!c.empty() && (c.back() & 0xf) == (cigar[0] & 0xf)) {
		c.back() += (cigar[0] >> 4) << 4;
		c.insert(c.end(), cigar + 1, cigar + n);
	} else c.insert(c.end(), cigar, cigar + n);
}

static void fix_cigar(Reg *r, const uint8_t *qseq, const uint8_t *tseq, int *qshift, int *tshift)
{
	std::vector<uint32_t> &cg = r->p->cigar;
	int32_t toff = 0, qoff = 0, to_shrink = 0;
	uint32_t k, n_cigar = (uint32_t)cg.size();
	*qshift = *tshift = 0;
	if (n_cigar <= 1) return;
	for (k = 0; k < n_cigar; ++k) {
		uint32_t op = cg[k] & 0xf, len = cg[k] >> 4;
		if (len == 0) to_shrink = 1;
		if (op == 0) toff += len, qoff += len;
		else if (op == 1 || op == 2) {
			if (k > 0 && k < n_cigar - 1 && (cg[k-1] & 0xf) == 0 && (cg[k+1] & 0xf) == 0) {
				int l, prev_len = cg[k-1] >> 4;
				if (op == 1) { for (l = 0; l < prev_len; ++l) if (qseq[qoff - 1 - l] != qseq[qoff + len - 1 - l]) break; }
				else { for (l = 0; l < prev_len; ++l) if (tseq[toff - 1 - l] != tseq[toff + len - 1 - l]) break; }
				if (l > 0) cg[k-1] -= l << 4, cg[k+1] += l << 4, qoff -= l, toff -= l;
				if (l == prev_len) to_shrink = 1;
			}
			if (op == 1) qoff += len; else toff += len;
		} else if (op == 3) toff += len;
	}
	for (k = 0; k + 2 < n_cigar; ++k) {
		if ((cg[k] & 0xf) > 0 && (cg[k] & 0xf) + (cg[k+1] & 0xf) == 3) {
			uint32_t l, s[3] = {0, 0, 0};
			for (l = k; l < n_cigar; ++l) {
				uint32_t op = cg[l] & 0xf;
				if (op == 1 || op == 2 || cg[l] >> 4 == 0) s[op] += cg[l] >> 4;
				else break;
			}
			if (s[1] > 0 && s[2] > 0 && l - k > 2) {
				cg[k] = s[1] << 4 | 1;
				cg[k+1] = s[2] << 4 | 2;
				for (k += 2; k < l; ++k) cg[k] &= 0xf;
				to_shrink = 1;
			}
			k = l;
		}
	}
	if (to_shrink) {
		uint32_t l = 0;
		for (k = 0; k < n_cigar; ++k) if (cg[k] >> 4 != 0) cg[l++] = cg[k];
		n_cigar = l;
		for (k = l = 0; k < n_cigar; ++k)
			if (k == n_cigar - 1 || (cg[k] & 0xf) != (cg[k+1] & 0xf)) cg[l++] = cg[k];
			else cg[k+1] += cg[k] >> 4 << 4;
		n_cigar = l;
	}
	if ((cg[0] & 0xf) == 1 || (cg[0] & 0xf) == 2) {
		int32_t l = cg[0] >> 4;
		if ((cg[0] & 0xf) == 1) { if (r->rev) r->qe -= l; else r->qs += l; *qshift = l; }
		else r->rs += l, *tshift = l;
		--n_cigar;
		memmove(cg.data(), cg.data() + 1, (size_t)n_cigar * 4);
	}
	cg.resize(n_cigar);
}

static void gen_cs(const Reg *r, const uint8_t *qseq, const uint8_t *tseq, std::string &s);
static void gen_md(const Reg *r, const uint8_t *qseq, const uint8_t *tseq, std::string &s);

// U:align.c::mm_update_cigar_eqx (MM_F_EQX): every M becomes alternating runs of '=' (7) and 'X' (8); N vs N counts as '='
static void cigar_eqx(Reg *r, const uint8_t *qseq, const uint8_t *tseq)
{
	std::vector<uint32_t> &cg = r->p->cigar, nc;
	nc.reserve(cg.size() * 2);
	uint32_t toff = 0, qoff = 0, l;
	for (size_t k = 0; k < cg.size(); ++k) {
		uint32_t op = cg[k] & 0xf, len = cg[k] >> 4;
		if (op == 0) {
			while (len > 0) {
				for (l = 0; l < len && qseq[qoff + l] == tseq[toff + l]; ++l) {}
				if (l > 0) { nc.push_back(l << 4 | 7); len -= l; toff += l; qoff += l; }
				for (l = 0; l < len && qseq[qoff + l] != tseq[toff + l]; ++l) {}
				if (l > 0) { nc.push_back(l << 4 | 8); len -= l; toff += l; qoff += l; }
			}
		} else {
			if (op == 1) qoff += len; else if (op == 2 || op == 3) toff += len;
			nc.push_back(cg[k]);
		}
	}
	cg.swap(nc);
}

static void update_extra(Reg *r, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int8_t q, int8_t e, int out_flags, bool eqx, const ExtraLoc *loc = 0)
{
	int32_t qshift, tshift, toff = 0, qoff = 0;
	double s = 0.0, max = 0.0;
	Extra *p = r->p;
	if (p == 0) return;
	{ ProfScope pf2(PF_X2); fix_cigar(r, qseq, tseq, &qshift, &tshift); }
	qseq += qshift, tseq += tshift;
	r->blen = r->mlen = 0;
	if (loc && !eqx && !p->cigar.empty()) {   // the walk below, cs and MD: on the device, for all regions of the batch at once (k_extra); '=' / 'X' CIGARs are rewritten here
		p->deferred = true;
		p->x_strand = loc->strand; p->x_qst = loc->q_st + qshift; p->x_rid = loc->rid; p->x_tst = loc->t_st + tshift;
		return;
	}
	ProfScope *pf3 = new ProfScope(PF_X3);
	for (size_t k = 0; k < p->cigar.size(); ++k) {
		uint32_t op = p->cigar[k] & 0xf, len = p->cigar[k] >> 4, l;
		if (op == 0) {
			int n_ambi = 0, n_diff = 0;
			// runs of equal unambiguous bases in one step: s only grows along them, so the running maximum is taken at their end.
			// (s is a sum of small integers and of float-valued gap costs: every partial sum is exact in double, the order of the
			// additions cannot change it)
			const double a_m = mat[0];
			l = 0;
			while (l < len) {
				uint32_t run = 0;
				if (a_m > 0) {
					while (l + run + 8 <= len) {
						uint64_t tw, qw; memcpy(&tw, tseq + toff + l + run, 8); memcpy(&qw, qseq + qoff + l + run, 8);
						const uint64_t bad = (tw ^ qw) | ((tw | qw) & 0xfcfcfcfcfcfcfcfcULL);
						if (bad == 0) { run += 8; continue; }
						run += (uint32_t)(__builtin_ctzll(bad) >> 3);
						goto run_done;
					}
					while (l + run < len && tseq[toff + l + run] == qseq[qoff + l + run] && tseq[toff + l + run] < 4) ++run;
				}
			run_done:
				if (run > 0) { s += a_m * run; max = max > s? max : s; l += run; }
				if (l < len) {
					int cq = qseq[qoff + l], ct = tseq[toff + l];
					if (ct > 3 || cq > 3) ++n_ambi;
					else if (ct != cq) ++n_diff;
					s += mat[ct * 5 + cq];
					if (s < 0) s = 0; else max = max > s? max : s;
					++l;
				}
			}
			r->blen += len - n_ambi, r->mlen += len - (n_ambi + n_diff), p->n_ambi += n_ambi;
			toff += len, qoff += len;
		} else if (op == 1) {
			int n_ambi = 0;
			for (l = 0; l < len; ++l) if (qseq[qoff + l] > 3) ++n_ambi;
			r->blen += len - n_ambi, p->n_ambi += n_ambi;
			s -= q + (double)e * mm_log2f_approx((float)(1.0 + len));   // log_gap: long-read presets are never MM_F_SR
			if (s < 0) s = 0;
			qoff += len;
		} else if (op == 2) {
			int n_ambi = 0;
			for (l = 0; l < len; ++l) if (tseq[toff + l] > 3) ++n_ambi;
			r->blen += len - n_ambi, p->n_ambi += n_ambi;
			s -= q + (double)e * mm_log2f_approx((float)(1.0 + len));
			if (s < 0) s = 0;
			toff += len;
		} else if (op == 3) toff += len;
	}
	p->dp_max = (int32_t)(max + .499);
	delete pf3;
	ProfScope pf4(PF_X4);
	if (out_flags & MM355_OUT_CS) { p->cs.clear(); gen_cs(r, qseq, tseq, p->cs); }
	if (out_flags & MM355_OUT_MD) { p->md.clear(); gen_md(r, qseq, tseq, p->md); }
	if (eqx) cigar_eqx(r, qseq, tseq);   // after cs/MD: both read M, '=' and 'X' alike (U:format.c::write_cs_core / write_MD_core)
}

static int *collect_long_gaps(int as1, int cnt1, const mm128 *a, int min_gap, int *n_, std::vector<int> &K)
{
	K.clear(); *n_ = 0;
	for (int i = 1; i < cnt1; ++i) {
		int gap = (int)(((int32_t)a[as1 + i].y - a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - a[as1 + i - 1].x));
		if (gap < -min_gap || gap > min_gap) K.push_back(i);
	}
	if (K.size() <= 1) { K.clear(); return 0; }
	*n_ = (int)K.size();
	return K.data();
}

static void filter_bad_seeds(int as1, int cnt1, mm128 *a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt)
{
	int max_st, max_en, n, i, k, max;
	std::vector<int> Kv;
	int *K = collect_long_gaps(as1, cnt1, a, min_gap, &n, Kv);
	if (K == 0) return;
	max = 0, max_st = max_en = -1;
	for (k = 0;; ++k) {
		int gap, l, n_ins = 0, n_del = 0, qs, rs, max_diff = 0, max_diff_l = -1;
		if (k == n || k >= max_en) {
			if (max_en > 0) for (i = K[max_st]; i < K[max_en]; ++i) a[as1 + i].y |= MM355_SEED_IGNORE;
			max = 0, max_st = max_en = -1;
			if (k == n) break;
		}
		i = K[k];
		gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - (int32_t)(a[as1 + i].x - a[as1 + i - 1].x);
		if (gap > 0) n_ins += gap; else n_del += -gap;
		qs = (int32_t)a[as1 + i - 1].y;
		rs = (int32_t)a[as1 + i - 1].x;
		for (l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
			int j = K[l], diff;
			if ((int32_t)a[as1 + j].y - qs > max_ext_len || (int32_t)a[as1 + j].x - rs > max_ext_len) break;
			gap = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
			if (gap > 0) n_ins += gap; else n_del += -gap;
			diff = n_ins + n_del - abs(n_ins - n_del);
			if (max_diff < diff) max_diff = diff, max_diff_l = l;
		}
		if (max_diff > diff_thres && max_diff > max) max = max_diff, max_st = k, max_en = max_diff_l;
	}
}

static void filter_bad_seeds_alt(int as1, int cnt1, mm128 *a, int min_gap, int max_ext)
{
	int n, k;
	std::vector<int> Kv;
	int *K = collect_long_gaps(as1, cnt1, a, min_gap, &n, Kv);
	if (K == 0) return;
	for (k = 0; k < n;) {
		int i = K[k], l;
		int gap1 = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - (int32_t)a[as1 + i - 1].x);
		int re1 = (int32_t)a[as1 + i].x, qe1 = (int32_t)a[as1 + i].y;
		gap1 = gap1 > 0? gap1 : -gap1;
		for (l = k + 1; l < n; ++l) {
			int j = K[l], gap2, q_span_pre, rs2, qs2, m;
			if ((int32_t)a[as1 + j].y - qe1 > max_ext || (int32_t)a[as1 + j].x - re1 > max_ext) break;
			gap2 = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
			q_span_pre = (int)(a[as1 + j - 1].y >> 32 & 0xff);
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
			for (j = K[k]; j < end; ++j) a[as1 + j].y |= MM355_SEED_IGNORE;
			a[as1 + end].y |= MM355_SEED_LONG_JOIN;
		}
		k = l;
	}
}

static void fix_bad_ends(const Reg *r, const mm128 *a, int bw, int min_match, int32_t *as, int32_t *cnt)
{
	int32_t i, l, m;
	*as = r->as, *cnt = r->cnt;
	if (r->cnt < 3) return;
	m = l = (int32_t)(a[r->as].y >> 32 & 0xff);
	for (i = r->as + 1; i < r->as + r->cnt - 1; ++i) {
		int32_t lq, lr, min, max, q_span = (int32_t)(a[i].y >> 32 & 0xff);
		if (a[i].y & MM355_SEED_LONG_JOIN) break;
		lr = (int32_t)a[i].x - (int32_t)a[i-1].x;
		lq = (int32_t)a[i].y - (int32_t)a[i-1].y;
		min = lr < lq? lr : lq; max = lr > lq? lr : lq;
		if (max - min > l >> 1) *as = i;
		l += min;
		m += min < q_span? min : q_span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->mlen >> 1) break;
	}
	*cnt = r->as + r->cnt - *as;
	m = l = (int32_t)(a[r->as + r->cnt - 1].y >> 32 & 0xff);
	for (i = r->as + r->cnt - 2; i > *as; --i) {
		int32_t lq, lr, min, max, q_span = (int32_t)(a[i+1].y >> 32 & 0xff);
		if (a[i+1].y & MM355_SEED_LONG_JOIN) break;
		lr = (int32_t)a[i+1].x - (int32_t)a[i].x;
		lq = (int32_t)a[i+1].y - (int32_t)a[i].y;
		min = lr < lq? lr : lq; max = lr > lq? lr : lq;
		if (max - min > l >> 1) *cnt = i + 1 - *as;
		l += min;
		m += min < q_span? min : q_span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->mlen >> 1) break;
	}
}

// U:align.c::mm_adjust_minier (+ mm_get_hplen_back): where a seed cuts the alignment.  Plain index: the middle of the k-mer.  HPC index
// (MM_I_HPC: map-pb / ava-pb): a seed ends on the last base of a homopolymer run on both sequences, and the cut is the FIRST base of
// that run (the query scan stops at position 1, as the reference's `i > 0` does).
static inline void adjust_minier(const mm355_index *mi, const ReadState &rs, const mm128 *a, int32_t *r, int32_t *q)
{
	if (mi->flag & 1) {
		const uint8_t *qseq = rs.qc[a->x >> 63].data();
		int32_t i = (int32_t)a->y;
		const int c = qseq[i];
		for (--i; i > 0; --i) if (qseq[i] != c) break;
		*q = i + 1;
		const uint32_t rid = (uint32_t)(a->x << 1 >> 33);
		const int64_t off0 = (int64_t)mi->seq_off[rid], off = off0 + (int64_t)(uint32_t)a->x;
		const uint32_t *S = mi->S.data();
		const uint32_t ct = S[off >> 3] >> ((off & 7) << 2) & 0xf;
		int64_t j;
		for (j = off - 1; j >= off0; --j) if ((S[j >> 3] >> ((j & 7) << 2) & 0xf) != ct) break;
		*r = (int32_t)a->x + 1 - (int32_t)(off - j);
	} else {
		*r = (int32_t)a->x - (mi->k >> 1);
		*q = (int32_t)a->y - (mi->k >> 1);
	}
}

// one-off part of mm_align1: seed filtering and the extension windows
static void task_prepare(const mm355_index *mi, const mm355_mapopt_t *opt, ReadState &rs, const Reg *r, AlnTask &T)
{
	ProfScope pf(PF_TASK_PREPARE);
	mm128 *a = rs.a.data();
	const int qlen = rs.qlen, n_a = rs.n_a;
	int32_t i, l, rs1, qs1, re1, qe1;
	T.rid = (int32_t)(a[r->as].x << 1 >> 33); T.rev = (int32_t)(a[r->as].x >> 63);
	T.split_inv = (int)r->split_inv;
	T.bw = (int)(opt->bw * 1.5 + 1.);
	T.bw_long = (int)(opt->bw_long * 1.5 + 1.);
	if (T.bw_long < T.bw) T.bw_long = T.bw;
	if (!(opt->flag & MMF_NO_END_FLT)) fix_bad_ends(r, a, opt->bw, opt->min_chain_score * 2, &T.as1, &T.cnt1);
	else T.as1 = r->as, T.cnt1 = r->cnt;
	filter_bad_seeds(T.as1, T.cnt1, a, 10, 40, opt->max_gap >> 1, 10);
	filter_bad_seeds_alt(T.as1, T.cnt1, a, 30, opt->max_gap >> 1);
	int32_t rs_, qs_, re_, qe_;
	adjust_minier(mi, rs, &a[T.as1], &rs_, &qs_);
	adjust_minier(mi, rs, &a[T.as1 + T.cnt1 - 1], &re_, &qe_);
	int32_t rs0, qs0, re0, qe0;
	const int32_t rlen = (int32_t)mi->seq_len[T.rid];
	rs0 = (int32_t)a[r->as].x + 1 - (int32_t)(a[r->as].y >> 32 & 0xff);
	qs0 = (int32_t)a[r->as].y + 1 - (int32_t)(a[r->as].y >> 32 & 0xff);
	if (rs0 < 0) rs0 = 0;
	rs1 = qs1 = 0;
	for (i = r->as - 1, l = 0; i >= 0 && a[i].x >> 32 == a[r->as].x >> 32; --i) {
		int32_t x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff);
		int32_t y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
		if (x < rs0 && y < qs0) {
			if (++l > opt->min_cnt) {
				l = rs0 - x > qs0 - y? rs0 - x : qs0 - y;
				rs1 = rs0 - l, qs1 = qs0 - l;
				if (rs1 < 0) rs1 = 0;
				break;
			}
		}
	}
	if (qs_ > 0 && rs_ > 0) {
		l = qs_ < opt->max_gap? qs_ : opt->max_gap;
		qs1 = qs1 > qs_ - l? qs1 : qs_ - l;
		qs0 = qs0 < qs1? qs0 : qs1;
		l += l * opt->a > opt->q? (l * opt->a - opt->q) / opt->e : 0;
		l = l < opt->max_gap? l : opt->max_gap;
		l = l < rs_? l : rs_;
		rs1 = rs1 > rs_ - l? rs1 : rs_ - l;
		rs0 = rs0 < rs1? rs0 : rs1;
		rs0 = rs0 < rs_? rs0 : rs_;
	} else rs0 = rs_, qs0 = qs_;
	re0 = (int32_t)a[r->as + r->cnt - 1].x + 1;
	qe0 = (int32_t)a[r->as + r->cnt - 1].y + 1;
	re1 = rlen, qe1 = qlen;
	for (i = r->as + r->cnt, l = 0; i < n_a && a[i].x >> 32 == a[r->as].x >> 32; ++i) {
		int32_t x = (int32_t)a[i].x + 1, y = (int32_t)a[i].y + 1;
		if (x > re0 && y > qe0) {
			if (++l > opt->min_cnt) {
				l = x - re0 > y - qe0? x - re0 : y - qe0;
				re1 = re0 + l, qe1 = qe0 + l;
				break;
			}
		}
	}
	if (qe_ < qlen && re_ < rlen) {
		l = qlen - qe_ < opt->max_gap? qlen - qe_ : opt->max_gap;
		qe1 = qe1 < qe_ + l? qe1 : qe_ + l;
		qe0 = qe0 > qe1? qe0 : qe1;
		l += l * opt->a > opt->q? (l * opt->a - opt->q) / opt->e : 0;
		l = l < opt->max_gap? l : opt->max_gap;
		l = l < rlen - re_? l : rlen - re_;
		re1 = re1 < re_ + l? re1 : re_ + l;
		re0 = re0 > re1? re0 : re1;
	} else re0 = re_, qe0 = qe_;
	T.rs = rs_, T.qs = qs_, T.re = re_, T.qe = qe_;
	T.rs0 = rs0, T.qs0 = qs0, T.re0 = re0, T.qe0 = qe0;
	T.res.clear(); T.res.reserve(40);
	T.slot_of.assign(2 + 2 * (size_t)T.cnt1, -1);
	T.prepared = true;
}

static bool want(AlnTask &T, int slot, int read_id, int task_id, std::vector<DpReq> &reqs, int32_t qlen, int32_t tlen, int32_t q_st, int rev_strand,
                 uint32_t rid, int32_t t_st, int reversed, int32_t w, int32_t zdrop, int32_t end_bonus, int32_t flag)
{
	if (T.slot_of[slot] < 0) { T.slot_of[slot] = (int32_t)T.res.size(); T.res.emplace_back(); }
	EzRes &e = T.res[T.slot_of[slot]];
	if (e.state == 2) return true;
	if (e.state == 0) {
		DpReq q; q.read = read_id; q.task = task_id; q.slot = T.slot_of[slot]; q.qlen = qlen; q.tlen = tlen; q.q_st = q_st; q.rev_strand = rev_strand;
		q.rid = rid; q.t_st = t_st; q.reversed = reversed; q.w = w; q.zdrop = zdrop; q.end_bonus = end_bonus; q.flag = flag;
		reqs.push_back(q);
		e.state = 1;
	}
	return false;
}

static void request_left(const mm355_mapopt_t *opt, int read_id, int task_id, AlnTask &T, std::vector<DpReq> &reqs);

// replayable part of mm_align1.  Returns true when region `ri` is fully aligned (r, and *r2 when split, are final).
static bool task_run(const mm355_index *mi, const mm355_mapopt_t *opt, int read_id, ReadState &rs, int ri, int task_id, Reg *r2, std::vector<DpReq> &reqs)
{
	ProfScope pf_run(PF_TASK_RUN);
	AlnTask &T = rs.tasks[task_id];
	Reg *r = &rs.regs[ri];
	mm128 *a = rs.a.data();
	const int qlen = rs.qlen;
	const int32_t rid = T.rid, rev = T.rev, as1 = T.as1, cnt1 = T.cnt1;
	int32_t rs_ = T.rs, qs_ = T.qs, re_ = T.re, qe_ = T.qe;
	const int32_t qs0 = T.qs0, re0 = T.re0, qe0 = T.qe0;
	int32_t rs1, qs1, re1, qe1, dropped = 0;
	bool complete = true;
	int8_t mat[25];
	gen_simple_mat(mat, opt->a, opt->b, opt->sc_ambi);
	Extra tmp;        // alignment under construction; committed only when every needed DP result is present
	bool have_p = false;
	auto add_cigar = [&](const uint32_t *cg, int ncg) { if (ncg <= 0) return; ProfScope pf(PF_ADD_CIGAR); Reg t2; t2.p = &tmp; append_cigar(&t2, cg, ncg); have_p = true; };
	auto RES = [&](int slot) -> EzRes& { return T.res[T.slot_of[slot]]; };
	std::vector<uint8_t> tseq;
	r2->cnt = 0;
	if (r->cnt == 0) return true;
	int split_at = -1, split_code = 0;
	// left extension (requested by request_left(), which runs before every task_run)
	rs1 = rs_, qs1 = qs_;
	if (qs_ > 0 && rs_ > 0) {
		if (T.slot_of[0] >= 0 && RES(0).state == 2) {
			EzRes &e = RES(0);
			if (e.n_cigar > 0) { add_cigar(e.cigar, e.n_cigar); tmp.dp_score += e.max; }
			rs1 = rs_ - (e.reach_end? e.mqe_t + 1 : e.max_t + 1);
			qs1 = qs_ - (e.reach_end? qs_ - qs0 : e.max_q + 1);
		} else complete = false;
	}
	re1 = rs_, qe1 = qs_;
	int32_t rs_run = rs_, qs_run = qs_, re_run = re_, qe_run = qe_;
	for (int32_t i = 1; i < cnt1; ++i) {   // gap filling
		if ((a[as1 + i].y & (MM355_SEED_IGNORE | MM355_SEED_TANDEM)) && i != cnt1 - 1) continue;
		adjust_minier(mi, rs, &a[as1 + i], &re_run, &qe_run);
		re1 = re_run, qe1 = qe_run;
		if (i == cnt1 - 1 || (a[as1 + i].y & MM355_SEED_LONG_JOIN) || (qe_run - qs_run >= opt->min_ksw_len && re_run - rs_run >= opt->min_ksw_len)) {
			int bw1 = T.bw_long, zdrop_code = 0;
			if (a[as1 + i].y & MM355_SEED_LONG_JOIN) bw1 = qe_run - qs_run > re_run - rs_run? qe_run - qs_run : re_run - rs_run;
			const int sa = 2 + 2 * i, se = 3 + 2 * i;
			bool ok = want(T, sa, read_id, task_id, reqs, qe_run - qs_run, re_run - rs_run, qs_run, rev, (uint32_t)rid, rs_run, 0, bw1, opt->zdrop, -1, EZ_APPROX_MAX);
			if (!ok) { complete = false; rs_run = re_run, qs_run = qe_run; continue; }   // speculate: not dropped
			EzRes *e = &RES(sa);
			// U:align.c::mm_test_zdrop walks the path and returns 0 unless some score drop along it exceeds zdrop (or zdrop_inv).
			// A gap fill is a full-matrix global alignment (bw1 covers it), so e->score is the score of the path in the CIGAR:
			// a*M - score - (two-piece gap costs) = everything the matched columns lose, and no drop can exceed that loss plus the
			// one-piece gap costs the test itself charges.  If that bound is within both thresholds the walk is skipped.
			bool quiet = false;
			if (!e->zdropped && e->score > -0x20000000 && e->n_cigar > 0) {
				int64_t M = 0, G1 = 0, G2 = 0;
				for (int k = 0; k < e->n_cigar; ++k) {
					const int64_t op = e->cigar[k] & 0xf, len = e->cigar[k] >> 4;
					if (op == 0) M += len;
					else if (op == 1 || op == 2) { const int64_t c1 = opt->q + (int64_t)opt->e * len, c2 = opt->q2 + (int64_t)opt->e2 * len; G1 += c1; G2 += c1 < c2? c1 : c2; }
					else { M = -1; break; }
				}
				const int64_t loss = M >= 0? (int64_t)opt->a * M - G2 - e->score : -1;
				const int64_t lim = opt->zdrop < opt->zdrop_inv? opt->zdrop : opt->zdrop_inv;
				quiet = M >= 0 && loss >= 0 && loss + G1 <= lim;
			}
			if (!quiet) {
				tseq.resize((size_t)(re_run - rs_run) + 1);
				{ ProfScope pf(PF_GETSEQ); getseq(mi, (uint32_t)rid, rs_run, re_run, tseq.data()); }
				const uint8_t *qseq = rs.qc[rev].data() + qs_run;
				{ ProfScope pf(PF_TEST_ZDROP); zdrop_code = test_zdrop(opt, qseq, tseq.data(), e->cigar, e->n_cigar, mat); }
			}
			if (zdrop_code != 0) {
				ok = want(T, se, read_id, task_id, reqs, qe_run - qs_run, re_run - rs_run, qs_run, rev, (uint32_t)rid, rs_run, 0, bw1,
				          zdrop_code == 2? opt->zdrop_inv : opt->zdrop, -1, 0);
				if (!ok) { complete = false; rs_run = re_run, qs_run = qe_run; continue; }
				e = &RES(se);
			}
			if (e->n_cigar > 0) add_cigar(e->cigar, e->n_cigar);
			if (e->zdropped) {
				int32_t j;
				have_p = true;
				for (j = i - 1; j >= 0; --j) if ((int32_t)a[as1 + j].x <= rs_run + e->max_t) break;
				dropped = 1;
				if (j < 0) j = 0;
				tmp.dp_score += e->max;
				re1 = rs_run + (e->max_t + 1);
				qe1 = qs_run + (e->max_q + 1);
				if (cnt1 - (j + 1) >= opt->min_cnt) { split_at = as1 + j + 1 - r->as; split_code = zdrop_code; }
				break;
			} else tmp.dp_score += e->score;
			rs_run = re_run, qs_run = qe_run;
		}
	}
	re_ = re_run, qe_ = qe_run;
	if (!dropped && qe_ < qe0 && re_ < re0) {   // right extension
		bool ok = want(T, 1, read_id, task_id, reqs, qe0 - qe_, re0 - re_, qe_, rev, (uint32_t)rid, re_, 0, T.bw, opt->zdrop, opt->end_bonus, EZ_EXTZ_ONLY);
		if (ok) {
			EzRes &e = RES(1);
			if (e.n_cigar > 0) { add_cigar(e.cigar, e.n_cigar); tmp.dp_score += e.max; }
			re1 = re_ + (e.reach_end? e.mqe_t + 1 : e.max_t + 1);
			qe1 = qe_ + (e.reach_end? qe0 - qe_ : e.max_q + 1);
		} else complete = false;
	}
	if (!complete) return false;
	// ---- commit
	if (split_at >= 0) {
		split_reg(r, r2, split_at, qlen, a);
		if (split_code == 2) r2->split_inv = 1;
	}
	r->rs = rs1, r->re = re1;
	if (rev) r->qs = qlen - qe1, r->qe = qlen - qs1;
	else r->qs = qs1, r->qe = qe1;
	if (have_p) {
		ProfScope pf(PF_UPDATE_EXTRA);
		{ ProfScope pf1(PF_X1);
		r->p = new Extra(tmp);
		tseq.resize((size_t)(re1 - rs1) + 1);
		getseq(mi, (uint32_t)rid, rs1, re1, tseq.data()); }
		const ExtraLoc loc = { (int32_t)r->rev, qs1, rid, rs1 };
		update_extra(r, rs.qc[r->rev].data() + qs1, tseq.data(), mat, (int8_t)opt->q, (int8_t)opt->e, rs.out_flags, (opt->flag & MMF_EQX) != 0, rs.defer_extra? &loc : 0);
	}
	return true;
}

// left-extension request (both strings reversed; U:align.c: KSW_EZ_EXTZ_ONLY|KSW_EZ_RIGHT|KSW_EZ_REV_CIGAR)
static void request_left(const mm355_mapopt_t *opt, int read_id, int task_id, AlnTask &T, std::vector<DpReq> &reqs)
{
	if (T.qs > 0 && T.rs > 0)
		want(T, 0, read_id, task_id, reqs, T.qs - T.qs0, T.rs - T.rs0, T.qs0, T.rev, (uint32_t)T.rid, T.rs0, 1, T.bw,
		     T.split_inv? opt->zdrop_inv : opt->zdrop, opt->end_bonus, EZ_EXTZ_ONLY | EZ_RIGHT | EZ_REV_CIGAR);
}

// U:align.c::mm_align1_inv; uses slot machinery of the *second* region's task (inv_res)
static void ensure_rev(ReadState &rs)
{
	if ((int)rs.qc[1].size() == rs.qlen) return;
	rs.qc[1].resize(rs.qlen);
	for (int i = 0; i < rs.qlen; ++i) { uint8_t c = rs.qc[0][i]; rs.qc[1][rs.qlen - 1 - i] = c < 4? 3 - c : 4; }
}

static int align1_inv(const mm355_index *mi, const mm355_mapopt_t *opt, int read_id, ReadState &rs, int i, int task_id, Reg *r_inv, std::vector<DpReq> &reqs, bool *pending)
{
	ensure_rev(rs);
	const Reg *r1 = &rs.regs[i - 1], *r2 = &rs.regs[i];
	AlnTask &T = rs.tasks[task_id];
	const int qlen = rs.qlen;
	int tl, ql, score, q_off, t_off;
	int8_t mat[25];
	*pending = false;
	*r_inv = Reg();
	if (!(r1->split & 1) || !(r2->split & 2)) return 0;
	if (r1->id != r1->parent && r1->parent != PARENT_TMP_PRI) return 0;
	if (r2->id != r2->parent && r2->parent != PARENT_TMP_PRI) return 0;
	if (r1->rid != r2->rid || r1->rev != r2->rev) return 0;
	ql = r1->rev? r1->qs - r2->qe : r2->qs - r1->qe;
	tl = r2->rs - r1->re;
	if (ql < opt->min_chain_score || ql > opt->max_gap) return 0;
	if (tl < opt->min_chain_score || tl > opt->max_gap) return 0;
	gen_simple_mat(mat, opt->a, opt->b, opt->sc_ambi);
	std::vector<uint8_t> tseq(tl), qseq(ql);
	getseq(mi, (uint32_t)r1->rid, r1->re, r2->rs, tseq.data());
	const int q_strand = r1->rev? 0 : 1;
	const int32_t q_base = r1->rev? r2->qe : qlen - r2->qs;
	memcpy(qseq.data(), rs.qc[q_strand].data() + q_base, ql);
	std::reverse(qseq.begin(), qseq.end()); std::reverse(tseq.begin(), tseq.end());
	score = ksw_ll(ql, qseq.data(), tl, tseq.data(), mat, opt->q, opt->e, &q_off, &t_off);
	std::reverse(qseq.begin(), qseq.end()); std::reverse(tseq.begin(), tseq.end());
	if (score < opt->min_dp_max) return 0;
	q_off = ql - (q_off + 1), t_off = tl - (t_off + 1);
	if (q_off < 0 || t_off < 0) return 0;
	if (T.inv_res.state != 2) {
		if (T.inv_res.state == 0) {
			DpReq q; q.read = read_id; q.task = task_id; q.slot = -1; q.qlen = ql - q_off; q.tlen = tl - t_off; q.q_st = q_base + q_off; q.rev_strand = q_strand;
			q.rid = (uint32_t)r1->rid; q.t_st = r1->re + t_off; q.reversed = 0; q.w = (int)(opt->bw * 1.5); q.zdrop = opt->zdrop; q.end_bonus = -1; q.flag = EZ_EXTZ_ONLY;
			reqs.push_back(q);
			T.inv_res.state = 1;
		}
		*pending = true;
		return 0;
	}
	const EzRes &ez = T.inv_res;
	if (ez.n_cigar <= 0) return 0;
	append_cigar(r_inv, ez.cigar, ez.n_cigar);
	r_inv->p->dp_score = ez.max;
	r_inv->id = -1;
	r_inv->parent = PARENT_UNSET;
	r_inv->inv = 1;
	r_inv->rev = !r1->rev;
	r_inv->rid = r1->rid;
	r_inv->div = -1.0f;
	if (r_inv->rev == 0) { r_inv->qs = r2->qe + q_off; r_inv->qe = r_inv->qs + ez.max_q + 1; }
	else { r_inv->qe = r2->qs - q_off; r_inv->qs = r_inv->qe - (ez.max_q + 1); }
	r_inv->rs = r1->re + t_off;
	r_inv->re = r_inv->rs + ez.max_t + 1;
	const ExtraLoc loc = { q_strand, q_base + q_off, r1->rid, r1->re + t_off };
	update_extra(r_inv, &qseq[q_off], &tseq[t_off], mat, (int8_t)opt->q, (int8_t)opt->e, rs.out_flags, (opt->flag & MMF_EQX) != 0, rs.defer_extra? &loc : 0);
	return 1;
}

static int ensure_task(const mm355_index *mi, const mm355_mapopt_t *opt, ReadState &rs, int ri)
{
	Reg &r = rs.regs[ri];
	if (r.task < 0) {
		r.task = (int)rs.tasks.size();
		rs.tasks.emplace_back();
		rs.tasks.back().reg_uid = rs.next_uid++;
	}
	AlnTask &T = rs.tasks[r.task];
	if (!T.prepared && r.cnt > 0) task_prepare(mi, opt, rs, &r, T);
	return r.task;
}

bool mm355_glue_align_step(const mm355_index *mi, const mm355_mapopt_t *opt, int read_id, ReadState &rs, std::vector<DpReq> &reqs, int flags)
{
	rs.out_flags = flags;
	if (rs.aligned) return true;
	ProfScope pf_step(PF_STEP_TOTAL);
	// commit in skeleton order
	while (rs.cursor < (int)rs.regs.size()) {
		const int i = rs.cursor;
		Reg r2;
		int tid = -1;
		if (rs.regs[i].cnt > 0 && !rs.regs[i].inv) {
			tid = ensure_task(mi, opt, rs, i);
			AlnTask &T = rs.tasks[tid];
			if (!T.done) {
				request_left(opt, read_id, tid, T, reqs);
				if (!task_run(mi, opt, read_id, rs, i, tid, &r2, reqs)) break;
				rs.tasks[tid].done = true;
				if (r2.cnt > 0) { rs.regs.insert(rs.regs.begin() + i + 1, r2); }
			}
		}
		if (i > 0 && rs.regs[i].split_inv && !(opt->flag & MMF_NO_INV) && tid >= 0) {
			AlnTask &T = rs.tasks[tid];
			if (T.inv_state != 2) {
				Reg rinv; bool pending = false;
				int ok = align1_inv(mi, opt, read_id, rs, i, tid, &rinv, reqs, &pending);
				if (pending) break;
				T.inv_state = 2;
				if (ok) { rs.regs.insert(rs.regs.begin() + i + 1, rinv); ++rs.cursor; }
			}
		}
		++rs.cursor;
	}
	if (rs.cursor >= (int)rs.regs.size()) { rs.aligned = true; return true; }
	// look ahead: collect requests of later regions so that the next round serves them too
	for (int i = rs.cursor + 1; i < (int)rs.regs.size(); ++i) {
		Reg &r = rs.regs[i];
		if (r.cnt == 0 || r.inv) continue;
		int tid = ensure_task(mi, opt, rs, i);
		AlnTask &T = rs.tasks[tid];
		if (T.done) continue;
		request_left(opt, read_id, tid, T, reqs);
		// dry run: collects the fill / right-extension requests without committing anything
		Reg save = rs.regs[i];
		Reg r2;
		if (task_run(mi, opt, read_id, rs, i, tid, &r2, reqs)) {
			// everything was already cached: undo the commit, the skeleton loop will redo it in order
			if (rs.regs[i].p && rs.regs[i].p != save.p) delete rs.regs[i].p;
			rs.regs[i] = save;
		}
	}
	return false;
}

// ================================================================== cs / MD (U:format.c)
static void get_aln_seqs(const mm355_index *mi, const ReadState &rs, const Reg *r, std::vector<uint8_t> &q, std::vector<uint8_t> &t)
{
	q.resize((size_t)(r->qe - r->qs) + 1); t.resize((size_t)(r->re - r->rs) + 1);
	getseq(mi, (uint32_t)r->rid, r->rs, r->re, t.data());
	if (!r->rev) for (int i = r->qs; i < r->qe; ++i) q[i - r->qs] = rs.qc[0][i];
	else for (int i = r->qs; i < r->qe; ++i) { uint8_t c = rs.qc[0][i]; q[r->qe - i - 1] = c >= 4? 4 : 3 - c; }
}

static inline void put_uint(std::string &s, char lead, unsigned v)   // lead (if non-zero) followed by v in decimal
{
	char buf[12]; int n = 0;
	do { buf[n++] = (char)('0' + v % 10); v /= 10; } while (v);
	if (lead) s += lead;
	while (n > 0) s += buf[--n];
}
// length of the common prefix of q[0..len) and t[0..len), eight bases per step
static inline int match_run(const uint8_t *q, const uint8_t *t, int len)
{
	int j = 0;
	while (j + 8 <= len) {
		uint64_t qw, tw; memcpy(&qw, q + j, 8); memcpy(&tw, t + j, 8);
		const uint64_t x = qw ^ tw;
		if (x) return j + (int)(__builtin_ctzll(x) >> 3);
		j += 8;
	}
	while (j < len && q[j] == t[j]) ++j;
	return j;
}

static void gen_cs(const Reg *r, const uint8_t *qseq, const uint8_t *tseq, std::string &s)   // U:format.c::write_cs_core (short form, no introns)
{
	int q_off = 0, t_off = 0;
	s.reserve(s.size() + (size_t)(r->qe - r->qs) / 2 + 64);
	for (size_t i = 0; i < r->p->cigar.size(); ++i) {
		int op = r->p->cigar[i] & 0xf, len = r->p->cigar[i] >> 4;
		if (op == 0 || op == 7 || op == 8) {
			int j = 0;
			while (j < len) {
				const int m = match_run(qseq + q_off + j, tseq + t_off + j, len - j);
				if (m > 0) { put_uint(s, ':', (unsigned)m); j += m; }
				if (j < len) { s += '*'; s += "acgtn"[tseq[t_off + j]]; s += "acgtn"[qseq[q_off + j]]; ++j; }
			}
			q_off += len, t_off += len;
		} else if (op == 1) { s += '+'; for (int j = 0; j < len; ++j) s += "acgtn"[qseq[q_off + j]]; q_off += len; }
		else if (op == 2) { s += '-'; for (int j = 0; j < len; ++j) s += "acgtn"[tseq[t_off + j]]; t_off += len; }
		else t_off += len;
	}
}

static void gen_md(const Reg *r, const uint8_t *qseq, const uint8_t *tseq, std::string &s)   // U:format.c::write_MD_core
{
	int q_off = 0, t_off = 0, l_MD = 0;
	for (size_t i = 0; i < r->p->cigar.size(); ++i) {
		int op = r->p->cigar[i] & 0xf, len = r->p->cigar[i] >> 4;
		if (op == 0 || op == 7 || op == 8) {
			int j = 0;
			while (j < len) {
				const int m = match_run(qseq + q_off + j, tseq + t_off + j, len - j);
				l_MD += m; j += m;
				if (j < len) { put_uint(s, 0, (unsigned)l_MD); s += "ACGTN"[tseq[t_off + j]]; l_MD = 0; ++j; }
			}
			q_off += len, t_off += len;
		} else if (op == 1) q_off += len;
		else if (op == 2) {
			put_uint(s, 0, (unsigned)l_MD); s += '^';
			for (int j = 0; j < len; ++j) s += "ACGTN"[tseq[t_off + j]];
			l_MD = 0; t_off += len;
		} else if (op == 3) t_off += len;
	}
	if (l_MD > 0) put_uint(s, 0, (unsigned)l_MD);
}

// ================================================================== stage 3
void mm355_glue_finish(const mm355_index *mi, const mm355_mapopt_t *opt, ReadState &rs, int flags,
                       std::vector<mm355_hit_t> &hits, std::vector<uint32_t> &cigar, std::string &str)
{
	int n = (int)rs.regs.size();
	Reg *regs = rs.regs.data();
	filter_regs(opt, rs.qlen, &n, regs);
	hit_sort(&n, regs);
	if (!(opt->flag & MMF_ALL_CHAINS)) {
		set_parent(opt->mask_level, opt->mask_len, n, regs, opt->a * 2 + opt->b, (int)(opt->flag & MMF_HARD_MLEVEL));
		select_sub(opt->pri_ratio, mi->k * 2, opt->best_n, 0, (int)(opt->max_gap * 0.8), &n, regs);
		set_sam_pri(n, regs);
	}
	set_mapq(n, regs, opt->min_chain_score, opt->a, rs.rep_len);
	rs.regs.resize(n);
	std::vector<uint8_t> q, t;
	for (int i = 0; i < n; ++i) {
		const Reg *r = &regs[i];
		mm355_hit_t h;
		memset(&h, 0, sizeof(h));
		h.query_start = r->qs; h.query_end = r->qe; h.strand = r->rev? -1 : 1; h.rid = r->rid;
		h.target_len = (int32_t)mi->seq_len[r->rid]; h.target_start = r->rs; h.target_end = r->re;
		h.match_len = r->mlen; h.block_len = r->blen; h.mapq = r->mapq; h.is_primary = r->parent == r->id;
		h.cs_len = h.md_len = -1;
		h.score0 = r->score0; h.cnt = r->cnt; h.n_sub = r->n_sub; h.subsc = r->subsc;
		if (r->p) {
			h.NM = r->blen - r->mlen + (int32_t)r->p->n_ambi;
			h.n_cigar = (int32_t)r->p->cigar.size(); h.cigar_off = (int64_t)cigar.size();
			cigar.insert(cigar.end(), r->p->cigar.begin(), r->p->cigar.end());
			h.dp_max = r->p->dp_max; h.dp_max2 = r->p->dp_max2; h.dp_score = r->p->dp_score;
			if (flags & MM355_OUT_CS) { h.cs_off = (int64_t)str.size(); h.cs_len = (int64_t)r->p->cs.size(); str += r->p->cs; str += '\0'; }
			if (flags & MM355_OUT_MD) { h.md_off = (int64_t)str.size(); h.md_len = (int64_t)r->p->md.size(); str += r->p->md; str += '\0'; }
		}
		hits.push_back(h);
	}
}

// ---- regions whose mm_update_extra walk / cs / MD were left to the device
void mm355_glue_extra_count(const ReadState &rs, int64_t *n_regions, int64_t *n_segs, int64_t *n_cig, int64_t *n_cs)
{
	*n_regions = *n_segs = *n_cig = *n_cs = 0;
	for (const Reg &r : rs.regs) if (r.p && r.p->deferred) {
		int64_t cc = 0, mc = 0;
		++*n_regions; *n_segs += mm355_extra_split(r.p->cigar.data(), (int)r.p->cigar.size(), 0, 0, 0, 0, 0, 0, 0, 0, &cc, &mc);
		*n_cig += (int64_t)r.p->cigar.size(); *n_cs += cc + mc;
	}
}
// segs: this read's segment descriptors (global index seg0 + ...); seg_first[reg0 + k]: global index of region k's first segment
void mm355_glue_extra_fill(const ReadState &rs, int64_t q_base, Mm355ExtraJob *segs, int64_t *seg_first, int64_t reg0, int64_t seg0, uint32_t *cig, int64_t cig0, int64_t cs0)
{
	int64_t k = 0, g = 0;
	for (const Reg &r : rs.regs) if (r.p && r.p->deferred) {
		const std::vector<uint32_t> &cg = r.p->cigar;
		const int n = (int)cg.size();
		seg_first[reg0 + k] = seg0 + g;
		if (n) memcpy(cig + cig0, cg.data(), (size_t)n * 4);
		int64_t cc = 0, mc = 0;
		mm355_extra_split(cg.data(), n, 0, 0, 0, 0, 0, 0, 0, 0, &cc, &mc);
		g += mm355_extra_split(cg.data(), n, q_base + (r.p->x_strand? rs.qlen : 0) + r.p->x_qst, (uint32_t)r.p->x_rid, r.p->x_tst, cig0, cs0, cs0 + cc, (int32_t)(reg0 + k), segs + g, 0, 0);
		++k; cig0 += n; cs0 += cc + mc;
	}
}
void mm355_glue_extra_apply(ReadState &rs, const Mm355ExtraOut *out, const char *cs, int want)
{
	int64_t k = 0;
	for (Reg &r : rs.regs) if (r.p && r.p->deferred) {
		const Mm355ExtraOut &o = out[k++];
		r.mlen = o.mlen; r.blen = o.blen; r.p->n_ambi += (uint32_t)o.n_ambi; r.p->dp_max = o.dp_max;
		if (want & MM355_OUT_CS) r.p->cs.assign(cs + o.cs_dense, (size_t)o.cs_len);
		if (want & MM355_OUT_MD) r.p->md.assign(cs + o.cs_dense + o.cs_len, (size_t)o.md_len);
		r.p->deferred = false;
	}
}

void mm355_glue_release(ReadState &rs)
{
	for (Reg &r : rs.regs) { delete r.p; r.p = 0; }
	rs.regs.clear(); rs.tasks.clear();
}
