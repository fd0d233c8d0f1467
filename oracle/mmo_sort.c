/* ORACLE (test infrastructure).  Restates U:ksort.h (klib radix sort / heap /
 * quick-select) as instantiated by minimap2 2.26 U:misc.c:
 *   KRADIX_SORT_INIT(128x, mm128_t, sort_key_128x, 8)   key = .x
 *   KRADIX_SORT_INIT(64,   uint64_t, sort_key_64,  8)
 * The in-place MSD radix sort is deterministic but NOT stable; the order of
 * equal keys is defined only by this exact procedure and is
 * observable downstream (anchor order -> chaining, z[] -> backtrack order).
 * Reference call sites: every mm_map() at R:src/lib.rs:482 / :587.
 */
#include <string.h>
#include <assert.h>
#include "mmo.h"

#define RS_MIN_SIZE 64
#define RS_MAX_BITS 8

#define MMO_RADIX_SORT(name, rstype_t, rskey, sizeof_key) \
	typedef struct { rstype_t *b, *e; } rsbucket_##name##_t; \
	static void rs_insertsort_##name(rstype_t *beg, rstype_t *end) \
	{ \
		rstype_t *i; \
		for (i = beg + 1; i < end; ++i) \
			if (rskey(*i) < rskey(*(i - 1))) { \
				rstype_t *j, tmp = *i; \
				for (j = i; j > beg && rskey(tmp) < rskey(*(j-1)); --j) \
					*j = *(j - 1); \
				*j = tmp; \
			} \
	} \
	static void rs_sort_##name(rstype_t *beg, rstype_t *end, int n_bits, int s) \
	{ \
		rstype_t *i; \
		int size = 1<<n_bits, m = size - 1; \
		rsbucket_##name##_t *k, b[1<<RS_MAX_BITS], *be = b + size; \
		assert(n_bits <= RS_MAX_BITS); \
		for (k = b; k != be; ++k) k->b = k->e = beg; \
		for (i = beg; i != end; ++i) ++b[rskey(*i)>>s&m].e; \
		for (k = b + 1; k != be; ++k) \
			k->e += (k-1)->e - beg, k->b = (k-1)->e; \
		for (k = b; k != be;) { \
			if (k->b != k->e) { \
				rsbucket_##name##_t *l; \
				if ((l = b + (rskey(*k->b)>>s&m)) != k) { \
					rstype_t tmp = *k->b, swap; \
					do { \
						swap = tmp; tmp = *l->b; *l->b++ = swap; \
						l = b + (rskey(tmp)>>s&m); \
					} while (l != k); \
					*k->b++ = tmp; \
				} else ++k->b; \
			} else ++k; \
		} \
		for (b->b = beg, k = b + 1; k != be; ++k) k->b = (k-1)->e; \
		if (s) { \
			s = s > n_bits? s - n_bits : 0; \
			for (k = b; k != be; ++k) \
				if (k->e - k->b > RS_MIN_SIZE) rs_sort_##name(k->b, k->e, n_bits, s); \
				else if (k->e - k->b > 1) rs_insertsort_##name(k->b, k->e); \
		} \
	} \
	void mmo_radix_sort_##name(rstype_t *beg, rstype_t *end) \
	{ \
		if (end - beg <= RS_MIN_SIZE) rs_insertsort_##name(beg, end); \
		else rs_sort_##name(beg, end, RS_MAX_BITS, (sizeof_key - 1) * RS_MAX_BITS); \
	}

#define sort_key_128x(a) ((a).x)
MMO_RADIX_SORT(128x, mm128_t, sort_key_128x, 8)

#define sort_key_64(a) (a)
MMO_RADIX_SORT(64, uint64_t, sort_key_64, 8)

/* U:ksort.h::ks_ksmall (quick-select); only the selected VALUE is observable */
uint32_t mmo_ksmall_u32(size_t n, uint32_t *arr, size_t kk)
{
	uint32_t *low, *high, *k, *ll, *hh, *mid, t;
	low = arr; high = arr + n - 1; k = arr + kk;
	for (;;) {
		if (high <= low) return *k;
		if (high == low + 1) {
			if (*high < *low) t = *low, *low = *high, *high = t;
			return *k;
		}
		mid = low + (high - low) / 2;
		if (*high < *mid) t = *mid, *mid = *high, *high = t;
		if (*high < *low) t = *low, *low = *high, *high = t;
		if (*low < *mid) t = *mid, *mid = *low, *low = t;
		t = *mid, *mid = *(low+1), *(low+1) = t;
		ll = low + 1; hh = high;
		for (;;) {
			do ++ll; while (*ll < *low);
			do --hh; while (*low < *hh);
			if (hh < ll) break;
			t = *ll, *ll = *hh, *hh = t;
		}
		t = *low, *low = *hh, *hh = t;
		if (hh <= k) low = ll;
		if (hh >= k) high = hh - 1;
	}
}
