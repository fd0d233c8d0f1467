/* ORACLE (test infrastructure).  Restates U:sketch.c::mm_sketch (+hash64,
 * seq_nt4_table) of minimap2 2.26; reached from R:src/lib.rs:482 / :587 through
 * mm_map -> mm_map_frag -> collect_minimizers, and from R:src/lib.rs:407-410
 * (index build from FASTA).  PINNED bit-exactly by the reference fixture pair
 * test.fa <-> test.mmi (tests/test_oracle_golden.py).
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "mmo.h"

unsigned char mmo_seq_nt4_table[256] = {
	0, 1, 2, 3,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 0, 4, 1,  4, 4, 4, 2,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  3, 3, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 0, 4, 1,  4, 4, 4, 2,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  3, 3, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4
};

static inline uint64_t hash64(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask;
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask;
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask;
	key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

typedef struct { int front, count; int a[32]; } tiny_queue_t;
static inline void tq_push(tiny_queue_t *q, int x) { q->a[((q->count++) + q->front) & 0x1f] = x; }
static inline int tq_shift(tiny_queue_t *q)
{
	int x;
	if (q->count == 0) return -1;
	x = q->a[q->front++];
	q->front &= 0x1f;
	--q->count;
	return x;
}

static inline void mv_push(mm128_v *p, mm128_t v)
{
	if (p->n == p->m) {
		p->m = p->m? p->m<<1 : 256;
		p->a = (mm128_t*)realloc(p->a, p->m * sizeof(mm128_t));
	}
	p->a[p->n++] = v;
}

/* U:sketch.c::mm_sketch */
void mmo_sketch(const char *str, int len, int w, int k, uint32_t rid, int is_hpc, mm128_v *p)
{
	uint64_t shift1 = 2 * (k - 1), mask = (1ULL<<2*k) - 1, kmer[2] = {0,0};
	int i, j, l, buf_pos, min_pos, kmer_span = 0;
	mm128_t buf[256], min = { UINT64_MAX, UINT64_MAX };
	tiny_queue_t tq;

	assert(len > 0 && (w > 0 && w < 256) && (k > 0 && k <= 28));
	memset(buf, 0xff, w * 16);
	memset(&tq, 0, sizeof(tiny_queue_t));

	for (i = l = buf_pos = min_pos = 0; i < len; ++i) {
		int c = mmo_seq_nt4_table[(uint8_t)str[i]];
		mm128_t info = { UINT64_MAX, UINT64_MAX };
		if (c < 4) { /* not an ambiguous base */
			int z;
			if (is_hpc) {
				int skip_len = 1;
				if (i + 1 < len && mmo_seq_nt4_table[(uint8_t)str[i + 1]] == c) {
					for (skip_len = 2; i + skip_len < len; ++skip_len)
						if (mmo_seq_nt4_table[(uint8_t)str[i + skip_len]] != c)
							break;
					i += skip_len - 1;
				}
				tq_push(&tq, skip_len);
				kmer_span += skip_len;
				if (tq.count > k) kmer_span -= tq_shift(&tq);
			} else kmer_span = l + 1 < k? l + 1 : k;
			kmer[0] = (kmer[0] << 2 | c) & mask;
			kmer[1] = (kmer[1] >> 2) | (3ULL^c) << shift1;
			if (kmer[0] == kmer[1]) continue; /* skip "symmetric k-mers": strand unknown */
			z = kmer[0] < kmer[1]? 0 : 1;
			++l;
			if (l >= k && kmer_span < 256) {
				info.x = hash64(kmer[z], mask) << 8 | kmer_span;
				info.y = (uint64_t)rid<<32 | (uint32_t)i<<1 | z;
			}
		} else l = 0, tq.count = tq.front = 0, kmer_span = 0;
		buf[buf_pos] = info;
		if (l == w + k - 1 && min.x != UINT64_MAX) { /* first window: identical k-mers not stored yet */
			for (j = buf_pos + 1; j < w; ++j)
				if (min.x == buf[j].x && buf[j].y != min.y) mv_push(p, buf[j]);
			for (j = 0; j < buf_pos; ++j)
				if (min.x == buf[j].x && buf[j].y != min.y) mv_push(p, buf[j]);
		}
		if (info.x <= min.x) { /* a new minimum; then write the old min */
			if (l >= w + k && min.x != UINT64_MAX) mv_push(p, min);
			min = info, min_pos = buf_pos;
		} else if (buf_pos == min_pos) { /* old min has moved outside the window */
			if (l >= w + k - 1 && min.x != UINT64_MAX) mv_push(p, min);
			for (j = buf_pos + 1, min.x = UINT64_MAX; j < w; ++j)
				if (min.x >= buf[j].x) min = buf[j], min_pos = j; /* >= : rightmost wins */
			for (j = 0; j <= buf_pos; ++j)
				if (min.x >= buf[j].x) min = buf[j], min_pos = j;
			if (l >= w + k - 1 && min.x != UINT64_MAX) { /* write identical k-mers */
				for (j = buf_pos + 1; j < w; ++j)
					if (min.x == buf[j].x && min.y != buf[j].y) mv_push(p, buf[j]);
				for (j = 0; j <= buf_pos; ++j)
					if (min.x == buf[j].x && min.y != buf[j].y) mv_push(p, buf[j]);
			}
		}
		if (++buf_pos == w) buf_pos = 0;
	}
	if (min.x != UINT64_MAX)
		mv_push(p, min);
}
