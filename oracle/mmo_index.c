/* ORACLE (test infrastructure).  Restates U:index.c of minimap2 2.26:
 * mm_idx_get, mm_idx_getseq, mm_idx_cal_max_occ, mm_idx_load/dump (MMI\2), and
 * the FASTA -> index build (mm_idx_gen / worker_post).  Reference call sites:
 * mm_idx_reader_open/read R:src/lib.rs:397-410, mm_idx_index_name :416,
 * mm_idx_name2id :716, mm_idx_getseq :747.  The per-bucket hash table is this
 * file's own open-addressing table: only the (minimizer -> positions) contract
 * of mm_idx_get is normative, not khash's slot order.  Key/value encoding and the
 * on-disk format are PINNED by the reference's fixture test.mmi.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include <pthread.h>
#include "mmo.h"

#define MM_IDX_MAGIC "MMI\2"
#define mm_seq4_set(s, i, c) ((s)[(i)>>3] |= (uint32_t)(c) << (((i)&7)<<2))
#define mm_seq4_get(s, i)    ((s)[(i)>>3] >> (((i)&7)<<2) & 0xf)

static inline uint32_t key_hash(uint64_t key) /* any mixing is fine: layout is ours */
{
	uint64_t h = (key >> 1) * 0x9E3779B97F4A7C15ULL;
	return (uint32_t)(h >> 32);
}

static void bucket_resize(mmo_bucket_t *b, uint32_t n_keys)
{
	uint32_t cap = 4, i;
	while (cap < n_keys * 2u) cap <<= 1;
	b->cap = cap; b->n_keys = 0;
	b->keys = (uint64_t*)malloc(cap * 8);
	b->vals = (uint64_t*)malloc(cap * 8);
	for (i = 0; i < cap; ++i) b->keys[i] = UINT64_MAX;
}

static void bucket_put(mmo_bucket_t *b, uint64_t key, uint64_t val)
{
	uint32_t m = b->cap - 1, i = key_hash(key) & m;
	while (b->keys[i] != UINT64_MAX) {
		assert((b->keys[i]>>1) != (key>>1));
		i = (i + 1) & m;
	}
	b->keys[i] = key, b->vals[i] = val, ++b->n_keys;
}

/* U:index.c::mm_idx_get */
const uint64_t *mmo_idx_get(const mmo_idx_t *mi, uint64_t minier, int *n)
{
	int mask = (1<<mi->b) - 1;
	const mmo_bucket_t *b = &mi->B[minier&mask];
	uint64_t key = minier>>mi->b<<1;
	uint32_t m, i;
	*n = 0;
	if (b->cap == 0) return 0;
	m = b->cap - 1, i = key_hash(key) & m;
	while (b->keys[i] != UINT64_MAX) {
		if ((b->keys[i]>>1) == (key>>1)) {
			if (b->keys[i]&1) { /* singleton */
				*n = 1;
				return &b->vals[i];
			} else {
				*n = (uint32_t)b->vals[i];
				return &b->p[b->vals[i]>>32];
			}
		}
		i = (i + 1) & m;
	}
	return 0;
}

/* U:index.c::mm_idx_getseq */
int mmo_idx_getseq(const mmo_idx_t *mi, uint32_t rid, uint32_t st, uint32_t en, uint8_t *seq)
{
	uint64_t i, st1, en1;
	if (rid >= mi->n_seq || st >= mi->seq[rid].len) return -1;
	if (en > mi->seq[rid].len) en = mi->seq[rid].len;
	st1 = mi->seq[rid].offset + st;
	en1 = mi->seq[rid].offset + en;
	for (i = st1; i < en1; ++i)
		seq[i - st1] = mm_seq4_get(mi->S, i);
	return en - st;
}

/* U:index.c::mm_idx_name2id (hash in upstream; linear here, same answer) */
int mmo_idx_name2id(const mmo_idx_t *mi, const char *name)
{
	uint32_t i;
	for (i = 0; i < mi->n_seq; ++i)
		if (mi->seq[i].name && strcmp(mi->seq[i].name, name) == 0) return (int)i;
	return -1;
}

/* U:index.c::mm_idx_cal_max_occ */
int32_t mmo_idx_cal_max_occ(const mmo_idx_t *mi, float f)
{
	int i;
	size_t n = 0;
	uint32_t thres, *a, k;
	if (f <= 0.) return INT32_MAX;
	for (i = 0; i < 1<<mi->b; ++i) n += mi->B[i].n_keys;
	if (n == 0) return INT32_MAX;
	a = (uint32_t*)malloc(n * 4);
	for (i = 0, n = 0; i < 1<<mi->b; ++i) {
		const mmo_bucket_t *b = &mi->B[i];
		for (k = 0; k < b->cap; ++k) {
			if (b->keys[k] == UINT64_MAX) continue;
			a[n++] = b->keys[k]&1? 1 : (uint32_t)b->vals[k];
		}
	}
	thres = mmo_ksmall_u32(n, a, (uint32_t)((1. - f) * n)) + 1;
	free(a);
	return thres;
}

int64_t mmo_idx_n_minimizers(const mmo_idx_t *mi, int64_t *n_distinct)
{
	int i; uint32_t k; int64_t tot = 0, nd = 0;
	for (i = 0; i < 1<<mi->b; ++i) {
		const mmo_bucket_t *b = &mi->B[i];
		for (k = 0; k < b->cap; ++k) {
			if (b->keys[k] == UINT64_MAX) continue;
			++nd; tot += b->keys[k]&1? 1 : (uint32_t)b->vals[k];
		}
	}
	if (n_distinct) *n_distinct = nd;
	return tot;
}

void mmo_idx_destroy(mmo_idx_t *mi)
{
	uint32_t i;
	if (mi == 0) return;
	if (mi->B) {
		for (i = 0; i < 1U<<mi->b; ++i) {
			free(mi->B[i].p); free(mi->B[i].keys); free(mi->B[i].vals); free(mi->B[i].a.a);
		}
		free(mi->B);
	}
	for (i = 0; i < mi->n_seq; ++i) free(mi->seq[i].name);
	free(mi->seq); free(mi->S); free(mi);
}

static mmo_idx_t *idx_init(int w, int k, int b, int flag)
{
	mmo_idx_t *mi;
	if (k*2 < b) b = k * 2;
	if (w < 1) w = 1;
	mi = (mmo_idx_t*)calloc(1, sizeof(mmo_idx_t));
	mi->w = w, mi->k = k, mi->b = b, mi->flag = flag;
	mi->B = (mmo_bucket_t*)calloc((size_t)1<<b, sizeof(mmo_bucket_t));
	return mi;
}

/* U:index.c::mm_idx_load */
static mmo_idx_t *idx_load_mmi(FILE *fp)
{
	char magic[4];
	uint32_t x[5], i;
	uint64_t sum_len = 0;
	mmo_idx_t *mi;

	if (fread(magic, 1, 4, fp) != 4) return 0;
	if (strncmp(magic, MM_IDX_MAGIC, 4) != 0) return 0;
	if (fread(x, 4, 5, fp) != 5) return 0;
	mi = idx_init(x[0], x[1], x[2], x[4]);
	mi->n_seq = x[3];
	mi->seq = (mmo_idx_seq_t*)calloc(mi->n_seq, sizeof(mmo_idx_seq_t));
	for (i = 0; i < mi->n_seq; ++i) {
		uint8_t l;
		mmo_idx_seq_t *s = &mi->seq[i];
		if (fread(&l, 1, 1, fp) != 1) goto fail;
		if (l) {
			s->name = (char*)malloc(l + 1);
			if (fread(s->name, 1, l, fp) != l) goto fail;
			s->name[l] = 0;
		}
		if (fread(&s->len, 4, 1, fp) != 1) goto fail;
		s->offset = sum_len;
		sum_len += s->len;
	}
	for (i = 0; i < 1U<<mi->b; ++i) {
		mmo_bucket_t *b = &mi->B[i];
		uint32_t j, size;
		if (fread(&b->n, 4, 1, fp) != 1) goto fail;
		b->p = (uint64_t*)malloc((size_t)(b->n > 0? b->n : 1) * 8);
		if (b->n > 0 && fread(b->p, 8, b->n, fp) != (size_t)b->n) goto fail;
		if (fread(&size, 4, 1, fp) != 1) goto fail;
		if (size == 0) continue;
		bucket_resize(b, size);
		for (j = 0; j < size; ++j) {
			uint64_t kv[2];
			if (fread(kv, 8, 2, fp) != 2) goto fail;
			bucket_put(b, kv[0], kv[1]);
		}
	}
	if (!(mi->flag & MM_I_NO_SEQ)) {
		size_t n = (sum_len + 7) / 8;
		mi->S = (uint32_t*)malloc((n? n : 1) * 4);
		if (fread(mi->S, 4, n, fp) != n) goto fail;
	}
	return mi;
fail:
	mmo_idx_destroy(mi);
	return 0;
}

/* U:index.c::mm_idx_dump */
int mmo_idx_dump(const mmo_idx_t *mi, const char *fn)
{
	uint64_t sum_len = 0;
	uint32_t x[5], i;
	FILE *fp = fopen(fn, "wb");
	if (fp == 0) return -1;
	x[0] = mi->w, x[1] = mi->k, x[2] = mi->b, x[3] = mi->n_seq, x[4] = mi->flag;
	fwrite(MM_IDX_MAGIC, 1, 4, fp);
	fwrite(x, 4, 5, fp);
	for (i = 0; i < mi->n_seq; ++i) {
		uint8_t l = mi->seq[i].name? (uint8_t)strlen(mi->seq[i].name) : 0;
		fwrite(&l, 1, 1, fp);
		if (l) fwrite(mi->seq[i].name, 1, l, fp);
		fwrite(&mi->seq[i].len, 4, 1, fp);
		sum_len += mi->seq[i].len;
	}
	for (i = 0; i < 1U<<mi->b; ++i) {
		const mmo_bucket_t *b = &mi->B[i];
		uint32_t k, size = b->n_keys;
		fwrite(&b->n, 4, 1, fp);
		if (b->n > 0) fwrite(b->p, 8, b->n, fp);
		fwrite(&size, 4, 1, fp);
		for (k = 0; k < b->cap; ++k) {
			uint64_t kv[2];
			if (b->keys[k] == UINT64_MAX) continue;
			kv[0] = b->keys[k], kv[1] = b->vals[k];
			fwrite(kv, 8, 2, fp);
		}
	}
	if (!(mi->flag & MM_I_NO_SEQ))
		fwrite(mi->S, 4, (sum_len + 7) / 8, fp);
	fclose(fp);
	return 0;
}

/* U:index.c::worker_post */
static void bucket_post(mmo_idx_t *mi, mmo_bucket_t *b)
{
	int n, n_keys;
	size_t j, start_a, start_p;
	if (b->a.n == 0) return;
	mmo_radix_sort_128x(b->a.a, b->a.a + b->a.n);
	for (j = 1, n = 1, n_keys = 0, b->n = 0; j <= b->a.n; ++j) {
		if (j == b->a.n || b->a.a[j].x>>8 != b->a.a[j-1].x>>8) {
			++n_keys;
			if (n > 1) b->n += n;
			n = 1;
		} else ++n;
	}
	bucket_resize(b, n_keys);
	b->p = (uint64_t*)calloc(b->n > 0? b->n : 1, 8);
	for (j = 1, n = 1, start_a = start_p = 0; j <= b->a.n; ++j) {
		if (j == b->a.n || b->a.a[j].x>>8 != b->a.a[j-1].x>>8) {
			mm128_t *p = &b->a.a[j-1];
			uint64_t key = p->x>>8>>mi->b<<1;
			assert(j == start_a + n);
			if (n == 1) {
				bucket_put(b, key | 1, p->y);
			} else {
				int k;
				for (k = 0; k < n; ++k)
					b->p[start_p + k] = b->a.a[start_a + k].y;
				mmo_radix_sort_64(&b->p[start_p], &b->p[start_p + n]); /* sort by position */
				bucket_put(b, key, (uint64_t)start_p<<32 | n);
				start_p += n;
			}
			start_a = j, n = 1;
		} else ++n;
	}
	assert(b->n == (int32_t)start_p);
	free(b->a.a);
	b->a.n = b->a.m = 0, b->a.a = 0;
}

static void idx_add(mmo_idx_t *mi, int n, const mm128_t *a)
{
	int i, mask = (1<<mi->b) - 1;
	for (i = 0; i < n; ++i) {
		mm128_v *p = &mi->B[a[i].x>>8&mask].a;
		if (p->n == p->m) {
			p->m = p->m? p->m<<1 : 8;
			p->a = (mm128_t*)realloc(p->a, p->m * sizeof(mm128_t));
		}
		p->a[p->n++] = a[i];
	}
}

/* add one sequence: U:index.c::worker_pipeline steps 0-2 */
static void idx_add_seq(mmo_idx_t *mi, uint64_t *sum_len, size_t *m_S, const char *name, const char *seq, uint32_t len)
{
	mmo_idx_seq_t *s;
	uint64_t need = (*sum_len + len + 7) / 8;
	uint32_t j;
	mm128_v a = {0,0,0};
	mi->seq = (mmo_idx_seq_t*)realloc(mi->seq, (mi->n_seq + 1) * sizeof(mmo_idx_seq_t));
	s = &mi->seq[mi->n_seq];
	s->name = (mi->flag & MM_I_NO_NAME) || name == 0? 0 : strdup(name);
	s->len = len, s->offset = *sum_len, s->is_alt = 0;
	if (!(mi->flag & MM_I_NO_SEQ)) {
		if (need > *m_S) {
			size_t old = *m_S;
			*m_S = need + (need>>1) + 1024;
			mi->S = (uint32_t*)realloc(mi->S, *m_S * 4);
			memset(mi->S + old, 0, (*m_S - old) * 4);
		}
		for (j = 0; j < len; ++j) {
			uint64_t o = *sum_len + j;
			int c = mmo_seq_nt4_table[(uint8_t)seq[j]];
			mm_seq4_set(mi->S, o, c);
		}
	}
	*sum_len += len;
	if (len > 0) {
		mmo_sketch(seq, len, mi->w, mi->k, mi->n_seq, mi->flag&MM_I_HPC, &a);
		idx_add(mi, (int)a.n, a.a);
		free(a.a);
	}
	++mi->n_seq;
}

static void idx_finish(mmo_idx_t *mi)
{
	uint32_t i;
	for (i = 0; i < 1U<<mi->b; ++i) bucket_post(mi, &mi->B[i]);
}

mmo_idx_t *mmo_idx_build_mem(int w, int k, int b, int flag, int n_seq, const char **seqs, const int *lens, const char **names)
{
	mmo_idx_t *mi = idx_init(w, k, b, flag);
	uint64_t sum_len = 0; size_t m_S = 0; int i;
	for (i = 0; i < n_seq; ++i)
		idx_add_seq(mi, &sum_len, &m_S, names? names[i] : 0, seqs[i], lens[i]);
	idx_finish(mi);
	return mi;
}

/* Threaded build (U:index.c::mm_idx_gen runs its pipeline with kt_for over the sequences of a mini-batch for mm_sketch and kt_for
 * over the buckets for worker_post; same two parallel loops here, with plain pthreads).  A worker sketches whole sequences and
 * files the minimizers into its OWN per-bucket lists; the bucket loop concatenates the workers' lists of a bucket and runs
 * bucket_post on it.  The index does not depend on the insertion order: worker_post sorts a bucket by minimizer and each
 * position run by position.  Used by bench.py's cpu_baseline on the GRCh38-scale genome (a single-threaded build of 3.1 Gbp
 * does not fit the bench's time budget); seqs[i] may be ASCII or raw codes 0..4 (seq_nt4_table maps both). */
typedef struct {
	mmo_idx_t *mi; int n_seq, n_threads, tid;
	const char **seqs; const int *lens;
	volatile int *next_seq; volatile int *next_bucket;
	mm128_v *local;            /* [1<<b] lists of this worker */
	mm128_v **all_local;       /* [n_threads] */
} bld_worker_t;

static void *bld_sketch_worker(void *arg)
{
	bld_worker_t *w = (bld_worker_t*)arg;
	mmo_idx_t *mi = w->mi;
	int mask = (1<<mi->b) - 1;
	for (;;) {
		int i = __sync_fetch_and_add(w->next_seq, 1);
		mm128_v a = {0,0,0};
		size_t j;
		if (i >= w->n_seq) break;
		if (w->lens[i] <= 0) continue;
		if (!(mi->flag & MM_I_NO_SEQ)) {   /* pack S; sequences may share a 32-bit word at their border: atomic OR on the first and last word */
			uint64_t o0 = mi->seq[i].offset, o1 = o0 + (uint64_t)w->lens[i], o;
			for (o = o0; o < o1; ++o) {
				int c = mmo_seq_nt4_table[(uint8_t)w->seqs[i][o - o0]];
				uint32_t v = (uint32_t)c << ((o&7)<<2);
				if ((o>>3) == (o0>>3) || (o>>3) == ((o1-1)>>3)) __sync_fetch_and_or(&mi->S[o>>3], v);
				else mi->S[o>>3] |= v;
			}
		}
		mmo_sketch(w->seqs[i], w->lens[i], mi->w, mi->k, (uint32_t)i, mi->flag&MM_I_HPC, &a);
		for (j = 0; j < a.n; ++j) {
			mm128_v *p = &w->local[a.a[j].x>>8&mask];
			if (p->n == p->m) {
				p->m = p->m? p->m<<1 : 8;
				p->a = (mm128_t*)realloc(p->a, p->m * sizeof(mm128_t));
			}
			p->a[p->n++] = a.a[j];
		}
		free(a.a);
	}
	return 0;
}

static void *bld_post_worker(void *arg)
{
	bld_worker_t *w = (bld_worker_t*)arg;
	mmo_idx_t *mi = w->mi;
	for (;;) {
		int i = __sync_fetch_and_add(w->next_bucket, 1), t;
		size_t tot = 0;
		mmo_bucket_t *b;
		if (i >= 1<<mi->b) break;
		b = &mi->B[i];
		for (t = 0; t < w->n_threads; ++t) tot += w->all_local[t][i].n;
		if (tot == 0) continue;
		b->a.a = (mm128_t*)malloc(tot * sizeof(mm128_t)); b->a.m = tot; b->a.n = 0;
		for (t = 0; t < w->n_threads; ++t) {
			mm128_v *p = &w->all_local[t][i];
			if (p->n) memcpy(b->a.a + b->a.n, p->a, p->n * sizeof(mm128_t));
			b->a.n += p->n;
			free(p->a); p->a = 0; p->n = p->m = 0;
		}
		bucket_post(mi, b);
	}
	return 0;
}

mmo_idx_t *mmo_idx_build_mem_mt(int w, int k, int b, int flag, int n_seq, const char **seqs, const int *lens, const char **names, int n_threads)
{
	mmo_idx_t *mi = idx_init(w, k, b, flag);
	uint64_t sum_len = 0;
	volatile int next_seq = 0, next_bucket = 0;
	int i, t;
	pthread_t *th;
	bld_worker_t *ws;
	mm128_v **locals;
	if (n_threads < 1) n_threads = 1;
	mi->seq = (mmo_idx_seq_t*)calloc(n_seq > 0? n_seq : 1, sizeof(mmo_idx_seq_t));
	for (i = 0; i < n_seq; ++i) {
		mi->seq[i].name = (mi->flag & MM_I_NO_NAME) || names == 0 || names[i] == 0? 0 : strdup(names[i]);
		mi->seq[i].len = lens[i], mi->seq[i].offset = sum_len, mi->seq[i].is_alt = 0;
		sum_len += lens[i];
	}
	mi->n_seq = n_seq;
	if (!(mi->flag & MM_I_NO_SEQ)) mi->S = (uint32_t*)calloc((sum_len + 7) / 8 + 1, 4);
	th = (pthread_t*)calloc(n_threads, sizeof(pthread_t));
	ws = (bld_worker_t*)calloc(n_threads, sizeof(bld_worker_t));
	locals = (mm128_v**)calloc(n_threads, sizeof(mm128_v*));
	for (t = 0; t < n_threads; ++t) {
		locals[t] = (mm128_v*)calloc((size_t)1<<mi->b, sizeof(mm128_v));
		ws[t].mi = mi; ws[t].n_seq = n_seq; ws[t].n_threads = n_threads; ws[t].tid = t; ws[t].seqs = seqs; ws[t].lens = lens;
		ws[t].next_seq = &next_seq; ws[t].next_bucket = &next_bucket; ws[t].local = locals[t]; ws[t].all_local = locals;
	}
	for (t = 0; t < n_threads; ++t) pthread_create(&th[t], 0, bld_sketch_worker, &ws[t]);
	for (t = 0; t < n_threads; ++t) pthread_join(th[t], 0);
	for (t = 0; t < n_threads; ++t) pthread_create(&th[t], 0, bld_post_worker, &ws[t]);
	for (t = 0; t < n_threads; ++t) pthread_join(th[t], 0);
	for (t = 0; t < n_threads; ++t) free(locals[t]);
	free(locals); free(ws); free(th);
	return mi;
}

/* minimal FASTA/FASTQ reader (U:bseq.c / kseq.h semantics: name = up to first whitespace) */
static mmo_idx_t *idx_build_fastx(FILE *fp, const mmo_idxopt_t *io)
{
	mmo_idx_t *mi = idx_init(io->w, io->k, io->bucket_bits, io->flag);
	uint64_t sum_len = 0; size_t m_S = 0;
	char *line = 0, *name = 0, *seq = 0; size_t m_line = 0, l_seq = 0, m_seq = 0;
	ssize_t n; int in_qual = 0, is_fq = 0; size_t l_qual = 0;
	while ((n = getline(&line, &m_line, fp)) >= 0) {
		while (n > 0 && (line[n-1] == '\n' || line[n-1] == '\r')) line[--n] = 0;
		if (in_qual) { /* FASTQ quality: skip l_seq chars */
			l_qual += n;
			if (l_qual >= l_seq) in_qual = 0;
			continue;
		}
		if (line[0] == '>' || (line[0] == '@' && (name == 0 || is_fq))) {
			char *p;
			if (name) idx_add_seq(mi, &sum_len, &m_S, name, seq, (uint32_t)l_seq);
			free(name);
			is_fq = line[0] == '@';
			for (p = line + 1; *p && *p != ' ' && *p != '\t'; ++p);
			*p = 0;
			name = strdup(line + 1);
			l_seq = 0;
		} else if (line[0] == '+' && is_fq) {
			in_qual = 1, l_qual = 0;
			if (l_seq == 0) in_qual = 0;
		} else if (name) {
			ssize_t i;
			if (l_seq + n + 1 > m_seq) { m_seq = (l_seq + n + 1) * 2; seq = (char*)realloc(seq, m_seq); }
			for (i = 0; i < n; ++i)
				if (line[i] > ' ') seq[l_seq++] = line[i];
		}
	}
	if (name) idx_add_seq(mi, &sum_len, &m_S, name, seq, (uint32_t)l_seq);
	free(name); free(seq); free(line);
	idx_finish(mi);
	return mi;
}

/* U:index.c::mm_idx_reader_open + mm_idx_reader_read (single part: batch_size = i64::MAX, R:src/lib.rs:340) */
mmo_idx_t *mmo_idx_load(const char *fn, const mmo_idxopt_t *io)
{
	FILE *fp;
	char magic[4];
	mmo_idx_t *mi;
	size_t n;
	fp = fopen(fn, "rb");
	if (fp == 0) return 0;
	n = fread(magic, 1, 4, fp);
	rewind(fp);
	if (n == 4 && strncmp(magic, MM_IDX_MAGIC, 4) == 0) mi = idx_load_mmi(fp);
	else if (n == 0) mi = 0;
	else mi = idx_build_fastx(fp, io);
	fclose(fp);
	if (mi && mi->n_seq == 0) { mmo_idx_destroy(mi); mi = 0; }
	return mi;
}
