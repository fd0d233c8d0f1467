/* ORACLE (test infrastructure).  Restates U:format.c::mm_gen_cs (write_cs_core,
 * no_iden=1 short form) and mm_gen_MD (write_MD_core) of minimap2 2.26, as
 * called by the L2 crate when `cs`/`MD` are requested (R:src/lib.rs:482-488;
 * cs=true is hard-wired in the batch worker, R:src/lib.rs:589).
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <assert.h>
#include "mmo.h"

typedef struct { size_t l, m; char *s; } kstr_t;

static void ks_put(kstr_t *s, const char *p, size_t l)
{
	if (s->l + l + 1 > s->m) {
		s->m = (s->l + l + 1) * 2;
		s->s = (char*)realloc(s->s, s->m);
	}
	memcpy(s->s + s->l, p, l);
	s->l += l;
	s->s[s->l] = 0;
}
static void ks_putd(kstr_t *s, int d) { char buf[16]; int l = snprintf(buf, 16, "%d", d); ks_put(s, buf, l); }
static void ks_putc(kstr_t *s, char c) { ks_put(s, &c, 1); }

static void get_seqs(const mmo_idx_t *mi, const mmo_reg1_t *r, const char *seq, uint8_t **qseq_, uint8_t **tseq_)
{
	int i;
	uint8_t *qseq = (uint8_t*)malloc(r->qe - r->qs + 1), *tseq = (uint8_t*)malloc(r->re - r->rs + 1);
	mmo_idx_getseq(mi, r->rid, r->rs, r->re, tseq);
	if (!r->rev) {
		for (i = r->qs; i < r->qe; ++i)
			qseq[i - r->qs] = mmo_seq_nt4_table[(uint8_t)seq[i]];
	} else {
		for (i = r->qs; i < r->qe; ++i) {
			uint8_t c = mmo_seq_nt4_table[(uint8_t)seq[i]];
			qseq[r->qe - i - 1] = c >= 4? 4 : 3 - c;
		}
	}
	*qseq_ = qseq, *tseq_ = tseq;
}

/* U:format.c::write_cs_core on given code strings (0..4) and CIGAR; also the stage entry the parity test of the device walk (k_extra) calls */
char *mmo_cs_core(const uint32_t *cigar, int n_cigar, const uint8_t *qseq, const uint8_t *tseq, int no_iden, int *q_len, int *t_len)
{
	int i, q_off, t_off;
	kstr_t s = {0,0,0};
	ks_put(&s, "", 0);
	for (i = q_off = t_off = 0; i < n_cigar; ++i) {
		int j, op = cigar[i]&0xf, len = cigar[i]>>4;
		if (op == MM_CIGAR_MATCH || op == 7 || op == 8) {
			int l_tmp = 0;
			for (j = 0; j < len; ++j) {
				if (qseq[q_off + j] != tseq[t_off + j]) {
					if (l_tmp > 0) {
						if (!no_iden) {
							int k;
							ks_putc(&s, '=');
							for (k = l_tmp; k > 0; --k) ks_putc(&s, "ACGTN"[qseq[q_off + j - k]]);
						} else ks_putc(&s, ':'), ks_putd(&s, l_tmp);
						l_tmp = 0;
					}
					ks_putc(&s, '*'); ks_putc(&s, "acgtn"[tseq[t_off + j]]); ks_putc(&s, "acgtn"[qseq[q_off + j]]);
				} else ++l_tmp;
			}
			if (l_tmp > 0) {
				if (!no_iden) {
					int k;
					ks_putc(&s, '=');
					for (k = l_tmp; k > 0; --k) ks_putc(&s, "ACGTN"[qseq[q_off + len - k]]);
				} else ks_putc(&s, ':'), ks_putd(&s, l_tmp);
			}
			q_off += len, t_off += len;
		} else if (op == MM_CIGAR_INS) {
			ks_putc(&s, '+');
			for (j = 0; j < len; ++j) ks_putc(&s, "acgtn"[qseq[q_off + j]]);
			q_off += len;
		} else if (op == MM_CIGAR_DEL) {
			ks_putc(&s, '-');
			for (j = 0; j < len; ++j) ks_putc(&s, "acgtn"[tseq[t_off + j]]);
			t_off += len;
		} else { /* intron: not produced on the long-read genomic path */
			t_off += len;
		}
	}
	if (q_len) *q_len = q_off;
	if (t_len) *t_len = t_off;
	return s.s;
}

char *mmo_gen_cs(const mmo_idx_t *mi, const mmo_reg1_t *r, const char *seq, int no_iden)
{
	int q_off, t_off;
	uint8_t *qseq, *tseq;
	char *out;
	if (r->p == 0) { kstr_t s = {0,0,0}; ks_put(&s, "", 0); return s.s; }
	get_seqs(mi, r, seq, &qseq, &tseq);
	out = mmo_cs_core(r->p->cigar, (int)r->p->n_cigar, qseq, tseq, no_iden, &q_off, &t_off);
	assert(t_off == r->re - r->rs && q_off == r->qe - r->qs);
	free(qseq); free(tseq);
	return out;
}

/* U:format.c::write_MD_core on given code strings (0..4) and CIGAR; also the stage entry the parity test of the device walk (k_extra) calls */
char *mmo_md_core(const uint32_t *cigar, int n_cigar, const uint8_t *qseq, const uint8_t *tseq, int *q_len, int *t_len)
{
	int i, q_off, t_off, l_MD = 0;
	kstr_t s = {0,0,0};
	ks_put(&s, "", 0);
	for (i = q_off = t_off = 0; i < n_cigar; ++i) {
		int j, op = cigar[i]&0xf, len = cigar[i]>>4;
		if (op == MM_CIGAR_MATCH || op == 7 || op == 8) {
			for (j = 0; j < len; ++j) {
				if (qseq[q_off + j] != tseq[t_off + j]) {
					ks_putd(&s, l_MD); ks_putc(&s, "ACGTN"[tseq[t_off + j]]);
					l_MD = 0;
				} else ++l_MD;
			}
			q_off += len, t_off += len;
		} else if (op == MM_CIGAR_INS) {
			q_off += len;
		} else if (op == MM_CIGAR_DEL) {
			ks_putd(&s, l_MD); ks_putc(&s, '^');
			for (j = 0; j < len; ++j) ks_putc(&s, "ACGTN"[tseq[t_off + j]]);
			l_MD = 0;
			t_off += len;
		} else if (op == MM_CIGAR_N_SKIP) {
			t_off += len;
		}
	}
	if (l_MD > 0) ks_putd(&s, l_MD);
	if (q_len) *q_len = q_off;
	if (t_len) *t_len = t_off;
	return s.s;
}

char *mmo_gen_MD(const mmo_idx_t *mi, const mmo_reg1_t *r, const char *seq)
{
	int q_off, t_off;
	uint8_t *qseq, *tseq;
	char *out;
	if (r->p == 0) { kstr_t s = {0,0,0}; ks_put(&s, "", 0); return s.s; }
	get_seqs(mi, r, seq, &qseq, &tseq);
	out = mmo_md_core(r->p->cigar, (int)r->p->n_cigar, qseq, tseq, &q_off, &t_off);
	assert(t_off == r->re - r->rs && q_off == r->qe - r->qs);
	free(qseq); free(tseq);
	return out;
}
