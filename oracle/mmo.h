/* mmo.h -- ORACLE (test infrastructure only; never linked into the product).
 *
 * A plain-C, single-threaded CPU restatement of the per-read mapping path that
 * mappy-rs reaches through `minimap2::Aligner::map` (R:src/lib.rs:482-488 and
 * R:src/lib.rs:587-593), i.e. minimap2 v2.26's mm_map() as pinned by
 * `minimap2-sys = "0.1.15+minimap2.2.26"` (R:Cargo.toml:24,29).  The minimap2 C
 * sources are an un-vendored dependency and are absent from /root/reference, so
 * every function here restates the published algorithm ("U:file::function" =
 * upstream minimap2 2.26 unit it follows) and cites the reference call site.
 *
 * Parity pinning: sketch + index encoding + .mmi format are pinned bit-exactly
 * by the reference's own fixture pair resources/test/test.fa <-> test.mmi, and
 * the one alignment the reference asserts (map_one, R:src/lib.rs:1094-1106).
 * Everything else is "parity unpinned" by the reference.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything in this directory.
 */
#ifndef MMO_H
#define MMO_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t x, y; } mm128_t;
typedef struct { size_t n, m; mm128_t *a; } mm128_v;

/* index flags (U:minimap.h) */
#define MM_I_HPC     0x1
#define MM_I_NO_SEQ  0x2
#define MM_I_NO_NAME 0x4

/* mapping flags (U:minimap.h) */
#define MM_F_NO_DIAG       0x001LL
#define MM_F_NO_DUAL       0x002LL
#define MM_F_CIGAR         0x004LL
#define MM_F_OUT_SAM       0x008LL
#define MM_F_NO_QUAL       0x010LL
#define MM_F_OUT_CG        0x020LL
#define MM_F_OUT_CS        0x040LL
#define MM_F_SPLICE        0x080LL
#define MM_F_SPLICE_FOR    0x100LL
#define MM_F_SPLICE_REV    0x200LL
#define MM_F_NO_LJOIN      0x400LL
#define MM_F_OUT_CS_LONG   0x800LL
#define MM_F_SR            0x1000LL
#define MM_F_FRAG_MODE     0x2000LL
#define MM_F_NO_PRINT_2ND  0x4000LL
#define MM_F_2_IO_THREADS  0x8000LL
#define MM_F_LONG_CIGAR    0x10000LL
#define MM_F_INDEPEND_SEG  0x20000LL
#define MM_F_SPLICE_FLANK  0x40000LL
#define MM_F_SOFTCLIP      0x80000LL
#define MM_F_FOR_ONLY      0x100000LL
#define MM_F_REV_ONLY      0x200000LL
#define MM_F_HEAP_SORT     0x400000LL
#define MM_F_ALL_CHAINS    0x800000LL
#define MM_F_OUT_MD        0x1000000LL
#define MM_F_COPY_COMMENT  0x2000000LL
#define MM_F_EQX           0x4000000LL
#define MM_F_PAF_NO_HIT    0x8000000LL
#define MM_F_NO_END_FLT    0x10000000LL
#define MM_F_HARD_MLEVEL   0x20000000LL
#define MM_F_SAM_HIT_ONLY  0x40000000LL
#define MM_F_RMQ           0x80000000LL
#define MM_F_QSTRAND       0x100000000LL
#define MM_F_NO_INV        0x200000000LL
#define MM_F_NO_HASH_NAME  0x400000000LL

#define MM_SEED_LONG_JOIN (1ULL<<40)
#define MM_SEED_IGNORE    (1ULL<<41)
#define MM_SEED_TANDEM    (1ULL<<42)
#define MM_SEED_SELF      (1ULL<<43)
#define MM_SEED_SEG_SHIFT 48
#define MM_SEED_SEG_MASK  (0xffULL<<(MM_SEED_SEG_SHIFT))

#define MM_PARENT_UNSET   (-1)
#define MM_PARENT_TMP_PRI (-2)

#define MM_CIGAR_MATCH 0
#define MM_CIGAR_INS   1
#define MM_CIGAR_DEL   2
#define MM_CIGAR_N_SKIP 3

/* U:minimap.h::mm_idxopt_t */
typedef struct {
	short k, w, flag, bucket_bits;
	int64_t mini_batch_size;
	uint64_t batch_size;
} mmo_idxopt_t;

/* U:minimap.h::mm_mapopt_t (fields the long-read path reads) */
typedef struct {
	int64_t flag;
	int seed;
	int sdust_thres;
	int max_qlen;
	int bw, bw_long;
	int max_gap, max_gap_ref;
	int max_frag_len;
	int max_chain_skip, max_chain_iter;
	int min_cnt;
	int min_chain_score;
	float chain_gap_scale;
	float chain_skip_scale;
	int rmq_size_cap, rmq_inner_dist;
	int rmq_rescue_size;
	float rmq_rescue_ratio;
	float mask_level;
	int mask_len;
	float pri_ratio;
	int best_n;
	float alt_drop;
	int a, b, q, e, q2, e2;
	int sc_ambi;
	int noncan;
	int junc_bonus;
	int zdrop, zdrop_inv;
	int end_bonus;
	int min_dp_max;
	int min_ksw_len;
	int anchor_ext_len, anchor_ext_shift;
	float max_clip_ratio;
	int rank_min_len;
	float rank_frac;
	int pe_ori, pe_bonus;
	float mid_occ_frac;
	float q_occ_frac;
	int32_t min_mid_occ, max_mid_occ;
	int32_t mid_occ;
	int32_t max_occ, max_max_occ, occ_dist;
	int64_t mini_batch_size;
	int64_t max_sw_mat;
	int64_t cap_kalloc;
} mmo_mapopt_t;

typedef struct {
	char *name;
	uint64_t offset;
	uint32_t len;
	uint32_t is_alt;
} mmo_idx_seq_t;

/* Own open-addressing table per bucket: only mm_idx_get()'s (key -> ptr,count)
 * contract is normative (U:index.c::mm_idx_get); the khash internals are not. */
typedef struct {
	int32_t n;          /* size of p[] */
	uint64_t *p;        /* positions of minimizers occurring >1 times */
	uint32_t n_keys, cap; /* hash capacity (power of two) */
	uint64_t *keys;     /* key = minier>>b<<1 | is_singleton ; UINT64_MAX = empty */
	uint64_t *vals;
	mm128_v a;          /* build-time (minimizer, position) list */
} mmo_bucket_t;

typedef struct {
	int32_t b, w, k, flag;
	uint32_t n_seq;
	int32_t n_alt;
	mmo_idx_seq_t *seq;
	uint32_t *S;        /* 4-bit packed bases */
	mmo_bucket_t *B;
} mmo_idx_t;

/* U:minimap.h::mm_extra_t */
typedef struct {
	uint32_t capacity;
	int32_t dp_score, dp_max, dp_max2;
	uint32_t n_ambi:30, trans_strand:2;
	uint32_t n_cigar;
	uint32_t cigar[];
} mmo_extra_t;

/* U:minimap.h::mm_reg1_t */
typedef struct {
	int32_t id;
	int32_t cnt;
	int32_t rid;
	int32_t score;
	int32_t qs, qe, rs, re;
	int32_t parent, subsc;
	int32_t as;
	int32_t mlen, blen;
	int32_t n_sub;
	int32_t score0;
	uint32_t mapq:8, split:2, rev:1, inv:1, sam_pri:1, proper_frag:1, pe_thru:1, seg_split:1, seg_id:8, split_inv:1, is_alt:1, strand_retained:1, dummy:5;
	uint32_t hash;
	float div;
	mmo_extra_t *p;
} mmo_reg1_t;

/* U:ksw2.h::ksw_extz_t */
#define KSW_NEG_INF -0x40000000
#define KSW_EZ_SCORE_ONLY  0x01
#define KSW_EZ_RIGHT       0x02
#define KSW_EZ_GENERIC_SC  0x04
#define KSW_EZ_APPROX_MAX  0x08
#define KSW_EZ_APPROX_DROP 0x10
#define KSW_EZ_EXTZ_ONLY   0x40
#define KSW_EZ_REV_CIGAR   0x80

typedef struct {
	uint32_t max:31, zdropped:1;
	int max_q, max_t;
	int mqe, mqe_t;
	int mte, mte_q;
	int score;
	int m_cigar, n_cigar;
	int reach_end;
	uint32_t *cigar;
} mmo_extz_t;

/* ---- sort.c (U:ksort.h) ---- */
void mmo_radix_sort_128x(mm128_t *beg, mm128_t *end);
void mmo_radix_sort_64(uint64_t *beg, uint64_t *end);
uint32_t mmo_ksmall_u32(size_t n, uint32_t *arr, size_t kk);

/* ---- sketch.c (U:sketch.c) ---- */
extern unsigned char mmo_seq_nt4_table[256];
void mmo_sketch(const char *str, int len, int w, int k, uint32_t rid, int is_hpc, mm128_v *p);

/* ---- index.c (U:index.c) ---- */
mmo_idx_t *mmo_idx_load(const char *fn, const mmo_idxopt_t *io);   /* .mmi or FASTA, decided by magic */
mmo_idx_t *mmo_idx_build_mem(int w, int k, int b, int flag, int n_seq, const char **seqs, const int *lens, const char **names);
mmo_idx_t *mmo_idx_build_mem_mt(int w, int k, int b, int flag, int n_seq, const char **seqs, const int *lens, const char **names, int n_threads);
void mmo_idx_destroy(mmo_idx_t *mi);
const uint64_t *mmo_idx_get(const mmo_idx_t *mi, uint64_t minier, int *n);
int mmo_idx_getseq(const mmo_idx_t *mi, uint32_t rid, uint32_t st, uint32_t en, uint8_t *seq);
int mmo_idx_name2id(const mmo_idx_t *mi, const char *name);
int32_t mmo_idx_cal_max_occ(const mmo_idx_t *mi, float f);
int64_t mmo_idx_n_minimizers(const mmo_idx_t *mi, int64_t *n_distinct);
int mmo_idx_dump(const mmo_idx_t *mi, const char *fn); /* writes MMI\2 */

/* ---- options.c (U:options.c) ---- */
void mmo_idxopt_init(mmo_idxopt_t *opt);
void mmo_mapopt_init(mmo_mapopt_t *opt);
int mmo_set_opt(const char *preset, mmo_idxopt_t *io, mmo_mapopt_t *mo);
void mmo_mapopt_update(mmo_mapopt_t *opt, const mmo_idx_t *mi);

/* ---- seed.c / map.c ---- */
typedef struct {
	uint32_t n;
	uint32_t q_pos;
	uint32_t q_span:31, flt:1;
	uint32_t seg_id:31, is_tandem:1;
	const uint64_t *cr;
} mmo_seed_t;
void mmo_seed_mz_flt(mm128_v *mv, int32_t q_occ_max, float q_occ_frac);
mm128_t *mmo_collect_seed_hits(const mmo_mapopt_t *opt, int max_occ, const mmo_idx_t *mi, const mm128_v *mv, int qlen,
                               int64_t *n_a, int *rep_len, int *n_mini_pos, uint64_t **mini_pos, int sorted);

/* ---- lchain.c ---- */
mm128_t *mmo_lchain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc,
                       float chn_pen_gap, float chn_pen_skip, int is_cdna, int n_seg, int64_t n, mm128_t *a,
                       int *n_u_, uint64_t **_u);
/* DP fill only (for kernel parity): writes f[n], p[n] (int64), v[n], t[n] */
void mmo_lchain_dp_fill(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, float chn_pen_gap, float chn_pen_skip,
                        int64_t n, const mm128_t *a, int32_t *f, int64_t *p, int32_t *v, int32_t *t);
mm128_t *mmo_lchain_rmq(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, int min_cnt, int min_sc,
                        float chn_pen_gap, float chn_pen_skip, int64_t n, mm128_t *a, int *n_u_, uint64_t **_u);

/* ---- hit.c / esterr.c ---- */
mmo_reg1_t *mmo_gen_regs(uint32_t hash, int qlen, int n_u, uint64_t *u, mm128_t *a, int is_qstrand);
void mmo_split_reg(mmo_reg1_t *r, mmo_reg1_t *r2, int n, int qlen, mm128_t *a, int is_qstrand);
void mmo_set_parent(float mask_level, int mask_len, int n, mmo_reg1_t *r, int sub_diff, int hard_mask_level, float alt_diff_frac);
void mmo_select_sub(float pri_ratio, int min_diff, int best_n, int check_strand, int min_strand_sc, int *n_, mmo_reg1_t *r);
void mmo_hit_sort(int *n_regs, mmo_reg1_t *r, float alt_diff_frac);
int mmo_set_sam_pri(int n, mmo_reg1_t *r);
void mmo_sync_regs(int n_regs, mmo_reg1_t *regs);
void mmo_filter_regs(const mmo_mapopt_t *opt, int qlen, int *n_regs, mmo_reg1_t *regs);
int mmo_filter_strand_retained(int n_regs, mmo_reg1_t *r);
int mmo_squeeze_a(int n_regs, mmo_reg1_t *regs, mm128_t *a);
void mmo_set_mapq(int n_regs, mmo_reg1_t *regs, int min_chain_sc, int match_sc, int rep_len, int is_sr);
void mmo_est_err(const mmo_idx_t *mi, int qlen, int n_regs, mmo_reg1_t *regs, const mm128_t *a, int32_t n, const uint64_t *mini_pos);

/* ---- ksw2.c ---- */
void mmo_ksw_reset_extz(mmo_extz_t *ez);
void mmo_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, mmo_extz_t *ez);
int mmo_ksw_ll(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat, int gapo, int gape, int *qe, int *te);
void mmo_ksw_gen_simple_mat(int m, int8_t *mat, int8_t a, int8_t b, int8_t sc_ambi);

/* ---- align.c ---- */
mmo_reg1_t *mmo_align_skeleton(const mmo_mapopt_t *opt, const mmo_idx_t *mi, int qlen, const char *qstr, int *n_regs_, mmo_reg1_t *regs, mm128_t *a);

int mmo_test_zdrop(const mmo_mapopt_t *opt, const uint8_t *qseq, const uint8_t *tseq, uint32_t n_cigar, uint32_t *cigar, const int8_t *mat);   /* test hook */

/* ---- format.c ---- */
char *mmo_gen_cs(const mmo_idx_t *mi, const mmo_reg1_t *r, const char *seq, int no_iden);  /* malloc'd, NUL-terminated */
char *mmo_md_core(const uint32_t *cigar, int n_cigar, const uint8_t *qseq, const uint8_t *tseq, int *q_len, int *t_len);
char *mmo_cs_core(const uint32_t *cigar, int n_cigar, const uint8_t *qseq, const uint8_t *tseq, int no_iden, int *q_len, int *t_len);
void mmo_extra_walk(const uint32_t *cigar, int n_cigar, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int8_t q, int8_t e, int log_gap,
                    int32_t *mlen, int32_t *blen, int32_t *n_ambi, int32_t *dp_max, int32_t *q_len, int32_t *t_len);
char *mmo_gen_MD(const mmo_idx_t *mi, const mmo_reg1_t *r, const char *seq);

/* ---- map.c ---- */
mmo_reg1_t *mmo_map(const mmo_idx_t *mi, int qlen, const char *seq, int *n_regs, const mmo_mapopt_t *opt, const char *qname);

/* per-stage counters of the last mmo_map call on this thread (for bench/roofline accounting) */
typedef struct {
	int64_t n_mz, n_hit, n_a, n_a_multi, chain_pairs, dp_cells, n_dp_calls;
	int32_t rep_len, n_chain0, n_chain1, did_rmq;
} mmo_stats_t;
extern __thread mmo_stats_t mmo_stats;

/* flat result record for bindings; mirrors mappy_rs::Mapping (R:src/lib.rs:109-154) */
typedef struct {
	int32_t query_start, query_end;
	int32_t strand;           /* +1 / -1 */
	int32_t rid;
	int32_t target_len, target_start, target_end;
	int32_t match_len, block_len;
	uint32_t mapq;
	int32_t is_primary;
	int32_t NM;
	int32_t n_cigar;
	int64_t cigar_off;        /* offset into cigar arena (u32 len<<4|op) */
	int64_t cs_off, cs_len;   /* offset into string arena; len<0 => None */
	int64_t md_off, md_len;
	int32_t score0, dp_max, dp_max2, dp_score, cnt, n_sub, subsc, sam_pri;
} mmo_hit_t;

typedef struct {
	int n_hits;
	mmo_hit_t *hits;
	size_t n_cigar, m_cigar; uint32_t *cigar;
	size_t n_str, m_str; char *str;
} mmo_result_t;

/* Equivalent of minimap2::Aligner::map (L2 crate) as called at R:src/lib.rs:482 / :587.
 * returns 0 ok, -1 "No index", -2 "Sequence is empty". */
int mmo_map_flat(const mmo_idx_t *mi, const mmo_mapopt_t *opt, const char *seq, int len, int with_cs, int with_md, mmo_result_t *res);
void mmo_result_free(mmo_result_t *res);

#ifdef __cplusplus
}
#endif
#endif
