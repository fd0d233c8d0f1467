/* mm355.h -- C-ABI of the MI355X-native mapping path (libmm355.so).
 *
 * Drop-in boundary for the ONE hot path of Adoni5/mappy-rs: the per-read minimap2
 * mapping call.  In the reference that call is `minimap2::Aligner::map` ->
 * `mm_map` of minimap2-sys (reference call sites /root/reference/src/lib.rs:482-488
 * for Aligner.map and :587-593 for the map_batch worker), surrounded by the
 * index/option FFI at lib.rs:333-416 and the sequence accessors at :716/:747.
 * Each entry point below names the reference FFI symbol it replaces.
 *
 * Plain pointers and sizes only; no torch / HIP types cross this boundary.
 * All functions return 0 on success or a negative MM355_E* code; mm355_strerror
 * gives the message.  The library refuses to run (MM355_ENODEV) when no gfx950
 * device is visible: there is no CPU fallback.
 */
#ifndef MM355_H
#define MM355_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM355_OK        0
#define MM355_ENODEV   (-1)   /* no HIP device / kernel image not loadable */
#define MM355_EINVAL   (-2)
#define MM355_ENOMEM   (-3)
#define MM355_EIO      (-4)   /* cannot open / parse index or FASTA */
#define MM355_ENOIDX   (-5)   /* "No index" (L2 crate error string) */
#define MM355_EEMPTY   (-6)   /* "Sequence is empty" (L2 crate error string) */
#define MM355_EUNSUP   (-7)   /* option outside the long-read hot path (sr/splice presets, query-strand / heap-sort flags) */
#define MM355_EHIP     (-8)   /* a HIP runtime call failed */

typedef struct mm355_index mm355_index_t;     /* replaces mm_idx_t* (lib.rs:400-410) */
typedef struct mm355_ctx   mm355_ctx_t;       /* replaces mm_tbuf_t: one per host thread / GPU */

/* replaces mm_idxopt_t (lib.rs:332) */
typedef struct {
	int16_t k, w, flag, bucket_bits;
	int64_t mini_batch_size;
	uint64_t batch_size;
} mm355_idxopt_t;

/* replaces mm_mapopt_t (lib.rs:331); same field meaning as minimap2 2.26 */
typedef struct {
	int64_t flag;
	int32_t seed, sdust_thres, max_qlen;
	int32_t bw, bw_long, max_gap, max_gap_ref, max_frag_len;
	int32_t max_chain_skip, max_chain_iter, min_cnt, min_chain_score;
	float chain_gap_scale, chain_skip_scale;
	int32_t rmq_size_cap, rmq_inner_dist, rmq_rescue_size;
	float rmq_rescue_ratio, mask_level;
	int32_t mask_len;
	float pri_ratio;
	int32_t best_n;
	float alt_drop;
	int32_t a, b, q, e, q2, e2, sc_ambi;
	int32_t zdrop, zdrop_inv, end_bonus, min_dp_max, min_ksw_len;
	float max_clip_ratio;
	float mid_occ_frac, q_occ_frac;
	int32_t min_mid_occ, max_mid_occ, mid_occ, max_occ, max_max_occ, occ_dist;
	int64_t max_sw_mat;
} mm355_mapopt_t;

/* one alignment; mirrors mappy_rs::Mapping (lib.rs:109-154) filled from mm_reg1_t */
typedef struct {
	int32_t query_start, query_end;
	int32_t strand;                 /* +1 forward, -1 reverse */
	int32_t rid;                    /* index into mm355_index_info names */
	int32_t target_len, target_start, target_end;
	int32_t match_len, block_len;
	uint32_t mapq;
	int32_t is_primary;
	int32_t NM;
	int32_t n_cigar;
	int64_t cigar_off;              /* into mm355_hits_t::cigar, u32 = len<<4|op */
	int64_t cs_off, cs_len;         /* into mm355_hits_t::str; cs_len < 0 => None */
	int64_t md_off, md_len;
	int32_t score0, dp_max, dp_max2, dp_score, cnt, n_sub, subsc, reserved;
} mm355_hit_t;

/* result of one batch; owned by the library until mm355_free_hits */
typedef struct {
	int64_t n_reads;
	int64_t *hit_off;               /* n_reads+1 offsets into hits[] */
	int32_t *status;                /* per read: 0 ok, MM355_EEMPTY for an empty sequence */
	mm355_hit_t *hits;
	uint32_t *cigar;
	char *str;
	int64_t n_hits, n_cigar, n_str;
} mm355_hits_t;

/* --- options: replaces mm_set_opt (lib.rs:333,336) and mm_mapopt_update (lib.rs:414) --- */
int mm355_set_opt(const char *preset, mm355_idxopt_t *io, mm355_mapopt_t *mo);
int mm355_mapopt_update(mm355_mapopt_t *mo, const mm355_index_t *idx);

/* --- index: replaces mm_idx_reader_open/read/close + mm_idx_index_name (lib.rs:397-416) --- */
int mm355_index_load(const char *path, const mm355_idxopt_t *io, int n_threads, mm355_index_t **out);
int mm355_index_build(const mm355_idxopt_t *io, int n_seq, const char *const *seqs, const int64_t *lens,
                      const char *const *names, int n_threads, mm355_index_t **out);
/* same index, built on GPU `device` (sketch + radix sort + table fill in HBM; replaces the FASTA branch of
 * mm_idx_reader_read for large references).  The table stays resident on that device; other devices get peer copies (mm355_upload). */
int mm355_index_build_device(const mm355_idxopt_t *io, int n_seq, const uint8_t *const *seqs, const int64_t *lens,
                             const char *const *names, int device, mm355_index_t **out);
void mm355_index_free(mm355_index_t *idx);
/* header fields read at lib.rs:655-670 (k, w, n_seq) */
int mm355_index_info(const mm355_index_t *idx, int32_t *k, int32_t *w, int32_t *b, int32_t *flag, uint32_t *n_seq);
const char *mm355_index_seq_name(const mm355_index_t *idx, uint32_t rid);   /* lib.rs:447-455 */
int64_t mm355_index_seq_len(const mm355_index_t *idx, uint32_t rid);
int mm355_index_name2id(const mm355_index_t *idx, const char *name);        /* mm_idx_name2id, lib.rs:716 */
int mm355_index_getseq(const mm355_index_t *idx, uint32_t rid, uint32_t st, uint32_t en, uint8_t *seq); /* mm_idx_getseq, lib.rs:747 */
/* host-side diagnostic equivalent of mm_idx_get(): occurrences of one minimizer (returns the count) */
int mm355_index_get(const mm355_index_t *idx, uint64_t minier, uint64_t *vals, int cap);
int mm355_index_stat(const mm355_index_t *idx, int64_t *n_minimizers, int64_t *n_distinct, int64_t *table_bytes, int64_t *pos_bytes);

/* --- multi-GPU: the reference shares ONE read-only mm_idx_t between its N worker threads (lib.rs:541-546, `self.aligner.clone()`
 * per thread); the GPU analogue is one replica of the index in the HBM of every device, shared by all contexts of that device.
 * mm355_upload replicates the index to the listed devices (H2D from the host image; a device-built index is copied device-to-device);
 * idempotent.  mm355_ctx_create replicates lazily when its device has no replica yet.  No collective is involved. --- */
int mm355_upload(mm355_index_t *idx, const int *device_ids, int n);

/* --- device context (one per host thread / GPU): replaces mm_tbuf_init/destroy --- */
int mm355_ctx_create(const mm355_index_t *idx, int device_id, mm355_ctx_t **out);   /* uses (or creates) the replica of device_id */
void mm355_ctx_destroy(mm355_ctx_t *ctx);

/* --- the hot path: replaces mm_map (+ mm_gen_cs / mm_gen_MD) for a whole batch of reads.
 * seqs[i] need not be NUL-terminated.  flags: bit0 = cs (short form), bit1 = MD. --- */
#define MM355_OUT_CS 1
#define MM355_OUT_MD 2
int mm355_map_batch(mm355_ctx_t *ctx, const mm355_mapopt_t *mo, int64_t n_reads, const char *const *seqs,
                    const int32_t *lens, int flags, mm355_hits_t **out);
/* the same call split in two, for callers that keep a batch resident in HBM (bench.py times mm355_map_resident):
 * mm355_map_batch == mm355_batch_upload + mm355_map_resident */
int mm355_batch_upload(mm355_ctx_t *ctx, int64_t n_reads, const char *const *seqs, const int32_t *lens);
/* several resident batches per context: make batch `slot` (0..63) the current one; upload / map_resident act on the current batch */
int mm355_batch_select(mm355_ctx_t *ctx, int slot);
int mm355_map_resident(mm355_ctx_t *ctx, const mm355_mapopt_t *mo, int flags, mm355_hits_t **out);
void mm355_free_hits(mm355_hits_t *hits);

/* --- per-stage entry points (same kernels as mm355_map_batch; used by the parity tests and
 * by bench.py to time one kernel with HIP events).  Outputs are caller-allocated host buffers. --- */
typedef struct {
	int64_t n_reads, n_bases;
	int64_t n_mz, n_hit, n_a, n_a_multi;       /* SURVEY 8(d) counters of the seed stage */
	int64_t chain_pairs, dp_cells, n_dp_jobs;
	double ms_sketch, ms_seed, ms_sort, ms_chain, ms_backtrack, ms_dp, ms_host, ms_total;
	double ms_seed_lookup, ms_seed_expand;
	int64_t n_launch_seed;
	int64_t n_launch_dp;                         /* extension launch groups (one per round and HBM-budget chunk) */
	/* per extension kernel of a launch group, timed with HIP events on the stream it is launched on; group = 2 * size class + exact,
	 * size classes: targets <= 128, 256, 512, 1024 (k_ksw_reg<1|2|4|8, exact>), <= 4096, <= 12288, larger (k_ksw_extd2<512>; one launch
	 * for groups 8-9, timed as 8, and one for 10-13, timed as 10); 14 / 15 / 16 = k_ksw_row<2|4|8> (full-band approximate gap fills), 17 k_ksw_rowl,
	 * 18 k_ksw_regw8, 19 / 20 / 21 = k_ksw_band<1|2|4> (the same fills on a band of 128 / 256 / 512 diagonals), 22 = k_ksw_band2 (64 diagonals, two problems per wave),
	 * 23 = second run of band problems */
	double ms_dp_group[24];
	int64_t dp_cells_group[24], n_launch_group[24];
	int64_t n_ext_rounds;                        /* extension rounds of the last call (the reference has no bound on them) */
	int64_t n_sort_fast_reads, n_sort_tie_reads; /* anchor sort: reads of anchor-rich batches (cull + per-read LDS sort) / of those, reads whose surviving
	                                                anchors contain equal keys and went through the literal radix_sort_128x emulation */
	/* mg_lchain_rmq on the device (row a9): kernel time, reads re-chained there, reads handed to the literal host implementation because the
	 * device could not prove its range-minimum answer unique (or ran out of LDS capacity), window elements looked at */
	double ms_rmq;
	int64_t n_rmq_reads, n_rmq_host, rmq_scanned;
	double host_cpu_ms;                          /* CPU time (not wall) the host tail of the last call spent, summed over the pool threads */
	int64_t n_a_kept;                            /* anchors left after the cull of the anchor-rich sort path (x-components too small to chain dropped);
	                                                0 when the batch took the literal path for every read */
	/* every kernel outside the extension rounds, timed alone with a HIP-event pair on the stream it is launched on (the extension kernels:
	 * ms_dp_group).  Slots: 0 sketch, 1 mz_flt, 2 seed lookup (tile list + probes), 3 seed select, 4 seed expand, 5 anchor cull, 6 anchor
	 * sort (LDS), 7 literal radix_sort_128x emulation (reads with equal keys), 8 chain segments, 9 chain (long segments, a wave each),
	 * 10 chain (short segments, a lane each), 11 chain backtrack, 12 / 13 / 14 mg_lchain_rmq sort / recurrence / backtrack, 15 extension
	 * gather, 16 extension backtrack (CIGAR), 17 mm_update_extra + cs walk, 18 read codes, 19 chain / anchor pack; the literal emulation by kernel:
	 * 7 k_sort_level_mw<1024> (all its levels), 20 k_sort_level_mw<256>, 21 k_sort_tasks, 22 its plain sort / copy / tcnt of the tie reads */
	double ms_kernel[24];
	int64_t chain_pairs_big;                     /* k_chain_big's share of chain_pairs */
	int64_t n_a_literal;                         /* anchors (all of them, culled ones included) of the reads that were sorted literally */
	int64_t n_v_rmq;                             /* anchors mg_lchain_rmq chained on the device */
	int64_t n_dp_band, n_dp_band_redo;           /* gap fills run on a diagonal band with a sufficiency proof / of those, run again on the full matrix */
	int64_t n_rounds_split;                      /* extension rounds whose direction matrices did not fit the HBM budget and were cut into several launches */
} mm355_stats_t;

/* sketch: minimizers of each read (mm_sketch). mz_off[n_reads+1] host array is filled; mz = (x,y) pairs */
int mm355_stage_sketch(mm355_ctx_t *ctx, int64_t n_reads, const char *const *seqs, const int32_t *lens,
                       int64_t *mz_off, uint64_t *mz, int64_t mz_cap);
/* seeds: sketch + mm_seed_mz_flt + mm_collect_matches + collect_seed_hits; anchors in generation order
 * (sorted = 0), after the radix_sort_128x emulation (sorted = 1: the reference's whole sorted array) or as the mapping path hands them
 * to the chainer (sorted = 2: anchor-rich batches drop the x-components that are too small to chain, see mm355_cullsort.hip; a_off
 * then describes the shorter arrays). */
int mm355_stage_anchors(mm355_ctx_t *ctx, const mm355_mapopt_t *mo, int64_t n_reads, const char *const *seqs,
                        const int32_t *lens, int sorted, int64_t *a_off, uint64_t *a, int64_t a_cap,
                        int32_t *rep_len, int32_t *n_mini_pos);
/* chaining DP fill (mg_lchain_dp) on the sorted anchors: f, p, v per anchor (p as int32, -1 = none) */
int mm355_stage_chain(mm355_ctx_t *ctx, const mm355_mapopt_t *mo, int64_t n_reads, const char *const *seqs,
                      const int32_t *lens, int64_t *a_off, uint64_t *a, int32_t *f, int32_t *p, int32_t *v, int64_t a_cap);
/* chains after backtrack + compact_a: u (score<<32|cnt) and the compacted anchors */
int mm355_stage_chains(mm355_ctx_t *ctx, const mm355_mapopt_t *mo, int64_t n_reads, const char *const *seqs,
                       const int32_t *lens, int64_t *u_off, uint64_t *u, int64_t u_cap,
                       int64_t *a_off, uint64_t *a, int64_t a_cap);
/* chains after the long-join re-chain (mg_lchain_rmq on the chained anchors when U:map.c::mm_map_frag's rescue test fires) or, for MM_F_RMQ
 * presets, after mg_lchain_rmq as the primary chainer: u and the compacted anchors as mm355_stage_chains returns them, plus state[r]:
 * 0 = not re-chained, 1 = re-chained on the device, 2 = the long-join re-chain is left to the literal host implementation (equal range-minimum
 * priorities): a[] then holds the read's chained anchors sorted by x and u_off[r+1] == u_off[r]; 3 = every mg_lchain_rmq call of the read is
 * left to the host (MM_F_RMQ presets whose primary pass was handed back, or MM355_RMQ_ON_HOST=1): a[] = the read's sorted anchors */
int mm355_stage_rmq(mm355_ctx_t *ctx, const mm355_mapopt_t *mo, int64_t n_reads, const char *const *seqs,
                    const int32_t *lens, int64_t *u_off, uint64_t *u, int64_t u_cap,
                    int64_t *a_off, uint64_t *a, int64_t a_cap, int32_t *state);
/* one batch of banded extension problems (ksw_extd2_sse semantics); see mm355_dpjob_t */
typedef struct {
	int32_t qlen, tlen;
	int64_t qoff, toff;     /* offsets into the code arrays (0..4 per byte) */
	int32_t w, zdrop, end_bonus, flag;
} mm355_dpjob_t;
typedef struct {
	int32_t max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end, n_cigar;
	int64_t cigar_off;
} mm355_dpres_t;
int mm355_stage_dp(mm355_ctx_t *ctx, const mm355_mapopt_t *mo, int64_t n_jobs, const mm355_dpjob_t *jobs,
                   const uint8_t *qcodes, int64_t n_q, const uint8_t *tcodes, int64_t n_t,
                   mm355_dpres_t *res, uint32_t *cigar, int64_t cigar_cap);
/* the per-base walk of an aligned region (minimap2 align.c::mm_update_extra after mm_fix_cigar: mlen, blen, n_ambi, dp_max) and its cs
   string (format.c::write_cs_core, short form), as the mapping path runs them on the device for all regions of a batch (k_extra).
   Stage entry for parity tests: the query codes (0..4 per byte) come from the caller, the target from the index (contig rid, from t_st).
   want_cs: bit 0 = cs, bit 1 = MD (format.c::write_MD_core); both strings land in `cs` (cs_off / md_off). */
typedef struct { int64_t q_off; int64_t cigar_off; int32_t rid, t_st, n_cigar, pad; } mm355_extrajob_t;
typedef struct { int32_t mlen, blen, n_ambi, dp_max; int64_t cs_off; int32_t cs_len, pad; int64_t md_off; int32_t md_len, pad2; } mm355_extrares_t;
int mm355_stage_extra(mm355_ctx_t *ctx, const mm355_mapopt_t *mo, int64_t n_regions, const mm355_extrajob_t *jobs,
                      const uint8_t *qcodes, int64_t n_q, const uint32_t *cigar, int64_t n_cigar, int want_cs,
                      mm355_extrares_t *res, char *cs, int64_t cs_cap);

int mm355_get_stats(mm355_ctx_t *ctx, mm355_stats_t *st);   /* counters/timers of the last call on ctx */
int mm355_device_count(void);
int mm355_device_synchronize(int device_id);                /* drains every stream of the device (hipDeviceSynchronize): bench.py brackets its timed region with it */
const char *mm355_strerror(int code);
const char *mm355_version(void);

#ifdef __cplusplus
}
#endif
#endif
