// mm355_glue.h -- host-resident tail of the path (small, strictly sequential per read; SURVEY.md 8a rows a9-a11,
// a13): long-join re-chaining, chains -> regions, primary/secondary selection, the DP work generator that feeds
// k_ksw_extd2 in rounds, CIGAR stitching, MAPQ, cs/MD.  Batch-oriented: one ReadState per read of the batch.
#pragma once
#include "mm355_extra.h"
#include <stdint.h>
#include <vector>
#include <string>
#include "mm355_host.h"

struct Extra {
	int32_t dp_score = 0, dp_max = 0, dp_max2 = 0;
	uint32_t n_ambi = 0;
	std::vector<uint32_t> cigar;
	std::string cs, md;     // written when the region is committed (the code strings are at hand there)
	// deferred = the per-base walk of U:align.c::mm_update_extra (mlen / blen / n_ambi / dp_max) and the cs string are left to the device
	// (k_extra, one launch for all regions of the batch after the last extension round): where the walk starts in the read's code strings
	// and in the reference (after mm_fix_cigar's shifts)
	bool deferred = false;
	int32_t x_strand = 0, x_qst = 0, x_rid = 0, x_tst = 0;
};

struct ExtraLoc { int32_t strand, q_st, rid, t_st; };   // start of a region's query / target strings

struct Reg {                // U:minimap.h::mm_reg1_t
	int32_t id = 0, cnt = 0, rid = 0, score = 0;
	int32_t qs = 0, qe = 0, rs = 0, re = 0;
	int32_t parent = 0, subsc = 0, as = 0, mlen = 0, blen = 0, n_sub = 0, score0 = 0;
	uint32_t mapq = 0, split = 0, rev = 0, inv = 0, sam_pri = 0, seg_split = 0, split_inv = 0, is_alt = 0, strand_retained = 0;
	uint32_t hash = 0;
	float div = -1.0f;
	Extra *p = 0;
	int task = -1;          // alignment task bound to this region (-1: none yet)
};

struct EzRes {              // U:ksw2.h::ksw_extz_t as returned by the DP kernel
	int32_t max = 0, zdropped = 0, max_q = -1, max_t = -1, mqe = 0, mqe_t = -1, mte = 0, mte_q = -1, score = 0, reach_end = 0;
	const uint32_t *cigar = 0; int32_t n_cigar = 0;   // points into the per-round CIGAR arena kept alive by the batch
	int state = 0;          // 0 none, 1 requested, 2 done
};

struct DpReq {              // one extension problem requested by a region task
	int read, task, slot;    // slot: index into AlnTask::res
	int32_t qlen, tlen;
	int32_t q_st;            // start on the strand-adjusted query code string (qseq0[rev])
	int rev_strand;          // which query strand buffer
	uint32_t rid; int32_t t_st;
	int reversed;            // both strings reversed (left extension)
	int32_t w, zdrop, end_bonus, flag;
};

struct AlnTask {            // U:align.c::mm_align1 split into a one-off preamble and a replayable DP walk
	int reg_uid;             // identifies the region (stable across inserts)
	int32_t rid, rev, as1, cnt1;
	int32_t rs, qs, re, qe;            // anchor-midpoint bounds of the first/last seed
	int32_t rs0, qs0, re0, qe0;
	int32_t bw, bw_long;
	int split_inv;
	bool prepared = false, done = false;
	std::vector<EzRes> res;            // compact: only the extension problems that are actually requested
	std::vector<int32_t> slot_of;      // logical slot (0 left, 1 right, 2+2*i approx fill at seed i, 3+2*i exact fill) -> index into res, -1 = none
	// inversion attempt (U:align.c::mm_align1_inv)
	EzRes inv_res; int inv_state = 0;  // 0 not tried, 1 waiting for DP, 2 finished
};

struct ReadState {
	int out_flags = 0;                 // MM355_OUT_CS / MM355_OUT_MD requested for this batch
	int32_t qlen = 0;
	const char *seq = 0;
	std::vector<uint8_t> qc[2];        // query codes forward / reverse-complement
	std::vector<mm128> a;              // chained anchors (after compact_a, or after the RMQ re-chain)
	std::vector<uint64_t> u;
	std::vector<uint64_t> mini_pos;
	int32_t rep_len = 0, n_a = 0;
	std::vector<Reg> regs;
	std::vector<AlnTask> tasks;
	int cursor = 0;                    // U:align.c::mm_align_skeleton loop index
	bool aligned = false;
	int next_uid = 0;
	bool defer_extra = false;          // leave mm_update_extra's walk and cs to the device (set per batch by the caller)
};

struct GlueStats { int64_t n_rmq = 0, n_rounds = 0, n_jobs = 0; };

// stage 0, MM_F_RMQ only: rs.a = the read's sorted anchors -> chained anchors + rs.u (U:lchain.c::mg_lchain_rmq as primary chainer)
void mm355_glue_chain_rmq(const mm355_index *mi, const mm355_mapopt_t *opt, ReadState &rs);
// stage 1 (after the chain kernels): re-chain (if triggered), regions, pre-DP selection; leaves rs.regs ready for DP.
// rmq_state: what the device stage (mm355_run_rmq) did with this read's long-join re-chain -- MM355_RMQ_KEEP / _DONE: nothing left to do
// here; MM355_RMQ_HOST: rs.a holds the chained anchors already sorted by x, mg_lchain_rmq runs here; -1: the stage did not run, the
// rescue test, the sort and mg_lchain_rmq all run here (MM355_RMQ_ON_HOST=1)
void mm355_glue_pre_align(const mm355_index *mi, const mm355_mapopt_t *opt, ReadState &rs, int rmq_state);
// stage 2: advance the skeleton of one read as far as cached DP results allow; appends missing DP problems to `reqs`.
// returns true when the read needs no more DP.
bool mm355_glue_align_step(const mm355_index *mi, const mm355_mapopt_t *opt, int read_id, ReadState &rs, std::vector<DpReq> &reqs, int flags);
// stage 3: post-DP filtering, sorting, selection and MAPQ; emits hit records
void mm355_glue_finish(const mm355_index *mi, const mm355_mapopt_t *opt, ReadState &rs, int flags,
                       std::vector<mm355_hit_t> &hits, std::vector<uint32_t> &cigar, std::string &str);
void mm355_glue_release(ReadState &rs);
// between the last extension round and stage 3, when rs.defer_extra: the regions whose walk was left to the device.
// count: number of such regions, of their CIGAR operations and of cs bytes to reserve; fill: descriptors + CIGARs at the given offsets;
// apply: results back into the regions (same order as fill)
void mm355_glue_extra_count(const ReadState &rs, int64_t *n_regions, int64_t *n_segs, int64_t *n_cig, int64_t *n_cs);
void mm355_glue_extra_fill(const ReadState &rs, int64_t q_base, Mm355ExtraJob *segs, int64_t *seg_first, int64_t reg0, int64_t seg0, uint32_t *cig, int64_t cig0, int64_t cs0);
void mm355_glue_extra_apply(ReadState &rs, const Mm355ExtraOut *out, const char *cs, int want);   // want: MM355_OUT_CS | MM355_OUT_MD
