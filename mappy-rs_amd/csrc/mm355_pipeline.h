// mm355_pipeline.h -- per-GPU batch context: device buffers, streams, stage drivers
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <stdlib.h>
#include <functional>
#include "mm355_host.h"
#include "mm355_dev.h"

struct DBuf {
	void *p = 0; size_t cap = 0;
	int ensure(size_t bytes, int slack_div = 2) {
		if (bytes <= cap) return 0;
		if (p) (void)hipFree(p);
		// grow-only with 50 % headroom: a re-allocation is a hipFree + hipMalloc (both synchronise the device and stall every other context), and
		// the sub-batches of a read stream differ by tens of per cent in anchors and extension cells; a context's buffers are ~10 GB of 288 GB
		// (the direction matrices of the extension rounds -- the one buffer of ten and more GB -- take 25 %)
		static const int slack_env = [] { const char *e = getenv("MM355_BUF_SLACK_DIV"); return e? atoi(e) : 0; }();   // (experiments: 8 = 12.5 % headroom)
		if (slack_env > 0 && slack_env > slack_div) slack_div = slack_env;
		const size_t slack = bytes / (size_t)slack_div;
		size_t want = bytes + slack + 256;
		if (hipMalloc(&p, want) != hipSuccess) { p = 0; cap = 0; return -1; }
		cap = want; return 0;
	}
	void release() { if (p) (void)hipFree(p); p = 0; cap = 0; }
	template <typename T> T *as() const { return (T*)p; }
};

struct HBuf {                 // pinned host staging buffer (grow-only)
	void *p = 0; size_t cap = 0;
	int ensure(size_t bytes) {
		if (bytes <= cap) return 0;
		if (p) (void)hipHostFree(p);
		size_t want = bytes + bytes / 2 + 4096;
		if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = 0; cap = 0; return -1; }
		cap = want; return 0;
	}
	void release() { if (p) (void)hipHostFree(p); p = 0; cap = 0; }
};

struct HostBatch {            // packed reads of one sub-batch
	int64_t n_reads = 0, n_bytes = 0, n_bases = 0;
	std::vector<uint8_t> seq; std::vector<int64_t> roff; std::vector<int32_t> rlen, order;
	std::vector<int32_t> n_mz, n_a, rep_len, n_mini, n_u, n_v, status;
	std::vector<uint8_t> rmq_state;   // per read after mm355_run_rmq: MM355_RMQ_KEEP / _DONE / _HOST (empty: the stage did not run, the host decides)
	std::vector<int64_t> aoff;
	int64_t tot_a = 0;
};

// a batch of reads resident in HBM that is not the context's current one (mm355_batch_select): the packed reads, their tables and
// the host copy; every working buffer stays with the context
struct ResidentBatch { HostBatch hb; DBuf seq, roff, rlen, order, ck_read, ck_start, ck_r0; int64_t n_chunks = 0; };

// Layout of mm355_ctx::counters (u64 words), one definition for every memset / kernel argument / read-back:
//   [0..7]    seed stage (n_a_multi (1), live lookup tiles (2)), the extension's total cells (4) and dense-arena pointer (5), k_chain_segments' two list lengths (6)
//   CTR_GCELLS_OFF   cells per extension launch group, [CTR_GROUPS][CTR_SPREAD] (slot = block & (CTR_SPREAD - 1): one word takes ~88 atomics / us)
//   CTR_PAIRS_OFF    chaining pair evaluations, CTR_PAIRS_WORDS slots (slot = block & 63)
//   CTR_RMQ_OFF      window elements looked at by k_rmq_dp, CTR_RMQ_WORDS slots
//   CTR_HITS_OFF     minimizers k_seed_lookup found in the index, CTR_HITS_WORDS slots
#define CTR_HEAD_WORDS   64
#define CTR_SPREAD       16
#define CTR_GROUPS       24
#define CTR_GCELLS_OFF   CTR_HEAD_WORDS
#define CTR_GCELLS_WORDS (CTR_GROUPS * CTR_SPREAD)
#define CTR_PAIRS_OFF    (CTR_GCELLS_OFF + CTR_GCELLS_WORDS)
#define CTR_PAIRS_WORDS  64
#define CTR_RMQ_OFF      (CTR_PAIRS_OFF + CTR_PAIRS_WORDS)
#define CTR_RMQ_WORDS    64
#define CTR_HITS_OFF     (CTR_RMQ_OFF + CTR_RMQ_WORDS)     // minimizers found in the index (k_seed_lookup), CTR_HITS_WORDS slots
#define CTR_HITS_WORDS   64
#define CTR_WORDS        (CTR_HITS_OFF + CTR_HITS_WORDS)
#define CTR_BYTES        (CTR_WORDS * 8)
static_assert(CTR_GCELLS_OFF >= 8 && CTR_PAIRS_OFF == CTR_GCELLS_OFF + CTR_GCELLS_WORDS && CTR_RMQ_OFF == CTR_PAIRS_OFF + CTR_PAIRS_WORDS, "counter regions must be disjoint");

struct mm355_ctx {
	const mm355_index *mi = 0;
	std::vector<ResidentBatch> slots; int cur_slot = 0;
	int dev = 0;
	hipStream_t st = 0;
	DevIndex dix;
	// per-batch device buffers
	DBuf heavy, seq, roff, rlen, order, ck_read, ck_start, ck_n, ck_r0;
	int64_t n_chunks = 0;
	int prio_low = 0, prio_high = 0; bool use_prio = false; int ord = 0;   // ord: creation ordinal of the context
	DBuf sort_flag, tie_list, n_keep, aoff2, cs_list, tie_a, tie_b, tie_f, tie_p, tie_t8, tie_tcnt; HBuf h_cs;   // cull + sort of anchor-rich batches (mm355_cullsort.hip)
	int n_heavy = 0; hipStream_t aux_st = 0; hipEvent_t aux_ev = 0, aux_ev2 = 0; DBuf sort_tasks;
	DBuf mz, mz_tmp, n_mz, sn, sv, sflt, hl, soff, n_a, rep_len, n_mini, mini_pos, counters, err;
	DBuf aoff, a, f, p, v, z, t8, vi, b, wk, u, u2, n_u, n_v;
	// dp buffers
	DBuf dp_jobs, dp_res, dp_q, dp_t, dp_bt, dp_cig, dp_work, dp_H, dp_dense, dp_gather, pack;
	DBuf dp_bt2, dp_fail; HBuf h_fail;      // band kernels: direction matrices of the problems that are run again on the full matrix, their list
	HBuf h_res, h_cig, h_pu, h_pa, h_pm, h_seq;
	HBuf h_tasks;                          // whole-array tasks of the literal anchor sort (pinned)
	HBuf h_chunks; DBuf d_chunks;          // chunk table of k_chain_segments
	HBuf h_jobs, h_gather, h_ids;          // pinned staging of the extension round (descriptors, launch orders)
	HBuf h_arena[8]; int n_arena = 0;
	hipEvent_t dp_up_ev = 0;      // dense CIGAR arenas of the launches of the current batch (results point into them)
	DBuf kprof;    // MM355_KPROF phase counters (64 x u64)
	DBuf rq;       // per-read query codes fwd|rev
	DBuf rmq_list, rmq_flag; HBuf h_rmq;   // device mg_lchain_rmq: listed reads, per-read state
	DBuf x_jobs, x_cig, x_cs, x_out, x_dense; HBuf h_xjobs, h_xcig, h_xout, h_xcs;   // k_extra (mm_update_extra's walk + cs on the device)
	mm355_stats_t stats;
	hipEvent_t ev0 = 0, ev1 = 0;
	std::vector<hipEvent_t> tev; std::vector<double*> tacc; int n_tpend = 0;   // lazy stage timers (EvTimer, mm355_kt)
	unsigned long long pairs_land[64] = {};   // landing zone of the chain stage's pair counters (mm355_run_backtrack)
	int pool_slot = -1;                // >= 0: st / aux_st are the device pool's (mm355_ctx_create), not this context's
	bool timers_on = true;             // off for calls of fewer than 16 reads (two event records per kernel are a fifth of a single-read call); MM355_TIMERS=1 / 0 forces
	int kt_open[KT_N] = {};   // open mm355_kt pair of a slot: its event-pair index + 1
	hipStream_t dp_st[16] = {}; hipEvent_t dp_ev[24] = {}, dp_ev0[24] = {}, dp_ev1[24] = {};
	HostBatch hb;
};

DevParams mm355_make_params(const mm355_mapopt_t *mo, const mm355_index *mi);
bool mm355_dp_shared_streams();
int mm355_dp_stream(mm355_ctx *c, int sidx, hipStream_t *out);   // stream of an extension kernel class (shared by the contexts of a device)
int mm355_check_opts(const mm355_mapopt_t *mo, const mm355_index *mi);

// stage drivers (each leaves its outputs resident on the device and the per-read counts in ctx->hb)
extern "C" int mm355_map_resident(mm355_ctx_t *c, const mm355_mapopt_t *mo, int flags, mm355_hits_t **out);
extern void (*mm355_parallel_hook)(int64_t n, const std::function<void(int64_t)> &f);   // the host pool's parallel loop (mm355_map.hip), or null
int mm355_run_pack(mm355_ctx *ctx, int64_t n_reads, const char *const *seqs, const int32_t *lens);
int mm355_run_sketch(mm355_ctx *ctx);
int mm355_run_seeds(mm355_ctx *ctx, const DevParams &pr);                  // mz_flt + lookup + select (+ D2H counts, anchor offsets)
int mm355_run_expand(mm355_ctx *ctx, const DevParams &pr);
int mm355_run_sort(mm355_ctx *ctx, const DevParams &pr, int cull = 1);
int mm355_run_chain(mm355_ctx *ctx, const DevParams &pr);
int mm355_run_backtrack(mm355_ctx *ctx, const DevParams &pr);
int mm355_run_chain_skip(mm355_ctx *c);
int mm355_run_rmq(mm355_ctx *c, const mm355_mapopt_t *mo, const DevParams &pr);   // mg_lchain_rmq on the device: long-join re-chain, or the primary chainer of MM_F_RMQ presets

// time one launch group on the context's stream with HIP events (the stream the kernels are launched on)
// Stage timers.  EvTimer records a pair of events around the launches of a stage and does NOT synchronise: the pairs are turned into
// milliseconds by mm355_timers_resolve() once the call has synchronised its stream anyway (end of mm355_map_resident, the stage entry
// points, mm355_get_stats) -- a host synchronisation per stage costs a single-read call ~0.3 ms.  EvTimer2 is the synchronising form
// (extension rounds: the round ends with a synchronisation in any case).
struct EvTimer2 {
	mm355_ctx *c; double *acc;
	EvTimer2(mm355_ctx *c_, double *a) : c(c_), acc(a) { (void)hipEventRecord(c->ev0, c->st); }
	~EvTimer2() { float ms = 0; (void)hipEventRecord(c->ev1, c->st); (void)mm355_wait_stream(c->st); (void)hipEventElapsedTime(&ms, c->ev0, c->ev1); *acc += ms; }
};
void mm355_timers_resolve(mm355_ctx *c);
void mm355_kprof_dump(mm355_ctx *c);
struct EvTimer {
	mm355_ctx *c; int slot;
	EvTimer(mm355_ctx *c_, double *a) : c(c_), slot(-1)
	{
		if (!c->timers_on) return;
		if (c->n_tpend >= 120) mm355_timers_resolve(c);
		slot = c->n_tpend++;
		while ((int)c->tev.size() < 2 * (slot + 1)) { hipEvent_t e = 0; (void)hipEventCreate(&e); c->tev.push_back(e); }
		if ((int)c->tacc.size() <= slot) c->tacc.resize(slot + 1);
		c->tacc[slot] = a;
		(void)hipEventRecord(c->tev[2 * slot], c->st);
	}
	~EvTimer() { if (slot >= 0) (void)hipEventRecord(c->tev[2 * slot + 1], c->st); }
};

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[mm355] HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return MM355_EHIP; } } while (0)

int mm355_sort_levels(const mm355_index *mi);
bool mm355_cull_sort_fits(const mm355_ctx *c);   // the 8-byte words of mm355_cullsort.hip can hold this batch (position bits + index bits <= 64)
int mm355_cull_sort(mm355_ctx *c, const DevParams &pr, int cull);   // mm355_cullsort.hip: anchors that cannot chain dropped, the rest sorted per read in LDS
hipError_t mm355_wait_stream(hipStream_t st);   // polls hipStreamQuery with short naps (MM355_BLOCKING_WAIT=0: hipStreamSynchronize, =1: blocking-sync event)
void mm355_trace_add(const void *ctx, const char *phase, double t0, double t1);   // MM355_TRACE timeline (no-op when unset)
double mm355_now_ms();
