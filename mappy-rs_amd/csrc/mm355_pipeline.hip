// mm355_pipeline.hip -- per-GPU context and the stage drivers of the seeding + chaining half of the path.
// Replaces the reference's per-thread mm_tbuf_t + the worker loop body at /root/reference/src/lib.rs:587-593:
// instead of N OS threads each calling mm_map on one read, one context owns one GPU, the index is uploaded
// once into its HBM, and a whole batch of reads moves through the kernels of mm355_kernels.hip.
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <numeric>
#include <mutex>
#include <atomic>
#include <functional>
#include "mm355_pipeline.h"
#include "mm355_rmq.h"

// ------------------------------------------------------------------ options -> kernel parameters
int mm355_check_opts(const mm355_mapopt_t *mo, const mm355_index *mi)
{
	if (mi->flag & 2) return MM355_EUNSUP;                                           // MM_I_NO_SEQ with MM_F_CIGAR (always set, lib.rs:339): "No sequence in this index" (lib.rs:710-714)
	if (mo->flag & (MMF_SPLICE | 0x100LL | 0x200LL | MMF_SR | MMF_QSTRAND | MMF_HEAP_SORT)) return MM355_EUNSUP;   // 0x100/0x200: SPLICE_FOR/REV imply SPLICE (U:options.c::mm_mapopt_update)
	if (!(mo->flag & MMF_CIGAR)) return MM355_EUNSUP;                                // the reference always sets it (lib.rs:339)
	if (mo->max_chain_iter > 8000 || mo->max_chain_iter < 1) return MM355_EUNSUP;    // LDS mark window of k_chain
	if (mi->w > 64 || mi->k > 28 || mi->k < 1) return MM355_EUNSUP;
	if (mo->sdust_thres > 0) return MM355_EUNSUP;
	{   // U:ksw2_extd2_sse.c: "if (-min_sc > 2 * (q + e)) return;" -- the kernel then returns an empty result which U:align.c::mm_align1
		// dereferences (r->p->dp_score with r->p == NULL: the reference crashes or adds KSW_NEG_INF).  No defined result to reproduce.
		const int b = mo->b > 0? mo->b : -mo->b, amb = mo->sc_ambi > 0? mo->sc_ambi : -mo->sc_ambi;
		const int ge1 = mo->q + mo->e, ge2 = mo->q2 + mo->e2;
		if ((b > amb? b : amb) > 2 * (ge1 < ge2? ge1 : ge2)) return MM355_EINVAL;
	}
	return 0;
}

DevParams mm355_make_params(const mm355_mapopt_t *mo, const mm355_index *mi)
{
	DevParams p;
	memset(&p, 0, sizeof(p));
	p.flag = mo->flag;
	p.mid_occ = mo->mid_occ; p.max_max_occ = mo->max_max_occ; p.occ_dist = mo->occ_dist;
	p.q_occ_frac = mo->q_occ_frac;
	p.max_gap = mo->max_gap; p.max_gap_ref = mo->max_gap_ref; p.max_frag_len = mo->max_frag_len;
	p.bw = mo->bw; p.max_chain_skip = mo->max_chain_skip; p.max_chain_iter = mo->max_chain_iter;
	p.min_cnt = mo->min_cnt; p.min_chain_score = mo->min_chain_score;
	p.pen_gap = (float)(mo->chain_gap_scale * 0.01 * mi->k);     // U:map.c: float <- double product
	p.pen_skip = (float)(mo->chain_skip_scale * 0.01 * mi->k);
	return p;
}

// ------------------------------------------------------------------ context
extern "C" int mm355_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

void mm355_timers_resolve(mm355_ctx *c)
{
	for (int i = 0; i < c->n_tpend; ++i) {
		float ms = 0;
		if (hipEventSynchronize(c->tev[2 * i + 1]) == hipSuccess && hipEventElapsedTime(&ms, c->tev[2 * i], c->tev[2 * i + 1]) == hipSuccess) *c->tacc[i] += ms;
	}
	c->n_tpend = 0;
}

// per-kernel timer (mm355_dev.h): the pair shares the lazy event slots of EvTimer
void mm355_kt(void *kt, int slot, int end, hipStream_t st)
{
	mm355_ctx *c = (mm355_ctx*)kt;
	if (c == 0 || slot < 0 || slot >= KT_N || !c->timers_on) return;
	if (!end) {
		if (c->n_tpend >= 120) mm355_timers_resolve(c);
		const int k = c->n_tpend++;
		while ((int)c->tev.size() < 2 * (k + 1)) { hipEvent_t e = 0; (void)hipEventCreate(&e); c->tev.push_back(e); }
		if ((int)c->tacc.size() <= k) c->tacc.resize(k + 1);
		c->tacc[k] = &c->stats.ms_kernel[slot];
		c->kt_open[slot] = k + 1;
		(void)hipEventRecord(c->tev[2 * k], st);
	} else if (c->kt_open[slot] > 0) {
		(void)hipEventRecord(c->tev[2 * (c->kt_open[slot] - 1) + 1], st);
		c->kt_open[slot] = 0;
	}
}

extern "C" int mm355_device_synchronize(int device_id)
{
	HIPCHK(hipSetDevice(device_id));
	HIPCHK(hipDeviceSynchronize());
	return 0;
}

// Streams of the extension rounds.  The runtime multiplexes the HIP streams of a priority level over GPU_MAX_HW_QUEUES (8) hardware queues,
// handed out round-robin at stream creation; a kernel waits for everything in front of it on its QUEUE, whatever stream that came from.
// With eight extension streams per context (round 2) the long latency chains of one context (k_ksw_regw8 / k_ksw_rowl: a few dozen
// alignments for 10-20 ms) sat on the queue of another context's k_ksw_row<2> -- a kernel of the TURN, which every other context's round is
// waiting for (rocprofv3 trace of round 3: a turn kernel started 17 ms late behind such a chain; the turn kernels covered 56 % of the time).
// The rounds take turns anyway, so the classes need no stream per context: one pool of eight per device, each class on a queue of its own
//   0 row<2> + approximate targets <= 256    2 row<8> + approximate 1024    3 row<4> + approximate 512        (the turn)
//   1 exact register classes   4 eight-wave LDS kernel (all long targets)   6 k_ksw_rowl   5 / 7 k_ksw_regw8 (contexts alternate)
// Measured (round 3, default bench, alternating runs on one box): shared pool 853 / 795 Mbases/s against 876 / 865 with eight streams per
// context -- the exact classes and the long chains of different contexts then wait for one another on their one stream, which costs more
// than the occasional held turn.  Kept as an experiment switch (MM355_DP_SHARED_STREAMS=1); the default is a set of streams per context.
// Priority of an extension stream (experiment, off by default).  The runtime keeps a pool of hardware queues PER PRIORITY LEVEL: with the wide
// grids of the turn (classes 0, 2, 3) and the latency chains (1 exact register classes, 4 / 5 long targets, 6 k_ksw_rowl, 7 k_ksw_regw8) on
// one level, a turn kernel of one context sometimes sits on the hardware queue of another context's k_ksw_rowl / k_ksw_regw8 and starts when
// that chain ends, 6-19 ms late, with every other context's round waiting for the turn (rocprofv3 trace of the round-4 default bench: 12 such
// starts in 96 turns).  A level of their own for the chains (MM355_DP_PRIO3=1: chains normal, turn least; =2: turn normal, chains least)
// removes that -- and costs more than it saves: 1276 1279 1336 (=1) and 1320 1319 (=2) against 1380 1435 1428 / 1406 1422 Mbases/s with one
// level for every extension stream (alternating runs on one box): a third level is eight more hardware queues, and more than sixteen in
// use were slower in every sweep of GPU_MAX_HW_QUEUES as well (profiles/r04_knob_sweeps.txt).
static int dp_stream_prio(const mm355_ctx *c, int sidx)
{
	static const int three = [] { const char *e = getenv("MM355_DP_PRIO3"); return e? atoi(e) : 0; }();   // 1: chains normal, turn least; 2: turn normal, chains least
	const bool chain = sidx == 1 || sidx >= 4;
	return three && (three == 1? chain : !chain) && c->prio_low - c->prio_high >= 2? (c->prio_low + c->prio_high) / 2 : c->prio_low;
}
struct StreamPool { hipStream_t main[8], aux[8]; bool ready; uint8_t used; };   // main + sort stream of up to eight contexts per device (mm355_ctx_create)
static StreamPool g_pool[16];
static std::mutex g_pool_mu;
bool mm355_dp_shared_streams() { static const bool on = [] { const char *e = getenv("MM355_DP_SHARED_STREAMS"); return e && atoi(e) != 0; }(); return on; }
int mm355_dp_stream(mm355_ctx *c, int sidx, hipStream_t *out)
{
	static std::mutex mu;
	static hipStream_t pool[16][8];
	static bool ready[16];
	if (!mm355_dp_shared_streams()) {
		hipStream_t *slot = &c->dp_st[sidx];
		if (*slot == 0) { if (c->use_prio) HIPCHK(hipStreamCreateWithPriority(slot, hipStreamNonBlocking, dp_stream_prio(c, sidx))); else HIPCHK(hipStreamCreateWithFlags(slot, hipStreamNonBlocking)); }
		*out = *slot;
		return 0;
	}
	const int d = c->dev & 15;
	{
		std::lock_guard<std::mutex> lk(mu);
		if (!ready[d]) {
			for (int i = 0; i < 8; ++i) {
				if (c->use_prio) HIPCHK(hipStreamCreateWithPriority(&pool[d][i], hipStreamNonBlocking, dp_stream_prio(c, i)));
				else HIPCHK(hipStreamCreateWithFlags(&pool[d][i], hipStreamNonBlocking));
			}
			ready[d] = true;
		}
	}
	if (sidx == 5) sidx = 4;                            // one stream for every long-target launch
	if (sidx == 7 && (c->ord & 1)) sidx = 5;            // k_ksw_regw8: two streams, the contexts alternate
	*out = pool[d][sidx];
	return 0;
}

extern "C" int mm355_ctx_create(const mm355_index_t *mi, int device_id, mm355_ctx_t **out)
{
	*out = 0;
	if (mi == 0) return MM355_ENOIDX;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n == 0 || device_id >= n) return MM355_ENODEV;
	HIPCHK(hipSetDevice(device_id));
	mm355_ctx *c = new mm355_ctx();
	c->mi = mi; c->dev = device_id;
	{   // the per-read front kernels are latency chains of single waves: their stream outranks the extension streams, whose wide
		// grids would otherwise occupy every CU slot and stretch the front of the other contexts (MM355_STREAM_PRIO=0 disables)
		int lo = 0, hi = 0;
		static const bool use_prio = [] { const char *e = getenv("MM355_STREAM_PRIO"); return !(e && atoi(e) == 0); }();
		(void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo = least, hi = greatest (numerically lower)
		// The main and the sort stream of the first eight contexts of a device come from a pool that is created in one go -- eight main streams, then
		// eight sort streams: the runtime multiplexes the streams of a priority level over eight hardware queues, handed out in turn at stream
		// creation, and a kernel waits for everything in front of it on its QUEUE.  Created context by context (main, sort, main, sort ...) the main
		// streams of contexts i and i + 4 shared a queue, and so did their sort streams: the front of one context waited for the other's kernels.
		// From the pool, the two streams of a context share a queue with each other and with no other context (1401 against 1350 Mbases/s, six and
		// four alternating runs; MM355_STREAM_POOL=0: streams of its own for every context, as before).
		static const bool pool_on = [] { const char *e = getenv("MM355_STREAM_POOL"); return !(e && atoi(e) == 0); }();
		if (pool_on && use_prio && hi < lo) {
			std::lock_guard<std::mutex> lk(g_pool_mu);
			StreamPool &P = g_pool[device_id & 15];
			if (!P.ready) {
				for (int i = 0; i < 8; ++i) HIPCHK(hipStreamCreateWithPriority(&P.main[i], hipStreamDefault, hi));
				for (int i = 0; i < 8; ++i) HIPCHK(hipStreamCreateWithPriority(&P.aux[i], hipStreamNonBlocking, hi));
				P.ready = true;
			}
			for (int k = 0; k < 8; ++k) if (!(P.used >> k & 1)) { P.used |= (uint8_t)(1u << k); c->pool_slot = k; c->st = P.main[k]; c->aux_st = P.aux[k]; break; }
		}
		if (c->st) {}
		else if (use_prio && hi < lo) HIPCHK(hipStreamCreateWithPriority(&c->st, hipStreamDefault, hi));
		else HIPCHK(hipStreamCreate(&c->st));
		c->prio_low = use_prio && hi < lo? lo : 0; c->prio_high = use_prio && hi < lo? hi : 0; c->use_prio = use_prio && hi < lo;
		// The streams of the extension classes are created here, back to back under a lock: the runtime hands out hardware queues round-robin at
		// stream creation, and the classes of one context must not share a queue (MM355_DP_SHARED_STREAMS=1: one pool per device, mm355_dp_stream).
		static std::mutex mk;
		std::lock_guard<std::mutex> lk(mk);
		static std::atomic<int> n_ctx(0);
		c->ord = n_ctx.fetch_add(1);
		if (!mm355_dp_shared_streams()) {
			// MM355_DP_QALIGN=1 (experiment): all eight extension streams at once, the four of the turn first (0, 2, 3 and the exact classes 1), then the
			// four chains (4 / 5 long targets, 6 k_ksw_rowl, 7 k_ksw_regw8) rotated by the context's ordinal: with eight queues handed out in turn, every
			// context's turn streams sit on queues 0-3 -- shared only with other contexts' turn streams, and turns exclude one another -- and a chain
			// of context j on queue 4 + (class + j) mod 4.
			static const bool qalign = [] { const char *e = getenv("MM355_DP_QALIGN"); return e && atoi(e) != 0; }();
			static const int turn_first[4] = { 0, 2, 3, 1 };
			for (int t = 0; t < (qalign? 8 : 6); ++t) {   // 0..3 the register classes, 4 and 5 the eight-wave kernel (mm355_dp_run): with the main and the sort stream, 8 per context
				const int i = !qalign? t : t < 4? turn_first[t] : 4 + ((t - c->ord) & 3);
				if (c->use_prio) HIPCHK(hipStreamCreateWithPriority(&c->dp_st[i], hipStreamNonBlocking, dp_stream_prio(c, i)));
				else HIPCHK(hipStreamCreateWithFlags(&c->dp_st[i], hipStreamNonBlocking));
			}
		} else { hipStream_t t; int rc = mm355_dp_stream(c, 0, &t); if (rc) { mm355_ctx_destroy(c); return rc; } }
		// the stream of the block-level sort of anchor-rich reads: same consideration (7 streams per context, 8 hardware queues)
		if (c->aux_st) {} else if (c->use_prio) HIPCHK(hipStreamCreateWithPriority(&c->aux_st, hipStreamNonBlocking, c->prio_high)); else HIPCHK(hipStreamCreateWithFlags(&c->aux_st, hipStreamNonBlocking));
		HIPCHK(hipEventCreateWithFlags(&c->aux_ev, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->aux_ev2, hipEventDisableTiming));
	}
	HIPCHK(hipEventCreate(&c->ev0)); HIPCHK(hipEventCreate(&c->ev1));
	// the index replica of this device (shared by all its contexts; created on first use: H2D from the host image or a peer copy)
	mm355_replica rp;
	{ int rc = mm355_index_replica(mi, device_id, &rp); if (rc) { mm355_ctx_destroy(c); return rc; } }
	c->dix.slots = (const mm355_slot*)rp.slots; c->dix.line_mask = mi->n_lines - 1;
	c->dix.pos = (const uint64_t*)rp.pos; c->dix.S2 = (const uint32_t*)rp.S2; c->dix.nr = (const uint64_t*)rp.nr; c->dix.n_nr = rp.n_nr;
	c->dix.seq_off = (const uint64_t*)rp.seq_off; c->dix.seq_len = (const uint32_t*)rp.seq_len;
	c->dix.k = mi->k; c->dix.w = mi->w; c->dix.b = mi->b; c->dix.flag = mi->flag; c->dix.n_seq = mi->n_seq;
	if (c->counters.ensure(CTR_BYTES) || c->err.ensure(16)) { mm355_ctx_destroy(c); return MM355_ENOMEM; }
	if (getenv("MM355_KPROF")) { if (c->kprof.ensure(512)) { mm355_ctx_destroy(c); return MM355_ENOMEM; } HIPCHK(hipMemset(c->kprof.p, 0, 512)); }
	c->n_tpend = 0; memset(&c->stats, 0, sizeof(c->stats));
	*out = c;
	return 0;
}

// MM355_KPROF=1: per-phase shader-cycle counters of the per-read latency kernels (sum over reads / slowest read), printed and reset
void mm355_kprof_dump(mm355_ctx *c)
{
	if (c->kprof.p == 0) return;
	unsigned long long h[64];
	if (hipMemcpy(h, c->kprof.p, 512, hipMemcpyDeviceToHost) != hipSuccess) return;
	(void)hipMemset(c->kprof.p, 0, 512);
	static const char *nm[32] = { "bt:zlist", "bt:zsort", "bt:walk", "bt:compact", "sel:hits", "sel:streaks", "sel:tail", "-",
	                              "srt:total", "-", "-", "-", "chb:total", "chs:total", "-", "-", "mzf:total", "rmq:windows", "rmq:dp", "-", "rbt:zlist", "rbt:zsort", "rbt:walk", "rbt:compact", "sk:piece", "sk:pack", "-", "-", "-", "-", "-", "-" };
	fprintf(stderr, "[mm355] kprof (Mcycles: sum over reads / slowest read):");
	for (int i = 0; i < 32; ++i) if (h[i]) fprintf(stderr, " %s %.1f/%.2f", nm[i], h[i] / 1e6, h[32 + i] / 1e6);
	fprintf(stderr, "\n");
}

// ------------------------------------------------------------------ index replicas (one per device)
// The reference in HBM is packed 2 bits per base (16 bases per word: half the bytes of the .mmi's 4-bit image for the same gather); the
// positions of ambiguous bases -- a few hundred runs in an assembly -- are a sorted interval table beside it.
__global__ void k_pack2(const uint32_t *S4, uint32_t *S2, uint64_t n_w2, uint64_t n_w4)
{
	const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= n_w2) return;
	uint32_t v = 0;
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const uint64_t w4 = 2 * w + h;
		const uint32_t x = w4 < n_w4? S4[w4] : 0u;
#pragma unroll
		for (int j = 0; j < 8; ++j) { const uint32_t c = x >> (4 * j) & 0xfu; v |= (c < 4? c : 0u) << (16 * h + 2 * j); }
	}
	S2[w] = v;
}
int mm355_replica_pack2(const mm355_index *mi, mm355_replica *rp)
{
	const uint64_t sum_len = mi->n_seq? mi->seq_off[mi->n_seq - 1] + mi->seq_len[mi->n_seq - 1] : 0;
	if (!mi->nrun_done) {   // the runs of ambiguous bases, from the host image (words without one are skipped eight bases at a time)
		std::vector<uint64_t> &nr = mi->nrun;
		nr.clear();
		const uint64_t nw = (sum_len + 7) / 8;
		bool in = false;
		for (uint64_t w = 0; w < nw && w < mi->S.size(); ++w) {
			const uint32_t x = mi->S[w];
			if ((x & 0xccccccccu) == 0) { if (in) { nr.push_back(w * 8); in = false; } continue; }   // (codes are 0..4: any of bits 2, 3 = ambiguous)
			for (int j = 0; j < 8; ++j) {
				const uint64_t o = w * 8 + j;
				if (o >= sum_len) break;
				const bool n = (x >> (4 * j) & 0xfu) > 3;
				if (n && !in) { nr.push_back(o); in = true; }
				else if (!n && in) { nr.push_back(o); in = false; }
			}
		}
		if (in) nr.push_back(sum_len);
		mi->nrun_done = true;
	}
	const uint64_t n_w4 = (sum_len + 7) / 8 + 2, n_w2 = (sum_len + 15) / 16 + 2;
	if (hipMalloc(&rp->S2, n_w2 * 4 + 16) != hipSuccess) return MM355_ENOMEM;
	if (hipMalloc(&rp->nr, mi->nrun.size() * 8 + 16) != hipSuccess) return MM355_ENOMEM;
	hipLaunchKernelGGL(k_pack2, dim3((unsigned)((n_w2 + 255) / 256)), dim3(256), 0, 0, (const uint32_t*)rp->S, (uint32_t*)rp->S2, n_w2, n_w4);
	if (hipGetLastError() != hipSuccess) return MM355_EHIP;
	if (!mi->nrun.empty() && hipMemcpy(rp->nr, mi->nrun.data(), mi->nrun.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return MM355_EHIP;
	if (hipDeviceSynchronize() != hipSuccess) return MM355_EHIP;
	rp->n_nr = (uint32_t)(mi->nrun.size() / 2);
	(void)hipFree(rp->S); rp->S = 0;
	return 0;
}
int mm355_index_replica(const mm355_index *mi, int dev, mm355_replica *out)
{
	std::lock_guard<std::mutex> lk(mi->rep_mu);
	for (const mm355_replica &r : mi->replicas) if (r.dev == dev) { *out = r; return 0; }
	int prev = 0; (void)hipGetDevice(&prev);
	HIPCHK(hipSetDevice(dev));
	mm355_replica rp; rp.dev = dev;
	const size_t sb = (size_t)mi->n_lines * MM355_SLOTS_PER_LINE * sizeof(mm355_slot);
	const size_t np = mi->dev_resident? (size_t)mi->n_pos : mi->pos.size(), pb = (np + 2) * 8;
	uint64_t sum_len = mi->n_seq? mi->seq_off[mi->n_seq - 1] + mi->seq_len[mi->n_seq - 1] : 0;
	const size_t Sw = (sum_len + 7) / 8 + 2, Sb = Sw * 4;
	auto fail = [&](int code) { if (rp.slots) (void)hipFree(rp.slots); if (rp.pos) (void)hipFree(rp.pos); if (rp.S) (void)hipFree(rp.S); if (rp.S2) (void)hipFree(rp.S2); if (rp.nr) (void)hipFree(rp.nr);
	                            if (rp.seq_off) (void)hipFree(rp.seq_off); if (rp.seq_len) (void)hipFree(rp.seq_len); (void)hipSetDevice(prev); return code; };
	if (hipMalloc(&rp.slots, sb) != hipSuccess || hipMalloc(&rp.pos, pb) != hipSuccess || hipMalloc(&rp.S, Sb + 16) != hipSuccess ||
	    hipMalloc(&rp.seq_off, (size_t)mi->n_seq * 8 + 8) != hipSuccess || hipMalloc(&rp.seq_len, (size_t)mi->n_seq * 4 + 8) != hipSuccess) return fail(MM355_ENOMEM);
	if (hipMemset(rp.S, 0, Sb + 16) != hipSuccess) return fail(MM355_EHIP);
	if (mi->dev_resident) {   // table and pos[] exist only in HBM of the build device: device-to-device over xGMI (or through the host if peers are not connected)
		const mm355_replica *src = 0;
		for (const mm355_replica &r : mi->replicas) if (r.dev == mi->dev_id) src = &r;
		if (src == 0) return fail(MM355_EINVAL);
		if (hipMemcpyPeer(rp.slots, dev, src->slots, src->dev, sb) != hipSuccess) return fail(MM355_EHIP);
		if (np && hipMemcpyPeer(rp.pos, dev, src->pos, src->dev, np * 8) != hipSuccess) return fail(MM355_EHIP);
	} else {
		if (hipMemcpy(rp.slots, mi->slots.data(), sb, hipMemcpyHostToDevice) != hipSuccess) return fail(MM355_EHIP);
		if (np && hipMemcpy(rp.pos, mi->pos.data(), np * 8, hipMemcpyHostToDevice) != hipSuccess) return fail(MM355_EHIP);
	}
	if (!mi->S.empty() && hipMemcpy(rp.S, mi->S.data(), std::min(Sb, mi->S.size() * 4), hipMemcpyHostToDevice) != hipSuccess) return fail(MM355_EHIP);
	if (hipMemcpy(rp.seq_off, mi->seq_off.data(), (size_t)mi->n_seq * 8, hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(rp.seq_len, mi->seq_len.data(), (size_t)mi->n_seq * 4, hipMemcpyHostToDevice) != hipSuccess) return fail(MM355_EHIP);
	{ const int rc2 = mm355_replica_pack2(mi, &rp); if (rc2) return fail(rc2); }
	mi->replicas.push_back(rp);
	(void)hipSetDevice(prev);
	*out = rp;
	return 0;
}

void mm355_index_free_replicas(mm355_index *mi)
{
	std::lock_guard<std::mutex> lk(mi->rep_mu);
	int prev = 0; (void)hipGetDevice(&prev);
	for (mm355_replica &r : mi->replicas) {
		(void)hipSetDevice(r.dev);
		if (r.slots) (void)hipFree(r.slots); if (r.pos) (void)hipFree(r.pos); if (r.S) (void)hipFree(r.S);
		if (r.seq_off) (void)hipFree(r.seq_off); if (r.seq_len) (void)hipFree(r.seq_len); if (r.S2) (void)hipFree(r.S2); if (r.nr) (void)hipFree(r.nr);
	}
	if (!mi->replicas.empty()) (void)hipSetDevice(prev);
	mi->replicas.clear();
	mi->d_slots = mi->d_pos = mi->d_S = 0;
}

// replicates the index into the HBM of every listed device (idempotent); contexts of those devices then share the replica
extern "C" int mm355_upload(mm355_index_t *mi, const int *device_ids, int n)
{
	if (mi == 0) return MM355_ENOIDX;
	if (n < 0 || (n > 0 && device_ids == 0)) return MM355_EINVAL;
	int nd = 0;
	if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) return MM355_ENODEV;
	for (int i = 0; i < n; ++i) {
		if (device_ids[i] < 0 || device_ids[i] >= nd) return MM355_ENODEV;
		mm355_replica rp;
		int rc = mm355_index_replica(mi, device_ids[i], &rp);
		if (rc) return rc;
	}
	return 0;
}

extern "C" void mm355_ctx_destroy(mm355_ctx_t *c)
{
	if (c == 0) return;
	(void)hipSetDevice(c->dev);
	DBuf *bufs[] = { &c->sort_tasks, &c->sort_flag, &c->tie_list, &c->n_keep, &c->aoff2, &c->cs_list, &c->tie_a, &c->tie_b, &c->tie_f, &c->tie_p, &c->tie_t8, &c->tie_tcnt, &c->heavy, &c->seq, &c->roff, &c->rlen, &c->order, &c->ck_read, &c->ck_start, &c->ck_n, &c->ck_r0,
		&c->mz, &c->mz_tmp, &c->n_mz, &c->sn, &c->sv, &c->sflt, &c->hl, &c->soff, &c->n_a, &c->rep_len, &c->n_mini, &c->mini_pos, &c->counters, &c->err,
		&c->aoff, &c->a, &c->f, &c->p, &c->v, &c->z, &c->t8, &c->vi, &c->b, &c->wk, &c->u, &c->u2, &c->n_u, &c->n_v,
		&c->kprof, &c->d_chunks, &c->dp_jobs, &c->dp_res, &c->dp_q, &c->dp_t, &c->dp_bt, &c->dp_bt2, &c->dp_fail, &c->dp_cig, &c->dp_work, &c->dp_H, &c->rq, &c->dp_dense, &c->dp_gather, &c->pack, &c->rmq_list, &c->rmq_flag, &c->x_jobs, &c->x_cig, &c->x_cs, &c->x_out, &c->x_dense };
	for (DBuf *b : bufs) b->release();
	for (ResidentBatch &r : c->slots) { r.seq.release(); r.roff.release(); r.rlen.release(); r.order.release(); r.ck_read.release(); r.ck_start.release(); r.ck_r0.release(); }
	c->h_fail.release(); c->h_cs.release(); c->h_rmq.release(); c->h_xjobs.release(); c->h_xcig.release(); c->h_xout.release(); c->h_xcs.release(); c->h_tasks.release(); c->h_chunks.release(); c->h_res.release(); c->h_jobs.release(); c->h_gather.release(); c->h_ids.release(); for (int i = 0; i < 8; ++i) c->h_arena[i].release(); c->h_cig.release(); c->h_pu.release(); c->h_pa.release(); c->h_pm.release(); c->h_seq.release();
	for (int i = 0; i < 16; ++i) if (c->dp_st[i]) (void)hipStreamDestroy(c->dp_st[i]);
	for (int i = 0; i < 24; ++i) { if (c->dp_ev[i]) (void)hipEventDestroy(c->dp_ev[i]); if (c->dp_ev0[i]) (void)hipEventDestroy(c->dp_ev0[i]); if (c->dp_ev1[i]) (void)hipEventDestroy(c->dp_ev1[i]); }
	if (c->dp_up_ev) (void)hipEventDestroy(c->dp_up_ev);
	if (c->aux_st && c->pool_slot < 0) (void)hipStreamDestroy(c->aux_st);
	if (c->aux_ev) (void)hipEventDestroy(c->aux_ev);
	if (c->aux_ev2) (void)hipEventDestroy(c->aux_ev2);
	for (hipEvent_t e : c->tev) (void)hipEventDestroy(e);
	if (c->ev0) (void)hipEventDestroy(c->ev0);
	if (c->ev1) (void)hipEventDestroy(c->ev1);
	if (c->st && c->pool_slot < 0) (void)hipStreamDestroy(c->st);
	if (c->pool_slot >= 0) { std::lock_guard<std::mutex> lk(g_pool_mu); g_pool[c->dev & 15].used &= (uint8_t)~(1u << c->pool_slot); }   // (the streams stay with the device's pool)
	delete c;
}

extern "C" int mm355_get_stats(mm355_ctx_t *c, mm355_stats_t *st) { if (c == 0) return MM355_EINVAL; mm355_timers_resolve(c); *st = c->stats; return 0; }

static DevBatch dev_batch(mm355_ctx *c)
{
	DevBatch b;
	b.n_reads = (int32_t)c->hb.n_reads; b.seq = c->seq.as<uint8_t>(); b.roff = c->roff.as<int64_t>();
	b.rlen = c->rlen.as<int32_t>(); b.order = c->order.as<int32_t>();
	b.prof = c->kprof.as<unsigned long long>();
	return b;
}
static DevSeeds dev_seeds(mm355_ctx *c)
{
	DevSeeds s;
	s.mz = c->mz.as<mm128>(); s.mz_tmp = c->mz_tmp.as<mm128>(); s.n_mz = c->n_mz.as<int32_t>();
	s.sn = c->sn.as<uint32_t>(); s.sv = c->sv.as<uint64_t>(); s.sflt = c->sflt.as<uint8_t>(); s.hl = c->hl.as<int32_t>();
	s.soff = c->soff.as<uint32_t>(); s.n_a = c->n_a.as<int32_t>(); s.rep_len = c->rep_len.as<int32_t>();
	s.n_mini = c->n_mini.as<int32_t>(); s.mini_pos = c->mini_pos.as<uint64_t>(); s.counters = c->counters.as<unsigned long long>();
	return s;
}
static DevAnchors dev_anchors(mm355_ctx *c)
{
	DevAnchors a;
	a.aoff = c->aoff.as<int64_t>(); a.a = c->a.as<mm128>(); a.f = c->f.as<int32_t>(); a.p = c->p.as<int32_t>(); a.v = c->v.as<int32_t>();
	a.z = c->z.as<uint64_t>(); a.t8 = c->t8.as<uint8_t>(); a.vi = c->vi.as<int32_t>(); a.b = c->b.as<mm128>(); a.wk = c->wk.as<mm128>();
	a.u = c->u.as<uint64_t>(); a.u2 = c->u2.as<uint64_t>(); a.n_u = c->n_u.as<int32_t>(); a.n_v = c->n_v.as<int32_t>();
	a.tcnt = 0;
	return a;
}

// ------------------------------------------------------------------ stage drivers
void (*mm355_parallel_hook)(int64_t n, const std::function<void(int64_t)> &f) = 0;
int mm355_run_pack(mm355_ctx *c, int64_t n_reads, const char *const *seqs, const int32_t *lens)
{
	HIPCHK(hipSetDevice(c->dev));
	HostBatch &hb = c->hb;
	hb.n_reads = n_reads; hb.roff.resize(n_reads + 1); hb.rlen.assign(lens, lens + n_reads); hb.order.resize(n_reads);
	int64_t off = 0, bases = 0;
	for (int64_t i = 0; i < n_reads; ++i) { hb.roff[i] = off; off += ((int64_t)lens[i] + 15) / 16 * 16; bases += lens[i]; }
	hb.roff[n_reads] = off; hb.n_bytes = off; hb.n_bases = bases;
	// the packed copy of the reads: reads in parallel on the host pool (mm355_parallel_hook, set by mm355_map.hip), only the padding is filled
	// (this is inside the timed region of the drop-in call: 70 MB per sub-batch were filled with 'N' and then copied by one thread)
	hb.seq.resize((size_t)off + 32);
	memset(&hb.seq[(size_t)off], 'N', 32);
	{
		auto one = [&](int64_t i) {
			const int64_t o = hb.roff[i], e = hb.roff[i + 1];
			const int64_t l = lens[i] > 0? lens[i] : 0;
			if (l) memcpy(&hb.seq[o], seqs[i], (size_t)l);
			if (e > o + l) memset(&hb.seq[o + l], 'N', (size_t)(e - o - l));
		};
		if (mm355_parallel_hook && n_reads >= 256) mm355_parallel_hook(n_reads, one);
		else for (int64_t i = 0; i < n_reads; ++i) one(i);
	}
	std::iota(hb.order.begin(), hb.order.end(), 0);
	std::stable_sort(hb.order.begin(), hb.order.end(), [&](int32_t x, int32_t y) { return hb.rlen[x] > hb.rlen[y]; });
	size_t nb = (size_t)off + 32, nr = (size_t)std::max<int64_t>(n_reads, 1);
	if (c->seq.ensure(nb) || c->roff.ensure((nr + 1) * 8) || c->rlen.ensure(nr * 4) || c->order.ensure(nr * 4)) return MM355_ENOMEM;
	size_t slots = (size_t)off + 64;   // one slot per (padded) base is the worst case for every per-minimizer array
	if (c->mz.ensure(slots * 16) || c->mz_tmp.ensure(slots * 16) || c->n_mz.ensure(nr * 4) || c->sn.ensure(slots * 4) || c->sv.ensure(slots * 8) ||
	    c->sflt.ensure(slots) || c->hl.ensure(slots * 4) || c->soff.ensure(slots * 4) || c->n_a.ensure(nr * 4) || c->rep_len.ensure(nr * 4) ||
	    c->n_mini.ensure(nr * 4) || c->mini_pos.ensure(slots * 8) || c->n_u.ensure(nr * 4) || c->n_v.ensure(nr * 4)) return MM355_ENOMEM;
	HIPCHK(hipMemcpyAsync(c->seq.p, hb.seq.data(), nb, hipMemcpyHostToDevice, c->st));
	HIPCHK(hipMemcpyAsync(c->roff.p, hb.roff.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, c->st));
	if (n_reads) {
		HIPCHK(hipMemcpyAsync(c->rlen.p, hb.rlen.data(), n_reads * 4, hipMemcpyHostToDevice, c->st));
		HIPCHK(hipMemcpyAsync(c->order.p, hb.order.data(), n_reads * 4, hipMemcpyHostToDevice, c->st));
	}
	HIPCHK(hipMemsetAsync(c->counters.p, 0, CTR_BYTES, c->st));
	HIPCHK(hipMemsetAsync(c->err.p, 0, 16, c->st));
	c->stats.n_reads = n_reads; c->stats.n_bases = bases;
	{   // chunk table of the sketch kernel: longest reads first so that a wave holds chunks of similar cost
		const int CH = mm355_sketch_chunk_size();
		std::vector<int32_t> cr, cst; std::vector<int64_t> r0(nr, 0);
		int64_t nc = 0;
		for (int64_t i = 0; i < n_reads; ++i) { r0[i] = nc; nc += (lens[i] + CH - 1) / CH; }
		cr.reserve(nc); cst.reserve(nc);
		for (int64_t i = 0; i < n_reads; ++i) for (int32_t s0 = 0; s0 < lens[i]; s0 += CH) { cr.push_back((int32_t)i); cst.push_back(s0); }
		c->n_chunks = nc;
		if (c->ck_read.ensure((nc + 1) * 4) || c->ck_start.ensure((nc + 1) * 4) || c->ck_n.ensure((nc + 1) * 4) || c->ck_r0.ensure(nr * 8)) return MM355_ENOMEM;
		if (nc) {
			HIPCHK(hipMemcpyAsync(c->ck_read.p, cr.data(), nc * 4, hipMemcpyHostToDevice, c->st));
			HIPCHK(hipMemcpyAsync(c->ck_start.p, cst.data(), nc * 4, hipMemcpyHostToDevice, c->st));
		}
		HIPCHK(hipMemcpyAsync(c->ck_r0.p, r0.data(), nr * 8, hipMemcpyHostToDevice, c->st));
		HIPCHK(mm355_wait_stream(c->st));   // the staging vectors above are about to go out of scope
	}
	return 0;
}

int mm355_run_sketch(mm355_ctx *c)
{
	DevBatch b = dev_batch(c); DevSeeds s = dev_seeds(c);
	{ EvTimer t(c, &c->stats.ms_sketch); mm355_launch_sketch(c->dix, b, s, c->ck_read.as<int32_t>(), c->ck_start.as<int32_t>(), (int)c->n_chunks,
	                                                         c->ck_r0.as<int64_t>(), c->ck_n.as<int32_t>(), c->st, c); }
	HIPCHK(hipGetLastError());
	return 0;
}

static int upload_heavy_order(mm355_ctx *c);
int mm355_run_seeds(mm355_ctx *c, const DevParams &pr)
{
	DevBatch b = dev_batch(c); DevSeeds s = dev_seeds(c);
	HostBatch &hb = c->hb;
	{ EvTimer t(c, &c->stats.ms_seed); mm355_launch_mzflt(pr, b, s, c->st, c); }
	{ EvTimer t(c, &c->stats.ms_seed_lookup); mm355_launch_seed_lookup(c->dix, b, s, c->ck_read.as<int32_t>(), c->ck_start.as<int32_t>(), (int)c->n_chunks,
	                                                                   c->counters.as<unsigned long long>() + CTR_HITS_OFF, (unsigned int*)(c->counters.as<unsigned long long>() + 2), c->st, c); }
	++c->stats.n_launch_seed;
	{ EvTimer t(c, &c->stats.ms_seed); mm355_launch_seed_select(c->dix, pr, b, s, c->st, c); }
	HIPCHK(hipGetLastError());
	int64_t n = hb.n_reads;
	hb.n_mz.resize(n); hb.n_a.resize(n); hb.rep_len.resize(n); hb.n_mini.resize(n); hb.aoff.resize(n + 1);
	if (n) {
		HIPCHK(hipMemcpyAsync(hb.n_mz.data(), c->n_mz.p, n * 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipMemcpyAsync(hb.n_a.data(), c->n_a.p, n * 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipMemcpyAsync(hb.rep_len.data(), c->rep_len.p, n * 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipMemcpyAsync(hb.n_mini.data(), c->n_mini.p, n * 4, hipMemcpyDeviceToHost, c->st));
	}
	unsigned long long ctr[8], hits[CTR_HITS_WORDS];   // (stack targets of asynchronous copies: the stream is synchronised before any return)
	hipError_t ce = hipMemcpyAsync(ctr, c->counters.p, 64, hipMemcpyDeviceToHost, c->st);
	if (ce == hipSuccess) ce = hipMemcpyAsync(hits, c->counters.as<unsigned long long>() + CTR_HITS_OFF, CTR_HITS_WORDS * 8, hipMemcpyDeviceToHost, c->st);
	const hipError_t we = mm355_wait_stream(c->st);
	if (ce != hipSuccess || we != hipSuccess) return MM355_EHIP;
	ctr[0] = 0;
	for (int k = 0; k < CTR_HITS_WORDS; ++k) ctr[0] += hits[k];
	int64_t tot = 0, tmz = 0;
	for (int64_t i = 0; i < n; ++i) { hb.aoff[i] = tot; tot += hb.n_a[i]; tmz += hb.n_mz[i]; }
	hb.aoff[n] = tot; hb.tot_a = tot;
	c->stats.n_mz += tmz; c->stats.n_hit += (int64_t)ctr[0]; c->stats.n_a_multi += (int64_t)ctr[1]; c->stats.n_a += tot;
	size_t na = (size_t)tot + 64;
	if (c->b.cap > c->a.cap) std::swap(c->a, c->b);   // (the sort of the previous batch left its short, culled array in `a`: the long buffer takes the new anchors)
	// a[] and two 8-byte words per anchor (the cull's position words, then chaining scratch) for every anchor; the arrays of the stages behind
	// the sort are sized there, for what the cull leaves (mm355_run_sort: a tenth of the anchors on a GRCh38-scale batch)
	if (c->aoff.ensure((n + 1) * 8) || c->a.ensure(na * 16) || c->z.ensure(na * 8) || c->u.ensure(na * 8)) return MM355_ENOMEM;
	HIPCHK(hipMemcpyAsync(c->aoff.p, hb.aoff.data(), (n + 1) * 8, hipMemcpyHostToDevice, c->st));
	return upload_heavy_order(c);
}

int mm355_run_expand(mm355_ctx *c, const DevParams &pr)
{
	DevBatch b = dev_batch(c); DevSeeds s = dev_seeds(c); DevAnchors a = dev_anchors(c);
	{ EvTimer t(c, &c->stats.ms_seed_expand); mm355_launch_seed_expand(c->dix, pr, b, s, a, c->st, c); }
	HIPCHK(hipGetLastError());
	return 0;
}

static int check_err(mm355_ctx *c)
{
	int e[4] = {0,0,0,0};
	HIPCHK(hipMemcpyAsync(e, c->err.p, 16, hipMemcpyDeviceToHost, c->st));
	HIPCHK(mm355_wait_stream(c->st));
	return e[0]? MM355_ENOMEM : 0;
}

// list levels the literal radix_sort_128x emulation can need for the anchors of this index: the bytes of x = strand << 63 | rid << 32 | rpos that vary
int mm355_sort_levels(const mm355_index *mi)
{
	uint32_t max_len = 1;
	for (uint32_t l : mi->seq_len) if (l > max_len) max_len = l;
	int n = 1;                                            // byte 7: the strand (and rid >> 24)
	for (int sh = 16; sh >= 0; sh -= 8) if (((uint64_t)(mi->n_seq > 0? mi->n_seq - 1 : 0) >> sh) != 0 || sh == 0) ++n;   // bytes 6, 5, 4: rid
	for (int sh = 24; sh >= 0; sh -= 8) if (((uint64_t)(max_len - 1) >> sh) != 0 || sh == 0) ++n;                          // bytes 3 .. 0: rpos
	return n + 1;                                         // (+ 1: slack for the list the last level writes)
}

static int upload_heavy_order(mm355_ctx *c)   // per-read kernels take the reads with the most anchors first: the slowest block starts at t = 0 instead of last
{
	HostBatch &hb = c->hb;
	const int64_t n = hb.n_reads;
	std::vector<int32_t> hv(n);
	std::iota(hv.begin(), hv.end(), 0);
	std::stable_sort(hv.begin(), hv.end(), [&](int32_t x, int32_t y) { return hb.n_a[x] > hb.n_a[y]; });
	c->n_heavy = 0;
	for (int64_t i = 0; i < n && hb.n_a[hv[i]] > mm355_sort_heavy_threshold(); ++i) ++c->n_heavy;
	if (c->heavy.ensure((size_t)(n + 1) * 4)) return MM355_ENOMEM;
	if (n) HIPCHK(hipMemcpyAsync(c->heavy.p, hv.data(), n * 4, hipMemcpyHostToDevice, c->st));
	HIPCHK(mm355_wait_stream(c->st));
	return 0;
}

// cull: anchor-rich batches drop the anchors that cannot chain before they sort (mm355_cullsort.hip); 0 = the full sorted array of every
// read (the stage entry points that hand out row a6's result; MM355_CULL=0 forces it everywhere)
static int ensure_chain_arrays(mm355_ctx *c, int64_t tot)   // everything the stages behind the sort index by anchor
{
	const size_t na = (size_t)tot + 64;
	if (c->f.ensure(na * 4) || c->p.ensure(na * 4) || c->v.ensure(na * 4) || c->z.ensure(na * 8) || c->t8.ensure(na) || c->vi.ensure(na * 4) ||
	    c->b.ensure(na * 16) || c->wk.ensure(na * 16) || c->u.ensure(na * 8) || c->u2.ensure(na * 8)) return MM355_ENOMEM;
	return 0;
}

int mm355_run_sort(mm355_ctx *c, const DevParams &pr, int cull)
{
	DevBatch b = dev_batch(c);
	{
		EvTimer t(c, &c->stats.ms_sort);
		const int n_reads = (int)c->hb.n_reads;
		// anchor-rich batches (GRCh38-scale): cull + per-read LDS sort; only the reads with equal keys among what is left -- where the tie order of
		// the reference's unstable sort is observable -- go through the literal emulation (MM355_FAST_SORT=0/1 forces the choice)
		static const int force = [] { const char *e = getenv("MM355_FAST_SORT"); return e? atoi(e) : -1; }();
		bool fast = force >= 0? force != 0 : (n_reads > 0 && c->hb.tot_a / n_reads >= 2048);
		if (fast && !mm355_cull_sort_fits(c)) fast = false;   // (position + index do not fit a 64-bit word: a 2^40-base reference with 2^22 anchors on a read -- the literal path sorts anything)
		if (fast) {
			const bool cull_env = [] { const char *e = getenv("MM355_CULL"); return !(e && atoi(e) == 0); }();   // (read per call: the tests switch it)
			const bool rmq_primary = (pr.flag & MMF_RMQ) != 0;   // mg_lchain_rmq as the primary chainer sees every anchor (its windows are not mg_lchain_dp's)
			int rc = mm355_cull_sort(c, pr, cull && cull_env && !rmq_primary? 1 : 0);
			if (rc) return rc;
			if ((rc = ensure_chain_arrays(c, c->hb.tot_a))) return rc;
			if ((rc = upload_heavy_order(c))) return rc;
		} else {
			if (ensure_chain_arrays(c, c->hb.tot_a)) return MM355_ENOMEM;   // (the literal emulation's scratch: b, f, p, t8)
			DevAnchors a = dev_anchors(c);
			// every emitted task covers > 64 elements, so tot_a / 64 (+ one whole-array task per read) bounds each list
			const size_t task_cap = (size_t)c->hb.tot_a / 64 + (size_t)c->hb.n_reads + 1024;
			if (c->sort_tasks.ensure(mm355_sort_buf_bytes(task_cap))) return MM355_ENOMEM;
			// whole-array tasks of the reads, by size class: 1024-thread levels, 256-thread levels, one wave
			const int big_min = mm355_sort_heavy_threshold(), med_min = mm355_sort_medium_threshold();
			int nb = 0, nm = 0, ns = 0;
			for (int i = 0; i < n_reads; ++i) {
				const int na = c->hb.n_a[i];
				if (na < 2) continue;
				if (na > big_min) ++nb; else if (na > med_min) ++nm; else ++ns;
			}
			const int n_list = nb + nm + ns;
			if (n_list) {
				if (c->h_tasks.ensure((size_t)n_list * sizeof(SortTask))) return MM355_ENOMEM;   // pinned: consumed by asynchronous copies
				SortTask *ht = (SortTask*)c->h_tasks.p;
				int ib = 0, im = nb, is = nb + nm;
				for (int i = 0; i < n_reads; ++i) {
					const int na = c->hb.n_a[i];
					if (na < 2) continue;
					SortTask tk; tk.read = i; tk.beg = 0; tk.end = (uint32_t)na; tk.s = 56;
					if (na > big_min) ht[ib++] = tk; else if (na > med_min) ht[im++] = tk; else ht[is++] = tk;
				}
				// big tasks first in their list: the longest level walks start at t = 0
				std::stable_sort(ht, ht + nb, [](const SortTask &x, const SortTask &y) { return x.end > y.end; });
				const double ts1 = mm355_now_ms();
				if (mm355_launch_sort(b, a, c->err.as<int>(), ht, nb, nm, ns, (size_t)c->hb.tot_a, c->sort_tasks.p, task_cap, c->st, c, mm355_sort_levels(c->mi))) return MM355_EHIP;
				mm355_trace_add(c, "s:levels", ts1, mm355_now_ms());
			}
		}
	}
	HIPCHK(hipGetLastError());
	const double ts2 = mm355_now_ms();
	const int rce = check_err(c);
	mm355_trace_add(c, "s:tail", ts2, mm355_now_ms());
	return rce;
}

int mm355_run_chain(mm355_ctx *c, const DevParams &pr)
{
	DevBatch b = dev_batch(c); DevAnchors a = dev_anchors(c);
	// chunk table of k_chain_segments: (read, first anchor) of every piece of mm355_chain_chunk() anchors
	const int CH = mm355_chain_chunk();
	size_t n_chunks = 0;
	for (int64_t i = 0; i < c->hb.n_reads; ++i) n_chunks += ((size_t)c->hb.n_a[i] + CH - 1) / CH;
	if (c->h_chunks.ensure(n_chunks * 8 + 8) || c->d_chunks.ensure(n_chunks * 8 + 8)) return MM355_ENOMEM;
	{
		int32_t *hc = (int32_t*)c->h_chunks.p; size_t k = 0;
		for (int64_t i = 0; i < c->hb.n_reads; ++i) for (int32_t c0 = 0; c0 < c->hb.n_a[i]; c0 += CH) { hc[2 * k] = (int32_t)i; hc[2 * k + 1] = c0; ++k; }
		if (n_chunks) HIPCHK(hipMemcpyAsync(c->d_chunks.p, hc, n_chunks * 8, hipMemcpyHostToDevice, c->st));
	}
	// segment lists live in scratch that is free at this point: z (8 B/anchor) and wk (16 B/anchor) hold >= tot_a/2 16-byte entries each
	{ EvTimer t(c, &c->stats.ms_chain); if (mm355_launch_chain(pr, b, a, c->counters.as<unsigned long long>() + CTR_PAIRS_OFF, c->z.p, c->wk.p, (unsigned int*)(c->counters.as<unsigned long long>() + 6), c->d_chunks.p, (int)n_chunks, c->st, c)) return MM355_EHIP; }
	HIPCHK(hipGetLastError());
	return 0;   // (no synchronisation here: the pair counters are statistics, mm355_run_backtrack reads them behind its own)
}

// MM_F_RMQ presets: no device chaining -- every sorted anchor of the read goes to the host chainer (mm355_glue_chain_rmq)
int mm355_run_chain_skip(mm355_ctx *c)
{
	HostBatch &hb = c->hb;
	const int64_t n = hb.n_reads;
	hb.n_u.assign(n, 0); hb.n_v = hb.n_a;
	if (n) {
		HIPCHK(hipMemsetAsync(c->n_u.p, 0, n * 4, c->st));
		HIPCHK(hipMemcpyAsync(c->n_v.p, c->n_a.p, n * 4, hipMemcpyDeviceToDevice, c->st));
	}
	return check_err(c);
}

int mm355_run_backtrack(mm355_ctx *c, const DevParams &pr)
{
	DevBatch b = dev_batch(c); DevAnchors a = dev_anchors(c);
	HostBatch &hb = c->hb;
	{ EvTimer t(c, &c->stats.ms_backtrack); mm355_launch_backtrack(pr, b, a, c->err.as<int>(), c->heavy.as<int32_t>(), (unsigned int*)(c->counters.as<unsigned long long>() + 7), c->st, c); }
	HIPCHK(hipGetLastError());
	int64_t n = hb.n_reads;
	hb.n_u.resize(n); hb.n_v.resize(n);
	if (n) {
		HIPCHK(hipMemcpyAsync(hb.n_u.data(), c->n_u.p, n * 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipMemcpyAsync(hb.n_v.data(), c->n_v.p, n * 4, hipMemcpyDeviceToHost, c->st));
	}
	unsigned long long *pairs = c->pairs_land;   // k_chain_*'s counters, spread over 64 words (slot = block & 63): one word takes ~88 atomics per microsecond
	// (they land in the context, not on this frame: check_err may return before it has synchronised the stream)
	HIPCHK(hipMemcpyAsync(pairs, c->counters.as<unsigned long long>() + CTR_PAIRS_OFF, CTR_PAIRS_WORDS * 8, hipMemcpyDeviceToHost, c->st));
	const int rc = check_err(c);   // (synchronises the stream)
	c->stats.chain_pairs = 0;
	for (int k = 0; k < CTR_PAIRS_WORDS; ++k) c->stats.chain_pairs += (int64_t)pairs[k];
	c->stats.chain_pairs_big = 0;
	for (int k = 32; k < CTR_PAIRS_WORDS; ++k) c->stats.chain_pairs_big += (int64_t)pairs[k];   // (k_chain_big's share)
	return rc;
}

// mg_lchain_rmq on the device (mm355_rmq.hip), as U:map.c::mm_map_frag calls it:
//   pass A (MM_F_RMQ presets only): the primary chainer over all sorted anchors of every read (band width bw);
//   pass B (bw_long > bw, no MM_F_NO_LJOIN): the long-join re-chain of every read that has more than one chain and passes the rescue test
//           (the kernel evaluates the test), band width bw_long.
// Leaves hb.n_u / hb.n_v / hb.rmq_state up to date.  MM355_RMQ_ON_HOST=1: the stage does nothing and the host tail chains as in round 2.
static int rmq_pass(mm355_ctx *c, const RmqParams &rp, const DevParams &pr, int nl, std::vector<uint8_t> &state)
{
	HostBatch &hb = c->hb;
	const int64_t n = hb.n_reads;
	int32_t *hl = (int32_t*)c->h_rmq.p; uint8_t *hf = (uint8_t*)c->h_rmq.p + (size_t)n * 4;
	state.assign(n, MM355_RMQ_KEEP);
	if (nl == 0) return 0;
	// the reads with the most anchors first: the longest dependence chain starts at t = 0
	std::stable_sort(hl, hl + nl, [&](int32_t x, int32_t y) { return (rp.primary? hb.n_a[x] > hb.n_a[y] : hb.n_v[x] > hb.n_v[y]); });
	if (!rp.primary) for (int k = 0; k < nl; ++k) c->stats.n_v_rmq += hb.n_v[hl[k]];   // (before the pass rewrites n_v)
	DevBatch b = dev_batch(c); DevAnchors a = dev_anchors(c);
	HIPCHK(hipMemcpyAsync(c->rmq_list.p, hl, (size_t)nl * 4, hipMemcpyHostToDevice, c->st));
	HIPCHK(hipMemsetAsync(c->rmq_flag.p, rp.primary? MM355_RMQ_DONE : MM355_RMQ_KEEP, (size_t)n, c->st));
	unsigned long long *ctr = c->counters.as<unsigned long long>() + CTR_RMQ_OFF;
	HIPCHK(hipMemsetAsync(ctr, 0, CTR_RMQ_WORDS * 8, c->st));
	{ EvTimer t(c, &c->stats.ms_rmq);
	  if (mm355_launch_rmq(rp, pr, b, a, c->rmq_list.as<int32_t>(), nl, c->rmq_flag.as<uint8_t>(), c->err.as<int>(), ctr, (unsigned int*)(c->counters.as<unsigned long long>() + 8), c->st, c)) return MM355_EHIP; }
	// (the copies below land in a stack array and in pageable vectors: no return between the first of them and the synchronisation in
	// check_err -- a failed call is remembered and reported behind it.  A pass that bails out leaves f / p / v / u2 / z partly rewritten: the
	// host fallback re-chains from a[] alone, the only array a HOST / HOST_ALL read relies on)
	unsigned long long hc[CTR_RMQ_WORDS];
	hipError_t ce = hipMemcpyAsync(hf, c->rmq_flag.p, (size_t)n, hipMemcpyDeviceToHost, c->st);
	if (ce == hipSuccess) ce = hipMemcpyAsync(hb.n_u.data(), c->n_u.p, n * 4, hipMemcpyDeviceToHost, c->st);
	if (ce == hipSuccess) ce = hipMemcpyAsync(hb.n_v.data(), c->n_v.p, n * 4, hipMemcpyDeviceToHost, c->st);
	if (ce == hipSuccess) ce = hipMemcpyAsync(hc, ctr, CTR_RMQ_WORDS * 8, hipMemcpyDeviceToHost, c->st);
	int rc = check_err(c);   // (synchronises the stream)
	if (ce != hipSuccess) return MM355_EHIP;
	if (rc) return rc;
	for (int k = 0; k < nl; ++k) c->stats.n_v_rmq += rp.primary? hb.n_a[hl[k]] : 0;
	for (int64_t i = 0; i < n; ++i) {
		state[i] = hf[i];
		if (hf[i] == MM355_RMQ_DONE) ++c->stats.n_rmq_reads; else if (hf[i] == MM355_RMQ_HOST) ++c->stats.n_rmq_host;
	}
	for (int k = 0; k < CTR_RMQ_WORDS; ++k) c->stats.rmq_scanned += (int64_t)hc[k];
	return 0;
}

int mm355_run_rmq(mm355_ctx *c, const mm355_mapopt_t *mo, const DevParams &pr)
{
	HostBatch &hb = c->hb;
	const int64_t n = hb.n_reads;
	hb.rmq_state.clear();
	const bool on_host = [] { const char *e = getenv("MM355_RMQ_ON_HOST"); return e && atoi(e) != 0; }();   // (read per call: the tests switch it)
	const bool primary = (mo->flag & MMF_RMQ) != 0;
	const bool ljoin = mo->bw_long > mo->bw && (mo->flag & (MMF_SPLICE | MMF_SR | MMF_NO_LJOIN)) == 0;
	if (on_host || n == 0) return 0;
	if (!primary && !ljoin) { hb.rmq_state.assign(n, MM355_RMQ_KEEP); return 0; }
	RmqParams rp; memset(&rp, 0, sizeof(rp));
	rp.max_dist = mo->max_gap; rp.max_dist_inner = mo->rmq_inner_dist; rp.max_chn_skip = mo->max_chain_skip;
	rp.cap = mo->rmq_size_cap; rp.pen_gap = pr.pen_gap; rp.pen_skip = pr.pen_skip; rp.rescue_size = mo->rmq_rescue_size; rp.rescue_ratio = mo->rmq_rescue_ratio;
	if (c->h_rmq.ensure((size_t)n * 4 + (size_t)n + 64) || c->rmq_list.ensure((size_t)n * 4 + 64) || c->rmq_flag.ensure((size_t)n + 64)) return MM355_ENOMEM;
	int32_t *hl = (int32_t*)c->h_rmq.p;
	std::vector<uint8_t> st_a, st_b;
	int rc;
	if (primary) {
		int nl = 0;
		for (int64_t i = 0; i < n; ++i) if (hb.n_a[i] > 0) hl[nl++] = (int32_t)i;
		rp.primary = 1; rp.bw = mo->bw;
		if ((rc = rmq_pass(c, rp, pr, nl, st_a))) return rc;
	}
	if (ljoin) {
		int nl = 0;
		for (int64_t i = 0; i < n; ++i) if (hb.n_u[i] > 1 && (!primary || st_a[i] == MM355_RMQ_DONE)) hl[nl++] = (int32_t)i;
		rp.primary = 0; rp.bw = mo->bw_long;
		if ((rc = rmq_pass(c, rp, pr, nl, st_b))) return rc;
	} else st_b.assign(n, MM355_RMQ_KEEP);
	hb.rmq_state.resize(n);
	for (int64_t i = 0; i < n; ++i)
		hb.rmq_state[i] = primary && st_a[i] == MM355_RMQ_HOST? MM355_RMQ_HOST_ALL : primary && st_a[i] == MM355_RMQ_KEEP && hb.n_a[i] > 0? MM355_RMQ_HOST_ALL : st_b[i];
	return 0;
}

// ------------------------------------------------------------------ per-stage C-ABI (tests + bench)
static int stage_prologue(mm355_ctx *c, const mm355_mapopt_t *mo, int64_t n_reads, const char *const *seqs, const int32_t *lens, DevParams *pr)
{
	if (c == 0) return MM355_EINVAL;
	if (mo) { int rc = mm355_check_opts(mo, c->mi); if (rc) return rc; *pr = mm355_make_params(mo, c->mi); }
	c->n_tpend = 0; memset(&c->stats, 0, sizeof(c->stats));
	return mm355_run_pack(c, n_reads, seqs, lens);
}

extern "C" int mm355_stage_sketch(mm355_ctx_t *c, int64_t n_reads, const char *const *seqs, const int32_t *lens, int64_t *mz_off, uint64_t *mz, int64_t mz_cap)
{
	DevParams pr; int rc;
	if ((rc = stage_prologue(c, 0, n_reads, seqs, lens, &pr))) return rc;
	if ((rc = mm355_run_sketch(c))) return rc;
	std::vector<int32_t> n_mz(n_reads);
	if (n_reads) HIPCHK(hipMemcpyAsync(n_mz.data(), c->n_mz.p, n_reads * 4, hipMemcpyDeviceToHost, c->st));
	HIPCHK(mm355_wait_stream(c->st));
	int64_t tot = 0;
	for (int64_t i = 0; i < n_reads; ++i) { mz_off[i] = tot; tot += n_mz[i]; }
	mz_off[n_reads] = tot;
	if (tot > mz_cap) return MM355_ENOMEM;
	for (int64_t i = 0; i < n_reads; ++i)
		if (n_mz[i]) HIPCHK(hipMemcpyAsync(mz + mz_off[i] * 2, c->mz.as<mm128>() + c->hb.roff[i], (size_t)n_mz[i] * 16, hipMemcpyDeviceToHost, c->st));
	HIPCHK(mm355_wait_stream(c->st));
	return 0;
}

extern "C" int mm355_stage_anchors(mm355_ctx_t *c, const mm355_mapopt_t *mo, int64_t n_reads, const char *const *seqs, const int32_t *lens, int sorted,
                                   int64_t *a_off, uint64_t *a, int64_t a_cap, int32_t *rep_len, int32_t *n_mini_pos)
{
	DevParams pr; int rc;
	if ((rc = stage_prologue(c, mo, n_reads, seqs, lens, &pr))) return rc;
	if ((rc = mm355_run_sketch(c))) return rc;
	if ((rc = mm355_run_seeds(c, pr))) return rc;
	if ((rc = mm355_run_expand(c, pr))) return rc;
	if (sorted && (rc = mm355_run_sort(c, pr, sorted == 2? 1 : 0))) return rc;
	HostBatch &hb = c->hb;
	for (int64_t i = 0; i <= n_reads; ++i) a_off[i] = hb.aoff[i];
	if (hb.tot_a > a_cap) return MM355_ENOMEM;
	if (hb.tot_a) HIPCHK(hipMemcpyAsync(a, c->a.p, (size_t)hb.tot_a * 16, hipMemcpyDeviceToHost, c->st));
	HIPCHK(mm355_wait_stream(c->st));
	for (int64_t i = 0; i < n_reads; ++i) { if (rep_len) rep_len[i] = hb.rep_len[i]; if (n_mini_pos) n_mini_pos[i] = hb.n_mini[i]; }
	return 0;
}

extern "C" int mm355_stage_chain(mm355_ctx_t *c, const mm355_mapopt_t *mo, int64_t n_reads, const char *const *seqs, const int32_t *lens,
                                 int64_t *a_off, uint64_t *a, int32_t *f, int32_t *p, int32_t *v, int64_t a_cap)
{
	DevParams pr; int rc;
	if ((rc = stage_prologue(c, mo, n_reads, seqs, lens, &pr))) return rc;
	if ((rc = mm355_run_sketch(c))) return rc;
	if ((rc = mm355_run_seeds(c, pr))) return rc;
	if ((rc = mm355_run_expand(c, pr))) return rc;
	if ((rc = mm355_run_sort(c, pr, 0))) return rc;   // (f / p / v are compared anchor by anchor with the oracle: the whole sorted array)
	if ((rc = mm355_run_chain(c, pr))) return rc;
	HostBatch &hb = c->hb;
	for (int64_t i = 0; i <= n_reads; ++i) a_off[i] = hb.aoff[i];
	if (hb.tot_a > a_cap) return MM355_ENOMEM;
	if (hb.tot_a) {
		HIPCHK(hipMemcpyAsync(a, c->a.p, (size_t)hb.tot_a * 16, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipMemcpyAsync(f, c->f.p, (size_t)hb.tot_a * 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipMemcpyAsync(p, c->p.p, (size_t)hb.tot_a * 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipMemcpyAsync(v, c->v.p, (size_t)hb.tot_a * 4, hipMemcpyDeviceToHost, c->st));
	}
	HIPCHK(mm355_wait_stream(c->st));
	return 0;
}

extern "C" int mm355_stage_chains(mm355_ctx_t *c, const mm355_mapopt_t *mo, int64_t n_reads, const char *const *seqs, const int32_t *lens,
                                  int64_t *u_off, uint64_t *u, int64_t u_cap, int64_t *a_off, uint64_t *a, int64_t a_cap)
{
	DevParams pr; int rc;
	if ((rc = stage_prologue(c, mo, n_reads, seqs, lens, &pr))) return rc;
	if ((rc = mm355_run_sketch(c))) return rc;
	if ((rc = mm355_run_seeds(c, pr))) return rc;
	if ((rc = mm355_run_expand(c, pr))) return rc;
	if ((rc = mm355_run_sort(c, pr))) return rc;
	if ((rc = mm355_run_chain(c, pr))) return rc;
	if ((rc = mm355_run_backtrack(c, pr))) return rc;
	HostBatch &hb = c->hb;
	int64_t tu = 0, tv = 0;
	for (int64_t i = 0; i < n_reads; ++i) { u_off[i] = tu; a_off[i] = tv; tu += hb.n_u[i]; tv += hb.n_v[i]; }
	u_off[n_reads] = tu; a_off[n_reads] = tv;
	if (tu > u_cap || tv > a_cap) return MM355_ENOMEM;
	for (int64_t i = 0; i < n_reads; ++i) {
		if (hb.n_u[i]) HIPCHK(hipMemcpyAsync(u + u_off[i], c->u.as<uint64_t>() + hb.aoff[i], (size_t)hb.n_u[i] * 8, hipMemcpyDeviceToHost, c->st));
		if (hb.n_v[i]) HIPCHK(hipMemcpyAsync(a + a_off[i] * 2, c->a.as<mm128>() + hb.aoff[i], (size_t)hb.n_v[i] * 16, hipMemcpyDeviceToHost, c->st));
	}
	HIPCHK(mm355_wait_stream(c->st));
	return 0;
}

extern "C" int mm355_stage_rmq(mm355_ctx_t *c, const mm355_mapopt_t *mo, int64_t n_reads, const char *const *seqs, const int32_t *lens,
                               int64_t *u_off, uint64_t *u, int64_t u_cap, int64_t *a_off, uint64_t *a, int64_t a_cap, int32_t *state)
{
	DevParams pr; int rc;
	if ((rc = stage_prologue(c, mo, n_reads, seqs, lens, &pr))) return rc;
	if ((rc = mm355_run_sketch(c))) return rc;
	if ((rc = mm355_run_seeds(c, pr))) return rc;
	if ((rc = mm355_run_expand(c, pr))) return rc;
	if ((rc = mm355_run_sort(c, pr))) return rc;
	if (mo->flag & MMF_RMQ) { if ((rc = mm355_run_chain_skip(c))) return rc; }
	else {
		if ((rc = mm355_run_chain(c, pr))) return rc;
		if ((rc = mm355_run_backtrack(c, pr))) return rc;
	}
	if ((rc = mm355_run_rmq(c, mo, pr))) return rc;
	HostBatch &hb = c->hb;
	int64_t tu = 0, tv = 0;
	for (int64_t i = 0; i < n_reads; ++i) {
		const int st = hb.rmq_state.empty()? MM355_RMQ_HOST_ALL : hb.rmq_state[i];
		state[i] = st;
		u_off[i] = tu; a_off[i] = tv; tu += st >= MM355_RMQ_HOST? 0 : hb.n_u[i]; tv += hb.n_v[i];
	}
	u_off[n_reads] = tu; a_off[n_reads] = tv;
	if (tu > u_cap || tv > a_cap) return MM355_ENOMEM;
	for (int64_t i = 0; i < n_reads; ++i) {
		if (u_off[i + 1] > u_off[i]) HIPCHK(hipMemcpyAsync(u + u_off[i], c->u.as<uint64_t>() + hb.aoff[i], (size_t)(u_off[i + 1] - u_off[i]) * 8, hipMemcpyDeviceToHost, c->st));
		if (hb.n_v[i]) HIPCHK(hipMemcpyAsync(a + a_off[i] * 2, c->a.as<mm128>() + hb.aoff[i], (size_t)hb.n_v[i] * 16, hipMemcpyDeviceToHost, c->st));
	}
	HIPCHK(mm355_wait_stream(c->st));
	return 0;
}
