// mm355_map.hip -- mm355_map_batch: the drop-in for the reference's per-read mm_map call
// (/root/reference/src/lib.rs:482-488 single read, :587-593 batch worker) over a whole batch of reads.
//   device : sketch -> mz_flt -> seed lookup -> seed select -> anchor expansion -> radix sort -> chaining DP ->
//            backtrack/compact -> [host: regions] -> banded extension in rounds (all pending problems of all reads
//            per launch) -> [host: stitch, select, MAPQ, cs]
//   host   : the O(#chains) sequential tail of mm355_glue.cpp on a few threads, never any per-base loop of the hot path.
// Reads are independent, so a node shards them over its GPUs with one context per GPU and no collective.
#include <stdio.h>
#include <stdlib.h>
#include <sched.h>
#include <time.h>
#include <string.h>
#include <thread>
#include <atomic>
#include <chrono>
#include <algorithm>
#include "mm355_pipeline.h"
#include "mm355_dp.h"
#include "mm355_glue.h"
#include "mm355_prof.h"

// packs u[], compacted anchors and mini_pos[] of all reads into three dense arrays (one D2H copy each)
__global__ __launch_bounds__(256) void k_pack_chains(int n_reads, const int64_t *aoff, const int64_t *roff, const int32_t *n_u, const int32_t *n_v, const int32_t *n_mini,
                                                     const int64_t *uo, const int64_t *vo, const int64_t *mo, const uint64_t *u, const mm128 *a, const uint64_t *mini_pos,
                                                     uint64_t *pu, mm128 *pa, uint64_t *pm)
{
	const int r = blockIdx.x;
	if (r >= n_reads) return;
	const int64_t ao = aoff[r], ro = roff[r];
	for (int i = threadIdx.x; i < n_u[r]; i += 256) pu[uo[r] + i] = u[ao + i];
	for (int i = threadIdx.x; i < n_v[r]; i += 256) pa[vo[r] + i] = a[ao + i];
	for (int i = threadIdx.x; i < n_mini[r]; i += 256) pm[mo[r] + i] = mini_pos[ro + i];
}

// One process-wide pool of host threads shared by every context: while some contexts wait for their kernels the others
// get all the cores for the host tail (a per-context pool would leave cores idle whenever contexts are out of phase).
#include <mutex>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
struct PJob { std::atomic<int64_t> next{0}; int64_t n = 0; std::function<void(int64_t, int)> f; std::atomic<int> pending{0}; std::atomic<int> tid{1};
              std::atomic<int64_t> helper_cpu_ns{0};         // CPU time the helper threads spent inside this job (the caller's own share is in its thread clock)
              std::mutex dm; std::condition_variable dcv; };   // the caller sleeps on dcv until the helpers that started have finished
static inline int64_t thread_cpu_ns() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (int64_t)ts.tv_sec * 1000000000LL + ts.tv_nsec; }
class HostPool {
public:
	// intentionally leaked: destroying a condition variable with parked workers at process exit would block in pthread_cond_destroy
	static HostPool &get() { static HostPool *p = new HostPool(); return *p; }
	int size() const { return (int)th.size() + 1; }
	// returns the CPU nanoseconds the helper threads spent in the job
	int64_t run(int64_t n, int want_threads, std::function<void(int64_t, int)> f) {
		if (n <= 0) return 0;
		int helpers = std::min<int64_t>((int64_t)std::min(want_threads, size()) - 1, n - 1);
		if (helpers <= 0) { for (int64_t i = 0; i < n; ++i) f(i, 0); return 0; }
		auto j = std::make_shared<PJob>();
		j->n = n; j->f = std::move(f); j->pending = helpers;
		{ std::lock_guard<std::mutex> lk(m); for (int h = 0; h < helpers; ++h) q.push_back(j); }
		cv.notify_all();
		for (;;) { int64_t i = j->next.fetch_add(1); if (i >= n) break; j->f(i, 0); }
		{   // withdraw tickets nobody picked up, then wait for the helpers that did start
			std::lock_guard<std::mutex> lk(m);
			for (auto it = q.begin(); it != q.end();) { if (it->get() == j.get()) { it = q.erase(it); j->pending.fetch_sub(1); } else ++it; }
		}
		for (int spin = 0; spin < 200 && j->pending.load(std::memory_order_acquire) > 0; ++spin) std::this_thread::yield();   // the usual case: a few microseconds
		if (j->pending.load(std::memory_order_acquire) > 0) {   // a helper is inside a long item: sleep instead of burning the CPU share
			std::unique_lock<std::mutex> lk(j->dm);
			j->dcv.wait(lk, [&]() { return j->pending.load(std::memory_order_acquire) <= 0; });
		}
		return j->helper_cpu_ns.load();
	}
private:
	HostPool() {
		// Default: the CPUs this process may use (cgroup quota, else the affinity mask) minus two for the context threads and the HIP runtime,
		// at most 32.  The path is host-bound on a 16-core share: fewer threads leave the GPU waiting, more than the
		// quota gets the whole cgroup throttled (measured on a 16-CPU quota: 10 threads 790, 14: 860, 16: 845, 20: 747 Mbases/s).
		const char *e = getenv("MM355_HOST_THREADS");
		int n = 14;
		if (e) n = atoi(e);
		else {
			double cpus = 0;
			cpu_set_t cs; CPU_ZERO(&cs);
			if (sched_getaffinity(0, sizeof(cs), &cs) == 0) cpus = (double)CPU_COUNT(&cs);
			if (FILE *fp = fopen("/sys/fs/cgroup/cpu.max", "r")) {
				char qb[64]; double per = 0;
				if (fscanf(fp, "%63s %lf", qb, &per) == 2 && strcmp(qb, "max") != 0 && per > 0) { const double qv = atof(qb) / per; if (cpus <= 0 || qv < cpus) cpus = qv; }
				fclose(fp);
			}
			if (cpus > 0) n = (int)cpus - 2;
			if (n > 32) n = 32;
			if (n < 4) n = 4;
		}
		if (n < 1) n = 1;
		if (n > 64) n = 64;
		for (int i = 1; i < n; ++i) th.emplace_back([this]() { worker(); });
		for (auto &t : th) t.detach();
	}
	void worker() {
		for (;;) {
			std::shared_ptr<PJob> j;
			{ std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&]() { return !q.empty(); }); j = q.front(); q.pop_front(); }
			const int tid = j->tid.fetch_add(1);
			const int64_t c0 = thread_cpu_ns();
			for (;;) { int64_t i = j->next.fetch_add(1); if (i >= j->n) break; j->f(i, tid); }
			j->helper_cpu_ns.fetch_add(thread_cpu_ns() - c0);
			if (j->pending.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> lk(j->dm); j->dcv.notify_all(); }
		}
	}
	std::vector<std::thread> th; std::mutex m; std::condition_variable cv; std::deque<std::shared_ptr<PJob>> q;
};

static thread_local int64_t tl_helper_cpu_ns = 0;   // helper CPU time of the jobs this (context) thread has issued
template <typename F>
static void parallel_for(int64_t n, int n_threads, F f)
{
	tl_helper_cpu_ns += HostPool::get().run(n, n_threads, std::function<void(int64_t, int)>(f));
}

// MM355_TRACE=<file>: host-side phase timeline (context, phase, start ms, end ms), written at process exit
struct TraceEv { const void *ctx; const char *phase; double t0, t1; };
static std::mutex g_trace_mu;
static std::vector<TraceEv> g_trace;
static const char *g_trace_path = getenv("MM355_TRACE");
static void trace_dump()
{
	FILE *fp = fopen(g_trace_path, "w");
	if (!fp) return;
	for (const TraceEv &e : g_trace) fprintf(fp, "%p\t%s\t%.3f\t%.3f\n", e.ctx, e.phase, e.t0, e.t1);
	fclose(fp);
}
void mm355_trace_add(const void *ctx, const char *phase, double t0, double t1)
{
	if (!g_trace_path) return;
	std::lock_guard<std::mutex> lk(g_trace_mu);
	if (g_trace.empty()) atexit(trace_dump);
	g_trace.push_back(TraceEv{ctx, phase, t0, t1});
}
static void trace_add(const void *ctx, const char *phase, double t0, double t1) { mm355_trace_add(ctx, phase, t0, t1); }
hipError_t mm355_wait_stream(hipStream_t st)
{
	// Default (MM355_BLOCKING_WAIT unset or 2): poll hipStreamQuery -- a few hundred quick queries (waits of a fraction of a millisecond
	// stay low-latency), then naps of 40 us.  hipStreamSynchronize (=0) spins on a core for the whole wait: with 8 context threads that is
	// half of a 16-CPU share burnt while the shared host pool needs it (+7 % throughput on the bench boxes, less run-to-run spread);
	// a blocking-sync event (=1) sleeps in the driver and wakes too slowly for the ~15 short waits of a sub-batch (-5 %).
	static const int mode = [] { const char *e = getenv("MM355_BLOCKING_WAIT"); return e? atoi(e) : 2; }();
	if (mode == 0) return hipStreamSynchronize(st);
	if (mode == 2) {
		static const int spins = [] { const char *e = getenv("MM355_WAIT_SPINS"); return e? atoi(e) : 300; }();
		static const int nap = [] { const char *e = getenv("MM355_WAIT_NAP_US"); return e? atoi(e) : 40; }();
		for (int i = 0;; ++i) {
			hipError_t q = hipStreamQuery(st);
			if (q == hipSuccess) return hipSuccess;
			if (q != hipErrorNotReady) return q;
			if (i < spins) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(nap));
		}
	}
	static thread_local hipEvent_t ev = 0;   // one per calling thread (leaked with it); valid for any stream of the current device
	static thread_local int ev_dev = -1;
	int dev = 0; (void)hipGetDevice(&dev);
	if (ev == 0 || ev_dev != dev) { hipError_t e = hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming); if (e != hipSuccess) return e; ev_dev = dev; }
	hipError_t e = hipEventRecord(ev, st);
	if (e != hipSuccess) return e;
	return hipEventSynchronize(ev);
}
double mm355_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int host_threads() { return HostPool::get().size(); }

static int host_threads();
// Deferred destruction of a batch's host state: one background thread frees what the mapping threads are done with (the process never joins
// it: it only ever frees memory).  At most a few batches are pending; bury() frees synchronously when the queue is long (memory bound).
struct Reaper {
	struct Item { virtual ~Item() {} };
	std::mutex m; std::condition_variable cv; std::vector<Item*> q;
	static Reaper &get() { static Reaper *r = [] { Reaper *x = new Reaper(); std::thread([x] { x->loop(); }).detach(); return x; }(); return *r; }
	void bury(Item *it) {
		{ std::unique_lock<std::mutex> lk(m); if (q.size() < 16) { q.push_back(it); it = 0; } }
		if (it) delete it; else cv.notify_one();
	}
	void loop() {
		for (;;) {
			std::vector<Item*> mine;
			{ std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return !q.empty(); }); mine.swap(q); }
			for (Item *it : mine) delete it;
		}
	}
};

static int run_dp_round(mm355_ctx *c, const mm355_mapopt_t *mo, std::vector<ReadState> &rs, const std::vector<DpReq> &reqs, std::vector<std::vector<uint32_t>> &arenas)
{
	// chunk the requests so that the direction matrices of one launch fit the HBM budget
	size_t free_b = 0, total_b = 0;
	(void)hipMemGetInfo(&free_b, &total_b);
	size_t budget = std::min<size_t>((size_t)24 << 30, free_b / 3);
	if (budget < ((size_t)256 << 20)) budget = (size_t)256 << 20;
	// what this context already holds for direction matrices costs nothing to use again: with six contexts' buffers resident the free
	// third alone is below one round's matrices, and a round cut in two is two launch cycles (two turns, twice the tails)
	if (c->dp_bt.cap > 64 && budget < c->dp_bt.cap - 64) budget = c->dp_bt.cap - 64;
	if (const char *e = getenv("MM355_DP_BUDGET_MB")) { if (atoll(e) > 0) budget = (size_t)atoll(e) << 20; }   // test hook (read per call): forces rounds to be cut
	const bool verbose = getenv("MM355_VERBOSE") != 0;
	const int nt = host_threads();
	std::vector<int64_t> qo, to;
	const DpConst dpc = mm355_dp_const(mo);
	size_t i = 0;
	while (i < reqs.size()) {
		const double tb0 = now_ms();
		// serial part: chunk end and the offsets of the code strings (a prefix sum); everything else is filled in parallel
		size_t q_tot = 0, t_tot = 0, p_tot = 0, j = i;
		qo.clear(); to.clear();
		for (; j < reqs.size(); ++j) {
			const DpReq &q = reqs[j];
			const size_t pb = mm355_dp_matrix_bytes(mo, dpc, q.qlen, q.tlen, q.w, q.flag);   // (as mm355_dp_run lays it out: tiled for the row sweep)
			if (j > i && p_tot + pb > budget) break;
			qo.push_back((int64_t)q_tot); to.push_back((int64_t)t_tot);
			q_tot += (size_t)(q.qlen > 0? q.qlen : 0) + 16; t_tot += (size_t)(q.tlen > 0? q.tlen : 0) + 16; p_tot += pb;
		}
		const size_t n = j - i;
		// a round whose direction matrices do not fit the budget is cut into several launch cycles (several turns, several tails): it costs
		// rate, silently -- so it is counted (mm355_stats_t::n_rounds_split; bench.py prints it, 0 on the GRCh38-scale workload)
		if (j < reqs.size() && i == 0) ++c->stats.n_rounds_split;
		if (c->h_gather.ensure(n * sizeof(DpGather)) || c->h_jobs.ensure(n * sizeof(DpJobDev))) return MM355_ENOMEM;
		DpGather *g = (DpGather*)c->h_gather.p; DpJobDev *jobs = (DpJobDev*)c->h_jobs.p;
		parallel_for(nt, nt, [&](int64_t part, int) {
			const size_t lo = n * (size_t)part / nt, hi = n * (size_t)(part + 1) / nt;
			for (size_t k = lo; k < hi; ++k) {
				const DpReq &q = reqs[i + k];
				DpGather gg; memset(&gg, 0, sizeof(gg));
				gg.qlen = q.qlen > 0? q.qlen : 0; gg.tlen = q.tlen > 0? q.tlen : 0;
				gg.qoff = qo[k]; gg.toff = to[k];
				gg.q_src = 2 * c->hb.roff[q.read] + (q.rev_strand? rs[q.read].qlen : 0) + q.q_st;
				gg.rid = q.rid; gg.t_st = q.t_st; gg.rev = q.reversed;
				g[k] = gg;
				DpJobDev jd; memset(&jd, 0, sizeof(jd));
				jd.qlen = q.qlen; jd.tlen = q.tlen; jd.qoff = gg.qoff; jd.toff = gg.toff; jd.w = q.w; jd.zdrop = q.zdrop; jd.end_bonus = q.end_bonus; jd.flag = q.flag;
				jobs[k] = jd;
			}
		});
		const double tg0 = now_ms();
		int rc = mm355_dp_gather(c, g, n, q_tot, t_tot);
		if (rc) return rc;
		const double tg1 = now_ms();
		const mm355_dpres_t *res = 0; const uint32_t *cig = 0;
		// the dense CIGAR arena of this launch stays alive until the batch is finished (results point into it): one pinned buffer per
		// launch, heap copies beyond the eighth
		HBuf *ab = c->n_arena < 8? &c->h_arena[c->n_arena] : &c->h_cig;
		rc = mm355_dp_run(c, mo, jobs, n, c->dp_q.as<uint8_t>(), c->dp_t.as<uint8_t>(), ab, &res, &cig);
		if (rc) return rc;
		const double tr1 = now_ms();
		const uint32_t *arena = cig;
		if (c->n_arena < 8) ++c->n_arena;
		else {
			size_t n_cg = 0;
			for (size_t k = 0; k < n; ++k) n_cg = std::max(n_cg, (size_t)(res[k].cigar_off + res[k].n_cigar));
			arenas.emplace_back(cig, cig + n_cg);
			arena = arenas.back().data();
		}
		parallel_for(nt, nt, [&](int64_t part, int) {   // static partition; every request owns a distinct (read, task, slot)
			const size_t lo = i + n * (size_t)part / nt, hi = i + n * (size_t)(part + 1) / nt;
			for (size_t k = lo; k < hi; ++k) {
				const DpReq &q = reqs[k];
				const mm355_dpres_t &r = res[k - i];
				AlnTask &T = rs[q.read].tasks[q.task];
				EzRes &e = q.slot >= 0? T.res[q.slot] : T.inv_res;
				e.max = r.max; e.zdropped = r.zdropped; e.max_q = r.max_q; e.max_t = r.max_t; e.mqe = r.mqe; e.mqe_t = r.mqe_t;
				e.mte = r.mte; e.mte_q = r.mte_q; e.score = r.score; e.reach_end = r.reach_end;
				e.cigar = arena + r.cigar_off; e.n_cigar = r.n_cigar;
				e.state = 2;
			}
		});
		if (verbose) fprintf(stderr, "[mm355]     dp chunk: %zu jobs, build %.1f ms, gather %.1f ms, run %.1f ms, distribute %.1f ms\n", n, tg0 - tb0, tg1 - tg0, tr1 - tg1, now_ms() - tr1);
		i = j;
	}
	return 0;
}

// Several batches can be resident in one context: select(slot) parks the current batch (packed reads + tables, device and host
// side) and makes the batch of `slot` current (an empty one the first time).  Working buffers are shared; they only grow.
extern "C" int mm355_batch_select(mm355_ctx_t *c, int slot)
{
	if (c == 0 || slot < 0 || slot >= 64) return MM355_EINVAL;
	if (slot == c->cur_slot) return 0;
	const size_t need = (size_t)std::max(slot, c->cur_slot) + 1;
	if (c->slots.size() < need) c->slots.resize(need);
	auto exchange = [&](ResidentBatch &r) {
		std::swap(r.hb, c->hb); std::swap(r.seq, c->seq); std::swap(r.roff, c->roff); std::swap(r.rlen, c->rlen); std::swap(r.order, c->order);
		std::swap(r.ck_read, c->ck_read); std::swap(r.ck_start, c->ck_start); std::swap(r.ck_r0, c->ck_r0); std::swap(r.n_chunks, c->n_chunks);
	};
	exchange(c->slots[c->cur_slot]);   // park the current batch in its own slot
	exchange(c->slots[slot]);          // and take the requested one
	c->cur_slot = slot;
	return 0;
}

static const bool g_parallel_hook_set = [] { mm355_parallel_hook = [](int64_t n, const std::function<void(int64_t)> &f) { parallel_for(n, host_threads(), [&](int64_t i, int) { f(i); }); }; return true; }();

extern "C" int mm355_batch_upload(mm355_ctx_t *c, int64_t n_reads, const char *const *seqs, const int32_t *lens)
{
	if (c == 0 || n_reads < 0) return MM355_EINVAL;
	std::vector<int32_t> dl(lens, lens + n_reads);
	c->hb.status.assign(n_reads, 0);
	for (int64_t i = 0; i < n_reads; ++i)
		if (lens[i] <= 0) { c->hb.status[i] = MM355_EEMPTY; dl[i] = 0; }   // "Sequence is empty" (L2 crate)
	return mm355_run_pack(c, n_reads, seqs, dl.data());
}

extern "C" int mm355_map_batch(mm355_ctx_t *c, const mm355_mapopt_t *mo, int64_t n_reads, const char *const *seqs, const int32_t *lens, int flags, mm355_hits_t **out)
{
	*out = 0;
	int rc = mm355_batch_upload(c, n_reads, seqs, lens);
	if (rc) return rc;
	return mm355_map_resident(c, mo, flags, out);
}

// maps the batch that mm355_batch_upload left resident in HBM (bench.py times this call: inputs already on the device)
extern "C" int mm355_map_resident(mm355_ctx_t *c, const mm355_mapopt_t *mo, int flags, mm355_hits_t **out)
{
	*out = 0;
	if (c == 0 || mo == 0) return MM355_EINVAL;
	if (c->mi == 0) return MM355_ENOIDX;
	int rc = mm355_check_opts(mo, c->mi);
	if (rc) return rc;
	HIPCHK(hipSetDevice(c->dev));
	const double t_start = now_ms();
	const int64_t cpu_start = thread_cpu_ns(); tl_helper_cpu_ns = 0;
	const mm355_index *mi = c->mi;
	DevParams pr = mm355_make_params(mo, mi);
	const int64_t n_reads = c->hb.n_reads;
	c->n_tpend = 0; memset(&c->stats, 0, sizeof(c->stats));
	{ static const int tm = [] { const char *e = getenv("MM355_TIMERS"); return e? atoi(e) : -1; }(); c->timers_on = tm > 0 || (tm < 0 && n_reads >= 16); }
	c->n_arena = 0;
	c->stats.n_reads = n_reads; c->stats.n_bases = c->hb.n_bases;
	HIPCHK(hipMemsetAsync(c->counters.p, 0, CTR_BYTES, c->st));
	HIPCHK(hipMemsetAsync(c->err.p, 0, 16, c->st));
	const std::vector<int32_t> &status = c->hb.status;
	std::vector<int32_t> dl(c->hb.rlen);
	// reads longer than max_qlen are not mapped (U:map.c::mm_map_frag)
	for (int64_t i = 0; i < n_reads; ++i) if (mo->max_qlen > 0 && dl[i] > mo->max_qlen) dl[i] = 0;
	std::vector<const char*> seqs(n_reads);
	for (int64_t i = 0; i < n_reads; ++i) seqs[i] = (const char*)&c->hb.seq[c->hb.roff[i]];
	const bool verbose = getenv("MM355_VERBOSE") != 0;
	const bool rmq_chain = (mo->flag & MMF_RMQ) != 0;
	double tv0 = now_ms(), tv_front, tv_pack, tv_pre, tv_steps = 0, tv_dp = 0, tv_fin, tv_asm, tv_extra = 0;
	trace_add(c, "prolog", t_start, tv0);
	{
		double ts = now_ms();
#define FRONT_STAGE(name, call) do { if ((rc = (call))) return rc; if (g_trace_path) { const double te = now_ms(); trace_add(c, name, ts, te); ts = te; } } while (0)
		FRONT_STAGE("f:sketch", mm355_run_sketch(c));      // (asynchronous: its time shows up in the next stage's wait)
		FRONT_STAGE("f:seeds", mm355_run_seeds(c, pr));
		FRONT_STAGE("f:expand", mm355_run_expand(c, pr));
		FRONT_STAGE("f:sort", mm355_run_sort(c, pr));
		if (rmq_chain) FRONT_STAGE("f:chain", mm355_run_chain_skip(c));   // asm presets: mg_lchain_rmq is the primary chainer (next stage)
		else {
			FRONT_STAGE("f:chain", mm355_run_chain(c, pr));
			FRONT_STAGE("f:backtrack", mm355_run_backtrack(c, pr));
		}
		FRONT_STAGE("f:rmq", mm355_run_rmq(c, mo, pr));                  // row a9: long-join re-chain / primary RMQ chainer on the device
		FRONT_STAGE("f:codes", mm355_run_read_codes(c));
#undef FRONT_STAGE
	}
	HostBatch &hb = c->hb;
	tv_front = now_ms() - tv0; trace_add(c, "front", tv0, now_ms()); tv0 = now_ms();
	// pack chains / anchors / mini_pos and bring them to the host
	std::vector<int64_t> uo(n_reads + 1), vo(n_reads + 1), mo_(n_reads + 1);
	int64_t tu = 0, tv = 0, tm = 0;
	for (int64_t i = 0; i < n_reads; ++i) { uo[i] = tu; vo[i] = tv; mo_[i] = tm; tu += hb.n_u[i]; tv += hb.n_v[i]; tm += hb.n_mini[i]; }
	uo[n_reads] = tu; vo[n_reads] = tv; mo_[n_reads] = tm;
	if (c->h_pu.ensure((size_t)(tu + 1) * 8) || c->h_pm.ensure((size_t)(tm + 1) * 8) || c->h_pa.ensure((size_t)(tv + 1) * 16)) return MM355_ENOMEM;
	uint64_t *pu = (uint64_t*)c->h_pu.p, *pm = (uint64_t*)c->h_pm.p; mm128 *pa = (mm128*)c->h_pa.p;
	if (n_reads) {
		DBuf &scr = c->b;   // b[] (compact_a scratch) is free again: reuse it for the packed copies
		const size_t nr2 = ((size_t)n_reads + 2) & ~(size_t)1, tu2 = ((size_t)tu + 1) & ~(size_t)1;   // keep every sub-array 16-B aligned
		size_t need = nr2 * 8 * 3 + tu2 * 8 + (size_t)tv * 16 + (size_t)tm * 8 + 256;
		DBuf &pack = c->pack; if (pack.ensure(need)) return MM355_ENOMEM;
		(void)scr;
		int64_t *d_uo = pack.as<int64_t>(), *d_vo = d_uo + nr2, *d_mo = d_vo + nr2;
		uint64_t *d_pu = (uint64_t*)(d_mo + nr2); mm128 *d_pa = (mm128*)(d_pu + tu2); uint64_t *d_pm = (uint64_t*)(d_pa + tv);
		HIPCHK(hipMemcpyAsync(d_uo, uo.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, c->st));
		HIPCHK(hipMemcpyAsync(d_vo, vo.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, c->st));
		HIPCHK(hipMemcpyAsync(d_mo, mo_.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, c->st));
		mm355_kt(c, KT_PACK, 0, c->st);
		hipLaunchKernelGGL(k_pack_chains, dim3((unsigned)n_reads), dim3(256), 0, c->st, (int)n_reads, c->aoff.as<int64_t>(), c->roff.as<int64_t>(),
		                   c->n_u.as<int32_t>(), c->n_v.as<int32_t>(), c->n_mini.as<int32_t>(), d_uo, d_vo, d_mo, c->u.as<uint64_t>(), c->a.as<mm128>(),
		                   c->mini_pos.as<uint64_t>(), d_pu, d_pa, d_pm);
		mm355_kt(c, KT_PACK, 1, c->st);
		HIPCHK(hipGetLastError());
		if (tu) HIPCHK(hipMemcpyAsync(pu, d_pu, (size_t)tu * 8, hipMemcpyDeviceToHost, c->st));
		if (tv) HIPCHK(hipMemcpyAsync(pa, d_pa, (size_t)tv * 16, hipMemcpyDeviceToHost, c->st));
		if (tm) HIPCHK(hipMemcpyAsync(pm, d_pm, (size_t)tm * 8, hipMemcpyDeviceToHost, c->st));
		HIPCHK(mm355_wait_stream(c->st));
	}
	mm355_timers_resolve(c);   // the stream is idle here: the stage timers of the front turn into milliseconds without waiting
	tv_pack = now_ms() - tv0; trace_add(c, "pack", tv0, now_ms()); tv0 = now_ms();
	const double t_host0 = now_ms();
	const int nt = host_threads();
	// mm_update_extra's per-base walk and the cs string: on the device for all regions of the batch at once (k_extra), unless MD or '=' / 'X'
	// CIGARs are asked for (those stay with the host walk) or MM355_EXTRA_HOST=1
	const bool extra_host = [] { const char *e = getenv("MM355_EXTRA_HOST"); return e && atoi(e) != 0; }();   // (read per call: the tests switch it)
	// ... and only for batches of a thousand reads or more: the extra launch (CIGARs up, three kernels, two synchronisations, cs down) costs a
	// single read 0.65 ms and a batch of 256 reads 1.3 ms more than the host walk; at 4096 reads it is 7 ms cheaper
	const int64_t extra_min_reads = [] { const char *e = getenv("MM355_EXTRA_MIN_READS"); return (int64_t)(e? atoll(e) : 1024); }();
	const bool defer_extra = !extra_host && !(mo->flag & MMF_EQX) && n_reads >= extra_min_reads;
	std::vector<ReadState> rs(n_reads);
	struct RegGuard { std::vector<ReadState> &v; ~RegGuard() { for (ReadState &r : v) mm355_glue_release(r); } } reg_guard{rs};   // every return below frees the regions' Extra records
	parallel_for(n_reads, nt, [&](int64_t i, int) {
		ReadState &r = rs[i];
		r.qlen = dl[i]; r.seq = seqs[i]; r.rep_len = hb.rep_len[i]; r.defer_extra = defer_extra;
		{ ProfScope pf(PF_PRE_COPY);
		r.u.assign(pu + uo[i], pu + uo[i + 1]);
		r.a.assign(pa + vo[i], pa + vo[i + 1]);
		r.mini_pos.assign(pm + mo_[i], pm + mo_[i + 1]); }
		if (r.qlen > 0) {
			int rst = hb.rmq_state.empty()? 3 : (int)hb.rmq_state[i];   // 3 = MM355_RMQ_HOST_ALL
			if (rst == 3) { if (rmq_chain) mm355_glue_chain_rmq(mi, mo, r); rst = -1; }   // every mg_lchain_rmq call of this read on the host
			mm355_glue_pre_align(mi, mo, r, rst);
		} else r.aligned = true;
	});
	double ms_host = now_ms() - t_host0;
	tv_pre = now_ms() - tv0; trace_add(c, "pre", tv0, now_ms());
	int n_rounds = 0;
	std::vector<std::vector<uint32_t>> arenas;   // CIGAR arenas of all extension launches of this batch
	// extension rounds: every pending problem of every read goes into the same launches.  No cap on the number of rounds (the reference
	// has none: every z-drop split of a long divergent read costs a round or two); progress is guaranteed because a round with open reads
	// and no request is reported as an error below.
	for (int round = 0;; ++round) {
		const double th0 = now_ms();
		std::vector<std::vector<DpReq>> treq(nt);
		std::atomic<int64_t> n_open(0);
		parallel_for(n_reads, nt, [&](int64_t i, int tid) {
			if (!mm355_glue_align_step(mi, mo, (int)i, rs[i], treq[tid], flags)) n_open.fetch_add(1);
		});
		std::vector<DpReq> reqs;
		for (auto &v : treq) reqs.insert(reqs.end(), v.begin(), v.end());
		ms_host += now_ms() - th0;
		tv_steps += now_ms() - th0; trace_add(c, "align", th0, now_ms());
		if (n_open.load() == 0) break;
		if (reqs.empty()) return MM355_EINVAL;   // a read is waiting for a result nobody requested: logic error
		const double td0 = now_ms();
		if ((rc = run_dp_round(c, mo, rs, reqs, arenas))) return rc;
		tv_dp += now_ms() - td0; ++n_rounds; trace_add(c, "dp", td0, now_ms());
		if (verbose) fprintf(stderr, "[mm355]   round %d: %zu jobs, %lld reads open\n", round, reqs.size(), (long long)n_open.load());
	}
	if (defer_extra && n_reads > 0) {   // row f2: one launch for every aligned region of the batch
		const double tx0 = now_ms();
		std::vector<int64_t> xr((size_t)n_reads + 1), xg((size_t)n_reads + 1), xc((size_t)n_reads + 1), xs((size_t)n_reads + 1);
		parallel_for(n_reads, nt, [&](int64_t i, int) { mm355_glue_extra_count(rs[i], &xr[i], &xg[i], &xc[i], &xs[i]); });
		int64_t tr = 0, tg = 0, tc = 0, ts = 0;
		for (int64_t i = 0; i < n_reads; ++i) {
			const int64_t a = xr[i], g = xg[i], b = xc[i], d = xs[i];
			xr[i] = tr; xg[i] = tg; xc[i] = tc; xs[i] = ts; tr += a; tg += g; tc += b; ts += d;
		}
		xr[n_reads] = tr; xg[n_reads] = tg; xc[n_reads] = tc; xs[n_reads] = ts;
		if (tr > 0) {
			const size_t seg_b = ((size_t)tg * sizeof(Mm355ExtraJob) + 63) & ~(size_t)63;
			if (c->h_xjobs.ensure(seg_b + ((size_t)tr + 1) * 8 + 64) || c->h_xcig.ensure(((size_t)tc + 16) * 4)) return MM355_ENOMEM;
			Mm355ExtraJob *xsegs = (Mm355ExtraJob*)c->h_xjobs.p; int64_t *xfirst = (int64_t*)((char*)c->h_xjobs.p + seg_b); uint32_t *xcig = (uint32_t*)c->h_xcig.p;
			xfirst[tr] = tg;
			parallel_for(n_reads, nt, [&](int64_t i, int) {
				if (xr[i + 1] > xr[i]) mm355_glue_extra_fill(rs[i], 2 * c->hb.roff[i], xsegs + xg[i], xfirst, xr[i], xg[i], xcig, xc[i], xs[i]);
			});
			const Mm355ExtraOut *xo = 0; const char *xcs = 0;
			const int want_cs = flags & (MM355_OUT_CS | MM355_OUT_MD);
			ms_host += now_ms() - tx0; trace_add(c, "x:fill", tx0, now_ms());
			if ((rc = mm355_extra_run(c, mo, xsegs, (size_t)tg, xfirst, (size_t)tr, xcig, (size_t)tc, (size_t)ts, want_cs, &xo, &xcs))) return rc;
			const double tx1 = now_ms();
			parallel_for(n_reads, nt, [&](int64_t i, int) { if (xr[i + 1] > xr[i]) mm355_glue_extra_apply(rs[i], xo + xr[i], xcs, want_cs); });
			ms_host += now_ms() - tx1; trace_add(c, "x:apply", tx1, now_ms());
		}
		tv_extra = now_ms() - tx0; trace_add(c, "extra", tx0, now_ms());
	}
	const double th1 = now_ms();
	std::vector<std::vector<mm355_hit_t>> rh(n_reads);
	std::vector<std::vector<uint32_t>> rc_(n_reads);
	std::vector<std::string> rstr(n_reads);
	parallel_for(n_reads, nt, [&](int64_t i, int) {
		ProfScope pf(PF_FINISH);
		if (rs[i].qlen > 0) mm355_glue_finish(mi, mo, rs[i], flags, rh[i], rc_[i], rstr[i]);
		mm355_glue_release(rs[i]);
	});
	tv_fin = now_ms() - th1; trace_add(c, "finish", th1, now_ms()); tv0 = now_ms();
	mm355_hits_t *H = (mm355_hits_t*)calloc(1, sizeof(mm355_hits_t));
	H->n_reads = n_reads;
	H->hit_off = (int64_t*)malloc((n_reads + 1) * 8);
	H->status = (int32_t*)malloc((n_reads > 0? n_reads : 1) * 4);
	int64_t nh = 0, nc = 0, ns = 0;
	for (int64_t i = 0; i < n_reads; ++i) { H->hit_off[i] = nh; H->status[i] = status[i]; nh += (int64_t)rh[i].size(); nc += (int64_t)rc_[i].size(); ns += (int64_t)rstr[i].size(); }
	H->hit_off[n_reads] = nh; H->n_hits = nh; H->n_cigar = nc; H->n_str = ns;
	H->hits = (mm355_hit_t*)malloc((nh > 0? nh : 1) * sizeof(mm355_hit_t));
	H->cigar = (uint32_t*)malloc((nc > 0? nc : 1) * 4);
	H->str = (char*)malloc(ns > 0? ns : 1);
	nh = nc = ns = 0;
	for (int64_t i = 0; i < n_reads; ++i) {
		for (mm355_hit_t h : rh[i]) {
			h.cigar_off += nc;
			if (h.cs_len >= 0) h.cs_off += ns;
			if (h.md_len >= 0) h.md_off += ns;
			H->hits[nh++] = h;
		}
		if (!rc_[i].empty()) memcpy(H->cigar + nc, rc_[i].data(), rc_[i].size() * 4);
		if (!rstr[i].empty()) memcpy(H->str + ns, rstr[i].data(), rstr[i].size());
		nc += (int64_t)rc_[i].size(); ns += (int64_t)rstr[i].size();
	}
	ms_host += now_ms() - th1;
	tv_asm = now_ms() - tv0; trace_add(c, "asm", tv0, now_ms());
	if (verbose) fprintf(stderr, "[mm355] map_resident: front %.1f ms | pack+d2h %.1f | pre_align %.1f | align_steps %.1f | dp rounds(%d) %.1f (kernel %.1f) | extra %.1f | finish %.1f | assemble %.1f | total %.1f\n",
	                     tv_front, tv_pack, tv_pre, tv_steps, n_rounds, tv_dp, c->stats.ms_dp, tv_extra, tv_fin, tv_asm, now_ms() - t_start);
	c->stats.ms_host = ms_host; c->stats.n_ext_rounds = n_rounds;
	c->stats.ms_total = now_ms() - t_start;
	c->stats.host_cpu_ms = (double)(thread_cpu_ns() - cpu_start + tl_helper_cpu_ns) * 1e-6;   // this thread (driver calls, waits, serial parts, its share of the jobs) + the pool's helpers
	if (verbose) mm355_prof_dump(n_reads);
	if (verbose) mm355_kprof_dump(c);
	*out = H;
	{   // the per-read host state (a dozen heap blocks per read: 20 ms of free() per 6144-read sub-batch, MM355_TRACE "teardown") is handed to the
		// reaper thread instead of being destroyed behind the return on this thread's critical path
		const double td = now_ms();
		struct Grave : Reaper::Item { std::vector<ReadState> rs; std::vector<std::vector<mm355_hit_t>> rh; std::vector<std::vector<uint32_t>> rc, ar; std::vector<std::string> st; };
		Grave *g = new Grave();
		g->rs.swap(rs); g->rh.swap(rh); g->rc.swap(rc_); g->st.swap(rstr); g->ar.swap(arenas);
		Reaper::get().bury(g);
		trace_add(c, "teardown", td, now_ms());
	}
	return 0;
}

extern "C" void mm355_free_hits(mm355_hits_t *h)
{
	if (h == 0) return;
	free(h->hit_off); free(h->status); free(h->hits); free(h->cigar); free(h->str); free(h);
}
