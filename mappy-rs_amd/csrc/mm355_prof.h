// mm355_prof.h -- optional host-side section timers (MM355_PROF=1): CPU time per named section, summed over threads.
// TSC based, thread-local accumulation (no shared cache lines on the hot path).
#pragma once
#include <stdint.h>
#include <x86intrin.h>
enum { PF_PRE_COPY, PF_PRE_RMQ, PF_PRE_REGS, PF_PRE_ESTERR, PF_PRE_CODES, PF_PRE_SQUEEZE, PF_TASK_PREPARE, PF_GETSEQ, PF_TEST_ZDROP, PF_ADD_CIGAR,
       PF_UPDATE_EXTRA, PF_TASK_RUN, PF_STEP_TOTAL, PF_FINISH, PF_DISTRIBUTE, PF_GATHER_BUILD, PF_ASSEMBLE, PF_X1, PF_X2, PF_X3, PF_X4, PF_X5, PF_N };
struct ProfThread { uint64_t tsc[PF_N], cnt[PF_N]; ProfThread *next; };
extern bool g_prof_on;
ProfThread *mm355_prof_thread();   // registers the calling thread on first use
struct ProfScope {
	int id; uint64_t t0;
	explicit ProfScope(int id_) : id(id_), t0(0) { if (g_prof_on) t0 = __rdtsc(); }
	~ProfScope() { if (g_prof_on) { ProfThread *p = mm355_prof_thread(); p->tsc[id] += __rdtsc() - t0; ++p->cnt[id]; } }
};
void mm355_prof_dump(int64_t n_reads);
