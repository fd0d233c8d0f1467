#include "mm355_pipeline.h"
extern "C" int mm355_stage_dp(mm355_ctx_t *ctx, const mm355_mapopt_t *mo, int64_t n_jobs, const mm355_dpjob_t *jobs,
                   const uint8_t *qcodes, int64_t n_q, const uint8_t *tcodes, int64_t n_t,
                   mm355_dpres_t *res, uint32_t *cigar, int64_t cigar_cap) { return MM355_EUNSUP; }
