#include "mm355_pipeline.h"
extern "C" int mm355_map_batch(mm355_ctx_t *ctx, const mm355_mapopt_t *mo, int64_t n_reads, const char *const *seqs,
                    const int32_t *lens, int flags, mm355_hits_t **out) { return MM355_EUNSUP; }
extern "C" void mm355_free_hits(mm355_hits_t *hits) {}
