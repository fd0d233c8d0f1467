// mm355_dp.h -- descriptors of the banded-extension stage (host <-> mm355_dp.hip)
#pragma once
#include <stdint.h>
#include <vector>
#include "mm355_pipeline.h"

struct DpConst {            // scoring constants after ksw_extd2_sse's (q,e)/(q2,e2) ordering
	int32_t q, e, q2, e2, qe_preswap;
	int8_t sc_mch, sc_mis, sc_N;
	int32_t long_thres, long_diff, valid;
};

struct DpJobDev {
	int32_t qlen, tlen;
	int64_t qoff, toff;      // into the device code buffers
	int32_t w, zdrop, end_bonus, flag;
	int32_t skip, pad;       // pad: layout of the direction matrix -- 0 the reference's anti-diagonal rows, 1 the row sweep's tiles, 2 the band's tiles (mm355_dpband.h), +4: the matrix lives in the redo buffer
	int64_t p_off, off_off, cig_off, st_off;
	int32_t dlo, lmin;       // band kernels: first diagonal of the band, least end score that proves the band sufficient
	int32_t bw, rsv;         // ... and the band's width in diagonals (128, 256, 512)
};

struct DpGather {           // where the code strings of a job come from
	int32_t qlen, tlen;
	int64_t qoff, toff;      // destination offsets
	int64_t q_src;           // offset into the per-read code buffer (already strand-adjusted)
	uint32_t rid; int32_t t_st;
	int32_t rev, pad;
};

DpConst mm355_dp_const(const mm355_mapopt_t *mo);
size_t mm355_dp_matrix_bytes(const mm355_mapopt_t *mo, const DpConst &dc, int qlen, int tlen, int w, int flag);   // direction-matrix bytes mm355_dp_run will lay out for one problem
int mm355_dp_run(mm355_ctx *c, const mm355_mapopt_t *mo, DpJobDev *jobs, size_t n, const uint8_t *d_q, const uint8_t *d_t, HBuf *arena,
                 const mm355_dpres_t **res_out, const uint32_t **cigar_out);   // results live in pinned host buffers (c->h_res, *arena)
int mm355_dp_gather(mm355_ctx *c, const DpGather *g, size_t n, size_t q_tot, size_t t_tot);   // g: pinned, valid until the next call
int mm355_run_read_codes(mm355_ctx *c);

#include "mm355_extra.h"
// segs / seg_first / cig: pinned host arrays (mm355_glue_extra_fill); results: out[n_regions] (pinned, c->h_xout) and the compacted cs bytes (c->h_xcs)
int mm355_extra_run(mm355_ctx *c, const mm355_mapopt_t *mo, const Mm355ExtraJob *segs, size_t n_segs, const int64_t *seg_first, size_t n_regions,
                    const uint32_t *cig, size_t n_cig, size_t cs_cap, int want, const Mm355ExtraOut **out, const char **cs);   // want: bit 0 cs, bit 1 MD
