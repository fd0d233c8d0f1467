// mm355_extra.h -- descriptors shared by the host tail (mm355_glue.cpp) and k_extra (mm355_dp.hip)
#pragma once
#include <stdint.h>
// ---- row f2: U:align.c::mm_update_extra's per-base walk and U:format.c::write_cs_core on the device (k_extra)
#define MM355_EXTRA_SEG 64   // CIGAR operations per segment
struct Mm355ExtraJob {      // one SEGMENT (up to MM355_EXTRA_SEG consecutive CIGAR operations) of an aligned region: one lane of k_extra
	int64_t q_src;          // offset of its first query base in the per-read code buffer (strand-adjusted, like DpGather::q_src)
	int64_t cig_off;        // first CIGAR operation in the uploaded array
	int64_t cs_off;         // where its piece of the cs string may be written (worst-case sized slot)
	uint32_t rid; int32_t t_st;   // first target base
	int32_t n_cigar, region;      // operations of the segment; index of the region it belongs to
};
// what a segment leaves: counts, its cs piece, and the score walk as a max-plus transform -- with A_i the sum of the first i score steps,
// s after step i is A_i + max(s_in, -m_i) (m_i = min of A_1..A_i), so A = A_n, m = m_n give s_out and C = max A_i, P = max (A_i - m_i)
// give the largest s inside the segment: max(s_in + C, P)
struct Mm355ExtraSegOut { double A, m, C, P; int32_t mlen, blen, n_ambi, cs_len; int64_t cs_dense; };
struct Mm355ExtraOut { int32_t mlen, blen, n_ambi, dp_max; int64_t cs_dense; int32_t cs_len, pad; };   // cs_dense: offset in the compacted cs arena
struct Mm355ExtraScore { int8_t mat[25]; int8_t q, e; };
