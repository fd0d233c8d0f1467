// mm355_extra.h -- descriptors shared by the host tail (mm355_glue.cpp) and k_extra (mm355_dp.hip)
#pragma once
#include <stdint.h>
// ---- row f2: U:align.c::mm_update_extra's per-base walk and U:format.c::write_cs_core on the device (k_extra)
#define MM355_EXTRA_SEG 64   // CIGAR operations per segment (at most)
#define MM355_EXTRA_SEG_COLS 2048   // ... and columns per segment, cut at an operation boundary: a lane's walk is a chain of dependent loads
#define MM355_EXTRA_MAX_OP 2048     // regions with a longer operation, or with long operations on average (HiFi: a few 500-base matches), keep the
#define MM355_EXTRA_AVG_OP 64       // host walk -- one lane would walk them alone, and the host compares eight bases per step on such runs
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

// worst-case cs bytes of a CIGAR: "*xy" per aligned base, "+" / "-" and the bases per gap, ":<number>" per match run
static inline int64_t mm355_extra_cs_cap(const uint32_t *cg, int n)
{
	int64_t tot = 0;
	for (int c = 0; c < n; ++c) tot += cg[c] >> 4;
	return (3 * tot + 12 * (int64_t)n + 31) & ~(int64_t)15;
}
// is a region's walk worth leaving to the device?  (short operations: ONT-like CIGARs)
static inline bool mm355_extra_device_ok(const uint32_t *cg, int n)
{
	int64_t tot = 0; uint32_t mx = 0;
	for (int c = 0; c < n; ++c) { const uint32_t len = cg[c] >> 4; tot += len; mx = len > mx? len : mx; }
	return n > 0 && mx <= MM355_EXTRA_MAX_OP && tot <= (int64_t)MM355_EXTRA_AVG_OP * n;
}
// number of segments mm355_extra_split will cut a region into
static inline int mm355_extra_n_segs(const uint32_t *cg, int n)
{
	int g = 0, ops = 0; int64_t cols = 0;
	for (int c = 0; c < n; ++c) {
		const int64_t len = cg[c] >> 4;
		if (ops > 0 && (ops >= MM355_EXTRA_SEG || cols + len > MM355_EXTRA_SEG_COLS)) { ++g; ops = 0; cols = 0; }
		++ops; cols += len;
	}
	return g + (ops > 0? 1 : 0);
}
// cuts one region into segments (at most MM355_EXTRA_SEG operations and, unless a single operation is longer, MM355_EXTRA_SEG_COLS columns);
// returns the number of segments written
static inline int mm355_extra_split(const uint32_t *cg, int n, int64_t q_src, uint32_t rid, int64_t t_st, int64_t cig_off, int64_t cs_off, int32_t region, Mm355ExtraJob *segs)
{
	int g = 0;
	int64_t qoff = 0, toff = 0, cso = 0;
	int c0 = 0;
	while (c0 < n) {
		int c1 = c0; int64_t cols = 0;
		while (c1 < n && c1 - c0 < MM355_EXTRA_SEG && (c1 == c0 || cols + (int64_t)(cg[c1] >> 4) <= MM355_EXTRA_SEG_COLS)) { cols += cg[c1] >> 4; ++c1; }
		Mm355ExtraJob j;
		j.q_src = q_src + qoff; j.cig_off = cig_off + c0; j.cs_off = cs_off + cso; j.rid = rid; j.t_st = (int32_t)(t_st + toff); j.n_cigar = c1 - c0; j.region = region;
		segs[g++] = j;
		int64_t tot = 0;
		for (int c = c0; c < c1; ++c) {
			const uint32_t op = cg[c] & 0xf, len = cg[c] >> 4;
			if (op == 0 || op == 7 || op == 8) qoff += len, toff += len;
			else if (op == 1) qoff += len;
			else if (op == 2 || op == 3) toff += len;
			tot += len;
		}
		cso += 3 * tot + 12 * (int64_t)(c1 - c0);
		c0 = c1;
	}
	return g;
}
