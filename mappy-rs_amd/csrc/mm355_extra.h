// mm355_extra.h -- descriptors shared by the host tail (mm355_glue.cpp) and k_extra (mm355_dp.hip)
#pragma once
#include <stdint.h>
// ---- row f2: U:align.c::mm_update_extra's per-base walk, U:format.c::write_cs_core and write_MD_core on the device (k_extra)
#define MM355_EXTRA_SEG 64          // CIGAR operations per segment (at most)
#define MM355_EXTRA_SEG_COLS 2048   // ... and columns per segment: a lane's walk is a chain of dependent loads.  A match operation longer than
                                    // that is CUT (HiFi: 500-base and longer matches); insertions and deletions stay whole (<= max_gap columns)
#define MM355_EXTRA_MIN_PIECE 256   // a match operation is not cut to fill less than this much of a segment
struct Mm355ExtraJob {      // one SEGMENT of an aligned region: one lane of k_extra
	int64_t q_src;          // offset of its first query base in the per-read code buffer (strand-adjusted, like DpGather::q_src)
	int64_t cig_off;        // first CIGAR operation in the uploaded array
	int64_t cs_off, md_off; // where its pieces of the cs / MD strings may be written (worst-case sized slots)
	uint32_t rid; int32_t t_st;   // first target base
	int32_t n_cigar, region;      // operations of the segment (the first and the last may be partial); index of the region it belongs to
	int32_t skip0, end_last;      // columns of the first operation that earlier segments walked; column where this segment leaves its last
	                              // operation (0 = at its end).  Only match operations are ever partial
};
// What a segment leaves.  Counts add up.  The score walk s <- max(0, s + d) is a max-plus transform: with A_i the sum of the first i score
// steps, s after step i is A_i + max(s_in, -m_i) (m_i = min of A_1..A_i), so A = A_n, m = m_n give s_out and C = max A_i, P = max (A_i - m_i)
// give the largest s inside the segment: max(s_in + C, P).  The strings: a run of matches may begin before the segment and end behind it, so
// the FIRST number a segment would print is left to the composer (lead = matches counted up to the first flush; the number printed there is
// the carry of the earlier segments + lead), and the matches pending at its end go on as the next segment's carry (tail); a segment that
// never flushes only adds to the carry.  cs flushes at a mismatch and at the end of every match operation (only a number > 0 is printed); MD
// flushes at a mismatch and at a deletion (the number is always printed) and carries its count across operations.
struct Mm355ExtraSegOut {
	double A, m, C, P;
	int32_t mlen, blen, n_ambi;
	int32_t cs_len, cs_lead, cs_tail, md_len, md_lead, md_tail, flushed;   // *_len: body bytes in the slot; flushed: bit 0 cs, bit 1 MD
	int32_t cs_pre, pad;           // cs bytes written BEFORE the first flush (the text of leading insertions / deletions): they stand in front of the number
	int32_t cs_num, md_num;        // filled by k_extra_compose: the number in front of the body (-1: none)
	int64_t cs_dense, md_dense;    // ... and where [number][body] goes inside the region's string
};
struct Mm355ExtraOut { int32_t mlen, blen, n_ambi, dp_max; int64_t cs_dense; int32_t cs_len, md_len, md_end_num, pad; };   // cs at cs_dense, MD right behind it
struct Mm355ExtraScore { int8_t mat[25]; int8_t q, e; };

// Cuts one region into segments (at most MM355_EXTRA_SEG operations and MM355_EXTRA_SEG_COLS columns; a longer match operation is cut,
// a longer gap gets a segment of its own).  segs == 0: count only.  Returns the number of segments; *cs_cap / *md_cap: bytes of string slots used
static inline int mm355_extra_split(const uint32_t *cg, int n, int64_t q_src, uint32_t rid, int64_t t_st, int64_t cig_off, int64_t cs_off, int64_t md_off, int32_t region,
                                    Mm355ExtraJob *segs, int64_t *cs_cap, int64_t *md_cap)
{
	int g = 0;
	int64_t qoff = 0, toff = 0, cso = 0, mdo = 0;
	int c = 0; int64_t done = 0;     // next operation and the columns of it already assigned
	while (c < n) {
		Mm355ExtraJob j;
		j.q_src = q_src + qoff; j.cig_off = cig_off + c; j.cs_off = cs_off + cso; j.md_off = md_off + mdo; j.rid = rid; j.t_st = (int32_t)(t_st + toff);
		j.region = region; j.skip0 = (int32_t)done; j.end_last = 0;
		int ops = 0; int64_t cols = 0;
		while (c < n && ops < MM355_EXTRA_SEG) {
			const uint32_t op = cg[c] & 0xf; const int64_t len = (int64_t)(cg[c] >> 4), rest = len - done;
			const bool is_m = op == 0 || op == 7 || op == 8;
			int64_t take = rest;
			if (cols + rest > MM355_EXTRA_SEG_COLS) {
				if (is_m) { take = MM355_EXTRA_SEG_COLS - cols; if (take < MM355_EXTRA_MIN_PIECE && ops > 0) break; if (take < MM355_EXTRA_MIN_PIECE) take = rest < MM355_EXTRA_SEG_COLS? rest : MM355_EXTRA_SEG_COLS; }
				else if (ops > 0) break;
			}
			++ops; cols += take;
			if (is_m) qoff += take, toff += take; else if (op == 1) qoff += take; else if (op == 2 || op == 3) toff += take;
			if (take < rest) { done += take; j.end_last = (int32_t)done; break; }
			done = 0; ++c;
		}
		j.n_cigar = ops;
		if (segs) segs[g] = j;
		++g;
		cso += 3 * cols + 12 * (int64_t)ops + 16; mdo += 2 * cols + 12 * (int64_t)ops + 16;
	}
	if (cs_cap) *cs_cap = (cso + 15) & ~(int64_t)15;
	if (md_cap) *md_cap = (mdo + 15) & ~(int64_t)15;
	return g;
}
