// mm355_sketch.h -- the chunked (w,k)-minimizer machine shared by the read sketch kernel (k_sketch) and the device
// index builder (k_sketch_contig).  See the comment above k_sketch in mm355_kernels.hip for the exactness argument.
#pragma once
#include "mm355_core.h"
#define SK_CHUNK 384
struct BaseReader {
	const uint8_t *s; uint64_t wd; int wi;
	__device__ int operator()(int i) {
		int q = i >> 3;
		if (q != wi) { wd = ((const uint64_t*)s)[q]; wi = q; }   // reads start 16-B aligned and are padded
		return mm_nt4((uint8_t)(wd >> ((i & 7) * 8)));
	}
};

// HPC (MM_I_HPC indexes: map-pb / ava-pb; mm_sketch_seq in mm355_core.h is the sequential form): a homopolymer run is one base of the
// k-mer at the position of its last base, the span the sum of the last k run lengths.  What the proof needs on top of the plain one:
//   * a lane may start inside a run: it sees the same compressed base at the same end position, only its first run length is short --
//     spans are exact once that run has left the queue (k + 1 runs pushed) or an ambiguous base has emptied the queue on both sides
//     (span_bad counts down); records are only COUNTED as exact (cnt_c, and from there the ring writes) behind that point;
//   * a run that starts before the chunk and ends inside it is a record of the chunk: the state must be trusted before that run is taken.
template <bool HPC>
__device__ int sketch_chunk(const uint8_t *seq, int len, int w, int k, int cs, int ce, mm128 *out, mm128 *buf, int bstride)
{
	const uint64_t shift1 = 2 * (k - 1), mask = (1ULL << 2 * k) - 1;
#define BUF(j) buf[(j) * bstride]
	// first attempt: what the proof below needs on plain sequence ((w + k) counted k-mers, then w + 1 ring writes) plus a margin; the proof, not
	// this length, is what makes the chunk exact -- a start that does not complete it is retried four times further back
	int warm = (w + k) + (w + 1) + (k & 1? 8 : k + 8);
	if (HPC) warm = 2 * (warm + k + 1);                        // runs, not bases: ~ 4/3 bases per run on plain sequence
	for (;;) {
		int s0 = cs - warm;
		if (s0 < 0) s0 = 0;
		const bool from_start = s0 == 0;
		BaseReader get = { seq, 0, -1 };
		uint64_t kmer[2] = {0, 0};
		int i, j, l = 0, buf_pos = 0, min_pos = 0, kmer_span = 0, n = 0;
		uint16_t tq[HPC? 32 : 1]; int tq_front = 0, tq_count = 0, span_bad = from_start? 0 : k + 1;
		mm128 min = { UINT64_MAX, UINT64_MAX };
		for (j = 0; j < w; ++j) BUF(j).x = BUF(j).y = UINT64_MAX;
		// proof state
		int n_real = 0, cnt_c = 0, writes_ok = 0, writes_after = 0;
		// odd k: a k-mer never equals its reverse complement, so no skip decision depends on bases before the warm-up
		bool jstar = from_start || (k & 1), seen_n = false, ok = from_start, trust = from_start, retry = false;
#define MM_EMIT(v) do { uint32_t pp_ = (uint32_t)(v).y >> 1; if ((int)pp_ >= cs && (int)pp_ < ce) out[n++] = (v); } while (0)
		for (i = s0; i < len; ++i) {
			if ((HPC? i >= cs : i == cs) && !trust) { retry = true; break; }
			int c = get(i);
			mm128 info = { UINT64_MAX, UINT64_MAX };
			if (c < 4) {
				int z;
				if (!jstar && ++n_real >= k) jstar = true;
				if (HPC) {
					int skip_len = 1;
					while (i + skip_len < len && get(i + skip_len) == c) ++skip_len;
					i += skip_len - 1;
					if (i >= cs && !trust) { retry = true; break; }   // this run's record belongs to the chunk
					if (skip_len > MM355_HPC_RUN_CAP) skip_len = MM355_HPC_RUN_CAP;
					tq[(tq_count + tq_front) & 31] = (uint16_t)skip_len; ++tq_count;
					kmer_span += skip_len;
					if (tq_count > k) { kmer_span -= tq[tq_front]; tq_front = (tq_front + 1) & 31; --tq_count; }
					if (span_bad > 0) --span_bad;
				} else
				kmer_span = l + 1 < k? l + 1 : k;
				kmer[0] = (kmer[0] << 2 | c) & mask;
				kmer[1] = (kmer[1] >> 2) | (3ULL^c) << shift1;
				if (kmer[0] == kmer[1]) continue;
				z = kmer[0] < kmer[1]? 0 : 1;
				++l;
				if (jstar && (!HPC || span_bad == 0)) ++cnt_c;
				if (l >= k && kmer_span < 256) {
					info.x = mm_hash64(kmer[z], mask) << 8 | kmer_span;
					info.y = (uint64_t)(uint32_t)i << 1 | z;
				}
			} else { l = 0, kmer_span = 0; if (jstar) seen_n = true; if (HPC) tq_count = tq_front = 0, span_bad = 0; }
			if (!ok) ok = jstar && (seen_n || cnt_c >= w + k);
			if (ok && !trust) { if (++writes_ok > w) trust = true; }   // this write and w earlier ones are exact
			BUF(buf_pos) = info;
			if (l == w + k - 1 && min.x != UINT64_MAX) {
				for (j = buf_pos + 1; j < w; ++j)
					if (min.x == BUF(j).x && BUF(j).y != min.y) MM_EMIT(BUF(j));
				for (j = 0; j < buf_pos; ++j)
					if (min.x == BUF(j).x && BUF(j).y != min.y) MM_EMIT(BUF(j));
			}
			if (info.x <= min.x) {
				if (l >= w + k && min.x != UINT64_MAX) MM_EMIT(min);
				min = info, min_pos = buf_pos;
			} else if (buf_pos == min_pos) {
				if (l >= w + k - 1 && min.x != UINT64_MAX) MM_EMIT(min);
				for (j = buf_pos + 1, min.x = UINT64_MAX; j < w; ++j)
					if (min.x >= BUF(j).x) min = BUF(j), min_pos = j;
				for (j = 0; j <= buf_pos; ++j)
					if (min.x >= BUF(j).x) min = BUF(j), min_pos = j;
				if (l >= w + k - 1 && min.x != UINT64_MAX) {
					for (j = buf_pos + 1; j < w; ++j)
						if (min.x == BUF(j).x && min.y != BUF(j).y) MM_EMIT(BUF(j));
					for (j = 0; j <= buf_pos; ++j)
						if (min.x == BUF(j).x && min.y != BUF(j).y) MM_EMIT(BUF(j));
				}
			}
			if (++buf_pos == w) buf_pos = 0;
			if (i >= ce && ++writes_after >= w) break;   // every ring record now lies beyond the chunk
		}
		if (retry) { warm = from_start? warm : warm * 4; continue; }
		if (i >= len && min.x != UINT64_MAX) MM_EMIT(min);   // the sequential final flush
#undef MM_EMIT
#undef BUF
		return n;
	}
}

