// Test harness (not product): the chunked (w,k)-minimizer machine of the sketch kernels -- mm355_sketch.h, the header k_sketch /
// k_sketch_hpc / k_sketch_contig compile for the device -- compiled for the HOST with g++, so that the exactness proof of its warm-up
// (plain and homopolymer-compressed) is exercised by the CPU suite too: one call sketches a sequence chunk by chunk exactly as the kernels
// do (a lane per chunk of SK_CHUNK bases, outputs at the chunk's slot range, then packed in chunk order) and tests/test_chunk_machine.py
// holds the result against the oracle's sequential U:sketch.c::mm_sketch.
#define __device__
#include <vector>
#include "../../mappy-rs_amd/csrc/mm355_sketch.h"

extern "C" int chunk_sketch_host(const uint8_t *seq, int len, int w, int k, int hpc, int piece, uint64_t *out /* 2 * len words */)
{
	if (len <= 0) return 0;
	// reads on the device start 16-B aligned and are padded: BaseReader fetches 8 bytes at a time
	std::vector<uint8_t> buf((size_t)len + 64, 'N');
	memcpy(buf.data(), seq, (size_t)len);
	std::vector<mm128> slots((size_t)len + 64), ring((size_t)w);
	int n = 0;
	const int step = piece > 0? piece : SK_CHUNK;       // the small-batch kernel cuts a chunk into pieces of SK_CHUNK / 12 bases
	for (int cs = 0; cs < len; cs += step) {
		int ce = cs + step; if (ce > len) ce = len;
		const int m = hpc? sketch_chunk<true>(buf.data(), len, w, k, cs, ce, slots.data() + cs, ring.data(), 1)
		                 : sketch_chunk<false>(buf.data(), len, w, k, cs, ce, slots.data() + cs, ring.data(), 1);
		for (int i = 0; i < m; ++i) { out[2 * n] = slots[cs + i].x; out[2 * n + 1] = slots[cs + i].y; ++n; }
	}
	return n;
}
