// mm355_core.h -- scalar building blocks shared by the host side and the HIP kernels
// of the MI355X mapping path (compiled as __host__ __device__ under hipcc, plain
// inline under g++).  Every routine states which minimap2 2.26 unit's behaviour it
// reproduces (U:file::function; the C sources are an un-vendored dependency of the
// reference, reached at /root/reference/src/lib.rs:482 and :587).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MM_HD __host__ __device__ inline
#else
#define MM_HD inline
#endif

struct mm128 { uint64_t x, y; };

#define MM355_SEED_LONG_JOIN (1ULL<<40)
#define MM355_SEED_IGNORE    (1ULL<<41)
#define MM355_SEED_TANDEM    (1ULL<<42)
#define MM355_SEED_SEG_SHIFT 48

// flags the path tests (U:minimap.h)
#define MMF_CIGAR        0x004LL
#define MMF_SPLICE       0x080LL
#define MMF_NO_LJOIN     0x400LL
#define MMF_SR           0x1000LL
#define MMF_FOR_ONLY     0x100000LL
#define MMF_REV_ONLY     0x200000LL
#define MMF_HEAP_SORT    0x400000LL
#define MMF_ALL_CHAINS   0x800000LL
#define MMF_EQX          0x4000000LL
#define MMF_NO_END_FLT   0x10000000LL
#define MMF_HARD_MLEVEL  0x20000000LL
#define MMF_RMQ          0x80000000LL
#define MMF_QSTRAND      0x100000000LL
#define MMF_NO_INV       0x200000000LL

// ---- base encoding (U:sketch.c::seq_nt4_table): A/a 0, C/c 1, G/g 2, T/t/U/u 3, else 4 ----
MM_HD int mm_nt4(uint8_t c)
{
	uint8_t d = c | 0x20;
	int r = d == 'a'? 0 : d == 'c'? 1 : d == 'g'? 2 : (d == 't' || d == 'u')? 3 : 4;
	return c < 4? (int)c : r;   // raw codes 0..3 map to themselves, as in the upstream table
}

// ---- U:sketch.c::hash64 (invertible mix, masked to 2k bits) ----
MM_HD uint64_t mm_hash64(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask;
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask;
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask;
	key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

// ---- hash of the flat HBM table (layout is this project's own; only mm_idx_get's contract is normative) ----
MM_HD uint64_t mm_table_hash(uint64_t minier)
{
	uint64_t h = minier * 0x9E3779B97F4A7C15ULL;
	h ^= h >> 29;
	return h;
}

// One slot of the flat table: 16 B.  key = minimizer<<1 | is_singleton, UINT64_MAX = empty.
// val = the y word (rid<<32|pos<<1|strand) for a singleton, else offset<<32|count into pos[].
struct mm355_slot { uint64_t key, val; };
#define MM355_SLOTS_PER_LINE 8      // 8 x 16 B = one 128-B line: a probe sequence stays inside one HBM fetch

// ---- sequential (w,k)-minimizer state machine: U:sketch.c::mm_sketch ----
// is_hpc (MM_I_HPC, the map-pb / ava-pb presets): a homopolymer run counts as ONE base of the k-mer, at the position of its last
// base; the span of a k-mer is the sum of its k run lengths (the reference's tiny_queue of the last k of them).  Run lengths are kept
// clamped to 4096: a span that holds such a run is >= 256 either way (no record), and what is added is what is subtracted later.
#define MM355_HPC_RUN_CAP 4096
// The window logic is order-dependent (rightmost minimum, duplicate emission rules, symmetric
// k-mers skipped without advancing the ring), so one lane runs it per read; out[] receives
// at most `cap` entries, the return value is the number that would have been written.
// `buf` is caller-provided ring storage of w entries spaced `bstride` apart (LDS on the device).
template <typename GetBase>
MM_HD int64_t mm_sketch_seq(GetBase get, int len, int w, int k, uint32_t rid, mm128 *out, int64_t cap, mm128 *buf, int bstride, bool is_hpc = false)
{
	uint64_t shift1 = 2 * (k - 1), mask = (1ULL<<2*k) - 1, kmer[2] = {0,0};
	int i, j, l, buf_pos, min_pos, kmer_span = 0;
	uint16_t tq[32]; int tq_front = 0, tq_count = 0;   // run lengths of the last k runs (HPC)
	int64_t n = 0;
	mm128 min = { UINT64_MAX, UINT64_MAX };
#define BUF(j) buf[(j) * bstride]
	for (j = 0; j < w; ++j) BUF(j).x = BUF(j).y = UINT64_MAX;
#define MM_PUSH(v) do { if (n < cap) out[n] = (v); ++n; } while (0)
	for (i = l = buf_pos = min_pos = 0; i < len; ++i) {
		int c = get(i);
		mm128 info = { UINT64_MAX, UINT64_MAX };
		if (c < 4) {
			int z;
			if (is_hpc) {
				int skip_len = 1;
				while (i + skip_len < len && get(i + skip_len) == c) ++skip_len;
				i += skip_len - 1;
				if (skip_len > MM355_HPC_RUN_CAP) skip_len = MM355_HPC_RUN_CAP;
				tq[(tq_count + tq_front) & 31] = (uint16_t)skip_len; ++tq_count;
				kmer_span += skip_len;
				if (tq_count > k) { kmer_span -= tq[tq_front]; tq_front = (tq_front + 1) & 31; --tq_count; }
			} else kmer_span = l + 1 < k? l + 1 : k;
			kmer[0] = (kmer[0] << 2 | c) & mask;
			kmer[1] = (kmer[1] >> 2) | (3ULL^c) << shift1;
			if (kmer[0] == kmer[1]) continue;
			z = kmer[0] < kmer[1]? 0 : 1;
			++l;
			if (l >= k && kmer_span < 256) {
				info.x = mm_hash64(kmer[z], mask) << 8 | kmer_span;
				info.y = (uint64_t)rid<<32 | (uint32_t)i<<1 | z;
			}
		} else l = 0, kmer_span = 0, tq_count = tq_front = 0;
		BUF(buf_pos) = info;
		if (l == w + k - 1 && min.x != UINT64_MAX) {
			for (j = buf_pos + 1; j < w; ++j)
				if (min.x == BUF(j).x && BUF(j).y != min.y) MM_PUSH(BUF(j));
			for (j = 0; j < buf_pos; ++j)
				if (min.x == BUF(j).x && BUF(j).y != min.y) MM_PUSH(BUF(j));
		}
		if (info.x <= min.x) {
			if (l >= w + k && min.x != UINT64_MAX) MM_PUSH(min);
			min = info, min_pos = buf_pos;
		} else if (buf_pos == min_pos) {
			if (l >= w + k - 1 && min.x != UINT64_MAX) MM_PUSH(min);
			for (j = buf_pos + 1, min.x = UINT64_MAX; j < w; ++j)
				if (min.x >= BUF(j).x) min = BUF(j), min_pos = j;
			for (j = 0; j <= buf_pos; ++j)
				if (min.x >= BUF(j).x) min = BUF(j), min_pos = j;
			if (l >= w + k - 1 && min.x != UINT64_MAX) {
				for (j = buf_pos + 1; j < w; ++j)
					if (min.x == BUF(j).x && min.y != BUF(j).y) MM_PUSH(BUF(j));
				for (j = 0; j <= buf_pos; ++j)
					if (min.x == BUF(j).x && min.y != BUF(j).y) MM_PUSH(BUF(j));
			}
		}
		if (++buf_pos == w) buf_pos = 0;
	}
	if (min.x != UINT64_MAX) MM_PUSH(min);
#undef MM_PUSH
#undef BUF
	return n;
}

// ---- U:ksort.h radix_sort_128x: in-place MSD byte radix, insertion sort <= 64; NOT stable.
// The permutation of equal keys is observable downstream, so the exact procedure is kept.
// Sequential form (one lane / host); `Key` maps an element to its u64 key.
#define MM355_RS_MIN_SIZE 64
template <typename T, typename Key>
MM_HD void mm_rs_insertsort(T *beg, T *end, Key key)
{
	for (T *i = beg + 1; i < end; ++i)
		if (key(*i) < key(*(i - 1))) {
			T *j, tmp = *i;
			for (j = i; j > beg && key(tmp) < key(*(j-1)); --j)
				*j = *(j - 1);
			*j = tmp;
		}
}

// one level of the cycle-leader permutation on [beg,end) by byte (key>>s)&255.
// cnt[256] must hold the histogram on entry; on exit bb[k]..be[k] are bucket bounds (element indices).
template <typename T, typename Key>
MM_HD void mm_rs_permute(T *beg, int64_t n, int s, const uint32_t *cnt, uint32_t *bb, uint32_t *be, Key key)
{
	uint32_t acc = 0;
	for (int k = 0; k < 256; ++k) { bb[k] = acc; acc += cnt[k]; be[k] = acc; }
	(void)n;
	for (int k = 0; k < 256;) {
		if (bb[k] != be[k]) {
			int l = (int)(key(beg[bb[k]]) >> s & 255);
			if (l != k) {
				T tmp = beg[bb[k]], swap;
				do {
					swap = tmp; tmp = beg[bb[l]]; beg[bb[l]++] = swap;
					l = (int)(key(tmp) >> s & 255);
				} while (l != k);
				beg[bb[k]++] = tmp;
			} else ++bb[k];
		} else ++k;
	}
	// restore bucket starts
	acc = 0;
	for (int k = 0; k < 256; ++k) { bb[k] = acc; acc += cnt[k]; }
}

// full sequential sort (host use, and device fallback for tiny arrays)
template <typename T, typename Key>
MM_HD void mm_rs_sort_seq(T *beg, T *end, int s, Key key)
{
	uint32_t cnt[256], bb[256], be[256];
	for (int k = 0; k < 256; ++k) cnt[k] = 0;
	for (T *i = beg; i != end; ++i) ++cnt[key(*i) >> s & 255];
	mm_rs_permute(beg, end - beg, s, cnt, bb, be, key);
	if (s) {
		s = s > 8? s - 8 : 0;
		for (int k = 0; k < 256; ++k) {
			uint32_t sz = be[k] - bb[k];
			if (sz > MM355_RS_MIN_SIZE) mm_rs_sort_seq(beg + bb[k], beg + be[k], s, key);
			else if (sz > 1) mm_rs_insertsort(beg + bb[k], beg + be[k], key);
		}
	}
}

template <typename T, typename Key>
MM_HD void mm_radix_sort(T *beg, T *end, Key key)
{
	if (end - beg <= MM355_RS_MIN_SIZE) mm_rs_insertsort(beg, end, key);
	else mm_rs_sort_seq(beg, end, 56, key);
}

struct mm_key_x { MM_HD uint64_t operator()(const mm128 &a) const { return a.x; } };
struct mm_key_u64 { MM_HD uint64_t operator()(const uint64_t &a) const { return a; } };

// ---- U:mmpriv.h::mg_log2 and U:lchain.c::comput_sc (float32, no FMA contraction) ----
MM_HD float mm_log2f_approx(float x)
{
	union { float f; uint32_t i; } z = { x };
	float log_2 = (float)((int)((z.i >> 23) & 255) - 128);
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	float t = -0.34484843f * z.f;
	t = t + 2.02466578f;
	t = t * z.f;
	t = t - 0.67487759f;
	log_2 = log_2 + t;
	return log_2;
}

#define MM355_SC_NONE INT32_MIN
MM_HD int32_t mm_comput_sc(uint64_t xi, uint64_t yi, uint64_t xj, uint64_t yj, int32_t max_dist_x, int32_t max_dist_y, int32_t bw, float pen_gap, float pen_skip)
{
	int32_t dq = (int32_t)yi - (int32_t)yj, dr, dd, dg, q_span, sc;
	if (dq <= 0 || dq > max_dist_x) return MM355_SC_NONE;
	dr = (int32_t)(xi - xj);
	if (dr == 0 || dq > max_dist_y) return MM355_SC_NONE;
	dd = dr > dq? dr - dq : dq - dr;
	if (dd > bw) return MM355_SC_NONE;
	dg = dr < dq? dr : dq;
	q_span = (int32_t)(yj>>32&0xff);
	sc = q_span < dg? q_span : dg;
	if (dd || dg > q_span) {
		float lin_pen, log_pen, a1, a2;
		a1 = pen_gap * (float)dd;
		a2 = pen_skip * (float)dg;
		lin_pen = a1 + a2;
		log_pen = dd >= 1? mm_log2f_approx((float)(dd + 1)) : 0.0f;
		a1 = .5f * log_pen;
		a1 = lin_pen + a1;
		sc -= (int)a1;
	}
	return sc;
}
