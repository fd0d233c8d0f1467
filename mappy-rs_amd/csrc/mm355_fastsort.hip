// mm355_fastsort.hip -- row a6 for anchor-rich reads (GRCh38-scale): the anchors of a read are sorted by x with the unstable in-place
// radix sort of U:ksort.h (radix_sort_128x), whose only observable difference from ANY other sort by x is the order of anchors with
// EQUAL x.  Reads without equal keys (≈95 % of them even on a repeat-rich genome) therefore get the same array from a plain parallel
// sort.  This file sorts every read out of place with one rocPRIM segmented radix sort (all reads at once, HBM-bound), flags the
// reads that contain equal keys, and copies the result back for the others; the flagged reads go through the literal emulation
// (k_sort_anchors / k_sort_level_mw in mm355_kernels.hip), starting from their untouched generation-order anchors.
#include <cstring>
#include <cstdio>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "mm355_pipeline.h"

// The sort key is the anchor's x = strand << 63 | rid << 32 | rpos with its zero bits squeezed out (strand | rid in rb bits | rpos in pb
// bits: 34 bits for GRCh38 instead of 64), an order-preserving bijection: the radix sort then runs half the digit passes.
// With the index of the read on top of that (read << kb | key: 46 bits for 4096 reads) ONE device-wide radix sort orders every read's
// anchors at once and leaves the reads where they were: the library's global onesweep sort moves the data in far fewer, fully
// coalesced passes than a segmented sort that handles each read in its own block.
struct FsKey { int pb, rb, kb; };
__device__ __forceinline__ uint64_t fs_pack(uint64_t x, FsKey k) { return (x >> 63) << (k.rb + k.pb) | ((x >> 32) & 0x7fffffffULL) << k.pb | (x & 0xffffffffULL); }
__device__ __forceinline__ uint64_t fs_unpack(uint64_t p, FsKey k) { return ((p >> (k.rb + k.pb)) & 1ULL) << 63 | ((p >> k.pb) & ((1ULL << k.rb) - 1)) << 32 | (p & ((1ULL << k.pb) - 1)); }

__global__ __launch_bounds__(256) void k_fs_split(const int64_t *aoff, const mm128 *a, uint64_t *kx, uint64_t *ky, int n_reads, FsKey fk)
{
	const int r = blockIdx.x;
	if (r >= n_reads) return;
	const int64_t b = aoff[r], e = aoff[r + 1];
	const uint64_t top = (uint64_t)r << fk.kb;
	for (int64_t i = b + threadIdx.x; i < e; i += 256) { const mm128 v = a[i]; kx[i] = top | fs_pack(v.x, fk); ky[i] = v.y; }
}

// one block per read: does the sorted key array of the read contain two equal neighbours?
// (also writes tf[i] = 1 where sorted element i equals its left neighbour inside the read; the inclusive scan of tf is the tcnt[] the
// literal sort uses to skip buckets without equal keys)
__global__ __launch_bounds__(256) void k_fs_ties(const int64_t *aoff, const uint64_t *kx, uint8_t *flag, int32_t *tf, int n_reads)
{
	const int r = blockIdx.x;
	if (r >= n_reads) return;
	const int64_t b = aoff[r], e = aoff[r + 1];
	bool tie = false;
	for (int64_t i = b + threadIdx.x; i < e; i += 256) { const bool t = i > b && kx[i] == kx[i - 1]; tf[i] = t? 1 : 0; tie |= t; }
	const int any = __syncthreads_or(tie);
	// (an array of up to 64 elements is insertion-sorted by the reference: stable, i.e. what the plain sort already produced)
	if (threadIdx.x == 0) flag[r] = any && e - b > 64? 1 : 0;
}

__global__ __launch_bounds__(256) void k_fs_merge(const int64_t *aoff, const uint64_t *kx, const uint64_t *ky, mm128 *a, const uint8_t *flag, int n_reads, FsKey fk)
{
	const int r = blockIdx.x;
	if (r >= n_reads || flag[r]) return;   // reads with equal keys keep their generation-order anchors for the literal sort
	const int64_t b = aoff[r], e = aoff[r + 1];
	for (int64_t i = b + threadIdx.x; i < e; i += 256) { mm128 v; v.x = fs_unpack(kx[i], fk); v.y = ky[i]; a[i] = v; }
}

// reads sorted literally (flag = 1): only the positions inside equal-key runs keep what the emulation produced; every other position has a
// unique occupant, taken from the plain sort (the emulation skipped the buckets that contain no equal keys)
__global__ __launch_bounds__(256) void k_fs_fix(const int64_t *aoff, const uint64_t *kx, const uint64_t *ky, mm128 *a, const uint8_t *flag, int n_reads, FsKey fk)
{
	const int r = blockIdx.x;
	if (r >= n_reads || !flag[r]) return;
	const int64_t b = aoff[r], e = aoff[r + 1];
	for (int64_t i = b + threadIdx.x; i < e; i += 256) {
		const uint64_t x = kx[i];
		const bool in_run = (i > b && kx[i - 1] == x) || (i + 1 < e && kx[i + 1] == x);
		if (!in_run) { mm128 v; v.x = fs_unpack(x, fk); v.y = ky[i]; a[i] = v; }
	}
}

static FsKey fs_key(const mm355_ctx *c)   // bit widths of rpos (longest contig) and rid (number of contigs)
{
	uint32_t max_len = 1;
	for (uint32_t l : c->mi->seq_len) if (l > max_len) max_len = l;
	FsKey k; k.pb = 1; k.rb = 1;
	while (k.pb < 32 && (1ULL << k.pb) < (uint64_t)max_len + 1) ++k.pb;
	while (k.rb < 31 && (1ULL << k.rb) < (uint64_t)c->mi->n_seq) ++k.rb;
	k.kb = 1 + k.rb + k.pb;
	return k;
}

int mm355_fast_sort_fix(mm355_ctx *c, int n_reads)
{
	if (n_reads <= 0 || c->hb.tot_a <= 0) return 0;
	const int64_t tot = c->hb.tot_a;
	const FsKey fk = fs_key(c);
	const uint64_t *kx_out = c->wk.as<uint64_t>(), *ky_out = kx_out + tot;
	hipLaunchKernelGGL(k_fs_fix, dim3((unsigned)n_reads), dim3(256), 0, c->st, c->aoff.as<int64_t>(), kx_out, ky_out, c->a.as<mm128>(), c->sort_flag.as<uint8_t>(), n_reads, fk);
	return hipGetLastError() == hipSuccess? 0 : MM355_EHIP;
}

// sorts a[] of every tie-free read; h_flag[r] = 1 for the reads that still have to be sorted literally.  Scratch: b[] and wk[] (16 B per anchor each).
int mm355_fast_sort(mm355_ctx *c, int64_t tot, int n_reads, std::vector<uint8_t> &h_flag)
{
	h_flag.assign((size_t)n_reads, 0);
	if (tot <= 0 || n_reads <= 0) return 0;
	uint64_t *kx_in = c->b.as<uint64_t>(), *ky_in = kx_in + tot, *kx_out = c->wk.as<uint64_t>(), *ky_out = kx_out + tot;
	const int64_t *aoff = c->aoff.as<int64_t>();
	const FsKey fk = fs_key(c);
	int read_bits = 1;
	while ((1LL << read_bits) < (long long)n_reads) ++read_bits;
	hipLaunchKernelGGL(k_fs_split, dim3((unsigned)n_reads), dim3(256), 0, c->st, aoff, c->a.as<mm128>(), kx_in, ky_in, n_reads, fk);
	size_t tb = 0;
	if (c->sort_flag.ensure((size_t)n_reads + 64)) return MM355_ENOMEM;
	if (fk.kb + read_bits <= 64) {   // one device-wide sort, the read index is the top of the key
		const unsigned int end_bit = (unsigned int)(fk.kb + read_bits);
		if (rocprim::radix_sort_pairs(nullptr, tb, kx_in, kx_out, ky_in, ky_out, (size_t)tot, 0u, end_bit, c->st) != hipSuccess) return MM355_EHIP;
		if (c->sort_tmp.ensure(tb + 256)) return MM355_ENOMEM;
		if (rocprim::radix_sort_pairs(c->sort_tmp.p, tb, kx_in, kx_out, ky_in, ky_out, (size_t)tot, 0u, end_bit, c->st) != hipSuccess) return MM355_EHIP;
	} else {                         // (more than 2^(64 - kb) reads in a batch: per-read segments)
		const unsigned int end_bit = (unsigned int)fk.kb;
		if (rocprim::segmented_radix_sort_pairs(nullptr, tb, kx_in, kx_out, ky_in, ky_out, (unsigned int)tot, (unsigned int)n_reads, aoff, aoff + 1, 0u, end_bit, c->st) != hipSuccess) return MM355_EHIP;
		if (c->sort_tmp.ensure(tb + 256)) return MM355_ENOMEM;
		if (rocprim::segmented_radix_sort_pairs(c->sort_tmp.p, tb, kx_in, kx_out, ky_in, ky_out, (unsigned int)tot, (unsigned int)n_reads, aoff, aoff + 1, 0u, end_bit, c->st) != hipSuccess) return MM355_EHIP;
	}
	int32_t *tf = c->p.as<int32_t>(), *tcnt = c->v.as<int32_t>();   // p[] and v[] (4 B per anchor) are free until chaining; v[] = tcnt stays for the literal sort
	hipLaunchKernelGGL(k_fs_ties, dim3((unsigned)n_reads), dim3(256), 0, c->st, aoff, kx_out, c->sort_flag.as<uint8_t>(), tf, n_reads);
	{
		size_t sb = 0;
		if (rocprim::inclusive_scan(nullptr, sb, tf, tcnt, (size_t)tot, rocprim::plus<int32_t>(), c->st) != hipSuccess) return MM355_EHIP;
		if (sb > tb) { if (c->sort_tmp.ensure(sb + 256)) return MM355_ENOMEM; }
		if (rocprim::inclusive_scan(c->sort_tmp.p, sb, tf, tcnt, (size_t)tot, rocprim::plus<int32_t>(), c->st) != hipSuccess) return MM355_EHIP;
	}
	hipLaunchKernelGGL(k_fs_merge, dim3((unsigned)n_reads), dim3(256), 0, c->st, aoff, kx_out, ky_out, c->a.as<mm128>(), c->sort_flag.as<uint8_t>(), n_reads, fk);
	if (hipMemcpyAsync(h_flag.data(), c->sort_flag.p, (size_t)n_reads, hipMemcpyDeviceToHost, c->st) != hipSuccess) return MM355_EHIP;
	if (mm355_wait_stream(c->st) != hipSuccess) return MM355_EHIP;
	return hipGetLastError() == hipSuccess? 0 : MM355_EHIP;
}
