// mm355_idxbuild.hip -- device index builder (replaces U:index.c::mm_idx_gen, reached in the reference
// through mm_idx_reader_read at /root/reference/src/lib.rs:407-410 when the input is a FASTA).
//   1. every contig is sketched by the chunked minimizer kernel (same machine as the read path, rid = contig id);
//   2. (minimizer, position) pairs are appended in (rid, pos) order, so ONE stable device radix sort by minimizer leaves
//      the positions of a minimizer ascending -- the order U:index.c::worker_post produces with radix_sort_64;
//   3. run detection + scans give per-minimizer counts and pos[] offsets; a CAS kernel fills the 128-B-line table.
// The table and pos[] never leave HBM (a GRCh38 table is ~16 GB); the host keeps S (4-bit bases), names and the tail of
// the occurrence-count distribution (for mm_idx_cal_max_occ).
#include <cstring>
#include <cstdio>
#include <algorithm>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "mm355_pipeline.h"
#include "mm355_sketch.h"

#define WAVE 64

template <bool HPC>
__device__ __forceinline__ void sketch_contig_chunk(const uint8_t *seq, int len, int w, int k, int n_chunks, mm128 *slots, int32_t *chunk_n)
{
	extern __shared__ mm128 ring[];
	int t = blockIdx.x * WAVE + threadIdx.x;
	if (t >= n_chunks) return;
	const int cs = t * SK_CHUNK;
	int ce = cs + SK_CHUNK;
	if (ce > len) ce = len;
	chunk_n[t] = sketch_chunk<HPC>(seq, len, w, k, cs, ce, slots + cs, ring + threadIdx.x, WAVE);
}
__global__ __launch_bounds__(WAVE) void k_sketch_contig(const uint8_t *seq, int len, int w, int k, int n_chunks, mm128 *slots, int32_t *chunk_n)
{
	sketch_contig_chunk<false>(seq, len, w, k, n_chunks, slots, chunk_n);
}
__global__ __launch_bounds__(WAVE) void k_sketch_contig_hpc(const uint8_t *seq, int len, int w, int k, int n_chunks, mm128 *slots, int32_t *chunk_n)   // MM_I_HPC
{
	sketch_contig_chunk<true>(seq, len, w, k, n_chunks, slots, chunk_n);
}

__global__ __launch_bounds__(WAVE) void k_gather_pairs(const mm128 *slots, const int32_t *chunk_n, const uint32_t *chunk_off, int n_chunks, uint32_t rid,
                                                       uint64_t *keys, uint64_t *vals, uint64_t base, uint64_t cap, int *err)
{
	const int c = blockIdx.x;
	if (c >= n_chunks) return;
	const int n = chunk_n[c];
	const uint64_t dst = base + chunk_off[c];
	if (dst + n > cap) { if (threadIdx.x == 0) *err = 1; return; }
	const mm128 *src = slots + (size_t)c * SK_CHUNK;
	for (int i = threadIdx.x; i < n; i += WAVE) {
		mm128 m = src[i];
		keys[dst + i] = m.x >> 8;
		vals[dst + i] = (uint64_t)rid << 32 | (uint32_t)m.y;
	}
}

__global__ void k_pack4(const uint8_t *seq, int64_t len, uint64_t off, uint32_t *S)
{   // 8 bases per u32; contigs are concatenated without padding, so neighbouring contigs can share a word: atomicOr
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // i indexes output words of this contig's span
	uint64_t w0 = off >> 3, w = w0 + (uint64_t)i;
	if (w > (off + (uint64_t)len - 1) >> 3) return;
	uint32_t v = 0;
	for (int j = 0; j < 8; ++j) {
		uint64_t o = (w << 3) + j;
		if (o >= off && o < off + (uint64_t)len) v |= (uint32_t)mm_nt4(seq[o - off]) << (j << 2);
	}
	if (v) atomicOr(&S[w], v);
}

__global__ void k_run_flags(const uint64_t *keys, uint64_t n, uint32_t *flag)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1])? 1u : 0u;
}
__global__ void k_run_starts(const uint32_t *flag, const uint32_t *run_id, uint64_t n, uint64_t *starts)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && flag[i]) starts[run_id[i] - 1] = i;   // run_id = inclusive scan of the start flags = run index + 1
}
__global__ void k_run_lens(const uint64_t *starts, uint64_t n_runs, uint64_t n, uint32_t *len, uint64_t *multi)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_runs) return;
	uint64_t e = r + 1 < n_runs? starts[r + 1] : n;
	uint32_t l = (uint32_t)(e - starts[r]);
	len[r] = l; multi[r] = l > 1? l : 0;
}
__global__ void k_table_insert(const uint64_t *keys, const uint64_t *vals, const uint64_t *starts, const uint32_t *len, const uint64_t *moff, uint64_t n_runs,
                               mm355_slot *slots, uint64_t line_mask)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_runs) return;
	const uint64_t s = starts[r], minier = keys[s];
	const uint32_t n = len[r];
	const uint64_t key = minier << 1 | (n == 1? 1 : 0), val = n == 1? vals[s] : (moff[r] << 32 | n);
	uint64_t line = mm_table_hash(minier) & line_mask;
	for (;;) {
		mm355_slot *ln = slots + line * MM355_SLOTS_PER_LINE;
		for (int q = 0; q < MM355_SLOTS_PER_LINE; ++q) {
			unsigned long long old = atomicCAS((unsigned long long*)&ln[q].key, ~0ULL, (unsigned long long)key);
			if (old == ~0ULL) { ln[q].val = val; return; }
		}
		line = (line + 1) & line_mask;
	}
}
__global__ void k_fill_pos(const uint64_t *vals, const uint32_t *run_id, const uint64_t *starts, const uint32_t *len, const uint64_t *moff, uint64_t n, uint64_t *pos)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t r = run_id[i] - 1;
	if (len[r] > 1) pos[moff[r] + (i - starts[r])] = vals[i];
}

static void free_build_buffers(mm355_index *mi)   // a failed build: nothing was registered as a replica yet
{
	(void)hipSetDevice(mi->dev_id);
	if (mi->d_slots) (void)hipFree(mi->d_slots);
	if (mi->d_pos) (void)hipFree(mi->d_pos);
	if (mi->d_S) (void)hipFree(mi->d_S);
	mi->d_slots = mi->d_pos = mi->d_S = 0;
}

#define GRID(n, b) dim3((unsigned)(((n) + (b) - 1) / (b)))
#include <chrono>
static double ib_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define IB_LOG(what) do { if (getenv("MM355_VERBOSE")) { (void)hipStreamSynchronize(st); fprintf(stderr, "[mm355] index build: %-28s %.2f s\n", what, ib_now() - t_ib); t_ib = ib_now(); } } while (0)

extern "C" int mm355_index_build_device(const mm355_idxopt_t *io, int n_seq, const uint8_t *const *seqs, const int64_t *lens, const char *const *names,
                                        int device, mm355_index_t **out)
{
	*out = 0;
	if (n_seq <= 0) return MM355_EINVAL;
	if (io->k <= 0 || io->k > 28 || io->w <= 0 || io->w >= 256) return MM355_EINVAL;   // U:sketch.c::mm_sketch asserts the same ranges (e.g. options never initialised with mm355_set_opt(NULL, ..))
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) return MM355_ENODEV;
	HIPCHK(hipSetDevice(device));
	hipStream_t st; HIPCHK(hipStreamCreate(&st));
	double t_ib = ib_now();
	mm355_index *mi = new mm355_index();
	mi->w = io->w < 1? 1 : io->w; mi->k = io->k; mi->b = io->bucket_bits; mi->flag = io->flag; mi->n_seq = n_seq;
	if (mi->k * 2 < mi->b) mi->b = mi->k * 2;
	uint64_t sum_len = 0; int64_t max_len = 0;
	for (int i = 0; i < n_seq; ++i) {
		if (lens[i] > 0x7fffffffLL) { delete mi; return MM355_EINVAL; }
		mi->names.emplace_back(names && names[i]? names[i] : "");
		mi->seq_off.push_back(sum_len); mi->seq_len.push_back((uint32_t)lens[i]);
		sum_len += lens[i]; max_len = std::max(max_len, lens[i]);
	}
	for (uint32_t i = 0; i < mi->n_seq; ++i) mi->name2id.emplace(mi->names[i], (int)i);
	const uint64_t cap = (uint64_t)((double)sum_len * 2.0 / (mi->w + 1) * 1.25) + (1u << 20);
	const int64_t max_chunks = (max_len + SK_CHUNK - 1) / SK_CHUNK;
	DBuf d_seq, d_slots16, d_cn, d_co, d_keys, d_vals, d_keys2, d_vals2, d_tmp, d_err;
	size_t Sw = (sum_len + 7) / 8 + 2;
	int rc = 0;
	uint64_t n = 0;
	void *dS = 0;
#define FAIL(code) do { rc = (code); if (getenv("MM355_VERBOSE")) fprintf(stderr, "[mm355] index build failed at %s:%d (%s)\n", __FILE__, __LINE__, hipGetErrorString(hipGetLastError())); goto done; } while (0)
	if (hipMalloc(&dS, Sw * 4) != hipSuccess) FAIL(MM355_ENOMEM);
	(void)hipMemsetAsync(dS, 0, Sw * 4, st);
	if (d_seq.ensure((size_t)max_len + 64) || d_slots16.ensure(((size_t)max_len + 64) * 16) || d_cn.ensure((max_chunks + 1) * 4) || d_co.ensure((max_chunks + 1) * 4) ||
	    d_keys.ensure(cap * 8) || d_vals.ensure(cap * 8) || d_err.ensure(16)) FAIL(MM355_ENOMEM);
	(void)hipMemsetAsync(d_err.p, 0, 16, st);
	IB_LOG("alloc");
	for (int i = 0; i < n_seq; ++i) {
		const int64_t len = lens[i];
		if (len <= 0) continue;
		const int nch = (int)((len + SK_CHUNK - 1) / SK_CHUNK);
		if (hipMemcpyAsync(d_seq.p, seqs[i], (size_t)len, hipMemcpyHostToDevice, st) != hipSuccess) FAIL(MM355_EHIP);
		(void)hipMemsetAsync((uint8_t*)d_seq.p + len, 4, 32, st);   // padding reads as 'N'
		hipLaunchKernelGGL(k_pack4, GRID((len >> 3) + 2, 256), dim3(256), 0, st, d_seq.as<uint8_t>(), len, mi->seq_off[i], (uint32_t*)dS);
		hipLaunchKernelGGL((mi->flag & 1)? k_sketch_contig_hpc : k_sketch_contig, GRID(nch, WAVE), dim3(WAVE), (size_t)mi->w * WAVE * sizeof(mm128), st, d_seq.as<uint8_t>(), (int)len, mi->w, mi->k, nch,
		                   d_slots16.as<mm128>(), d_cn.as<int32_t>());
		if (hipGetLastError() != hipSuccess) FAIL(MM355_EHIP);   // k_pack4 / k_sketch_contig launch
		size_t tb = 0;
		(void)rocprim::exclusive_scan(nullptr, tb, d_cn.as<uint32_t>(), d_co.as<uint32_t>(), 0u, (size_t)nch + 1, rocprim::plus<uint32_t>(), st);
		if (d_tmp.ensure(tb + 256)) FAIL(MM355_ENOMEM);
		(void)hipMemsetAsync(d_cn.as<int32_t>() + nch, 0, 4, st);
		if (rocprim::exclusive_scan(d_tmp.p, tb, d_cn.as<uint32_t>(), d_co.as<uint32_t>(), 0u, (size_t)nch + 1, rocprim::plus<uint32_t>(), st) != hipSuccess) FAIL(MM355_EHIP);
		hipLaunchKernelGGL(k_gather_pairs, dim3((unsigned)nch), dim3(WAVE), 0, st, d_slots16.as<mm128>(), d_cn.as<int32_t>(), d_co.as<uint32_t>(), nch, (uint32_t)i,
		                   d_keys.as<uint64_t>(), d_vals.as<uint64_t>(), n, cap, d_err.as<int>());
		if (hipGetLastError() != hipSuccess) FAIL(MM355_EHIP);   // k_gather_pairs launch
		uint32_t tot = 0;
		if (hipMemcpyAsync(&tot, d_co.as<uint32_t>() + nch, 4, hipMemcpyDeviceToHost, st) != hipSuccess) FAIL(MM355_EHIP);
		if (hipStreamSynchronize(st) != hipSuccess) FAIL(MM355_EHIP);
		if (getenv("MM355_VERBOSE") && i < 4) fprintf(stderr, "[mm355] contig %d: len %lld, %d chunks, %u minimizers\n", i, (long long)len, nch, tot);
		n += tot;
		if (n > cap) FAIL(MM355_ENOMEM);
	}
	{
		int e = 0;
		(void)hipMemcpy(&e, d_err.p, 4, hipMemcpyDeviceToHost);
		if (e) FAIL(MM355_ENOMEM);
	}
	IB_LOG("sketch contigs");
	d_seq.release(); d_slots16.release();
	mi->n_minimizers = (int64_t)n;
	if (n == 0) FAIL(MM355_EIO);
	{
		// stable sort by minimizer (2k bits); values (rid,pos) stay ascending inside a run
		if (d_keys2.ensure(n * 8) || d_vals2.ensure(n * 8)) FAIL(MM355_ENOMEM);
		size_t tb = 0;
		(void)rocprim::radix_sort_pairs(nullptr, tb, d_keys.as<uint64_t>(), d_keys2.as<uint64_t>(), d_vals.as<uint64_t>(), d_vals2.as<uint64_t>(), (size_t)n, 0u, (unsigned)(2 * mi->k), st);
		if (d_tmp.ensure(tb + 256)) FAIL(MM355_ENOMEM);
		{ hipError_t e = rocprim::radix_sort_pairs(d_tmp.p, tb, d_keys.as<uint64_t>(), d_keys2.as<uint64_t>(), d_vals.as<uint64_t>(), d_vals2.as<uint64_t>(), (size_t)n, 0u, (unsigned)(2 * mi->k), st);
		  if (e != hipSuccess) { if (getenv("MM355_VERBOSE")) fprintf(stderr, "[mm355] radix_sort_pairs(n=%lld, tmp=%zu): %s\n", (long long)n, tb, hipGetErrorString(e)); FAIL(MM355_EHIP); } }
		if (hipStreamSynchronize(st) != hipSuccess) FAIL(MM355_EHIP);
		IB_LOG("radix sort");
		d_keys.release(); d_vals.release();
		uint64_t *keys = d_keys2.as<uint64_t>(), *vals = d_vals2.as<uint64_t>();
		// runs
		DBuf d_flag, d_rid, d_starts, d_len, d_multi, d_moff;
		if (d_flag.ensure((n + 1) * 4) || d_rid.ensure((n + 1) * 4)) FAIL(MM355_ENOMEM);
		hipLaunchKernelGGL(k_run_flags, GRID(n, 256), dim3(256), 0, st, keys, n, d_flag.as<uint32_t>());
		tb = 0;
		(void)rocprim::inclusive_scan(nullptr, tb, d_flag.as<uint32_t>(), d_rid.as<uint32_t>(), (size_t)n, rocprim::plus<uint32_t>(), st);
		if (d_tmp.ensure(tb + 256)) FAIL(MM355_ENOMEM);
		if (rocprim::inclusive_scan(d_tmp.p, tb, d_flag.as<uint32_t>(), d_rid.as<uint32_t>(), (size_t)n, rocprim::plus<uint32_t>(), st) != hipSuccess) FAIL(MM355_EHIP);
		uint32_t n_runs32 = 0;
		if (hipMemcpyAsync(&n_runs32, d_rid.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, st) != hipSuccess) FAIL(MM355_EHIP);
		if (hipStreamSynchronize(st) != hipSuccess) FAIL(MM355_EHIP);
		const uint64_t n_runs = n_runs32;
		mi->n_distinct = (int64_t)n_runs;
		if (d_starts.ensure((n_runs + 1) * 8) || d_len.ensure((n_runs + 1) * 4) || d_multi.ensure((n_runs + 1) * 8) || d_moff.ensure((n_runs + 1) * 8)) FAIL(MM355_ENOMEM);
		hipLaunchKernelGGL(k_run_starts, GRID(n, 256), dim3(256), 0, st, d_flag.as<uint32_t>(), d_rid.as<uint32_t>(), n, d_starts.as<uint64_t>());
		hipLaunchKernelGGL(k_run_lens, GRID(n_runs, 256), dim3(256), 0, st, d_starts.as<uint64_t>(), n_runs, n, d_len.as<uint32_t>(), d_multi.as<uint64_t>());
		(void)hipMemsetAsync(d_multi.as<uint64_t>() + n_runs, 0, 8, st);
		tb = 0;
		(void)rocprim::exclusive_scan(nullptr, tb, d_multi.as<uint64_t>(), d_moff.as<uint64_t>(), (uint64_t)0, (size_t)n_runs + 1, rocprim::plus<uint64_t>(), st);
		if (d_tmp.ensure(tb + 256)) FAIL(MM355_ENOMEM);
		if (rocprim::exclusive_scan(d_tmp.p, tb, d_multi.as<uint64_t>(), d_moff.as<uint64_t>(), (uint64_t)0, (size_t)n_runs + 1, rocprim::plus<uint64_t>(), st) != hipSuccess) FAIL(MM355_EHIP);
		uint64_t n_pos = 0;
		if (hipMemcpyAsync(&n_pos, d_moff.as<uint64_t>() + n_runs, 8, hipMemcpyDeviceToHost, st) != hipSuccess) FAIL(MM355_EHIP);
		if (hipStreamSynchronize(st) != hipSuccess) FAIL(MM355_EHIP);
		if (n_pos >= (1ULL << 32)) FAIL(MM355_EUNSUP);   // offset<<32|count packing
		mi->n_pos = n_pos;
		IB_LOG("runs + scans");
		// table
		uint64_t want = (uint64_t)(n_runs / 0.55) + MM355_SLOTS_PER_LINE, n_lines = 1;
		while (n_lines * MM355_SLOTS_PER_LINE < want) n_lines <<= 1;
		mi->n_lines = n_lines;
		if (hipMalloc(&mi->d_slots, n_lines * MM355_SLOTS_PER_LINE * sizeof(mm355_slot)) != hipSuccess) FAIL(MM355_ENOMEM);
		if (hipMalloc(&mi->d_pos, (n_pos + 2) * 8) != hipSuccess) FAIL(MM355_ENOMEM);
		(void)hipMemsetAsync(mi->d_slots, 0xff, n_lines * MM355_SLOTS_PER_LINE * sizeof(mm355_slot), st);
		hipLaunchKernelGGL(k_table_insert, GRID(n_runs, 256), dim3(256), 0, st, keys, vals, d_starts.as<uint64_t>(), d_len.as<uint32_t>(), d_moff.as<uint64_t>(), n_runs,
		                   (mm355_slot*)mi->d_slots, n_lines - 1);
		hipLaunchKernelGGL(k_fill_pos, GRID(n, 256), dim3(256), 0, st, vals, d_rid.as<uint32_t>(), d_starts.as<uint64_t>(), d_len.as<uint32_t>(), d_moff.as<uint64_t>(), n,
		                   (uint64_t*)mi->d_pos);
		IB_LOG("table insert + pos fill");
		// occurrence-count tail for mm_idx_cal_max_occ: sort counts descending, keep the top 2M
		DBuf d_len2;
		if (d_len2.ensure((n_runs + 1) * 4)) FAIL(MM355_ENOMEM);
		tb = 0;
		(void)rocprim::radix_sort_keys_desc(nullptr, tb, d_len.as<uint32_t>(), d_len2.as<uint32_t>(), (size_t)n_runs, 0u, 32u, st);
		if (d_tmp.ensure(tb + 256)) FAIL(MM355_ENOMEM);
		if (rocprim::radix_sort_keys_desc(d_tmp.p, tb, d_len.as<uint32_t>(), d_len2.as<uint32_t>(), (size_t)n_runs, 0u, 32u, st) != hipSuccess) FAIL(MM355_EHIP);
		size_t keep = (size_t)std::min<uint64_t>(n_runs, 2u << 20);
		mi->top_counts.resize(keep);
		if (hipMemcpyAsync(mi->top_counts.data(), d_len2.p, keep * 4, hipMemcpyDeviceToHost, st) != hipSuccess) FAIL(MM355_EHIP);
		if (hipStreamSynchronize(st) != hipSuccess) FAIL(MM355_EHIP);
		if (hipGetLastError() != hipSuccess) FAIL(MM355_EHIP);
		d_flag.release(); d_rid.release(); d_starts.release(); d_len.release(); d_multi.release(); d_moff.release(); d_len2.release();
	}
	IB_LOG("count tail");
	mi->S.resize(Sw);
	if (hipMemcpy(mi->S.data(), dS, Sw * 4, hipMemcpyDeviceToHost) != hipSuccess) FAIL(MM355_EHIP);
	IB_LOG("S to host");
	mi->d_S = dS; dS = 0;
	mi->dev_resident = true; mi->dev_id = device;
	{   // the build device holds the first replica; other devices get peer copies (mm355_upload / mm355_ctx_create)
		mm355_replica rp; rp.dev = device; rp.slots = mi->d_slots; rp.pos = mi->d_pos; rp.S = mi->d_S;
		if (hipMalloc(&rp.seq_off, (size_t)n_seq * 8) != hipSuccess || hipMalloc(&rp.seq_len, (size_t)n_seq * 4) != hipSuccess) { if (rp.seq_off) (void)hipFree(rp.seq_off); mi->dev_resident = false; FAIL(MM355_ENOMEM); }
		(void)hipMemcpy(rp.seq_off, mi->seq_off.data(), (size_t)n_seq * 8, hipMemcpyHostToDevice);
		(void)hipMemcpy(rp.seq_len, mi->seq_len.data(), (size_t)n_seq * 4, hipMemcpyHostToDevice);
		{   // 2-bit image + N-run table for the kernels; the 4-bit image leaves HBM (the host keeps it for mm_idx_getseq and the .mmi)
			std::lock_guard<std::mutex> lk(mi->rep_mu);
			rc = mm355_replica_pack2(mi, &rp);
			mi->d_S = rp.S;
		}
		if (rc) {   // (the table buffers still belong to the build: free_build_buffers releases them, the replica only its own pieces)
			(void)hipFree(rp.seq_off); (void)hipFree(rp.seq_len); if (rp.S2) (void)hipFree(rp.S2); if (rp.nr) (void)hipFree(rp.nr);
			mi->dev_resident = false; goto done;
		}
		mi->replicas.push_back(rp);
	}
done:
	d_seq.release(); d_slots16.release(); d_cn.release(); d_co.release(); d_keys.release(); d_vals.release(); d_keys2.release(); d_vals2.release(); d_tmp.release(); d_err.release();
	if (dS) (void)hipFree(dS);
	(void)hipStreamDestroy(st);
	if (rc) { mi->dev_id = device; free_build_buffers(mi); delete mi; return rc; }
	*out = mi;
	return 0;
}
