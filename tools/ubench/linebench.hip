// linebench.hip -- what the memory system of an MI355X delivers for UNIFORMLY RANDOM line reads of a table far larger than the
// Infinity Cache: the practical roof of k_seed_lookup (one 128-byte line per minimizer out of a 16 GB table; 64-byte buckets as the
// alternative layout).  Eight lanes fetch one 128-B line (16 B each) or four lanes one 64-B bucket; LINES_IN_FLIGHT independent fetches
// per lane group; a streaming read of the same table for reference.  Output: GB/s of fetched bytes (requested granule x requests).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o linebench linebench.hip && ./linebench [table GiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int GRAN, int UNROLL>    // GRAN = 128 or 64 bytes per request; lanes per request = GRAN / 16
__global__ __launch_bounds__(256) void k_lines(const uint4 *tab, uint64_t n_gran, uint64_t n_req_per_group, uint32_t *out, uint64_t seed)
{
	const int LPR = GRAN / 16;
	const uint64_t group = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / LPR;
	const int sub = threadIdx.x % LPR;
	uint32_t acc = 0;
	for (uint64_t it = 0; it < n_req_per_group; it += UNROLL) {
		uint4 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) {
			const uint64_t g = mix(seed + group * n_req_per_group + it + u) % n_gran;
			v[u] = tab[g * LPR + sub];
		}
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
	}
	if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_stream(const uint4 *tab, uint64_t n16, uint32_t *out)
{
	uint32_t acc = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) { const uint4 v = tab[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
	if (acc == 0x12345678u) out[0] = acc;
}
template <int GRAN, int UNROLL>
static double run(const uint4 *tab, uint64_t bytes, uint32_t *out, int blocks_per_cu)
{
	const uint64_t n_gran = bytes / GRAN, groups = (uint64_t)256 * blocks_per_cu * 256 / (GRAN / 16), per = 4096 / UNROLL * UNROLL;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL((k_lines<GRAN, UNROLL>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, tab, n_gran, per, out, 1ull);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL((k_lines<GRAN, UNROLL>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, tab, n_gran, per, out, 77ull);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	return (double)groups * per * GRAN / (ms * 1e-3) / 1e9;
}
int main(int argc, char **argv)
{
	const uint64_t gib = argc > 1? strtoull(argv[1], 0, 10) : 16, bytes = gib << 30;
	uint4 *tab; uint32_t *out;
	if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
	hipMemset(tab, 1, bytes);
	hipDeviceSynchronize();
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(256), 0, 0, tab, bytes / 16, out); hipDeviceSynchronize();
	hipEventRecord(e0); hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(256), 0, 0, tab, bytes / 16, out); hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	printf("{\"table_GiB\": %llu, \"streaming_read_GBs\": %.1f", (unsigned long long)gib, bytes / (ms * 1e-3) / 1e9);
	double best128 = 0, best64 = 0;
	for (int bpc = 2; bpc <= 8; bpc *= 2) {
		const double a = run<128, 4>(tab, bytes, out, bpc), b = run<128, 8>(tab, bytes, out, bpc), c = run<128, 16>(tab, bytes, out, bpc);
		const double d = run<64, 4>(tab, bytes, out, bpc), e = run<64, 8>(tab, bytes, out, bpc), f = run<64, 16>(tab, bytes, out, bpc);
		printf(", \"random_128B_bpc%d_unroll4_8_16\": [%.1f, %.1f, %.1f], \"random_64B_bpc%d_unroll4_8_16\": [%.1f, %.1f, %.1f]", bpc, a, b, c, bpc, d, e, f);
		if (a > best128) best128 = a; if (b > best128) best128 = b; if (c > best128) best128 = c;
		if (d > best64) best64 = d; if (e > best64) best64 = e; if (f > best64) best64 = f;
	}
	printf(", \"random_128B_lines_GBs\": %.1f, \"random_64B_buckets_GBs\": %.1f}\n", best128, best64);
	return 0;
}
