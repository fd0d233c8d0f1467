// sortbench.hip -- what rocPRIM's device radix sort delivers on an MI355X for the anchor sort of one sub-batch (row a6): n (key, value)
// pairs, 47 significant key bits, with the library's default onesweep configuration and with wider digits / narrower values.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o sortbench sortbench.hip && ./sortbench [n_millions=102] [bits=47]
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
__global__ void k_fill(uint64_t *k, uint64_t *v, uint32_t *v32, size_t n, int bits)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		uint64_t x = i * 0x9e3779b97f4a7c15ULL; x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 32;
		k[i] = x & ((1ULL << bits) - 1); v[i] = i; v32[i] = (uint32_t)i;
	}
}
template <class Config, class V>
static float run(uint64_t *ki, uint64_t *ko, V *vi, V *vo, size_t n, int bits, void *tmp, size_t tmp_cap)
{
	size_t tb = 0;
	const hipError_t e = rocprim::radix_sort_pairs<Config>(nullptr, tb, ki, ko, vi, vo, n, 0u, (unsigned)bits, 0);
	if (e != hipSuccess || tb > tmp_cap) { fprintf(stderr, "size query: %d, %zu bytes\n", (int)e, tb); return -1; }
	rocprim::radix_sort_pairs<Config>(tmp, tb, ki, ko, vi, vo, n, 0u, (unsigned)bits, 0);
	hipDeviceSynchronize();
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	rocprim::radix_sort_pairs<Config>(tmp, tb, ki, ko, vi, vo, n, 0u, (unsigned)bits, 0);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	return ms;
}
template <unsigned BS, unsigned IPT, unsigned RB>
using OS = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                      rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<BS, IPT>, RB, rocprim::block_radix_rank_algorithm::match>, 1024u * 1024u>;
int main(int argc, char **argv)
{
	const size_t n = (size_t)(argc > 1? atof(argv[1]) : 102) * 1000000;
	const int bits = argc > 2? atoi(argv[2]) : 47;
	uint64_t *ki, *ko, *vi, *vo; uint32_t *wi, *wo; void *tmp; const size_t tmp_cap = 4ull << 30;
	hipMalloc(&ki, n * 8); hipMalloc(&ko, n * 8); hipMalloc(&vi, n * 8); hipMalloc(&vo, n * 8); hipMalloc(&wi, n * 4); hipMalloc(&wo, n * 4); hipMalloc(&tmp, tmp_cap);
	hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, ki, vi, wi, n, bits); hipDeviceSynchronize();
	printf("{\"n\": %zu, \"key_bits\": %d", n, bits);
	printf(", \"default_u64_u64_ms\": %.2f", run<rocprim::default_config, uint64_t>(ki, ko, vi, vo, n, bits, tmp, tmp_cap));
	printf(", \"default_u64_u32_ms\": %.2f", run<rocprim::default_config, uint32_t>(ki, ko, wi, wo, n, bits, tmp, tmp_cap));
	printf(", \"os_256x12_8b_u64_ms\": %.2f", run<OS<256, 12, 8>, uint64_t>(ki, ko, vi, vo, n, bits, tmp, tmp_cap));
	printf(", \"os_512x8_8b_u64_ms\": %.2f", run<OS<512, 8, 8>, uint64_t>(ki, ko, vi, vo, n, bits, tmp, tmp_cap));
	printf(", \"os_512x8_9b_u64_ms\": %.2f", run<OS<512, 8, 9>, uint64_t>(ki, ko, vi, vo, n, bits, tmp, tmp_cap));
	printf(", \"os_1024x4_10b_u64_ms\": %.2f", run<OS<1024, 4, 10>, uint64_t>(ki, ko, vi, vo, n, bits, tmp, tmp_cap));
	printf(", \"os_1024x4_10b_u32_ms\": %.2f", run<OS<1024, 4, 10>, uint32_t>(ki, ko, wi, wo, n, bits, tmp, tmp_cap));
	printf(", \"os_1024x6_10b_u32_ms\": %.2f", run<OS<1024, 6, 10>, uint32_t>(ki, ko, wi, wo, n, bits, tmp, tmp_cap));
	printf(", \"os_512x12_8b_u32_ms\": %.2f", run<OS<512, 12, 8>, uint32_t>(ki, ko, wi, wo, n, bits, tmp, tmp_cap));
	printf("}\n");
	return 0;
}
