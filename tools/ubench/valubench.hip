// valubench.hip -- what the vector pipes of an MI355X deliver for the instruction mix of the extension kernels: packed int16 add / max,
// 32-bit max, DPP row shifts.  One wave = a chain of N_CHAIN independent dependency chains (ILP), W waves per SIMD by the launch shape.
// Prints wave-instructions per second for every (instruction, waves per SIMD) pair; the roof used by bench.py's "valu" rooflines.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o valubench valubench.hip && ./valubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 4096
typedef short short2_t __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k_valu(uint32_t *out, uint32_t seed)
{
	uint32_t a[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) a[k] = seed * (k + 1) + threadIdx.x;
	const uint32_t b = seed ^ 0x00030005u;
	for (int it = 0; it < ITER; ++it) {
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			if (KIND == 0) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
			else if (KIND == 1) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
			else if (KIND == 2) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
			else if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
			else if (KIND == 4) asm volatile("v_max_i32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(b));
			else if (KIND == 5) asm volatile("v_pk_max_i16 %0, %0, %1\n\tv_pk_add_i16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
		}
	}
	uint32_t s = 0;
#pragma unroll
	for (int k = 0; k < 8; ++k) s ^= a[k];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
static void run(const char *name, int per_instr)
{
	uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int wps = 1; wps <= 8; wps *= 2) {            // waves per SIMD: blocks of 256 threads (4 waves = one per SIMD), wps blocks per CU
		const int blocks = 256 * wps;
		hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
		hipDeviceSynchronize();
		hipEventRecord(e0);
		for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), 0, 0, d, 12345u + r);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		const double winstr = 5.0 * blocks * 4 * (double)ITER * 8 * per_instr;
		printf("%-28s waves/SIMD %d  %.3f T wave-instr/s  (%.2f cycles per instruction per SIMD at 2.4 GHz)\n", name, wps, winstr / (ms * 1e-3) / 1e12,
		       2.4e9 * 1024 / (winstr / (ms * 1e-3)));
	}
	hipFree(d);
}
int main()
{
	run<0>("v_pk_add_i16", 1); run<1>("v_pk_max_i16", 1); run<2>("v_max_i32", 1); run<3>("v_add_u32", 1); run<4>("v_max_i32 dpp row_shr:1", 1); run<5>("v_pk_max_i16 + v_pk_add_i16", 2);
	return 0;
}
