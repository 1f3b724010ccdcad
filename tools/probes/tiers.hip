// probe: cost per chain-slot visit (12 dwords per lane: load + 6 fp64 FMAs) from the three tiers, one wave per SIMD, 4 waves per CU, all CUs busy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
#define AREAD(lo, hi, i) asm volatile("v_accvgpr_read_b32 %0, a[%c2]\n\tv_accvgpr_read_b32 %1, a[%c3]" : "=v"(lo), "=v"(hi) : "i"(2 * (i)), "i"(2 * (i) + 1))
template <int MODE>
__global__ void __launch_bounds__(256, 1) tier(double *out, const double *gsrc, int iters, unsigned long long *cyc)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	double *s_hl = (double *)smem + (size_t)wave * 10 * 64 * 6;
	for (int i = lane; i < 10 * 64 * 6; i += 64) s_hl[i] = 1.0 + 1e-3 * i;
	asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23");
	for (int i = 0; i < 24; i++) asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_write_b32 a3, %0\n\tv_accvgpr_write_b32 a4, %0\n\tv_accvgpr_write_b32 a5, %0\n\tv_accvgpr_write_b32 a6, %0\n\tv_accvgpr_write_b32 a7, %0\n\tv_accvgpr_write_b32 a8, %0\n\tv_accvgpr_write_b32 a9, %0\n\tv_accvgpr_write_b32 a10, %0\n\tv_accvgpr_write_b32 a11, %0\n\tv_accvgpr_write_b32 a12, %0\n\tv_accvgpr_write_b32 a13, %0\n\tv_accvgpr_write_b32 a14, %0\n\tv_accvgpr_write_b32 a15, %0\n\tv_accvgpr_write_b32 a16, %0\n\tv_accvgpr_write_b32 a17, %0\n\tv_accvgpr_write_b32 a18, %0\n\tv_accvgpr_write_b32 a19, %0\n\tv_accvgpr_write_b32 a20, %0\n\tv_accvgpr_write_b32 a21, %0\n\tv_accvgpr_write_b32 a22, %0\n\tv_accvgpr_write_b32 a23, %0" ::"v"(0x3ff00000));
	__syncthreads();
	double v[6] = {1.0, 1.1, 1.2, 1.3, 1.4, 1.5}, acc[4] = {0, 0, 0, 0};
	const double *g = gsrc + ((size_t)(blockIdx.x * 4 + wave) * 20) * 64 * 6;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; it++) {
		if (MODE == 0) {   // AGPR: 2 slots per trip (static registers)
#pragma unroll
			for (int sl = 0; sl < 2; sl++) {
				double h[6];
#pragma unroll
				for (int e = 0; e < 6; e++) { int lo, hi; if (sl == 0) { switch (e) { case 0: AREAD(lo, hi, 0); break; case 1: AREAD(lo, hi, 1); break; case 2: AREAD(lo, hi, 2); break; case 3: AREAD(lo, hi, 3); break; case 4: AREAD(lo, hi, 4); break; default: AREAD(lo, hi, 5); } } else { switch (e) { case 0: AREAD(lo, hi, 6); break; case 1: AREAD(lo, hi, 7); break; case 2: AREAD(lo, hi, 8); break; case 3: AREAD(lo, hi, 9); break; case 4: AREAD(lo, hi, 10); break; default: AREAD(lo, hi, 11); } } h[e] = __hiloint2double(hi, lo); }
				double a0 = 0, a1 = 0;
#pragma unroll
				for (int e = 0; e < 6; e++) { if (e & 1) a1 += h[e] * v[e]; else a0 += h[e] * v[e]; }
				acc[sl] += a0 + a1;
			}
		} else if (MODE == 1) {   // LDS: 2 slots per trip, loads of both issued first
			double h[2][6];
#pragma unroll
			for (int sl = 0; sl < 2; sl++) {
				const d2 *p = (const d2 *)(s_hl + ((size_t)((it * 2 + sl) % 10) * 64 + lane) * 6);
#pragma unroll
				for (int e = 0; e < 3; e++) { const d2 t = p[e]; h[sl][2 * e] = t.x; h[sl][2 * e + 1] = t.y; }
			}
#pragma unroll
			for (int sl = 0; sl < 2; sl++) {
				double a0 = 0, a1 = 0;
#pragma unroll
				for (int e = 0; e < 6; e++) { if (e & 1) a1 += h[sl][e] * v[e]; else a0 += h[sl][e] * v[e]; }
				acc[sl] += a0 + a1;
			}
		} else {   // global (L2 / MALL resident per wave: 20 slots of 3 KB), 4 slots per trip, dwordx4 loads of all four issued first
			double h[4][6];
#pragma unroll
			for (int sl = 0; sl < 4; sl++) {
				const d2 *p = (const d2 *)(g + (size_t)((it * 4 + sl) % 20) * 64 * 6);
#pragma unroll
				for (int e = 0; e < 3; e++) { const d2 t = p[e * 64 + lane]; h[sl][2 * e] = t.x; h[sl][2 * e + 1] = t.y; }
			}
#pragma unroll
			for (int sl = 0; sl < 4; sl++) {
				double a0 = 0, a1 = 0;
#pragma unroll
				for (int e = 0; e < 6; e++) { if (e & 1) a1 += h[sl][e] * v[e]; else a0 += h[sl][e] * v[e]; }
				acc[sl] += a0 + a1;
			}
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
	if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
	double *dd, *gs; unsigned long long *dc, c;
	hipMalloc(&dd, 256 * 256 * 8); hipMalloc(&dc, 8);
	const size_t gbytes = (size_t)256 * 4 * 20 * 64 * 6 * 8;
	hipMalloc(&gs, gbytes); hipMemset(gs, 0, gbytes);
	const int iters = 5000, lds = 4 * 10 * 64 * 6 * 8;
	hipFuncSetAttribute((const void *)tier<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
	hipFuncSetAttribute((const void *)tier<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
	hipFuncSetAttribute((const void *)tier<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
	for (int rep = 0; rep < 2; rep++) {
		hipLaunchKernelGGL(tier<0>, dim3(256), dim3(256), lds, 0, dd, gs, iters, dc); hipDeviceSynchronize(); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
		printf("AGPR  tier: %7.1f cycles per slot visit (12 accvgpr_read + 6 fma + 1 add)\n", (double)c / iters / 2);
		hipLaunchKernelGGL(tier<1>, dim3(256), dim3(256), lds, 0, dd, gs, iters, dc); hipDeviceSynchronize(); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
		printf("LDS   tier: %7.1f cycles per slot visit (3 ds_read_b128 + 6 fma + 1 add, 2 slots in flight)\n", (double)c / iters / 2);
		hipLaunchKernelGGL(tier<2>, dim3(256), dim3(256), lds, 0, dd, gs, iters, dc); hipDeviceSynchronize(); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
		printf("L2/MALL tier: %7.1f cycles per slot visit (3 global_load_dwordx4 + 6 fma + 1 add, 4 slots in flight, 61 MB footprint)\n", (double)c / iters / 4);
	}
	return 0;
}
