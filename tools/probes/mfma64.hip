// probe: issue rate / latency and lane layout of the fp64 matrix instructions on gfx950 (one wave per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) layout4(const double *a, const double *b, double *d)
{
	const int l = threadIdx.x;
	d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
template <int MODE>
__global__ void __launch_bounds__(256, 1) rate(double *out, int iters, unsigned long long *cyc)
{
	const int l = threadIdx.x & 63;
	double a = 1.0 + l * 1e-3, b = 2.0 - l * 1e-3;
	double c0 = 0, c1 = 0, c2 = 0, c3 = 0, v0 = a, v1 = b, v2 = a + b, v3 = a - b;
	d4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < iters; i++) {
		if (MODE == 0) {   // 4 independent 4x4x4
			c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
			c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
		} else if (MODE == 1) {   // 4 dependent 4x4x4 (one accumulator)
			c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
			c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
		} else if (MODE == 2) {   // 2 independent 16x16x4 (x2)
			e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e0, 0, 0, 0); e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e1, 0, 0, 0);
			e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e0, 0, 0, 0); e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e1, 0, 0, 0);
		} else if (MODE == 3) {   // 4 independent 4x4x4 + 16 independent fp64 FMAs on the VALU
			c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); v0 = fma(v0, a, b); v1 = fma(v1, a, b); v2 = fma(v2, a, b); v3 = fma(v3, a, b);
			c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0); v0 = fma(v0, a, b); v1 = fma(v1, a, b); v2 = fma(v2, a, b); v3 = fma(v3, a, b);
			c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0); v0 = fma(v0, a, b); v1 = fma(v1, a, b); v2 = fma(v2, a, b); v3 = fma(v3, a, b);
			c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0); v0 = fma(v0, a, b); v1 = fma(v1, a, b); v2 = fma(v2, a, b); v3 = fma(v3, a, b);
		} else if (MODE == 4) {   // 16 independent fp64 FMAs alone
			v0 = fma(v0, a, b); v1 = fma(v1, a, b); v2 = fma(v2, a, b); v3 = fma(v3, a, b); v0 = fma(v0, a, b); v1 = fma(v1, a, b); v2 = fma(v2, a, b); v3 = fma(v3, a, b);
			v0 = fma(v0, a, b); v1 = fma(v1, a, b); v2 = fma(v2, a, b); v3 = fma(v3, a, b); v0 = fma(v0, a, b); v1 = fma(v1, a, b); v2 = fma(v2, a, b); v3 = fma(v3, a, b);
		} else if (MODE == 5) {   // 16 dependent fp64 FMAs (one chain)
#pragma unroll
			for (int u = 0; u < 16; u++) v0 = fma(v0, a, b);
		} else if (MODE == 6) {   // 16 x (2 v_accvgpr_read + fma) : emulated with v_mov
#pragma unroll
			for (int u = 0; u < 4; u++) {
				int lo, hi;
				asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1" : "=v"(lo), "=v"(hi));
				v0 = fma(__hiloint2double(hi, lo), a, v0);
				asm volatile("v_accvgpr_read_b32 %0, a2\n\tv_accvgpr_read_b32 %1, a3" : "=v"(lo), "=v"(hi));
				v1 = fma(__hiloint2double(hi, lo), a, v1);
				asm volatile("v_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a5" : "=v"(lo), "=v"(hi));
				v2 = fma(__hiloint2double(hi, lo), a, v2);
				asm volatile("v_accvgpr_read_b32 %0, a6\n\tv_accvgpr_read_b32 %1, a7" : "=v"(lo), "=v"(hi));
				v3 = fma(__hiloint2double(hi, lo), a, v3);
			}
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + v0 + v1 + v2 + v3 + e0[0] + e0[1] + e0[2] + e0[3] + e1[0] + e1[1] + e1[2] + e1[3];
	if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
	double *da, *db, *dd; unsigned long long *dc;
	hipMalloc(&da, 64 * 8); hipMalloc(&db, 64 * 8); hipMalloc(&dd, 2048 * 256 * 8); hipMalloc(&dc, 8);
	// layout: A one-hot at lane la, B one-hot at lane lb -> where does the product land?
	std::vector<double> ha(64), hb(64), hd(64);
	printf("4x4x4_4b layout (A lane, B lane) -> D lanes, for block 0 lanes\n");
	for (int la = 0; la < 16; la++) {
		printf("A@%2d:", la);
		for (int lb = 0; lb < 16; lb++) {
			for (int i = 0; i < 64; i++) { ha[i] = 0; hb[i] = 0; }
			ha[la] = 1; hb[lb] = 1;
			hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice);
			hipLaunchKernelGGL(layout4, dim3(1), dim3(64), 0, 0, da, db, dd);
			hipMemcpy(hd.data(), dd, 512, hipMemcpyDeviceToHost);
			int hit = -1; for (int i = 0; i < 64; i++) if (hd[i] != 0) hit = i;
			printf(" %2d", hit);
		}
		printf("\n");
	}
	const int iters = 20000;
	const char *names[] = {"4 indep 4x4x4_4b", "4 dep 4x4x4_4b", "4 indep(2 acc) 16x16x4", "4 x (4x4x4 + 4 fma64)", "16 indep fma64", "16 dep fma64", "16 x (2 accread + fma64)"};
	unsigned long long c;
#define RUN(M) hipLaunchKernelGGL(rate<M>, dim3(256), dim3(256), 0, 0, dd, iters, dc); hipDeviceSynchronize(); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("%-28s %8.2f cycles per loop body\n", names[M], (double)c / iters);
	RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
	// two waves per SIMD (two 256-thread blocks per CU): does the fp64 pipe have headroom beyond what ONE wave can issue?
#define RUN2(M) hipLaunchKernelGGL(rate<M>, dim3(512), dim3(256), 0, 0, dd, iters, dc); hipDeviceSynchronize(); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("2 waves/SIMD: %-28s %8.2f cycles per loop body (per wave)\n", names[M], (double)c / iters);
	RUN2(4) RUN2(5) RUN2(6) RUN2(0)
	return 0;
}
