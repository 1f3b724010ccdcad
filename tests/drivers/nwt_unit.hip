// nwt_unit.hip -- GPU unit check of the band Cholesky / triangular solves of ntg_amd/csrc/newton.hpp against a host band
// Cholesky (test infrastructure: built and run by tests/test_gpu_newton.py on the GPU box).
//   usage: nwt_unit <ng> <hb>    prints "max rel err factor-solve: X" and exits 0 when X < 1e-9
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../../ntg_amd/csrc/solve_impl.hpp"

__global__ void __launch_bounds__(64) unit_kernel(double *K, int ng, int hb, double *yio, int *fail, long long *cyc)
{
	extern __shared__ double sm[];
	double *y = sm, *panel = sm + 16 * ((ng + 15) / 16) + 48;
	const int ylen = 16 * ((ng + 15) / 16) + 48;
	for (int i = threadIdx.x; i < ylen; i += 64) y[i] = i < ng ? yio[i] : 0.0;
	__syncthreads();
	const long long t0 = __builtin_amdgcn_s_memtime();
	const int f = nwt_factor_wave((nwt_glb_dp)K, ng, hb, (nwt_lds_dp)panel, 1, 1 << 20);
	__syncthreads();
	const long long t1 = __builtin_amdgcn_s_memtime();
	nwt_solve_wave((nwt_glb_cdp)K, ng, hb, (nwt_lds_dp)y);
	__syncthreads();
	const long long t2 = __builtin_amdgcn_s_memtime();
	if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
	for (int i = threadIdx.x; i < ng; i += 64) yio[i] = y[i];
	if (threadIdx.x == 0) *fail = f;
}

// two-sided factorisation (nwt_factor_pairs / nwt_solve_pairs): two waves, one group
__global__ void __launch_bounds__(128) pair_kernel(double *Kt, double *Kb, NwtPair q, double *yio, int *fail, long long *cyc)
{
	extern __shared__ double sm[];
	const int lena = 16 * (q.ja + 3) + 48, lenb = 16 * (q.jb + 3) + 48, sep = nwt_pair_sep(q);
	double *ya = sm, *yb = sm + lena, *panel = yb + lenb;
	__shared__ int flag;
	if (threadIdx.x == 0) flag = 0;
	for (int i = threadIdx.x; i < lena; i += 128) ya[i] = i < 16 * q.ja + sep ? yio[i] : 0.0;
	for (int i = threadIdx.x; i < lenb; i += 128) yb[i] = i < 16 * q.jb ? yio[q.n - 1 - i] : 0.0;
	__syncthreads();
	const long long t0 = __builtin_amdgcn_s_memtime();
	const int f = nwt_factor_pairs(Kt, Kb, 1, q, panel, 1, &flag);
	__syncthreads();
	const long long t1 = __builtin_amdgcn_s_memtime();
	nwt_solve_pairs(Kt, Kb, 1, q, ya, yb, [] {});
	const long long t2 = __builtin_amdgcn_s_memtime();
	if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; *fail = f + flag; }
	for (int i = threadIdx.x; i < q.n; i += 128) yio[i] = i < 16 * q.ja + sep ? ya[i] : yb[q.n - 1 - i];
}

int main(int argc, char **argv)
{
	const int ng = argc > 1 ? atoi(argv[1]) : 100, hb = argc > 2 ? atoi(argv[2]) : 17, ld = hb + 1;
	const double spread = argc > 3 ? atof(argv[3]) : 0.0;   // decades of symmetric diagonal scaling D K D (ill conditioning)
	std::vector<double> K((size_t)ng * ld, 0.0), rhs(ng), x(ng);
	srand(7);
	auto rnd = [] { return rand() / (double)RAND_MAX - 0.5; };
	// SPD band: random symmetric band + dominant diagonal
	for (int i = 0; i < ng; i++) for (int e = 0; e <= hb; e++) { const int j = i - hb + e; if (j < 0) continue; K[(size_t)i * ld + e] = (j == i) ? 2.0 * (hb + 1) + rnd() : rnd(); }
	{
		std::vector<double> sc(ng);
		for (int i = 0; i < ng; i++) sc[i] = pow(10.0, spread * rnd());
		for (int i = 0; i < ng; i++) for (int e = 0; e <= hb; e++) { const int j = i - hb + e; if (j >= 0) K[(size_t)i * ld + e] *= sc[i] * sc[j]; }
	}
	for (int i = 0; i < ng; i++) rhs[i] = rnd();
	// host reference: dense Cholesky solve
	std::vector<double> A((size_t)ng * ng, 0.0);
	for (int i = 0; i < ng; i++) for (int e = 0; e <= hb; e++) { const int j = i - hb + e; if (j < 0) continue; A[(size_t)i * ng + j] = K[(size_t)i * ld + e]; A[(size_t)j * ng + i] = K[(size_t)i * ld + e]; }
	std::vector<double> L(A);
	for (int j = 0; j < ng; j++) {
		double d = L[(size_t)j * ng + j];
		for (int k = 0; k < j; k++) d -= L[(size_t)j * ng + k] * L[(size_t)j * ng + k];
		d = sqrt(d); L[(size_t)j * ng + j] = d;
		for (int i = j + 1; i < ng; i++) { double s = L[(size_t)i * ng + j]; for (int k = 0; k < j; k++) s -= L[(size_t)i * ng + k] * L[(size_t)j * ng + k]; L[(size_t)i * ng + j] = s / d; }
	}
	x = rhs;
	for (int i = 0; i < ng; i++) { double s = x[i]; for (int k = 0; k < i; k++) s -= L[(size_t)i * ng + k] * x[k]; x[i] = s / L[(size_t)i * ng + i]; }
	for (int i = ng - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < ng; k++) s -= L[(size_t)k * ng + i] * x[k]; x[i] = s / L[(size_t)i * ng + i]; }
	// nwt_unit ng hb spread 3 ROW: the matrix is made indefinite at ROW (strict factorisations must report it: exit 0 when BOTH the one-sided
	// and the two-sided factorisation do, whichever part of the two-sided split ROW falls in)
	const int indef_row = (argc > 5 && atoi(argv[4]) == 3) ? atoi(argv[5]) : -1;
	if (indef_row >= 0 && indef_row < ng) K[(size_t)indef_row * ld + hb] = -fabs(K[(size_t)indef_row * ld + hb]);
	if (argc > 4 && (atoi(argv[4]) == 2 || atoi(argv[4]) == 3)) {   // two-sided: nwt_unit ng hb spread 2
		NwtPair q; q.n = ng; q.hb = hb; const int jt = (ng - 32) / 16; q.ja = (jt + 1) / 2; q.jb = jt / 2;
		const int sep = ng - 16 * (q.ja + q.jb), ngt = 16 * q.ja + sep, brows = 16 * q.jb + 48;
		std::vector<double> Kt((size_t)ng * ld, 0.0), Kb((size_t)brows * ld, 0.0);
		for (int i = 0; i < ng; i++) for (int e = 0; e <= hb; e++) {
			const int j = i - hb + e; if (j < 0) continue;
			if (i < ngt) Kt[(size_t)i * ld + e] = K[(size_t)i * ld + e];
			else Kb[(size_t)(ng - 1 - j) * ld + e] = K[(size_t)i * ld + e];
		}
		double *dKt, *dKb, *dy2; int *df2; long long *dc2, hc2[2] = {0, 0};
		hipMalloc(&dKt, Kt.size() * 8); hipMalloc(&dKb, Kb.size() * 8); hipMalloc(&dy2, ng * 8); hipMalloc(&df2, 4); hipMalloc(&dc2, 16);
		hipMemcpy(dKt, Kt.data(), Kt.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dKb, Kb.data(), Kb.size() * 8, hipMemcpyHostToDevice);
		hipMemcpy(dy2, rhs.data(), ng * 8, hipMemcpyHostToDevice);
		const size_t lds2 = (size_t)(16 * (q.ja + 3) + 48 + 16 * (q.jb + 3) + 48 + 2 * 48 * NWT_PSTRIDE) * 8;
		hipLaunchKernelGGL(pair_kernel, dim3(1), dim3(128), lds2, 0, dKt, dKb, q, dy2, df2, dc2);
		if (hipDeviceSynchronize() != hipSuccess) { printf("pair kernel failed\n"); return 2; }
		std::vector<double> y2(ng); int fail2 = -1;
		hipMemcpy(y2.data(), dy2, ng * 8, hipMemcpyDeviceToHost); hipMemcpy(&fail2, df2, 4, hipMemcpyDeviceToHost); hipMemcpy(hc2, dc2, 16, hipMemcpyDeviceToHost);
		double err2 = 0.0, nx2 = 0.0;
		for (int i = 0; i < ng; i++) { err2 = fmax(err2, fabs(y2[i] - x[i])); nx2 = fmax(nx2, fabs(x[i])); }
		printf("  two-sided: ja %d jb %d sep %d; clock ticks: factor %lld, solve %lld\n", q.ja, q.jb, sep, hc2[0], hc2[1]);
		printf("ng %d hb %d fail %d: max rel err two-sided factor-solve: %.3e\n", ng, hb, fail2, err2 / nx2);
		if (indef_row < 0) return (fail2 == 0 && err2 / nx2 < 1e-9) ? 0 : 1;
		if (fail2 == 0) { printf("two-sided factorisation did not report the indefinite matrix\n"); return 1; }
		// ... and the one-sided routine on the same matrix
		double *dK1, *dy1; int *df1; long long *dc1;
		hipMalloc(&dK1, K.size() * 8); hipMalloc(&dy1, ng * 8); hipMalloc(&df1, 4); hipMalloc(&dc1, 16);
		hipMemcpy(dK1, K.data(), K.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dy1, rhs.data(), ng * 8, hipMemcpyHostToDevice);
		const size_t lds1 = (size_t)(16 * ((ng + 15) / 16) + 48 + 48 * NWT_PSTRIDE) * 8;
		hipLaunchKernelGGL(unit_kernel, dim3(1), dim3(64), lds1, 0, dK1, ng, hb, dy1, df1, dc1);
		int fail1 = 0;
		if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
		hipMemcpy(&fail1, df1, 4, hipMemcpyDeviceToHost);
		printf("one-sided fail %d\n", fail1);
		return fail1 != 0 ? 0 : 1;
	}
	double *dK, *dy; int *df; long long *dc, hc[2] = {0, 0};
	hipMalloc(&dK, K.size() * 8); hipMalloc(&dy, ng * 8); hipMalloc(&df, 4); hipMalloc(&dc, 16);
	hipMemcpy(dK, K.data(), K.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dy, rhs.data(), ng * 8, hipMemcpyHostToDevice);
	const size_t lds = (size_t)(16 * ((ng + 15) / 16) + 48 + 48 * NWT_PSTRIDE) * 8;
	hipLaunchKernelGGL(unit_kernel, dim3(1), dim3(64), lds, 0, dK, ng, hb, dy, df, dc);
	std::vector<double> y(ng), Lg(K.size()); int fail = -1;
	if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
	hipMemcpy(y.data(), dy, ng * 8, hipMemcpyDeviceToHost); hipMemcpy(&fail, df, 4, hipMemcpyDeviceToHost); hipMemcpy(Lg.data(), dK, K.size() * 8, hipMemcpyDeviceToHost);
	hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
	printf("  clock ticks (s_memtime, 100 MHz): factor %lld (%.1f per block column), solve %lld (%.1f per block)\n", hc[0], (double)hc[0] / ((ng + 15) / 16), hc[1], (double)hc[1] / ((ng + 15) / 16));
	double errL = 0.0, err = 0.0, nx = 0.0;
	for (int i = 0; i < ng; i++) for (int e = 0; e <= hb; e++) { const int j = i - hb + e; if (j < 0) continue; const double ref = (i == j) ? 1.0 / L[(size_t)i * ng + i] : L[(size_t)i * ng + j]; errL = fmax(errL, fabs(Lg[(size_t)i * ld + e] - ref) / fabs(ref)); }
	for (int i = 0; i < ng; i++) { err = fmax(err, fabs(y[i] - x[i])); nx = fmax(nx, fabs(x[i])); }
	printf("ng %d hb %d fail %d: max rel err factor %.3e, max rel err factor-solve: %.3e\n", ng, hb, fail, errL, err / nx);
	return (fail == 0 && err / nx < 1e-9 && errL < 1e-9) ? 0 : 1;
}
