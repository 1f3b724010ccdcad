// probe of the v_mfma_f64_16x16x4_f64 register layout (diagnostic, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const double *A, const double *B, double *D, int *rowmap)
{
	const int l = threadIdx.x;
	// hypothesis: a = A[i = l%16][k = l/16], b = B[k = l/16][j = l%16]
	const double a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
	double4_t c = {0, 0, 0, 0};
	c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
	for (int r = 0; r < 4; r++) D[l * 4 + r] = c[r];
}
int main()
{
	double hA[64], hB[64], hD[256], ref[256];
	for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) hA[i * 4 + k] = 1.0 + i + 0.01 * k;
	for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) hB[k * 16 + j] = 1.0 + 0.1 * j + 7.0 * k;
	for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
	double *dA, *dB, *dD; int *dm;
	hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD); hipMalloc(&dm, 4);
	hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dm);
	hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
	// find for each (lane, r) which (i, j) it matches
	int ok = 1;
	for (int l = 0; l < 64; l += 5) for (int r = 0; r < 4; r++) {
		int fi = -1, fj = -1, cnt = 0;
		for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) if (fabs(ref[i * 16 + j] - hD[l * 4 + r]) < 1e-9) { fi = i; fj = j; cnt++; }
		printf("lane %2d r %d -> i %2d j %2d (matches %d)\n", l, r, fi, fj, cnt);
		if (cnt != 1) ok = 0;
	}
	printf("unique %d\n", ok);
	return 0;
}
