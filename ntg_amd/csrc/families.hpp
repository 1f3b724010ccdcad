// families.hpp -- device functors standing in for the user callbacks of ntg.h:81-83,90-92.
//
// The reference takes host function pointers (icf/ucf/fcf, nlicf/nltcf/nlfcf); those cannot
// run on the GPU, so the batched path selects a problem family by id.  A functor sees the flat
// flag of ONE breakpoint as z[iz[o]+r] (== the reference's zp[o][r], colloc.c:425-447) and
// returns the value and the gradient over the full stacked flag, exactly like the callbacks
// (cost.c:107-108, constraints.c:146-151).  All families assume maxderiv == 3 per output.
//
//   NTG_FAM_KINCAR     f = sum_o (z_o'')^2                      examples/kincar.c:105-117
//   NTG_FAM_VANDERPOL  f = (z^2 + z'^2 + u^2)/2, u = z''+z-(1-z^2)z'   examples/vanderpol.c:206-241
//   NTG_FAM_TESTFAM    synthetic, every slot populated (mirrors oracle/families.c family 2)
//   NTG_FAM_OBSTACLE   kincar cost (2 outputs) + trajectory constraint c = (x-20)^2 + (y-0.5)^2 (>= r^2 via bounds)
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ntg_amd.h"

template <int FAM> struct Family;

template <> struct Family<NTG_FAM_KINCAR> {
	static constexpr int NNLIC = 0, NNLTC = 0, NNLFC = 0;
	static __device__ __forceinline__ void ucf(int nout, int, const double *z, double &f, double *df)
	{
		double s = 0.0;
		for (int o = 0; o < nout; o++) {
			s += z[3 * o + 2] * z[3 * o + 2];
			df[3 * o] = 0.0; df[3 * o + 1] = 0.0; df[3 * o + 2] = 2.0 * z[3 * o + 2];
		}
		f = s;
	}
	static __device__ __forceinline__ void icf(int, const double *, double &f, double *) { f = 0.0; }
	static __device__ __forceinline__ void fcf(int, const double *, double &f, double *) { f = 0.0; }
	static __device__ __forceinline__ void nlicf(int, const double *, double *, double *) {}
	static __device__ __forceinline__ void nltcf(int, int, const double *, double *, double *) {}
	static __device__ __forceinline__ void nlfcf(int, const double *, double *, double *) {}
};

template <> struct Family<NTG_FAM_VANDERPOL> {
	static constexpr int NNLIC = 0, NNLTC = 0, NNLFC = 0;
	static __device__ __forceinline__ void ucf(int, int, const double *zz, double &f, double *df)
	{
		const double z = zz[0], zd = zz[1], zdd = zz[2], t1 = z * z;
		const double u = zdd + z - (1.0 - t1) * zd;
		f = t1 / 2.0 + zd * zd / 2.0 + u * u / 2.0;
		df[0] = z + u * (1.0 + 2.0 * z * zd);
		df[1] = zd - u * (1.0 - t1);
		df[2] = u;
	}
	static __device__ __forceinline__ void icf(int, const double *, double &f, double *) { f = 0.0; }
	static __device__ __forceinline__ void fcf(int, const double *, double &f, double *) { f = 0.0; }
	static __device__ __forceinline__ void nlicf(int, const double *, double *, double *) {}
	static __device__ __forceinline__ void nltcf(int, int, const double *, double *, double *) {}
	static __device__ __forceinline__ void nlfcf(int, const double *, double *, double *) {}
};

// dc is [ncon][nz] row-major (== the reference's dc[constraint][variable])
template <> struct Family<NTG_FAM_TESTFAM> {
	static constexpr int NNLIC = 1, NNLTC = 2, NNLFC = 1;
	static __device__ __forceinline__ void icf(int nout, const double *z, double &f, double *df)
	{
		const int L = nout - 1;
		double s = 0.0;
		for (int o = 0; o < nout; o++) {
			s += (z[3 * o] - 1.0) * (z[3 * o] - 1.0) + 0.5 * z[3 * o + 1] * z[3 * o + 1];
			df[3 * o] = 2.0 * (z[3 * o] - 1.0); df[3 * o + 1] = z[3 * o + 1]; df[3 * o + 2] = 0.0;
		}
		f = s + 0.25 * z[0] * z[3 * L + 1];
		df[0] += 0.25 * z[3 * L + 1]; df[3 * L + 1] += 0.25 * z[0];
	}
	static __device__ __forceinline__ void ucf(int nout, int, const double *z, double &f, double *df)
	{
		const int L = nout - 1;
		const double sn = sin(z[0]), cs = cos(z[0]);
		double s = 0.0;
		for (int o = 0; o < nout; o++) {
			s += z[3 * o] * z[3 * o] + 0.1 * z[3 * o + 1] * z[3 * o + 1] + z[3 * o + 2] * z[3 * o + 2];
			df[3 * o] = 2.0 * z[3 * o]; df[3 * o + 1] = 0.2 * z[3 * o + 1]; df[3 * o + 2] = 2.0 * z[3 * o + 2];
		}
		f = s + 0.3 * sn * z[3 * L + 1];
		df[0] += 0.3 * cs * z[3 * L + 1]; df[3 * L + 1] += 0.3 * sn;
	}
	static __device__ __forceinline__ void fcf(int nout, const double *z, double &f, double *df)
	{
		double s = 0.0;
		for (int o = 0; o < nout; o++) {
			s += z[3 * o] * z[3 * o + 1] + 0.5 * z[3 * o + 2] * z[3 * o + 2];
			df[3 * o] = z[3 * o + 1]; df[3 * o + 1] = z[3 * o]; df[3 * o + 2] = z[3 * o + 2];
		}
		f = s;
	}
	static __device__ __forceinline__ void nlicf(int nout, const double *z, double *c, double *dc)
	{
		const int L = nout - 1, nz = 3 * nout;
		c[0] = z[0] * z[0] + z[3 * L + 1];
		for (int v = 0; v < nz; v++) dc[v] = 0.0;
		dc[0] += 2.0 * z[0]; dc[3 * L + 1] += 1.0;
	}
	static __device__ __forceinline__ void nltcf(int nout, int, const double *z, double *c, double *dc)
	{
		const int L = nout - 1, nz = 3 * nout;
		c[0] = z[0] * z[0] + z[3 * L] * z[3 * L];
		c[1] = z[1] * z[3 * L + 2] - cos(z[0]);
		for (int v = 0; v < 2 * nz; v++) dc[v] = 0.0;
		dc[0] += 2.0 * z[0]; dc[3 * L] += 2.0 * z[3 * L];
		dc[nz + 1] += z[3 * L + 2]; dc[nz + 3 * L + 2] += z[1]; dc[nz + 0] += sin(z[0]);
	}
	static __device__ __forceinline__ void nlfcf(int nout, const double *z, double *c, double *dc)
	{
		const int L = nout - 1, nz = 3 * nout;
		c[0] = z[2] * z[0] + z[3 * L + 1] * z[3 * L + 1];
		for (int v = 0; v < nz; v++) dc[v] = 0.0;
		dc[2] += z[0]; dc[0] += z[2]; dc[3 * L + 1] += 2.0 * z[3 * L + 1];
	}
};

template <> struct Family<NTG_FAM_OBSTACLE> {
	static constexpr int NNLIC = 0, NNLTC = 1, NNLFC = 0;
	static __device__ __forceinline__ void ucf(int nout, int i, const double *z, double &f, double *df) { Family<NTG_FAM_KINCAR>::ucf(nout, i, z, f, df); }
	static __device__ __forceinline__ void icf(int, const double *, double &f, double *) { f = 0.0; }
	static __device__ __forceinline__ void fcf(int, const double *, double &f, double *) { f = 0.0; }
	static __device__ __forceinline__ void nlicf(int, const double *, double *, double *) {}
	static __device__ __forceinline__ void nlfcf(int, const double *, double *, double *) {}
	static __device__ __forceinline__ void nltcf(int nout, int, const double *z, double *c, double *dc)
	{
		const double dx = z[0] - 20.0, dy = z[3] - 0.5;
		c[0] = dx * dx + dy * dy;
		for (int v = 0; v < 3 * nout; v++) dc[v] = 0.0;
		dc[0] = 2.0 * dx; dc[3] = 2.0 * dy;
	}
};
