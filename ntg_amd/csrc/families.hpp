// families.hpp -- device functors standing in for the user callbacks of ntg.h:81-83,90-92.
//
// The reference takes host function pointers (icf/ucf/fcf, nlicf/nltcf/nlfcf); those cannot
// run on the GPU, so the batched path selects a problem family by id.  A functor sees the flat
// flag of ONE breakpoint as z[iz[o]+r] (== the reference's zp[o][r], colloc.c:425-447) and
// returns the value and the gradient over the full stacked flag, exactly like the callbacks
// (cost.c:107-108, constraints.c:146-151).  Every output of a family has maxderiv == Family::DM.
// Besides the dense callbacks every family offers the trajectory constraints in the two-step form the
// augmented-Lagrangian evaluation uses: nltc_val (values, plus Family::TAPE doubles it wants to keep) and nltc_vjp
// (df += J' t, with that tape), so that a family
// with many constraints never holds its dense [ncon][nz] Jacobian in registers.
//
//   NTG_FAM_KINCAR     f = sum_o (z_o'')^2                      examples/kincar.c:105-117
//   NTG_FAM_VANDERPOL  f = (z^2 + z'^2 + u^2)/2, u = z''+z-(1-z^2)z'   examples/vanderpol.c:206-241
//   NTG_FAM_TESTFAM    synthetic, every slot populated (mirrors oracle/families.c family 2)
//   NTG_FAM_OBSTACLE   kincar cost (2 outputs) + trajectory constraint c = (x-20)^2 + (y-0.5)^2 (>= r^2 via bounds)
//   NTG_FAM_QUADROTOR  4 outputs (x, y, z, yaw), maxderiv 5: snap^2 + yaw''^2; c0 = x''^2+y''^2+(z''+g)^2, c1 = |v|^2
//   NTG_FAM_MANIP      3 joints per planar arm, maxderiv 3: sum q''^2; per arm c = sin(qa)+sin(qa+qb)+sin(qa+qb+qc)
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ntg_amd.h"

template <int FAM> struct Family;

// nltc_val / nltc_vjp through the dense callback (families with a handful of constraints)
template <class Fam, int NZMAX>
struct DenseTraj {
	static __device__ __forceinline__ void val(int nout, int i, const double *z, double *c)
	{
		double dc[(Fam::NNLTC > 0 ? Fam::NNLTC : 1) * NZMAX];
		Fam::nltcf(nout, i, z, c, dc);
	}
	static __device__ __forceinline__ void vjp(int nout, int nz, int i, const double *z, const double *t, double *df)
	{
		double c[Fam::NNLTC > 0 ? Fam::NNLTC : 1], dc[(Fam::NNLTC > 0 ? Fam::NNLTC : 1) * NZMAX];
		Fam::nltcf(nout, i, z, c, dc);
		for (int j = 0; j < Fam::NNLTC; j++)
#pragma unroll
			for (int v = 0; v < NZMAX; v++) { if (v < nz) df[v] += t[j] * dc[j * nz + v]; }
	}
};

template <> struct Family<NTG_FAM_KINCAR> {
	static __device__ __forceinline__ int row_group(int j) { (void)j; return 0; }   // coupling group of trajectory row function j (QP-based SQP step)
	static constexpr int DM = 3, TAPE = 1;
	static constexpr u64 TCON_VARS = ~0ull;   // flag entries a trajectory constraint row can depend on (all: not declared)
	static constexpr bool PER_OUTPUT_COST = true;   // ucf(nout, ...) is a sum of identical terms over the outputs: a subset of the outputs gives its share
	static constexpr int COUPLE = 0, CG = 1;   // no structured Newton mode
	template <int NZMAX> static __device__ __forceinline__ void nltc_block(int, int, const double *, const double *, double, bool, double *) {}
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
	template <int NZMAX> static __device__ __forceinline__ void nltc_val(int, int, const double *, double *, double *) {}
	template <int NZMAX> static __device__ __forceinline__ void nltc_vjp(int, int, int, const double *, const double *, double *, const double *) {}
};

template <> struct Family<NTG_FAM_VANDERPOL> {
	static __device__ __forceinline__ int row_group(int j) { (void)j; return 0; }   // coupling group of trajectory row function j (QP-based SQP step)
	static constexpr int DM = 3, TAPE = 1;
	static constexpr u64 TCON_VARS = ~0ull;   // flag entries a trajectory constraint row can depend on (all: not declared)
	static constexpr bool PER_OUTPUT_COST = false;
	static constexpr int COUPLE = 0, CG = 1;   // no structured Newton mode
	template <int NZMAX> static __device__ __forceinline__ void nltc_block(int, int, const double *, const double *, double, bool, double *) {}
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
	template <int NZMAX> static __device__ __forceinline__ void nltc_val(int, int, const double *, double *, double *) {}
	template <int NZMAX> static __device__ __forceinline__ void nltc_vjp(int, int, int, const double *, const double *, double *, const double *) {}
};

// dc is [ncon][nz] row-major (== the reference's dc[constraint][variable])
template <> struct Family<NTG_FAM_TESTFAM> {
	static __device__ __forceinline__ int row_group(int j) { (void)j; return 0; }   // coupling group of trajectory row function j (QP-based SQP step)
	static constexpr int DM = 3, TAPE = 1;
	static constexpr u64 TCON_VARS = ~0ull;   // flag entries a trajectory constraint row can depend on (all: not declared)
	static constexpr bool PER_OUTPUT_COST = false;
	static constexpr int COUPLE = 0, CG = 1;   // no structured Newton mode
	template <int NZMAX> static __device__ __forceinline__ void nltc_block(int, int, const double *, const double *, double, bool, double *) {}
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
	template <int NZMAX> static __device__ __forceinline__ void nltc_val(int nout, int i, const double *z, double *c, double *) { DenseTraj<Family, NZMAX>::val(nout, i, z, c); }
	template <int NZMAX> static __device__ __forceinline__ void nltc_vjp(int nout, int nz, int i, const double *z, const double *t, double *df, const double *) { DenseTraj<Family, NZMAX>::vjp(nout, nz, i, z, t, df); }
};

template <> struct Family<NTG_FAM_OBSTACLE> {
	static __device__ __forceinline__ int row_group(int j) { (void)j; return 0; }   // coupling group of trajectory row function j (QP-based SQP step)
	static constexpr int DM = 3, TAPE = 1;
	static constexpr u64 TCON_VARS = ~0ull;   // flag entries a trajectory constraint row can depend on (all: not declared)
	static constexpr bool PER_OUTPUT_COST = false;
	static constexpr int COUPLE = 2, CG = 2;   // one group (x, y); constraint flag entries x, y
	// B (CG x CG) = mu a a' [row active] + t d2c/dz2 [curv]: the second-order model of the row's augmented-Lagrangian term
	template <int NZMAX> static __device__ __forceinline__ void nltc_block(int, int, const double *z, const double *t, double mu, bool curv, double *B)
	{
		const double dx = z[0] - 20.0, dy = z[3] - 0.5, a0 = 2.0 * dx, a1 = 2.0 * dy, m = t[0] != 0.0 ? mu : 0.0, h = curv ? 2.0 * t[0] : 0.0;
		B[0] = m * a0 * a0 + h; B[1] = m * a0 * a1; B[2] = m * a1 * a0; B[3] = m * a1 * a1 + h;
	}
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
	template <int NZMAX> static __device__ __forceinline__ void nltc_val(int nout, int i, const double *z, double *c, double *) { DenseTraj<Family, NZMAX>::val(nout, i, z, c); }
	template <int NZMAX> static __device__ __forceinline__ void nltc_vjp(int nout, int nz, int i, const double *z, const double *t, double *df, const double *) { DenseTraj<Family, NZMAX>::vjp(nout, nz, i, z, t, df); }
};

template <> struct Family<NTG_FAM_QUADROTOR> {
	static __device__ __forceinline__ int row_group(int j) { (void)j; return 0; }   // coupling group of trajectory row function j (QP-based SQP step)
	static constexpr int DM = 5, TAPE = 1;
	// flag entries (5 o + r) a trajectory constraint row can depend on: first and second derivatives of x, y, z
	static constexpr u64 TCON_VARS = (1ull << 1) | (1ull << 2) | (1ull << 6) | (1ull << 7) | (1ull << 11) | (1ull << 12);
	static constexpr bool PER_OUTPUT_COST = false;
	static constexpr int COUPLE = 3, CG = 6;   // one group (x, y, z), yaw is a free output; constraint flag entries in flag order: x', x'', y', y'', z', z''
	template <int NZMAX> static __device__ __forceinline__ void nltc_block(int, int, const double *z, const double *t, double mu, bool curv, double *B)
	{
		const double a0[6] = {0.0, 2.0 * z[2], 0.0, 2.0 * z[7], 0.0, 2.0 * (z[12] + G)};   // thrust^2
		const double a1[6] = {2.0 * z[1], 0.0, 2.0 * z[6], 0.0, 2.0 * z[11], 0.0};          // speed^2
		const double m0 = t[0] != 0.0 ? mu : 0.0, m1 = t[1] != 0.0 ? mu : 0.0;
#pragma unroll
		for (int i = 0; i < 6; i++)
#pragma unroll
			for (int j = 0; j < 6; j++) B[6 * i + j] = m0 * a0[i] * a0[j] + m1 * a1[i] * a1[j];
		if (curv) {
			B[7] += 2.0 * t[0]; B[21] += 2.0 * t[0]; B[35] += 2.0 * t[0];
			B[0] += 2.0 * t[1]; B[14] += 2.0 * t[1]; B[28] += 2.0 * t[1];
		}
	}
	static constexpr int NNLIC = 0, NNLTC = 2, NNLFC = 0;
	static constexpr double G = 9.81;
	static __device__ __forceinline__ void ucf(int, int, const double *z, double &f, double *df)
	{
#pragma unroll
		for (int v = 0; v < 20; v++) df[v] = 0.0;
		f = z[4] * z[4] + z[9] * z[9] + z[14] * z[14] + z[17] * z[17];
		df[4] = 2.0 * z[4]; df[9] = 2.0 * z[9]; df[14] = 2.0 * z[14]; df[17] = 2.0 * z[17];
	}
	static __device__ __forceinline__ void icf(int, const double *, double &f, double *) { f = 0.0; }
	static __device__ __forceinline__ void fcf(int, const double *, double &f, double *) { f = 0.0; }
	static __device__ __forceinline__ void nlicf(int, const double *, double *, double *) {}
	static __device__ __forceinline__ void nlfcf(int, const double *, double *, double *) {}
	static __device__ __forceinline__ void nltcf(int, int, const double *z, double *c, double *dc)
	{
		const double ax = z[2], ay = z[7], az = z[12] + G;
		c[0] = ax * ax + ay * ay + az * az;
		c[1] = z[1] * z[1] + z[6] * z[6] + z[11] * z[11];
#pragma unroll
		for (int v = 0; v < 40; v++) dc[v] = 0.0;
		dc[2] = 2.0 * ax; dc[7] = 2.0 * ay; dc[12] = 2.0 * az;
		dc[20 + 1] = 2.0 * z[1]; dc[20 + 6] = 2.0 * z[6]; dc[20 + 11] = 2.0 * z[11];
	}
	template <int NZMAX> static __device__ __forceinline__ void nltc_val(int, int, const double *z, double *c, double *)
	{
		const double ax = z[2], ay = z[7], az = z[12] + G;
		c[0] = ax * ax + ay * ay + az * az;
		c[1] = z[1] * z[1] + z[6] * z[6] + z[11] * z[11];
	}
	template <int NZMAX> static __device__ __forceinline__ void nltc_vjp(int, int, int, const double *z, const double *t, double *df, const double *)
	{
		df[2] += t[0] * (2.0 * z[2]); df[7] += t[0] * (2.0 * z[7]); df[12] += t[0] * (2.0 * (z[12] + G));
		df[1] += t[1] * (2.0 * z[1]); df[6] += t[1] * (2.0 * z[6]); df[11] += t[1] * (2.0 * z[11]);
	}
};

constexpr u64 ntg_manip_vars(int narms) { u64 m = 0; for (int j = 0; j < narms && 9 * j + 6 < 64; j++) m |= (1ull << (9 * j)) | (1ull << (9 * j + 3)) | (1ull << (9 * j + 6)); return m; }
template <> struct Family<NTG_FAM_MANIP> {
	static __device__ __forceinline__ int row_group(int j) { (void)j; return j; }   // coupling group of trajectory row function j (QP-based SQP step)
	static constexpr int DM = 3;
	static constexpr bool PER_OUTPUT_COST = false;
	static constexpr int MAXARMS = NTG_MAX_OUT / 3, TAPE = 3 * (NTG_MAX_OUT / 3);
	static constexpr u64 TCON_VARS = ntg_manip_vars(NTG_MAX_OUT / 3);   // the three joint angles of every arm (flag entries 9 j, 9 j + 3, 9 j + 6)
	static constexpr int COUPLE = 3, CG = 3;   // one group per arm; constraint flag entries qa, qb, qc
	template <int NZMAX> static __device__ __forceinline__ void nltc_block(int, int g, const double *z, const double *t, double mu, bool curv, double *B)
	{
		const double a1 = z[9 * g], a2 = a1 + z[9 * g + 3], a3 = a2 + z[9 * g + 6];
		double s1, s2, s3, c1, c2, c3;
		sincos(a1, &s1, &c1); sincos(a2, &s2, &c2); sincos(a3, &s3, &c3);
		const double a[3] = {c1 + c2 + c3, c2 + c3, c3}, m = t[g] != 0.0 ? mu : 0.0, tt = curv ? t[g] : 0.0;
		// d2c = -(s1 e1 e1' + s2 e2 e2' + s3 e3 e3'),  e1 = (1,0,0), e2 = (1,1,0), e3 = (1,1,1)
		const double h[9] = {s1 + s2 + s3, s2 + s3, s3, s2 + s3, s2 + s3, s3, s3, s3, s3};
#pragma unroll
		for (int i = 0; i < 3; i++)
#pragma unroll
			for (int j = 0; j < 3; j++) B[3 * i + j] = m * a[i] * a[j] - tt * h[3 * i + j];
	}
	static constexpr int NNLIC = 0, NNLTC = MAXARMS, NNLFC = 0;
	static __device__ __forceinline__ void ucf(int nout, int i, const double *z, double &f, double *df) { Family<NTG_FAM_KINCAR>::ucf(nout, i, z, f, df); }
	static __device__ __forceinline__ void icf(int, const double *, double &f, double *) { f = 0.0; }
	static __device__ __forceinline__ void fcf(int, const double *, double &f, double *) { f = 0.0; }
	static __device__ __forceinline__ void nlicf(int, const double *, double *, double *) {}
	static __device__ __forceinline__ void nlfcf(int, const double *, double *, double *) {}
	static __device__ __forceinline__ void nltcf(int nout, int, const double *z, double *c, double *dc)
	{
		const int narms = nout / 3, nz = 3 * nout;
		for (int j = 0; j < narms; j++) {
			const double a1 = z[9 * j], a2 = a1 + z[9 * j + 3], a3 = a2 + z[9 * j + 6];
			const double c1 = cos(a1), c2 = cos(a2), c3 = cos(a3);
			c[j] = sin(a1) + sin(a2) + sin(a3);
			for (int v = 0; v < nz; v++) dc[j * nz + v] = 0.0;
			dc[j * nz + 9 * j] = c1 + c2 + c3;
			dc[j * nz + 9 * j + 3] = c2 + c3;
			dc[j * nz + 9 * j + 6] = c3;
		}
	}
	// values, and the cosines of the three link angles of every arm on the tape: the vjp needs nothing else
	template <int NZMAX> static __device__ __forceinline__ void nltc_val(int nout, int, const double *z, double *c, double *tape)
	{
#pragma unroll
		for (int j = 0; j < MAXARMS; j++) {
			if (3 * j < nout) {
				const double a1 = z[9 * j], a2 = a1 + z[9 * j + 3], a3 = a2 + z[9 * j + 6];
				double s1, s2, s3;
				sincos(a1, &s1, &tape[3 * j]); sincos(a2, &s2, &tape[3 * j + 1]); sincos(a3, &s3, &tape[3 * j + 2]);
				c[j] = s1 + s2 + s3;
			}
		}
	}
	template <int NZMAX> static __device__ __forceinline__ void nltc_vjp(int nout, int, int, const double *, const double *t, double *df, const double *tape)
	{
#pragma unroll
		for (int j = 0; j < MAXARMS; j++) {
			if (3 * j < nout) {
				const double c1 = tape[3 * j], c2 = tape[3 * j + 1], c3 = tape[3 * j + 2];
				df[9 * j] += t[j] * (c1 + c2 + c3);
				df[9 * j + 3] += t[j] * (c2 + c3);
				df[9 * j + 6] += t[j] * c3;
			}
		}
	}
};
