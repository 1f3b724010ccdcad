// newton.hpp -- structured Newton step of sqp_kernel for problems with nonlinear rows (ntg_solve_opts.hessian = 2;
// DESIGN.md section 4c).  Stands where NPSOL's dense QP subproblem stands in the reference: the constraint Jacobian the
// reference assembles in constraints.c:120-162 and hands to npsol_ (ntg.c:217-220,250-253) is used here, per breakpoint,
// to build the second-order model of the augmented Lagrangian
//     K = sum_i M_i' B_i M_i,   B_i = 2 w_i diag(cost variables) + mu sum_{j active} a_j a_j' + sum_j t_j d2c_j/dz2
// (M_i = collocation rows of breakpoint i, a_j = dc_j/dz, t = multiplier estimates of the last evaluation).  With one spline
// spec per coupling group and the group's free coefficients interleaved by output (p = cl Go + o) K is banded, half
// bandwidth k Go - 1 <= 32.  The equality rows pin the first / last coefficients of every output (a square invertible
// block), so null(A_E) = {pinned coefficients = 0} and the reduced Hessian is the principal submatrix of K over the free
// coefficients: no Schur complement, no projection.
//
// Layout in HBM, per problem and group: compact lower band, row-major, Kc[p][e] = K(p, p - hb + e), e = 0..hb (e = hb is
// the diagonal); the Cholesky factor overwrites it.  Per breakpoint and group the CG x CG block B_i (Bz).
//
// Mapping to CDNA4: one wavefront per coupling group.  The factorisation walks the band in 16 x 16 tiles with a register
// window of six tiles in the accumulator layout of v_mfma_f64_16x16x4_f64; the panel of a block column (48 rows x 16) is
// factored with a lane per row (pivots and multipliers broadcast through SGPRs, v_readlane), the trailing update
// T -= X X' of the window is three products of 16 x 16 x 16 on the matrix cores.  Triangular solves are blocked the same
// way; the coefficients of the next block are in flight while the current one is eliminated.
#pragma once
#include <hip/hip_runtime.h>
#include "ntg_dev.hpp"

#define NWT_PSTRIDE 17   // LDS row stride (doubles) of the 48 x 16 panel

__device__ __forceinline__ double nwt_readlane(double v, int lane)
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
	return __hiloint2double(hi, lo);
}

// element (row, col) of the symmetric band matrix (lower part stored), identity padding beyond ng
__device__ __forceinline__ double nwt_band_get(const double *__restrict__ Kc, int ng, int hb, int row, int col)
{
	if (row >= ng) return row == col ? 1.0 : 0.0;
	if (col > row || col < 0 || row - col > hb) return 0.0;
	return Kc[(size_t)row * (hb + 1) + (col - row + hb)];
}
typedef double nwt_d4 __attribute__((ext_vector_type(4)));
// tile (I, J) in the accumulator layout: lane l, register r <-> element (16 I + 4 r + l/16, 16 J + l%16)
__device__ __forceinline__ nwt_d4 nwt_load_tile(const double *__restrict__ Kc, int ng, int hb, int I, int J, int lane)
{
	nwt_d4 t;
#pragma unroll
	for (int r = 0; r < 4; r++) t[r] = nwt_band_get(Kc, ng, hb, 16 * I + 4 * r + (lane >> 4), 16 * J + (lane & 15));
	return t;
}

// Band Cholesky K = L L' of one group by ONE wavefront, in place.  panel: LDS scratch of 48 * NWT_PSTRIDE doubles owned by
// this wave.  strict: a non-positive pivot is reported (return value 1, factor unusable); otherwise it is replaced by a tiny
// positive number (the Gauss-Newton matrix is positive definite up to rounding).
__device__ __attribute__((noinline)) int nwt_factor_wave(double *__restrict__ Kc, int ng, int hb, double *panel, int strict)
{
	const int lane = threadIdx.x & 63, ld = hb + 1, nbr = (ng + 15) >> 4;
	const int prow = lane < 48 ? lane : 47;   // lanes 48..63 shadow row 47 (their results are never stored)
	int fail = 0;
	nwt_d4 T00 = nwt_load_tile(Kc, ng, hb, 0, 0, lane), T10 = nwt_load_tile(Kc, ng, hb, 1, 0, lane), T11 = nwt_load_tile(Kc, ng, hb, 1, 1, lane);
	nwt_d4 T20 = nwt_load_tile(Kc, ng, hb, 2, 0, lane), T21 = nwt_load_tile(Kc, ng, hb, 2, 1, lane), T22 = nwt_load_tile(Kc, ng, hb, 2, 2, lane);
	for (int J = 0; J < nbr; J++) {
		// next block row of the window: in flight during the panel factorisation
		const nwt_d4 N0 = nwt_load_tile(Kc, ng, hb, J + 3, J + 1, lane), N1 = nwt_load_tile(Kc, ng, hb, J + 3, J + 2, lane), N2 = nwt_load_tile(Kc, ng, hb, J + 3, J + 3, lane);
		// accumulator layout -> one lane per panel row, through LDS
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int rr = 4 * r + (lane >> 4), cc = lane & 15;
			panel[rr * NWT_PSTRIDE + cc] = T00[r];
			panel[(16 + rr) * NWT_PSTRIDE + cc] = T10[r];
			panel[(32 + rr) * NWT_PSTRIDE + cc] = T20[r];
		}
		double a[16];
#pragma unroll
		for (int c = 0; c < 16; c++) a[c] = panel[prow * NWT_PSTRIDE + c];
		// column Cholesky of the 48 x 16 panel: lanes 0..15 hold the rows of the diagonal tile
#pragma unroll
		for (int j = 0; j < 16; j++) {
			double piv = nwt_readlane(a[j], j);
			if (!(piv > 0.0)) {
				if (strict) fail = 1;
				piv = strict ? 1.0 : 1e-30;
			}
			const double rinv = 1.0 / sqrt(piv);
			if (lane >= j) a[j] = lane == j ? piv * rinv : a[j] * rinv;
#pragma unroll
			for (int k = j + 1; k < 16; k++) {
				const double lkj = nwt_readlane(a[j], k);
				if (lane >= k) a[k] -= a[j] * lkj;
			}
		}
		// the finished block column of L: rows to HBM (band entries only) ...
		if (lane < 48) {
			const int row = 16 * J + lane;
			if (row < ng) {
#pragma unroll
				for (int c = 0; c < 16; c++) {
					const int col = 16 * J + c;
					if (col <= row && row - col <= hb) Kc[(size_t)row * ld + (col - row + hb)] = a[c];
				}
			}
		}
		// ... and the two sub-diagonal tiles X1, X2 back to LDS for the operand layout of the matrix instruction:
		// lane l supplies X[l%16][4 s + l/16] both as A[i][k] and as B[k][j] = X'[k][j]
		if (lane >= 16 && lane < 48) {
#pragma unroll
			for (int c = 0; c < 16; c++) panel[lane * NWT_PSTRIDE + c] = a[c];
		}
		double x1[4], x2[4];
#pragma unroll
		for (int s = 0; s < 4; s++) {
			x1[s] = panel[(16 + (lane & 15)) * NWT_PSTRIDE + 4 * s + (lane >> 4)];
			x2[s] = panel[(32 + (lane & 15)) * NWT_PSTRIDE + 4 * s + (lane >> 4)];
		}
#pragma unroll
		for (int s = 0; s < 4; s++) {
			T11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x1[s], x1[s], T11, 0, 0, 0);
			T21 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x2[s], x1[s], T21, 0, 0, 0);
			T22 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x2[s], x2[s], T22, 0, 0, 0);
		}
		T00 = T11; T10 = T21; T11 = T22; T20 = N0; T21 = N1; T22 = N2;
	}
	return fail;
}

// y <- L^-T L^-1 y for one group by ONE wavefront.  y: LDS, 16 nbr + 48 doubles, entries >= ng zero.
__device__ __attribute__((noinline)) void nwt_solve_wave(const double *__restrict__ Lc, int ng, int hb, double *y)
{
	const int lane = threadIdx.x & 63, q = lane & 15, part = lane >> 4, ld = hb + 1, nbr = (ng + 15) >> 4;
	// ---- forward: L w = y ----
	{
		double off[8], lrow[16];
		auto load = [&](int J, double (&o)[8], double (&lr)[16]) {
			const int R = 16 * J + q;
#pragma unroll
			for (int u = 0; u < 8; u++) {   // entries left of the diagonal block: columns R - hb + e < 16 J  <=>  e < hb - q
				const int e = part + 4 * u, col = R - hb + e;
				o[u] = (R < ng && e < hb - q && col >= 0) ? Lc[(size_t)R * ld + e] : 0.0;
			}
#pragma unroll
			for (int c = 0; c < 16; c++) {   // row R of the diagonal block (every part loads it: the values are lane-uniform per q)
				const int e = hb - (q - c);
				lr[c] = (R < ng && c <= q && e >= 0) ? Lc[(size_t)R * ld + e] : (c == q ? 1.0 : 0.0);
			}
		};
		load(0, off, lrow);
		for (int J = 0; J < nbr; J++) {
			double offn[8], lrown[16];
			if (J + 1 < nbr) load(J + 1, offn, lrown);
			const int R = 16 * J + q;
			double acc = 0.0;
#pragma unroll
			for (int u = 0; u < 8; u++) { const int col = R - hb + part + 4 * u; acc += off[u] * y[col >= 0 ? col : 0]; }
			acc += lane_xchg<16>(acc);
			acc += lane_xchg<32>(acc);
			double r = y[R] - acc;
#pragma unroll
			for (int j = 0; j < 16; j++) {
				if (q == j) r = r / lrow[j];
				const double yj = nwt_readlane(r, j);
				if (q > j) r -= lrow[j] * yj;
			}
			if (part == 0) y[R] = r;
			if (J + 1 < nbr) {
#pragma unroll
				for (int u = 0; u < 8; u++) off[u] = offn[u];
#pragma unroll
				for (int c = 0; c < 16; c++) lrow[c] = lrown[c];
			}
		}
	}
	// ---- backward: L' z = w ----
	{
		double off[8], lcol[16];
		auto load = [&](int J, double (&o)[8], double (&lc)[16]) {
			const int i = 16 * J + q;
#pragma unroll
			for (int u = 0; u < 8; u++) {   // rows below the diagonal block: j = 16 J + 16 + part + 4 u,  j - i <= hb
				const int j = 16 * J + 16 + part + 4 * u;
				o[u] = (j < ng && j - i <= hb) ? Lc[(size_t)j * ld + (i - j + hb)] : 0.0;
			}
#pragma unroll
			for (int c = 0; c < 16; c++) {   // column q of the diagonal block: L[16 J + c][16 J + q], c >= q
				const int j = 16 * J + c, e = hb - (c - q);
				lc[c] = (j < ng && c >= q && e >= 0) ? Lc[(size_t)j * ld + e] : (c == q ? 1.0 : 0.0);
			}
		};
		load(nbr - 1, off, lcol);
		for (int J = nbr - 1; J >= 0; J--) {
			double offn[8], lcoln[16];
			if (J > 0) load(J - 1, offn, lcoln);
			const int i = 16 * J + q;
			double acc = 0.0;
#pragma unroll
			for (int u = 0; u < 8; u++) acc += off[u] * y[16 * J + 16 + part + 4 * u];
			acc += lane_xchg<16>(acc);
			acc += lane_xchg<32>(acc);
			double r = y[i] - acc;
#pragma unroll
			for (int j = 15; j >= 0; j--) {
				if (q == j) r = r / lcol[j];
				const double zj = nwt_readlane(r, j);
				if (q < j) r -= lcol[j] * zj;
			}
			if (part == 0) y[i] = r;
			if (J > 0) {
#pragma unroll
				for (int u = 0; u < 8; u++) off[u] = offn[u];
#pragma unroll
				for (int c = 0; c < 16; c++) lcol[c] = lcoln[c];
			}
		}
	}
}

// K (compact band, every group) = cost model + sum over the breakpoints of M_i' B_i M_i; every lane takes band entries.
// Bz: [ngrp][P][cg*cg] of this problem, or nullptr (cost model alone: phase 0, mu == 0).
template <int NT>
__device__ __attribute__((noinline)) void nwt_assemble(const NtgDims &D, const NtgTables &T, const double *rowv, const int *chrow, const int *offt,
                                                       const int *tcomp, const double *__restrict__ Bz, double *__restrict__ Kc)
{
	const int ng = D.nwt_ng, hb = D.nwt_hb, ld = hb + 1, go = D.nwt_go, cg = D.nwt_cg, nco = D.ncoef[0], P = D.P, dm = D.d[0];
	const int per = ng * ld, total = D.nwt_ngrp * per;
	for (int e0 = threadIdx.x; e0 < total; e0 += NT) {
		const int g = e0 / per, rem = e0 - g * per, p = rem / ld, e = rem - p * ld, p2 = p - hb + e;
		double acc = 0.0;
		if (p2 >= 0) {
			acc = T.nwt_k0[e0];
			if (Bz) {
				const int c1 = T.nwt_map[g * ng + p], c2 = T.nwt_map[g * ng + p2];
				const int o1 = c1 / nco, cl1 = c1 - o1 * nco, o2 = c2 / nco, cl2 = c2 - o2 * nco;
				const int lo = max((int)T.nwt_lo[cl1], (int)T.nwt_lo[cl2]), hi = min((int)T.nwt_hi[cl1], (int)T.nwt_hi[cl2]);
				const double *Bg = Bz + (size_t)g * P * cg * cg;
				// the (derivative of output 1, derivative of output 2) pairs that carry a constraint flag entry: at most 4 are
				// taken per sweep over the breakpoints (quadrotor: velocity / acceleration of both outputs; one otherwise)
				int nc = 0, r1n = 0, r2n = 0;
				while (r1n < dm) {
					int ro1[4], ro2[4], bi[4];
					nc = 0;
					for (; r1n < dm && nc < 4; r1n++, r2n = 0) {
						const int a1 = tcomp[dm * o1 + r1n];
						if (a1 < 0) continue;
						for (; r2n < dm && nc < 4; r2n++) {
							const int a2 = tcomp[dm * o2 + r2n];
							if (a2 < 0) continue;
							ro1[nc] = chrow[r1n] + cl1 * P; ro2[nc] = chrow[r2n] + cl2 * P; bi[nc] = (a1 - g * cg) * cg + (a2 - g * cg);
							nc++;
						}
						if (r2n < dm) break;   // the sweep is full: resume at (r1n, r2n)
					}
#pragma unroll 4
					for (int i = lo; i < hi; i++) {
						const int sh = i - offt[i] * P;   // rowv[ch + (cl - off_i) P + i]
						const double *Bi = Bg + (size_t)i * cg * cg;
#pragma unroll
						for (int u = 0; u < 4; u++)
							if (u < nc) acc += rowv[ro1[u] + sh] * Bi[bi[u]] * rowv[ro2[u] + sh];
					}
				}
			}
		}
		Kc[e0] = acc;
	}
	(void)go;
}
