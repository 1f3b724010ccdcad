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

#define NWT_PSTRIDE 17   // LDS row stride (doubles) of a 48 x 16 panel in operand order (nwt_to_operand: diagnostics, tests/tools_mfma.hip)
#define NWT_PANEL 416    // LDS doubles a factoring wave owns (nwt_factor_wave: two buffers of four panel columns, stride 52)
#ifndef NWT_PIVOTS
#define NWT_PIVOTS 2     // pivots per LDS round trip of the factorisation's sweep (1, 2 or 4; 4 is no faster standalone and 15-30 % slower
                         // inside the solve kernels, whose instances compile the routine for at most 256 registers)
#endif
// LDS scratch of the out-of-line routines is passed as an address-space-3 pointer: through a generic `double *` the compiler has to
// emit FLAT loads and stores (it cannot see across the call that the pointer is LDS), which go through the vector-memory path -- every
// pivot step then waited on vmcnt(0), i.e. also on the prefetched band rows and the stores of the previous block column (measured:
// factorisation 13.0 k -> 12.0 k ticks per block column standalone; config E's Newton solve 272 -> 231 ms per 1024 problems).
typedef __attribute__((address_space(3))) double *nwt_lds_dp;
// ... and the band in HBM as an address-space-1 pointer: FLAT accesses also count in lgkmcnt, so the LDS hand-offs of the pivot sweep
// (s_waitcnt lgkmcnt(0)) waited for the band rows prefetched for the next block column.
typedef __attribute__((address_space(1))) double *nwt_glb_dp;
typedef const __attribute__((address_space(1))) double *nwt_glb_cdp;
// The assembly is inlined into the solve kernel; the factorisation (nwt_factor_wave) and the triangular solves (nwt_solve_wave) are out
// of line (own register allocation) and take every pointer with its address space in the type.
// Round 2 met HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in a build with all three routines out of line.  That build, reconstructed
// (commit 24d0f74 with NWT_FN = noinline, compiled for gfx950 and read statically, round 3): nwt_assemble took `const NtgDims &` /
// `const NtgTables &`; the kernel's by-value arguments live in a PRIVATE (scratch) copy once their address escapes, so the call passed
// a generic pointer into the private segment (`v_mov v0, 0xa0; v1 = src_private_base.hi` at the call site) and the callee read the
// dimensions with 31 FLAT loads through the private aperture -- under a 2 336 + 332 byte scratch frame, 297 VGPR and 359 SGPR spills
// around the calls, band and LDS scratch as generic pointers too (171 + 114 FLAT loads in the factorisation and the solves).  The
// descriptor was consistent (private segment 2 668 B = frame + largest callee frame, no dynamic stack, no AGPRs): nothing the compiler
// emitted was wrong on paper, and the fault itself cannot be replayed statically.  What can be said: the ONLY class of access the
// faulting build had and every clean build lacks is FLAT through the private aperture to a kernel-argument copy, reached through
// SGPR-spilled aperture bases.  The shipped code excludes the whole class by construction, and ntg_amd/call_audit.py checks the
// assembly of every translation unit for it at build time:
//   R1 no generic pointer into the private segment is ever formed (no src_private_base anywhere);
//   R2 out-of-line device functions contain no FLAT instruction (address-space-typed parameters, structs by value in registers);
//   R3 no dynamic stack.
#define NWT_FN __device__ __forceinline__
#ifdef NWT_CHECK   // debugging aid: report an out-of-range band index instead of touching memory
#define NWT_IDX(i, lim, tag) (((long long)(i) >= 0 && (long long)(i) < (long long)(lim)) ? (size_t)(i) : (printf("NWT index %s: %lld of %lld (blk %d thr %d)\n", tag, (long long)(i), (long long)(lim), (int)blockIdx.x, (int)threadIdx.x), (size_t)0))
#else
#define NWT_IDX(i, lim, tag) ((size_t)(i))
#endif

// Lanes of ONE wave exchange data through LDS without a workgroup barrier (the LDS queue of a wave is processed in order).
// The compiler has to be told: without a fence it may forward a lane's own store to its later load of the same address
// and never see what the other lanes wrote.
__device__ __forceinline__ void nwt_wave_sync()
{
	// compiler: no memory access moves across, nothing is forwarded; hardware: LDS operations done (the loads from HBM
	// that are in flight for the next block are NOT waited for, unlike with a fence)
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

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

// element (row, col) of the band for the accumulator-layout tile loads: no branches (the address is clamped, the value selected)
__device__ __forceinline__ double nwt_band_get2(const double *__restrict__ Kc, int ng, int hb, int row, int col)
{
	const bool inb = row < ng && col <= row && col >= 0 && row - col <= hb;
	const double v = Kc[NWT_IDX(inb ? (long long)row * (hb + 1) + (col - row + hb) : 0, (long long)ng * (hb + 1), "tile")];
	return inb ? v : ((row >= ng && row == col) ? 1.0 : 0.0);
}
__device__ __forceinline__ nwt_d4 nwt_load_tile2(const double *__restrict__ Kc, int ng, int hb, int I, int J, int lane)
{
	nwt_d4 t;
#pragma unroll
	for (int r = 0; r < 4; r++) t[r] = nwt_band_get2(Kc, ng, hb, 16 * I + 4 * r + (lane >> 4), 16 * J + (lane & 15));
	return t;
}

__device__ __forceinline__ double nwt_rcp(double p)      // 1/p: hardware estimate + two Newton steps with fused residuals
{
	double r = __builtin_amdgcn_rcp(p);
	r = fma(r, fma(-p, r, 1.0), r);
	r = fma(r, fma(-p, r, 1.0), r);
	return r;
}
__device__ __forceinline__ double nwt_rsqrt(double p)    // 1/sqrt(p): hardware estimate, then coupled Goldschmidt steps (g -> sqrt, h -> 1/(2 sqrt))
{
	const double y0 = __builtin_amdgcn_rsq(p);
	double g = p * y0, h = 0.5 * y0, rr = fma(-h, g, 0.5);
	g = fma(g, rr, g); h = fma(h, rr, h);
	rr = fma(-h, g, 0.5);
	g = fma(g, rr, g); h = fma(h, rr, h);
	rr = fma(-h, g, 0.5);
	h = fma(h, rr, h);
	return 2.0 * h;
}

// Band Cholesky K = L L' of one group by ONE wavefront, in place; the diagonal of the factor is stored INVERTED (1 / L_jj:
// the triangular solves multiply).  panel: LDS scratch of 48 * NWT_PSTRIDE doubles owned by this wave.  Returns the number of
// non-positive pivots: with strict the factor is then unusable (the sweep stops at that block column); otherwise each was
// replaced by a tiny positive number (the Gauss-Newton matrix is positive definite up to rounding).
// The six window tiles stay in the accumulator layout of the matrix instruction throughout (lane l, register r <->
// element (4 r + l/16, l%16)).  Per block column:
//   1. the 16 pivots run over the diagonal tile T00 and an identity tile E only: column j of both goes to LDS (the four lanes
//      that hold it), every lane reads back the pivot, the entry of its own column in row... (row l%16) and the multipliers of
//      its rows, and applies  T += c m,  m = -c_j[k] / pivot  on the columns k > j -- columns are NOT scaled inside the sweep
//      (no special case for column j, and the reciprocal of the pivot replaces a reciprocal square root on the critical path);
//      afterwards every lane scales its own column once by 1/sqrt(pivot): T00 = L00, E = L00^-T;
//   2. the sub-diagonal tiles by the matrix cores, X = T L00^-T: E in the accumulator layout IS the B operand, T goes
//      through LDS into the operand layout;
//   3. the trailing update T -= X X' of the window, 12 matrix instructions, X through LDS into the operand layout.
__device__ __forceinline__ void nwt_to_operand(const nwt_d4 &T, nwt_lds_dp xb, int li, int lk, double (&op)[4])
{
#pragma unroll
	for (int r = 0; r < 4; r++) xb[(4 * r + lk) * NWT_PSTRIDE + li] = T[r];
	nwt_wave_sync();
#pragma unroll
	for (int s = 0; s < 4; s++) op[s] = xb[li * NWT_PSTRIDE + 4 * s + lk];   // X[l%16][4 s + l/16]: A[i][k] as well as B[k][j] = X'[k][j]
	nwt_wave_sync();
}
// two tiles in one LDS round trip (xb: two tiles of 16 x NWT_PSTRIDE)
__device__ __forceinline__ void nwt_to_operand2(const nwt_d4 &Ta, const nwt_d4 &Tb, nwt_lds_dp xb, int li, int lk, double (&opa)[4], double (&opb)[4])
{
	nwt_lds_dp xb2 = xb + 16 * NWT_PSTRIDE;
#pragma unroll
	for (int r = 0; r < 4; r++) { xb[(4 * r + lk) * NWT_PSTRIDE + li] = Ta[r]; xb2[(4 * r + lk) * NWT_PSTRIDE + li] = Tb[r]; }
	nwt_wave_sync();
#pragma unroll
	for (int s = 0; s < 4; s++) { opa[s] = xb[li * NWT_PSTRIDE + 4 * s + lk]; opb[s] = xb2[li * NWT_PSTRIDE + 4 * s + lk]; }
	nwt_wave_sync();
}
// nstop: number of block columns to eliminate (>= the matrix's: all of it).  With nstop smaller the sweep stops there and the window -- the
// Schur complement of the eliminated columns on the next 48 rows -- is written back into the band (two-sided factorisation, nwt_factor_pair).
__device__ __attribute__((noinline)) int nwt_factor_wave(nwt_glb_dp __restrict__ Kc, int ng, int hb, nwt_lds_dp panel, int strict, int nstop)
{
	const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4, ld = hb + 1, nbr = (ng + 15) >> 4;
	constexpr int CS = 52;   // LDS stride (doubles) of a panel column of 48 rows: 104 words = 8 banks, the four columns of a group do not collide
	int fail = 0;
	// Addressing of the window relative to block column J: element (tile row a, register r) of a tile whose columns are block J + b sits
	// at  Kc[16 J ld + off],  off = (16 a + 4 r + lk) (ld - 1) + 16 b + li + hb  -- a lane constant; whether it lies inside the band
	// is a lane constant too.  Only "row < ng" (the end of the matrix) and "column >= 0" move with J.
	auto off_of = [&](int a, int b, int r) { return (16 * a + 4 * r + lk) * (ld - 1) + 16 * b + li + hb; };
	auto inband = [&](int a, int b, int r) { const int d = 16 * (a - b) + 4 * r + lk - li; return d >= 0 && d <= hb; };   // row - column
	auto load_tile = [&](int J, int a, int b) {   // tile (J + a, J + b) in the accumulator layout; identity beyond the end of the matrix
		nwt_d4 t;
		nwt_glb_cdp base = Kc + (size_t)16 * J * ld;
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int row = 16 * (J + a) + 4 * r + lk;
			const bool ok = inband(a, b, r) && row < ng;
			const double v = base[ok ? off_of(a, b, r) : 0];
			t[r] = ok ? v : ((row >= ng && a == b && 4 * r + lk == li) ? 1.0 : 0.0);
		}
		return t;
	};
	// the block row prefetched per block column (tiles (J + 3, J + 1 .. J + 3)): which entries lie in the band is a lane constant, kept
	// as a 0 / 1 factor and an offset clamped into the row (every address an initialised band entry of a row < ng); only "row < ng"
	// moves with J, and beyond the end the tile is the identity
	double nmask[3][4]; int noff[3][4]; double nident[4];
#pragma unroll
	for (int r = 0; r < 4; r++) {
#pragma unroll
		for (int b = 0; b < 3; b++) {
			const int d = 16 * (3 - (b + 1)) + 4 * r + lk - li;   // row - column
			const bool ok = d >= 0 && d <= hb;
			nmask[b][r] = ok ? 1.0 : 0.0; noff[b][r] = ok ? hb - d : hb;   // entry e = column - row + hb of the row
		}
		nident[r] = (4 * r + lk == li) ? 1.0 : 0.0;
	}
	auto load_next = [&](int J, int b) {
		nwt_d4 t;
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int row = 16 * (J + 3) + 4 * r + lk;
			const double v = Kc[(size_t)min(row, ng - 1) * ld + noff[b][r]];
			// the select sits on the factor, not on the loaded value: the load stays unconditional (in flight across the panel factorisation)
			t[r] = fma(v, row < ng ? nmask[b][r] : 0.0, (b == 2 && row >= ng) ? nident[r] : 0.0);
		}
		return t;
	};
	const double badpiv = strict ? 1.0 : 1e-30;
	// the lane's k index of the operand layout (lk) picks one of a group's four columns
	const bool k1 = (lk & 1) != 0, k2 = (lk & 2) != 0;
	auto pick = [&](const double (&v)[4]) { const double lo = k1 ? v[1] : v[0], hi = k1 ? v[3] : v[2]; return k2 ? hi : lo; };
	nwt_d4 T00 = load_tile(0, 0, 0), T10 = load_tile(0, 1, 0), T11 = load_tile(0, 1, 1);
	nwt_d4 T20 = load_tile(0, 2, 0), T21 = load_tile(0, 2, 1), T22 = load_tile(0, 2, 2);
	const int jend = min(nbr, nstop);
	for (int J = 0; J < jend; J++) {
		// next block row of the window: in flight during the panel factorisation
		const nwt_d4 N0 = load_next(J, 0), N1 = load_next(J, 1), N2 = load_next(J, 2);
		nwt_glb_dp base = Kc + (size_t)16 * J * ld;
		const int rows_left = ng - 16 * J;
		double x1[4], x2[4];   // the finished sub-diagonal tiles in the operand layout: x[s] = L[row li][column 4 s + lk]
#pragma unroll
		for (int s = 0; s < 4; s++) {
			const int g = 4 * s;
			// 1. columns g .. g + 3 of the panel (48 rows), from the accumulator layout (the 16 lanes that hold them) to LDS, read back by ROW
			nwt_lds_dp cb = panel + (s & 1) * (4 * CS);   // alternate buffers: the next group's writes never meet this group's reads
			// the third tile row holds band entries of this group's columns only if  row - column = 32 + li - (g + lk) <= hb  for some lane,
			// i.e.  g + hb >= 29  (wave uniform; half width 17: the last group only, 23: the last two): its share of the exchange, of the
			// elimination and of the products is skipped otherwise
			const bool t2 = g + hb >= 29;
			if ((li >> 2) == s) {
				const int o = (li & 3) * CS + lk;
#pragma unroll
				for (int r = 0; r < 4; r++) { cb[o + 4 * r] = T00[r]; cb[o + 16 + 4 * r] = T10[r]; }
				if (t2) {
#pragma unroll
					for (int r = 0; r < 4; r++) cb[o + 32 + 4 * r] = T20[r];
				}
			}
			nwt_wave_sync();
			double blk[4][4], c0[4], c1[4], c2[4];   // blk[p][q]: (row g + q, column g + p), q >= p;  c_t[p]: (row li of tile t, column g + p)
#pragma unroll
			for (int p = 0; p < 4; p++) {
#pragma unroll
				for (int q = p; q < 4; q++) blk[p][q] = cb[p * CS + g + q];
				c0[p] = cb[p * CS + li]; c1[p] = cb[p * CS + 16 + li]; c2[p] = 0.0;
			}
			if (t2) {
#pragma unroll
				for (int p = 0; p < 4; p++) c2[p] = cb[p * CS + 32 + li];
			}
			// 2. the four pivots of the group in registers: every lane eliminates the 4 x 4 pivot block (redundantly) and its own row of the
			// three tiles.  Columns stay unscaled (y = column of L times sqrt(pivot)): the reciprocal, not the reciprocal square root, is
			// on the critical path.
			double ip[4], dd[4];
#pragma unroll
			for (int p = 0; p < 4; p++) {
				double piv = blk[p][p];
				const bool bad = !(piv > 0.0);   // (branch-free: a select and an add)
				fail += bad ? 1 : 0; piv = bad ? badpiv : piv;
				dd[p] = piv; ip[p] = nwt_rcp(piv);
#pragma unroll
				for (int pp = p + 1; pp < 4; pp++) {
					const double mm = -blk[p][pp] * ip[p];
#pragma unroll
					for (int q = pp; q < 4; q++) blk[pp][q] = fma(blk[p][q], mm, blk[pp][q]);
					c0[pp] = fma(c0[p], mm, c0[pp]); c1[pp] = fma(c1[p], mm, c1[pp]);
					if (t2) c2[pp] = fma(c2[p], mm, c2[pp]);
				}
			}
			// 3. rank-4 update of the later columns on the matrix cores, in the accumulator layout:  T_t -= (Y_t D^-1) Y_0',  operands
			// A[i = li][k = lk] = y_t / d_k,  B[k = lk][j = li] = y_0 -- both are "row li, column g + lk": this lane's own values.  Rows of
			// the diagonal tile at or above the pivot are masked (their entries are the unused upper triangle); finished and current
			// columns receive no (finished) or unused (current) updates.
			const double ys0 = pick(c0), ys1 = pick(c1), ys2 = t2 ? pick(c2) : 0.0, ipk = pick(ip), dk = pick(dd);
			const int dg = li - (g + lk);   // row minus pivot row inside the diagonal tile
			const double y0 = dg > 0 ? ys0 : 0.0;
			if (s < 3) {
				T00 = __builtin_amdgcn_mfma_f64_16x16x4f64(-(y0 * ipk), y0, T00, 0, 0, 0);
				T10 = __builtin_amdgcn_mfma_f64_16x16x4f64(-(ys1 * ipk), y0, T10, 0, 0, 0);
				if (t2) T20 = __builtin_amdgcn_mfma_f64_16x16x4f64(-(ys2 * ipk), y0, T20, 0, 0, 0);
			}
			// 4. the group's columns of L (scaled once, off the critical path): to HBM by row (band entries only; the diagonal inverted), and
			// kept as the operands of the trailing update -- which is the layout they are already in
			const double rs = nwt_rsqrt(dk);
			x1[s] = ys1 * rs; x2[s] = ys2 * rs;
			{
				const int d1 = 16 + dg, d2 = 32 + dg;   // row - column of the two sub-diagonal tiles
				if (dg >= 0 && dg <= hb && li < rows_left) base[li * ld + (hb - dg)] = dg == 0 ? rs : y0 * rs;
				if (d1 <= hb && 16 + li < rows_left) base[(16 + li) * ld + (hb - d1)] = x1[s];
				if (t2 && d2 <= hb && 32 + li < rows_left) base[(32 + li) * ld + (hb - d2)] = x2[s];
			}
		}
		// trailing update of the window
#pragma unroll
		for (int s = 0; s < 4; s++) {
			T11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x1[s], x1[s], T11, 0, 0, 0);
			if (4 * s + hb >= 29) {   // x2[s] is zero otherwise (see t2)
				T21 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x2[s], x1[s], T21, 0, 0, 0);
				T22 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x2[s], x2[s], T22, 0, 0, 0);
			}
		}
		T00 = T11; T10 = T21; T11 = T22; T20 = N0; T21 = N1; T22 = N2;
		if (strict && fail) break;   // not positive definite: the caller repeats with the Gauss-Newton terms, the rest is not needed
	}
	if (jend < nbr && !(strict && fail)) {   // the window goes back to the band: rows 16 jend .. + 47, band entries inside the matrix
		nwt_glb_dp base = Kc + (size_t)16 * jend * ld;
		const int rows_left = ng - 16 * jend;
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int rr = 4 * r + lk;
			if (inband(0, 0, r) && rr < rows_left) base[off_of(0, 0, r)] = T00[r];
			if (inband(1, 0, r) && 16 + rr < rows_left) base[off_of(1, 0, r)] = T10[r];
			if (inband(2, 0, r) && 32 + rr < rows_left) base[off_of(2, 0, r)] = T20[r];
			if (inband(1, 1, r) && 16 + rr < rows_left) base[off_of(1, 1, r)] = T11[r];
			if (inband(2, 1, r) && 32 + rr < rows_left) base[off_of(2, 1, r)] = T21[r];
			if (inband(2, 2, r) && 32 + rr < rows_left) base[off_of(2, 2, r)] = T22[r];
		}
	}
	return fail;
}

// y <- L^-T L^-1 y for one group by ONE wavefront.  y: LDS, 16 nbr + 48 doubles, entries >= ng zero.
// Blocked by 16: lane (q, part) stands for row (forward) / column (backward) q of the block; the four parts split the products with the
// entries outside the diagonal block, the 16 steps inside it exchange the finished unknowns through v_readlane.  The routine is bound by
// instruction issue (one wave, dependent stream), so it is written for few instructions per block: entries a lane does not use are
// loaded as zeros ONCE (at load time), the elimination steps are then unconditional (mul, readlane, fma -- no per-step predicates),
// the coefficients of the next block are in flight in a second register set (the loop is unrolled by two: no copies), and the diagonal
// (stored inverted) is a separate load instead of a 16-way select.  (Before: ~600 instructions per block and direction, 7.0 k ticks
// per block; the arithmetic is 16 x 3.)
struct NwtBlk { double o[8], l[16], dq; };
// Block ranges (two-sided factorisation, nwt_solve_pair): the forward pass runs over blocks [f0, f1), the blocks from facc on only
// ACCUMULATE -- y[row] becomes the product of the row's entries left of its diagonal block with the unknowns found so far (the update a
// separator row receives from this side), nothing is solved; the backward pass runs over blocks [b0, b1) downwards.  An empty range skips
// the pass.  nwt_solve_wave(L, ng, hb, y) is the whole solve.
__device__ __attribute__((noinline)) void nwt_solve_range(nwt_glb_cdp __restrict__ Lc, int ng, int hb, nwt_lds_dp y, int f0, int f1, int facc, int b0, int b1)
{
	const int lane = threadIdx.x & 63, q = lane & 15, part = lane >> 4, ld = hb + 1, nbr = (ng + 15) >> 4;
	// Which entries of a block a lane uses is a lane constant: kept as 0 / 1 factors (a multiply per loaded value, no predicates -- as
	// booleans they were 24 SGPR pairs per direction, spilled to VGPR lanes and read back for every select) next to offsets clamped into
	// the row, so that every load reads an initialised (finite) band entry.  What depends on the block -- rows beyond the end of the
	// matrix, columns before its start -- needs no mask: those rows' unknowns are and stay zero (y is written for rows < ng only), and the
	// band slots of columns < 0 hold the zeros they were initialised with (the factorisation stores inside the matrix only).
	// ---- forward: L w = y ----
	{
		double mo[8], ml[16]; int eo[8], el[16];
#pragma unroll
		for (int u = 0; u < 8; u++) { const int e = part + 4 * u; const bool ok = e < hb - q; mo[u] = ok ? 1.0 : 0.0; eo[u] = ok ? e : hb; }   // left of the diagonal block
#pragma unroll
		for (int c = 0; c < 16; c++) { const int e = hb - (q - c); const bool ok = c < q && e >= 0; ml[c] = ok ? 1.0 : 0.0; el[c] = ok ? e : hb; }   // row of the block, left of the diagonal
		auto load = [&](int J, NwtBlk &b) {
			nwt_glb_cdp rowp = Lc + (size_t)min(16 * J + q, ng - 1) * ld;
#pragma unroll
			for (int u = 0; u < 8; u++) b.o[u] = rowp[eo[u]];
#pragma unroll
			for (int c = 0; c < 16; c++) b.l[c] = rowp[el[c]];
			b.dq = rowp[hb];
		};
		auto step = [&](int J, NwtBlk &b, NwtBlk &nxt) {
			load(min(J + 1, nbr - 1), nxt);   // always (the last block loads itself again): no branch around the prefetch
			const int R = 16 * J + q;
			double acc = 0.0;
#pragma unroll
			for (int u = 0; u < 8; u++) acc = fma(b.o[u] * mo[u], y[max(R - hb + part + 4 * u, 0)], acc);
			acc += lane_xchg<16>(acc);
			acc += lane_xchg<32>(acc);
			if (J >= facc) {   // (wave uniform) a separator block seen from this side: the update only
				if (part == 0 && R < ng) y[R] = acc;
				nwt_wave_sync();
				return;
			}
			double r = y[R] - acc;
#pragma unroll
			for (int c = 0; c < 16; c++) b.l[c] *= ml[c];
#pragma unroll
			for (int j = 0; j < 16; j++) r = fma(-b.l[j], nwt_readlane(r * b.dq, j), r);   // lanes q <= j: l[j] = 0
			if (part == 0 && R < ng) y[R] = r * b.dq;
			nwt_wave_sync();
		};
		if (f0 < f1) {
			NwtBlk A, B;
			load(f0, A);
			for (int J = f0; J < f1; J += 2) {
				step(J, A, B);
				if (J + 1 < f1) step(J + 1, B, A);
			}
		}
	}
	// ---- backward: L' z = w ----
	{
		// entry (row j, column i) sits at  j (ld - 1) + i + hb;  rows are clamped to the last one (whose unknown multiplies a zero)
		double mo[8], ml[16];
#pragma unroll
		for (int u = 0; u < 8; u++) mo[u] = (16 + part + 4 * u - q <= hb) ? 1.0 : 0.0;             // rows below the block: j - i <= hb
#pragma unroll
		for (int c = 0; c < 16; c++) ml[c] = (c > q && c - q <= hb) ? 1.0 : 0.0;                    // column of the block, below the diagonal
		auto load = [&](int J, NwtBlk &b) {
			// both indices clamped into the matrix: the address then lies inside the band's memory for used and unused entries alike
			const int ic = min(16 * J + q, ng - 1);
			nwt_glb_cdp colp = Lc + (ic + hb);
#pragma unroll
			for (int u = 0; u < 8; u++) b.o[u] = colp[(size_t)min(16 * J + 16 + part + 4 * u, ng - 1) * (ld - 1)];
#pragma unroll
			for (int c = 0; c < 16; c++) b.l[c] = colp[(size_t)min(16 * J + c, ng - 1) * (ld - 1)];
			b.dq = colp[(size_t)ic * (ld - 1)];
		};
		auto step = [&](int J, NwtBlk &b, NwtBlk &nxt) {
			load(max(J - 1, 0), nxt);
			const int i = 16 * J + q;
			double acc = 0.0;
#pragma unroll
			for (int u = 0; u < 8; u++) acc = fma(b.o[u] * mo[u], y[16 * J + 16 + part + 4 * u], acc);
			acc += lane_xchg<16>(acc);
			acc += lane_xchg<32>(acc);
			double r = y[i] - acc;
#pragma unroll
			for (int c = 0; c < 16; c++) b.l[c] *= ml[c];
#pragma unroll
			for (int j = 15; j >= 0; j--) r = fma(-b.l[j], nwt_readlane(r * b.dq, j), r);   // lanes q >= j: l[j] = 0
			if (part == 0 && i < ng) y[i] = r * b.dq;
			nwt_wave_sync();
		};
		if (b0 < b1) {
			NwtBlk A, B;
			load(b1 - 1, A);
			for (int J = b1 - 1; J >= b0; J -= 2) {
				step(J, A, B);
				if (J - 1 >= b0) step(J - 1, B, A);
			}
		}
	}
}
__device__ __forceinline__ void nwt_solve_wave(nwt_glb_cdp Lc, int ng, int hb, nwt_lds_dp y)
{
	const int nbr = (ng + 15) >> 4;
	nwt_solve_range(Lc, ng, hb, y, 0, nbr, nbr, 0, nbr);
}

// ---- two-sided factorisation: TWO wavefronts per coupling group ----
// The band Cholesky is a chain of block columns; one wave walks it at the pace of its dependent instruction stream while the other waves of
// the workgroup idle (config D: one group; config E: four groups on eight waves).  Split in the middle it is two chains of half the
// length: the top wave eliminates block columns 0 .. ja - 1 downwards, the bottom wave eliminates the LAST jb block columns upwards -- which
// is the same routine on the reversed matrix P K P, kept as a second band array (Kb: row p' = n - 1 - p; the assembly writes the entries of
// bottom rows there, newton.hpp nwt_assemble; the cost model's share comes reversed from the plan).  Both stop at the separator, the
// sep = n - 16 (ja + jb) rows in the middle (32 <= sep < 48 >= the half width, so no entry couples top and bottom directly): each writes
// its Schur complement on the separator back into its array (nwt_factor_wave, nstop), the top wave adds the bottom wave's (the same band
// offsets: entry (p, c) sits at row n - 1 - c of the reversed array), factors the separator and leaves L_SS in the top array.  The factor is
//     L = [ L_TT . . ; . L_BB . ; L_ST L_SB L_SS ]   (unknowns ordered top, bottom, separator),
// L_ST in the separator rows of the top array, L_SB in those of the bottom array.  Solves: both sides forward in parallel, the bottom wave
// also forms L_SB w_B (the accumulate-only blocks of nwt_solve_range), the top wave finishes the separator forward and backward, both sides
// backward in parallel.  Chain length 29 -> 13 + 3 block columns (config D), 34 -> 16 + 3 (config E).
struct NwtPair { int n, hb, ja, jb; };   // free coefficients of a group, half width, block columns eliminated from the top / from the bottom
__device__ __forceinline__ int nwt_pair_sep(const NwtPair &q) { return q.n - 16 * (q.ja + q.jb); }
__device__ __forceinline__ int nwt_pair_brows(const NwtPair &q) { return 16 * q.jb + 48; }   // rows of the reversed array (allocation)

// Called by EVERY wave of the workgroup (it contains workgroup barriers).  Waves [0, ngp) are the groups' top waves, [ngp, 2 ngp) their bottom
// waves.  Kt: [ngp][n][ld], Kb: [ngp][brows][ld].  panel: LDS, NWT_PANEL doubles per wave (2 ngp waves).  flag: LDS word, preset
// to 0 by the caller, set when a strict factorisation met a non-positive pivot.  Returns this wave's count of replaced pivots.
__device__ __forceinline__ int nwt_factor_pairs(double *Kt, double *Kb, int ngp, const NwtPair q, double *panel, int strict, int *flag)
{
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, ld = q.hb + 1, sep = nwt_pair_sep(q), brows = nwt_pair_brows(q);
	int f = 0;
	if (wave < ngp)
		f = nwt_factor_wave((nwt_glb_dp)(Kt + (size_t)wave * q.n * ld), 16 * q.ja + sep, q.hb, (nwt_lds_dp)(panel + (size_t)wave * NWT_PANEL), strict, q.ja);
	else if (wave < 2 * ngp)
		f = nwt_factor_wave((nwt_glb_dp)(Kb + (size_t)(wave - ngp) * brows * ld), 16 * q.jb + sep, q.hb, (nwt_lds_dp)(panel + (size_t)wave * NWT_PANEL), strict, q.jb);
	if (f && strict && lane == 0) *flag = 1;
	__syncthreads();
	if (wave < ngp && !(strict && *flag)) {
		double *kt = Kt + (size_t)wave * q.n * ld, *kb = Kb + (size_t)wave * brows * ld;
		for (int i = lane; i < sep * ld; i += 64) {
			const int r = i / ld, e = i - r * ld, p = 16 * q.ja + r, col = p - q.hb + e;
			if (col >= 16 * q.ja) {
				const size_t ib = (size_t)(q.n - 1 - col) * ld + e;
				kt[(size_t)p * ld + e] += kb[ib];
				kb[ib] = 0.0;   // the bottom wave's forward pass multiplies these slots with the separator's unknowns-to-be
			}
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's own stores, before its tile loads of the same entries
		const int f2 = nwt_factor_wave((nwt_glb_dp)(kt + (size_t)16 * q.ja * ld), sep, q.hb, (nwt_lds_dp)(panel + (size_t)wave * NWT_PANEL), strict, 1 << 20);
		if (f2 && strict && lane == 0) *flag = 1;
		f += f2;
	}
	return f;
}

// y <- K^-1 y for every group with the two-sided factor.  Called by EVERY wave of the workgroup (barriers).  ya: [ngp][lena], the group's top
// and separator entries in band order, zero padded (lena = 16 (ja + 3) + 48); yb: [ngp][lenb], the bottom entries REVERSED (yb[p'] = entry
// n - 1 - p'), the rest zero (lenb = 16 (jb + 3) + 48).  On return the solution sits in the same places (separator: ya).
// extra(): what the waves beyond 2 ngp do meanwhile (free outputs), called once.
template <class F>
__device__ __forceinline__ void nwt_solve_pairs(const double *Kt, const double *Kb, int ngp, const NwtPair q, double *ya, double *yb, F extra)
{
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, ld = q.hb + 1, sep = nwt_pair_sep(q), brows = nwt_pair_brows(q), nst = (sep + 15) >> 4;
	const int lena = 16 * (q.ja + 3) + 48, lenb = 16 * (q.jb + 3) + 48, never = 1 << 20;
	const int g = wave < ngp ? wave : wave - ngp;
	nwt_glb_cdp kt = (nwt_glb_cdp)(Kt + (size_t)g * q.n * ld), kb = (nwt_glb_cdp)(Kb + (size_t)g * brows * ld);
	nwt_lds_dp a = (nwt_lds_dp)(ya + (size_t)g * lena), b = (nwt_lds_dp)(yb + (size_t)g * lenb);
	if (wave < ngp) nwt_solve_range(kt, 16 * q.ja + sep, q.hb, a, 0, q.ja, never, 0, 0);
	else if (wave < 2 * ngp) nwt_solve_range(kb, 16 * q.jb + sep, q.hb, b, 0, q.jb + nst, q.jb, 0, 0);
	else extra();
	__syncthreads();
	if (wave < ngp) {
		if (lane < sep) a[16 * q.ja + lane] -= b[16 * q.jb + sep - 1 - lane];
		nwt_wave_sync();
		nwt_solve_range(kt, 16 * q.ja + sep, q.hb, a, q.ja, q.ja + nst, never, q.ja, q.ja + nst);
		if (lane < sep) b[16 * q.jb + sep - 1 - lane] = a[16 * q.ja + lane];
	}
	__syncthreads();
	if (wave < ngp) nwt_solve_range(kt, 16 * q.ja + sep, q.hb, a, 0, 0, never, 0, q.ja);
	else if (wave < 2 * ngp) nwt_solve_range(kb, 16 * q.jb + sep, q.hb, b, 0, 0, never, 0, q.jb);
	__syncthreads();
}

// K (compact band, every group) = cost model + sum over the breakpoints of M_i' B_i M_i, on the matrix cores.
// The breakpoints of one knot interval share their block of k Go coefficients, so their contribution is one small dense
// product  S (kg x kg) = Mst' V,  Mst = the collocation rows of the interval's breakpoints stacked (cnt cg x kg),
// V = blockdiag(B_i) Mst:  v_mfma_f64_16x16x4_f64 tiles, operands formed on the fly from the channel tables in LDS and
// the blocks B_i.  A wave takes whole (group, interval) items; intervals of one colour share no coefficient, so the
// read-modify-write of the band needs no atomics and the sum is deterministic.  Per item there is ONE round trip to
// HBM/L2: the interval's blocks (one coalesced load, staged in LDS) and the band entries the result is added to are
// requested together, before the products.
// Bz: [ngrp][P][cg*cg] of this problem, or nullptr (cost model alone: phase 0, mu == 0).  wbuf: LDS, 216 doubles per wave.
// CG: the family's block size (Family::CG) as a compile-time constant: the loop over a block's variables unrolls, kk / CG is a multiply,
// and what a variable is (its output, the LDS offset of its derivative channel, whether this lane's columns belong to its output) is
// decoded once per kernel instead of once per k-step and variable.
template <int NT, int CG>
NWT_FN void nwt_assemble(const NtgDims &D, const NtgTables &T, const double *rowv, const int *chrow, const int *offt,
                                                       const double *__restrict__ Bz, double *__restrict__ Kc, double *wbuf_all,
                                                       unsigned long long *tkx = nullptr)
{
	unsigned long long tx0 = tkx ? __builtin_amdgcn_s_memtime() : 0;
	constexpr int NW = NT / 64;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
	const int ngp = D.nwt_ngrp, ng = D.nwt_ng, hb = D.nwt_hb, ld = hb + 1, go = D.nwt_go, P = D.P;
	constexpr int cg = CG;
	const int kg = D.order[0] * go, cover = D.nwt_cover, nint = D.nwt_nint, clo = D.nwt_clo, chi = D.nwt_chi;
	// two-sided factorisation: the entries of rows below the separator live in the reversed array behind the groups' top arrays
	const bool tw = D.nwt_tw != 0;
	const int brows = 16 * D.nwt_jb + 48, ngt = tw ? ng - 16 * D.nwt_jb : ng;   // top rows + separator rows
	const int total = ngp * ng * ld + (tw ? ngp * brows * ld : 0);
	const u64 upack = D.nwt_upack;
	{	// the cost model into the band: 16-byte words, four requests in flight per lane (the copy is a latency chain otherwise: 306 KB per
		// refresh for config E, 5-8 % of a solve)
		const double2 *src = reinterpret_cast<const double2 *>(T.nwt_k0);
		double2 *dst = reinterpret_cast<double2 *>(Kc);
		const int n2 = ((((size_t)Kc | (size_t)T.nwt_k0) & 15) == 0) ? total >> 1 : 0;
		int e = tid;
		for (; e + 3 * NT < n2; e += 4 * NT) {
			const double2 a = src[e], b = src[e + NT], c = src[e + 2 * NT], d = src[e + 3 * NT];
			dst[e] = a; dst[e + NT] = b; dst[e + 2 * NT] = c; dst[e + 3 * NT] = d;
		}
		for (; e < n2; e += NT) dst[e] = src[e];
		for (int r = 2 * n2 + tid; r < total; r += NT) Kc[r] = T.nwt_k0[r];
	}
	__syncthreads();
	if (tkx) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); tkx[0] += t1 - tx0; tx0 = t1; }
	if (!Bz) return;
	double *wbuf = wbuf_all + wave * 216;
	const unsigned long long *act = (const unsigned long long *)(wbuf_all + NW * 216);   // written by the block pass (sqp_kernel, nwt_refresh)
	// the two local columns (block-coefficient index a = q go + o) this lane stands for in the operand tiles
	const int a0 = li, a1 = 16 + li;
	const int qa0 = a0 / go, oa0 = a0 - qa0 * go, qa1 = a1 / go, oa1 = a1 - qa1 * go;
	const bool va0 = a0 < kg, va1 = a1 < kg;
	// ... and the rows of its four accumulator registers
	int qr0[4], or0[4], qr1[4], or1[4];
#pragma unroll
	for (int r = 0; r < 4; r++) { const int ar0 = 4 * r + lk, ar1 = 16 + ar0; qr0[r] = ar0 / go; or0[r] = ar0 - qr0[r] * go; qr1[r] = ar1 / go; or1[r] = ar1 - qr1[r] * go; }
	// per block variable v: output, LDS offset of its channel's rows, and the offsets of this lane's two columns in that channel
	// (-1: the column belongs to another output)
	int v_out[CG], v_c0[CG], v_c1[CG];
#pragma unroll
	for (int v = 0; v < CG; v++) {
		const int ov = (int)((upack >> (8 * v + 4)) & 15u), rv2 = (int)((upack >> (8 * v)) & 15u), ch = chrow[rv2];
		v_out[v] = ov;
		v_c0[v] = (va0 && ov == oa0) ? ch + qa0 * P : -1;
		v_c1[v] = (va1 && ov == oa1) ? ch + qa1 * P : -1;
	}
	for (int color = 0; color < cover; color++) {
		const int ntc = (nint - color + cover - 1) / cover;   // intervals of this colour
		for (int w = wave; w < ngp * ntc; w += NW) {
			const int g = w / ntc, t = color + (w - g * ntc) * cover;
			const int bp0 = D.igb[t], cnt = D.igb[t + 1] - bp0;
			{	// nothing to add when none of the interval's breakpoints has a non-zero block (flags of the block pass, wave uniform)
				const unsigned long long *am = act + g * ((P + 63) >> 6);
				const int w0 = bp0 >> 6, sh = bp0 & 63;
				unsigned long long bits = am[w0] >> sh;
				if (sh + cnt > 64) bits |= am[w0 + 1] << (64 - sh);
				if (cnt < 64) bits &= (1ull << cnt) - 1ull;
				if (__builtin_amdgcn_readfirstlane((int)(bits != 0ull)) == 0) continue;
			}
			const int of = offt[bp0];
			const int kdim = cnt * cg, ksteps = (kdim + 3) >> 2, nb = cnt * cg * cg;
			// requests: the interval's blocks ...
			const double *Bsrc = Bz + ((size_t)g * P + bp0) * cg * cg;
			double bl[4];
#pragma unroll
			for (int u = 0; u < 4; u++) bl[u] = lane + 64 * u < nb ? Bsrc[lane + 64 * u] : 0.0;
			// ... and the band entries of the lower triangle of S: local index a -> free index p = (of + a/go - clo) go + a%go
			double *Kg = Kc + (size_t)g * ng * ld;
			const int kbd = (ngp - g) * ng * ld + g * brows * ld;
			const int cb0 = of + qa0, cb1 = of + qa1;
			const bool fb0 = va0 && cb0 >= clo && cb0 < chi, fb1 = va1 && cb1 >= clo && cb1 < chi;
			const int pb0 = (cb0 - clo) * go + oa0, pb1 = (cb1 - clo) * go + oa1;
			int idx[12]; double kv[12];
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const int cr0 = of + qr0[r], cr1 = of + qr1[r];
				const bool fr0 = 4 * r + lk < kg && cr0 >= clo && cr0 < chi, fr1 = 16 + 4 * r + lk < kg && cr1 >= clo && cr1 < chi;
				const int pr0 = (cr0 - clo) * go + or0[r], pr1 = (cr1 - clo) * go + or1[r];
				// entry (row pr, column pb) of the group's matrix: in the top array, or -- rows below the separator -- at row n - 1 - pb of the
				// reversed array (same band offset), whose distance from the group's top array is kbd
				auto at = [&](int pr, int pb) { return pr < ngt ? pr * ld + (pb - pr + hb) : kbd + (ng - 1 - pb) * ld + (pb - pr + hb); };
				idx[3 * r] = (fr0 && fb0 && pr0 >= pb0) ? at(pr0, pb0) : -1;
				idx[3 * r + 1] = (fr1 && fb0) ? at(pr1, pb0) : -1;
				idx[3 * r + 2] = (fr1 && fb1 && pr1 >= pb1) ? at(pr1, pb1) : -1;
			}
#pragma unroll
			for (int e = 0; e < 12; e++) kv[e] = Kg[NWT_IDX(idx[e] >= 0 ? idx[e] : 0, (long long)total, "asm")];
#pragma unroll
			for (int u = 0; u < 4; u++) if (lane + 64 * u < 216) wbuf[lane + 64 * u] = bl[u];
			nwt_wave_sync();
			nwt_d4 S00 = {0.0, 0.0, 0.0, 0.0}, S10 = {0.0, 0.0, 0.0, 0.0}, S11 = {0.0, 0.0, 0.0, 0.0};
			for (int s = 0; s < ksteps; s++) {
				const int kk = 4 * s + lk;
				const bool kvd = kk < kdim;
				const int bi = kvd ? kk / CG : 0, u = kvd ? kk - bi * CG : 0, bp = bp0 + bi;
				// A[a][kk] = Mst[kk][a]: the basis value of variable u's derivative at the breakpoint, if column a belongs to u's output
				int ua0 = -1, ua1 = -1;
#pragma unroll
				for (int v = 0; v < CG; v++) { if (u == v) { ua0 = v_c0[v]; ua1 = v_c1[v]; } }
				// (loads unconditional at a clamped offset, the value selected afterwards: no divergent branch around an LDS read)
				const double a0v = rowv[NWT_IDX(max(ua0, 0) + bp, D.row_total, "rvu")], a1v = rowv[NWT_IDX(max(ua1, 0) + bp, D.row_total, "rvu")];
				const double A0 = (kvd && ua0 >= 0) ? a0v : 0.0;
				const double A1 = (kvd && ua1 >= 0) ? a1v : 0.0;
				// B[kk][b] = V[kk][b] = sum_v B_i[u][v] Mst_i[v][b]
				double B0 = 0.0, B1 = 0.0;
				const double *Bi = wbuf + NWT_IDX((bi * CG + u) * CG, 216, "wbuf");
#pragma unroll
				for (int v = 0; v < CG; v++) {
					const double bb = kvd ? Bi[v] : 0.0;
					const double r0 = rowv[NWT_IDX(max(v_c0[v], 0) + bp, D.row_total, "rvv")], r1 = rowv[NWT_IDX(max(v_c1[v], 0) + bp, D.row_total, "rvv")];
					B0 += (v_c0[v] >= 0 ? bb : 0.0) * r0;
					B1 += (v_c1[v] >= 0 ? bb : 0.0) * r1;
				}
				S00 = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, B0, S00, 0, 0, 0);
				S10 = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, B0, S10, 0, 0, 0);
				S11 = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, B1, S11, 0, 0, 0);
			}
#pragma unroll
			for (int r = 0; r < 4; r++) {
				if (idx[3 * r] >= 0) Kg[idx[3 * r]] = kv[3 * r] + S00[r];
				if (idx[3 * r + 1] >= 0) Kg[idx[3 * r + 1]] = kv[3 * r + 1] + S10[r];
				if (idx[3 * r + 2] >= 0) Kg[idx[3 * r + 2]] = kv[3 * r + 2] + S11[r];
			}
			nwt_wave_sync();   // the staged blocks are consumed before the next item overwrites them
		}
		__syncthreads();
	}
}
