// kernels.hip -- hand-written HIP kernels for gfx950 (MI355X) and their launchers.
//
//   basis_kernel    PGS knots/interv/bsplvb/bsplvd at every collocation point
//                   (reference call sites colloc.c:92-111; the Fortran is absent from the
//                   reference tree, algorithm from de Boor's PGS)
//   linrows_kernel  banded rows of the linear-constraint matrix (constraints.c:198-261)
//   bounds_kernel   bounds() (constraints.c:5-33)
//   eval_kernel     NPfunobj + NPfuncon (ntg.c:274-371): Z = M C (colloc.c:318-367), cost and
//                   constraint functors, banded gradient / Jacobian assembly (cost.c:38-139,
//                   constraints.c:88-195, colloc.c:243-316), trapezoid quadrature (integrator.c)
//   sqp_kernel      the npsol_() call of ntg.c:250: one workgroup owns one problem for the
//                   whole solve (feasibility, projected inverse-BFGS, line search)
//
// Mapping to CDNA4: one workgroup per problem; breakpoints (then coefficients) across the
// lanes; basis tables, knots/breakpoints and the coefficient vector staged in LDS; the only
// HBM traffic inside a solve is the quasi-Newton history (coalesced, 8 B/lane); reductions are
// 64-lane wavefront shuffles + one LDS hop across the waves of the workgroup.
#include <hip/hip_runtime.h>
#include <math.h>
#include <algorithm>
#include "solve_impl.hpp"


// ------------------------------------------------------------------------------------------
// PGS on the device.  The augmented knot vector is never materialised: t(idx) is read through
// the break sequence (knots(): first/last break k times, interior breaks k-m times).
// ------------------------------------------------------------------------------------------
struct AugKnots {
	const double *brk; int l, k, m;
	__device__ __forceinline__ double operator()(int idx1) const // 1-based like the Fortran
	{
		const int idx = idx1 - 1, n = l * (k - m) + m;
		if (idx < k) return brk[0];
		if (idx >= n) return brk[l];
		return brk[1 + (idx - k) / (k - m)];
	}
};

// interv on the break sequence (2nd-edition rule at the right end), 1-based interval index
__device__ __forceinline__ int interv_breaks(const double *brk, int l, double x)
{
	const int lxt = l + 1;
	if (x < brk[0]) return 1;
	if (x >= brk[lxt - 1]) {
		for (int i = lxt - 1; i >= 1; i--)
			if (brk[i - 1] < brk[lxt - 1]) return i;
		return 1;
	}
	int lo = 1, hi = lxt;
	while (hi - lo > 1) {
		const int mid = (lo + hi) >> 1;
		if (x >= brk[mid - 1]) lo = mid; else hi = mid;
	}
	return lo;
}

// bsplvb/bsplvd (PGS).  db is [nderiv][k] (m-major) == Fortran dbiatx(k,nderiv) column-major.
__device__ void bsplvd_dev(const AugKnots &t, int k, double x, int left, int nderiv, double *db)
{
	double a[NTG_MAX_ORDER * NTG_MAX_ORDER];
	double deltal[NTG_MAX_ORDER], deltar[NTG_MAX_ORDER];
	int j = 1;
	const int mhigh = max(min(nderiv, k), 1), kp1 = k + 1;
#define DB(r, c) db[((c) - 1) * k + ((r) - 1)]
#define A_(r, c) a[((c) - 1) * k + ((r) - 1)]
	auto raise = [&](int jhigh, double *biatx) {
		while (j < jhigh) {
			const int jp1 = j + 1;
			deltar[j - 1] = t(left + j) - x;
			deltal[j - 1] = x - t(left + 1 - j);
			double saved = 0.0;
			for (int i = 1; i <= j; i++) {
				const double term = biatx[i - 1] / (deltar[i - 1] + deltal[jp1 - i - 1]);
				biatx[i - 1] = saved + deltar[i - 1] * term;
				saved = deltal[jp1 - i - 1] * term;
			}
			biatx[jp1 - 1] = saved;
			j = jp1;
		}
	};
	db[0] = 1.0;
	raise(kp1 - mhigh, db);
	if (mhigh == 1) return;
	int ideriv = mhigh;
	for (int m = 2; m <= mhigh; m++) {
		int jp1mid = 1;
		for (int jj = ideriv; jj <= k; jj++) { DB(jj, ideriv) = DB(jp1mid, 1); jp1mid++; }
		ideriv--;
		raise(kp1 - ideriv, db);
	}
	int jlow = 1;
	for (int i = 1; i <= k; i++) {
		for (int jj = jlow; jj <= k; jj++) A_(jj, i) = 0.0;
		jlow = i;
		A_(i, i) = 1.0;
	}
	for (int m = 2; m <= mhigh; m++) {
		const int kp1mm = kp1 - m;
		const double fkp1mm = (double)kp1mm;
		int il = left, i = k;
		for (int ld = 1; ld <= kp1mm; ld++) {
			const double factor = fkp1mm / (t(il + kp1mm) - t(il));
			for (int jj = 1; jj <= i; jj++) A_(i, jj) = (A_(i, jj) - A_(i - 1, jj)) * factor;
			il--; i--;
		}
		for (i = 1; i <= k; i++) {
			double sum = 0.0;
			const int jl = i > m ? i : m;
			for (int jj = jl; jj <= k; jj++) sum = A_(jj, i) * DB(jj, m) + sum;
			DB(i, m) = sum;
		}
	}
#undef DB
#undef A_
}

// one thread per (grid, breakpoint); the grid's break sequence is staged in LDS
__global__ void basis_kernel(int ngrids, int l, int k, int m, int d, int P,
                             const double *__restrict__ knots, const double *__restrict__ bps,
                             long long knots_stride, long long bps_stride,
                             double *__restrict__ blk, int *__restrict__ off)
{
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	double *sbrk = reinterpret_cast<double *>(smem_raw);
	const int g = blockIdx.y;
	if (g >= ngrids) return;
	for (int i = threadIdx.x; i <= l; i += blockDim.x) sbrk[i] = knots[g * knots_stride + i];
	__syncthreads();
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= P) return;
	const double x = bps[g * bps_stride + i];
	const int ivl = interv_breaks(sbrk, l, x);        // colloc.c:107 on the un-augmented knots
	const int left = k + (ivl - 1) * (k - m);         // == interv_ on the augmented knots (colloc.c:98)
	AugKnots t{sbrk, l, k, m};
	double db[NTG_MAX_ORDER * NTG_MAX_ORDER];
	bsplvd_dev(t, k, x, left, d, db);
	double *o = blk + ((size_t)g * P + i) * k * d;
	for (int q = 0; q < k; q++)
		for (int r = 0; r < d; r++) o[q * d + r] = db[r * k + q];   // FTranspose (colloc.c:100-101)
	off[(size_t)g * P + i] = (ivl - 1) * (k - m);                   // colloc.c:108
}

// ------------------------------------------------------------------------------------------
// linear-constraint rows (constraints.c:198-261 through colloc.c:243-316, band entries only)
// row order [lic; ltc constraint-major x breakpoint; lfc]; one thread per (row, band column)
// ------------------------------------------------------------------------------------------
__global__ void linrows_kernel(NtgDims D, NtgTables T, const double *__restrict__ lic,
                               const double *__restrict__ ltc, const double *__restrict__ lfc,
                               double *__restrict__ aband, int *__restrict__ rbp)
{
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= D.nclin * D.sumk) return;
	const int r = idx / D.sumk, kc = idx % D.sumk;
	int o = 0;
	while (o + 1 < D.nout && D.koff[o + 1] <= kc) o++;
	const int q = kc - D.koff[o];
	const double *row; int bp;
	if (r < D.nlic) { row = lic + (size_t)r * D.nz; bp = 0; }
	else if (r < D.nlic + D.nltc * D.P) { const int rr = r - D.nlic; row = ltc + (size_t)(rr / D.P) * D.nz; bp = rr % D.P; }
	else { row = lfc + (size_t)(r - D.nlic - D.nltc * D.P) * D.nz; bp = D.P - 1; }
	const int c = D.cls[o], k = D.order[o], d = D.d[o];
	const double *b = T.blk + D.cls_blk[c] + ((size_t)bp * k + q) * d;
	double acc = 0.0;
	for (int l = 0; l < d; l++) acc += row[D.iz[o] + l] * b[l];
	aband[idx] = acc;
	if (kc == 0) rbp[r] = bp;
}

// bounds(): constraints.c:5-33
__global__ void bounds_kernel(NtgDims D, int batch, const double *__restrict__ lower,
                              const double *__restrict__ upper, double *__restrict__ bl,
                              double *__restrict__ bu, double big)
{
	const int ntot = D.nC + D.nclin + D.ncnln;
	const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= (long long)batch * ntot) return;
	const int b = idx / ntot, e = idx % ntot;
	double lo, up;
	if (e < D.nC) { lo = -big; up = big; }
	else {
		int r = e - D.nC, s;
		if (r < D.nlic) s = r;
		else if ((r -= D.nlic) < D.nltc * D.P) s = D.nlic + r / D.P;
		else if ((r -= D.nltc * D.P) < D.nlfc) s = D.nlic + D.nltc + r;
		else if ((r -= D.nlfc) < D.nnlic) s = D.nlic + D.nltc + D.nlfc + r;
		else if ((r -= D.nnlic) < D.nnltc * D.P) s = D.nlic + D.nltc + D.nlfc + D.nnlic + r / D.P;
		else { r -= D.nnltc * D.P; s = D.nlic + D.nltc + D.nlfc + D.nnlic + D.nnltc + r; }
		lo = lower[(size_t)b * D.nbounds + s]; up = upper[(size_t)b * D.nbounds + s];
	}
	bl[idx] = lo; bu[idx] = up;
}

// ------------------------------------------------------------------------------------------
// host-callback path of the ntg() drop-in: the user's C function pointers run on the host
// between two kernels -- updateZ (colloc.c:344-367) and the banded assembly + quadrature.
// ------------------------------------------------------------------------------------------
// Z in the reference layout Z[iZ[o] + d_o*bp + r], iZ[o] = iz[o]*P (colloc.c:328-331); only the
// declared active variables are written (the buffer persists like GZ, ntg.c:119)
template <int NT>
__global__ void __launch_bounds__(NT)
hostz_kernel(NtgDims D, NtgTables T, SmemLayout L, const double *__restrict__ x, u64 maskI, u64 maskT,
             u64 maskF, double *__restrict__ Z)
{
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	Smem S(smem_raw, L, D, T);
	stage_tables<NT>(D, T, S, smem_raw, L);
	for (int i = threadIdx.x; i < D.nC; i += NT) S.x[i] = x[i];
	lds_sync();
	for (int i = threadIdx.x; i < D.P; i += NT) {
		const u64 mask = maskT | (i == 0 ? maskI : 0ull) | (i == D.P - 1 ? maskF : 0ull);
		if (!mask) continue;
		double z[NTG_MAX_NZ];
		compute_z<0, 0, 3>(D, S, S.x, i, mask, z);
		for (int o = 0; o < D.nout; o++)
			for (int r = 0; r < D.d[o]; r++)
				if ((mask >> (D.iz[o] + r)) & 1ull) Z[(size_t)D.iz[o] * D.P + (size_t)D.d[o] * i + r] = z[D.iz[o] + r];
	}
}

// fT [P], dfT [P][nz] = what ucf returned at each breakpoint; fdI/fdF [nz+1] = (df, f) of icf/fcf
template <int NT>
__global__ void __launch_bounds__(NT)
hostcost_kernel(NtgDims D, NtgTables T, SmemLayout L, const double *__restrict__ fT,
                const double *__restrict__ dfT, const double *__restrict__ fdI,
                const double *__restrict__ fdF, double *__restrict__ F, double *__restrict__ g)
{
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	Smem S(smem_raw, L, D, T);
	stage_tables<NT>(D, T, S, smem_raw, L);
	const int P = D.P, nz = D.nz;
	for (int i = threadIdx.x; i < P; i += NT) S.fvals[i] = D.nucf ? fT[i] : 0.0;
	lds_sync();
	for (int e = threadIdx.x; e < P * nz; e += NT) { const int i = e / nz, v = e % nz; if (D.tav_row[v] >= 0) S.dfz[D.tav_row[v] * (P + 1) + i] = D.nucf ? S.wts[i] * dfT[e] : 0.0; }
	for (int v = threadIdx.x; v <= nz; v += NT) { S.dfi[v] = D.nicf ? fdI[v] : 0.0; S.dff[v] = D.nfcf ? fdF[v] : 0.0; }
	double gn2;
	const double val = cost_phase2<0, 0, NT, 3, 4>(D, S, S.vecs, &gn2, CoefMap<4>(), D.nicf != 0, D.nfcf != 0, 0.0, 0.0, nullptr, nullptr);
	if (threadIdx.x == 0) *F = val;
	for (int i = threadIdx.x; i < D.nC; i += NT) g[i] = S.vecs[i];
}

// dc [ncnln][nz]: row r holds the user's dc[constraint][variable] for constraint row r
// (rows: initial; trajectory constraint-major x breakpoint; final).  One lane per row.
__global__ void hostcon_kernel(NtgDims D, NtgTables T, const double *__restrict__ dc,
                               double *__restrict__ jband, double *__restrict__ cjac)
{
	const int row = blockIdx.x * blockDim.x + threadIdx.x;
	if (row >= D.ncnln) return;
	int bp;
	if (row < D.nnlic) bp = 0;
	else if (row < D.nnlic + D.nnltc * D.P) bp = (row - D.nnlic) % D.P;
	else bp = D.P - 1;
	const double *dcrow = dc + (size_t)row * D.nz;
	for (int o = 0; o < D.nout; o++) {
		const int k = D.order[o], cc = D.cls[o], d = D.d[o];
		const int col0 = D.iC[o] + T.off[cc * D.P + bp];
		for (int q = 0; q < k; q++) {
			double a = 0.0;
			for (int r = 0; r < d; r++) {
				const int chr = T.chrow[cc * NTG_MAX_ORDER + r];
				if (chr >= 0) a += dcrow[D.iz[o] + r] * T.rowv[chr + q * D.P + bp];
			}
			if (jband) jband[(size_t)row * D.sumk + D.koff[o] + q] = a;
			if (cjac) cjac[(size_t)(col0 + q) * D.ncnln + row] = a;
		}
	}
}

hipError_t ntg_launch_hostz(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const double *x, u64 mI,
                            u64 mT, u64 mF, double *Z, hipStream_t st)
{
	auto kfn = hostz_kernel<128>;
	if (L.total > 64 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
	hipLaunchKernelGGL(kfn, dim3(1), dim3(128), L.total, st, D, T, L, x, mI, mT, mF, Z);
	return hipGetLastError();
}
hipError_t ntg_launch_hostcost(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const double *fT,
                               const double *dfT, const double *fdI, const double *fdF, double *F, double *g,
                               hipStream_t st)
{
	auto kfn = hostcost_kernel<128>;
	if (L.total > 64 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
	hipLaunchKernelGGL(kfn, dim3(1), dim3(128), L.total, st, D, T, L, fT, dfT, fdI, fdF, F, g);
	return hipGetLastError();
}
hipError_t ntg_launch_hostcon(const NtgDims &D, const NtgTables &T, const double *dc, double *jband, double *cjac,
                              hipStream_t st)
{
	if (D.ncnln == 0) return hipSuccess;
	hipLaunchKernelGGL(hostcon_kernel, dim3((D.ncnln + 63) / 64), dim3(64), 0, st, D, T, dc, jband, cjac);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// receding-horizon (MPC) shift, SURVEY.md §8f rank 2: after a solve, advance every problem to the
// breakpoint `sbp` of its own solution: (1) the linear initial-constraint bounds are re-pinned to
// the flat flag of the old solution at that breakpoint, b_r = sum_v lic[r][v] z_v(sbp) (the role
// NPSOL's warm start arrays istate/clambda/R were meant for, ntg.h:64-68, never used by the
// reference); (2) the coefficient vector is shifted by `sknot` knot intervals, C_new[j] =
// C_old[j + sknot (k-m)], the tail holding the last coefficient -- the warm start of the next solve.
// One workgroup per problem.
// ------------------------------------------------------------------------------------------
__global__ void mpc_shift_kernel(NtgDims D, NtgTables T, int batch, int sbp, int sknot, const double *__restrict__ lic,
                                 double *__restrict__ x, double *__restrict__ lower, double *__restrict__ upper)
{
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	double *sx = (double *)smem_raw, *z = sx + ((D.nC + 1) & ~1);
	const int b = blockIdx.x, tid = threadIdx.x;
	if (b >= batch) return;
	for (int i = tid; i < D.nC; i += blockDim.x) sx[i] = x[(size_t)b * D.nC + i];
	__syncthreads();
	// flat flag of the old solution at breakpoint sbp (every derivative, Zvalue of colloc.c:318-326)
	for (int v = tid; v < D.nz; v += blockDim.x) {
		int o = 0;
		while (o + 1 < D.nout && D.iz[o + 1] <= v) o++;
		const int r = v - D.iz[o], k = D.order[o], d = D.d[o], c = D.cls[o];
		const double *bq = T.blk + (size_t)b * T.pp_blk + D.cls_blk[c] + (size_t)sbp * k * d;   // (per-problem grids: this problem's blocks)
		const double *cx = sx + D.iC[o] + T.off[c * D.P + sbp];
		double a = 0.0;
		for (int q = 0; q < k; q++) a += bq[q * d + r] * cx[q];
		z[v] = a;
	}
	__syncthreads();
	for (int r = tid; r < D.nlic; r += blockDim.x) {
		double a = 0.0;
		for (int v = 0; v < D.nz; v++) a += lic[(size_t)r * D.nz + v] * z[v];
		lower[(size_t)b * D.nbounds + r] = a;
		upper[(size_t)b * D.nbounds + r] = a;
	}
	for (int c = tid; c < D.nC; c += blockDim.x) {
		int o = 0;
		while (o + 1 < D.nout && D.iC[o + 1] <= c) o++;
		const int cl = c - D.iC[o], sh = sknot * (D.order[o] - D.mult[o]), src = cl + sh;
		x[(size_t)b * D.nC + c] = sx[D.iC[o] + (src < D.ncoef[o] ? src : D.ncoef[o] - 1)];
	}
}

hipError_t ntg_launch_mpc_shift(const NtgDims &D, const NtgTables &T, int batch, int sbp, int sknot, const double *lic,
                                double *x, double *lower, double *upper, hipStream_t st)
{
	if (batch <= 0) return hipSuccess;
	const size_t sm = (size_t)(((D.nC + 1) & ~1) + D.nz + 2) * 8;
	hipLaunchKernelGGL(mpc_shift_kernel, dim3(batch), dim3(128), sm, st, D, T, batch, sbp, sknot, lic, x, lower, upper);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// launchers (called from plan.cpp)
// ------------------------------------------------------------------------------------------
static inline int align16(int x) { return (x + 15) & ~15; }

// derivative orders named by the trajectory-constraint active variables: the channels the row emission of the banded Jacobian lists
// (eval_constraints / emit_decode; an upper bound is all the layout needs)
static int ntg_emit_channels(const NtgDims &D)
{
	unsigned um = 0;
	for (int o = 0; o < D.nout; o++)
		for (int r = 0; r < D.d[o] && r < NTG_MAX_ORDER; r++)
			if ((D.tcon_mask >> (D.iz[o] + r)) & 1ull) um |= 1u << r;
	return std::max(1, __builtin_popcount(um));
}

SmemLayout ntg_make_layout(const NtgDims &D, int nthreads, int nvec, int with_x, int hrc_pairs, int qp)
{
	SmemLayout L;
	int p = 0;
	const int npad = (D.nC + 1) & ~1;
	L.rowv = p; p = align16(p + D.row_total * 8);
	L.colp = p; p = align16(p + D.col_total * 4);
	L.chrow = p; p = align16(p + D.nclass * NTG_MAX_ORDER * 4);
	L.chcol = p; p = align16(p + D.nclass * NTG_MAX_ORDER * 4);
	L.off = p; p = align16(p + D.nclass * D.P * 4);
	L.bps = p; p = align16(p + D.P * 8);
	L.wts = p; p = align16(p + D.P * 8);
	L.x = p; if (with_x) p = align16(p + npad * 8);
	// [row][P+1] + a zero tail: the column form reads W consecutive entries from a column's first breakpoint
	{
		// the structured Newton mode borrows this area between evaluations: solve vectors and factorisation panels of every group
		// evaluation layout (nvec == 0): the cost pass touches the cost's rows only, and the constraint pass wants room for the
		// derivative rows of ONE constraint (it chunks over the constraints)
		const int ncomp = __builtin_popcountll(D.tcon_mask);
		L.dfz_rows = nvec > 0 ? D.ntav : std::max(D.ntav_cost, (ncomp * D.P + D.P) / (D.P + 1));
		if (L.dfz_rows < 1) L.dfz_rows = 1;
		L.tav_rows = nvec > 0 ? D.ntav : D.ntav_cost;
		int dfz_bytes = (L.dfz_rows * (D.P + 1) + ntg_dfz_tail(D)) * 8;
		if (D.nwt_on && nvec > 0) dfz_bytes = std::max(dfz_bytes,   // (solve layouts only: the evaluation kernels never factor anything)
		                                   std::max(D.nwt_ngrp * ((D.nwt_tw ? 16 * (D.nwt_ja + 3) + 48 + 16 * (D.nwt_jb + 3) + 48 : 16 * ((D.nwt_ng + 15) / 16) + 48) + (D.nwt_tw ? 2 : 1) * 416 /* NWT_PANEL */) * 8 + D.nwt_nfo * (16 * ((D.nwt_ngf + 15) / 16) + 48) * 8 + (qp ? (D.nwt_ngrp * ntg_qp_doubles(ntg_qp_maxa(D.nwt_ngrp, nthreads)) + NTG_QP_RED) * 8 : 0) /* QP-based SQP step: the groups' slots, scratch of the entering-row search */, (nthreads / 64) * 216 * 8 + D.nwt_ngrp * ((D.P + 63) / 64) * 8));   // solve vectors + panels | the assembly's staging buffers + the block flags
		L.dfz = p; L.nwt_y = p; p = align16(p + dfz_bytes);
	}
	L.fvals = p; p = align16(p + D.P * 8);
	L.red = p; p = align16(p + (32 * (nthreads / 64) + 2) * 8);   // two halves of 16 values x waves (block_sum) + the Newton mode's flag words
	L.dfi = p; p = align16(p + (D.nz + 1) * 8);
	L.dff = p; p = align16(p + (D.nz + 1) * 8);
	// evaluation layouts (nvec == 0) carry no solver state: no vectors, multipliers or line-search records (config D's evaluation sits
	// 640 bytes under the two-workgroups-per-CU line)
	L.vecs = p; if (nvec > 0) p = align16(p + (nvec * npad + 2 * D.nclin + 2) * 8);
	L.lam = p; if (nvec > 0) p = align16(p + (D.nclin + 1) * 8);
	L.rho = L.c2 = p; L.hrc_n = hrc_pairs; p = align16(p + 2 * hrc_pairs * 8);   // (rho_i, c2_i) of a short quasi-Newton memory; longer ones keep them with the pair in HBM
	L.oinfo = p; p = align16(p + D.nout * 10 * 4);
	L.tavrow = p; p = align16(p + D.nz * 4);
	L.tcomp = p; p = align16(p + D.nz * 4);
	L.ls = p; if (nvec > 0) p = align16(p + 2 * (int)sizeof(LineSearch));   // double buffered (see sqp_kernel)
	L.tI = p; p = align16(p + (D.nI + 1) * 8);   // multiplier estimates of the linear inequality rows
	L.q_idx = L.q_col = L.q_val = p;
	L.with_lin = nvec > 0;
	if (D.q_use && L.with_lin) {
		L.q_idx = p; p = align16(p + D.nC * 2);
		L.q_col = p; p = align16(p + D.q_nt * D.q_w * 4);
		L.q_val = p; p = align16(p + D.q_nt * D.q_w * 8);
	}
	L.csr_ptr = L.csr_col = L.csr_val = L.csc_ptr = L.csc_row = L.csc_val = L.sinv_ptr = L.sinv_col = L.sinv_val = p;
	if (D.lin_lds && L.with_lin) {
		L.csr_ptr = p; p = align16(p + (D.nclin + 1) * 4);
		L.csr_col = p; p = align16(p + D.lin_nnz * 4);
		L.csr_val = p; p = align16(p + D.lin_nnz * 8);
		L.csc_ptr = p; p = align16(p + (D.nC + 1) * 4);
		L.csc_row = p; p = align16(p + D.lin_nnz * 4);
		L.csc_val = p; p = align16(p + D.lin_nnz * 8);
		L.sinv_ptr = p; p = align16(p + (D.nclin + 1) * 4);
		L.sinv_col = p; p = align16(p + D.sinv_nnz * 4);
		L.sinv_val = p; p = align16(p + D.sinv_nnz * 8);
	}
	// row emission of the banded Jacobian (eval_constraints): [64 lanes] (pair index | row slot) + [NTG_MAX_ORDER][64 lanes] (scratch offset,
	// table offset) + the number of listed channels -- decoded once per workgroup instead of once per problem
	L.emit = -1;
	if (nvec == 0 && D.nnltc > 0) { L.emit = p; p = align16(p + 64 * 4 + ntg_emit_channels(D) * 64 * 8 + 16); }
	L.total = p;
	return L;
}

// family dispatch: every family has its own translation unit (fam_*.hip) that picks between its tuned
// (compile-time nout / order) instances and the generic one
#define NTG_FAM_DECL(NAME)                                                                                         \
	hipError_t ntg_launch_eval_##NAME(const NtgDims &, const NtgTables &, const SmemLayout &, const EvalArgs &);   \
	hipError_t ntg_launch_sqp_##NAME(const NtgDims &, const NtgTables &, const SmemLayout &, const SolveParams &, const SqpArgs &);
NTG_FAM_DECL(kincar) NTG_FAM_DECL(vanderpol) NTG_FAM_DECL(testfam) NTG_FAM_DECL(obstacle) NTG_FAM_DECL(quadrotor) NTG_FAM_DECL(manip)
#undef NTG_FAM_DECL

hipError_t ntg_launch_eval(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const EvalArgs &a)
{
	switch (D.family) {
	case NTG_FAM_KINCAR: return ntg_launch_eval_kincar(D, T, L, a);
	case NTG_FAM_VANDERPOL: return ntg_launch_eval_vanderpol(D, T, L, a);
	case NTG_FAM_TESTFAM: return ntg_launch_eval_testfam(D, T, L, a);
	case NTG_FAM_OBSTACLE: return ntg_launch_eval_obstacle(D, T, L, a);
	case NTG_FAM_QUADROTOR: return ntg_launch_eval_quadrotor(D, T, L, a);
	case NTG_FAM_MANIP: return ntg_launch_eval_manip(D, T, L, a);
	}
	return hipErrorInvalidValue;
}

hipError_t ntg_launch_sqp(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const SolveParams &sp, const SqpArgs &a)
{
	switch (D.family) {
	case NTG_FAM_KINCAR: return ntg_launch_sqp_kincar(D, T, L, sp, a);
	case NTG_FAM_VANDERPOL: return ntg_launch_sqp_vanderpol(D, T, L, sp, a);
	case NTG_FAM_TESTFAM: return ntg_launch_sqp_testfam(D, T, L, sp, a);
	case NTG_FAM_OBSTACLE: return ntg_launch_sqp_obstacle(D, T, L, sp, a);
	case NTG_FAM_QUADROTOR: return ntg_launch_sqp_quadrotor(D, T, L, sp, a);
	case NTG_FAM_MANIP: return ntg_launch_sqp_manip(D, T, L, sp, a);
	}
	return hipErrorInvalidValue;
}

hipError_t ntg_launch_basis(int ngrids, int l, int k, int m, int d, int P, const double *knots, const double *bps,
                            long long knots_stride, long long bps_stride, double *blk, int *off, hipStream_t st)
{
	const int nt = 64;
	dim3 grid((P + nt - 1) / nt, ngrids);
	hipLaunchKernelGGL(basis_kernel, grid, dim3(nt), (size_t)(l + 1) * 8, st, ngrids, l, k, m, d, P, knots, bps,
	                   knots_stride, bps_stride, blk, off);
	return hipGetLastError();
}

// Receding horizon: the multiplier estimates of the trajectory rows move sbp breakpoints towards the start (row (j, i) <- row (j, i + sbp));
// what enters the window at its end starts at 0.  alw: [batch][2][nal], estimates in the second half (what a warm-started solve reads).
// One workgroup per (problem, trajectory constraint); the read of element i + sbp happens before any lane writes it (two phases).
__global__ void mpc_shift_lambda_kernel(NtgDims D, int batch, int sbp, double *__restrict__ alw)
{
	extern __shared__ double s_row[];
	const int b = blockIdx.x / D.nnltc, j = blockIdx.x % D.nnltc, P = D.P, nal = D.ncnln + D.nI;
	if (b >= batch) return;
	double *t = alw + (size_t)b * 2 * nal + nal + D.nnlic + (size_t)j * P;
	for (int i = threadIdx.x; i < P; i += blockDim.x) s_row[i] = i + sbp < P ? t[i + sbp] : 0.0;
	__syncthreads();
	for (int i = threadIdx.x; i < P; i += blockDim.x) t[i] = s_row[i];
}
hipError_t ntg_launch_mpc_shift_lambda(const NtgDims &D, int batch, int sbp, double *alw, hipStream_t st)
{
	if (D.nnltc <= 0) return hipSuccess;
	hipLaunchKernelGGL(mpc_shift_lambda_kernel, dim3(batch * D.nnltc), dim3(128), (size_t)D.P * 8, st, D, batch, sbp, alw);
	return hipGetLastError();
}

// SplineInterp (colloc.c:449-484) for a batch: flat flag of every problem at ntimes shared points in time.
// tblk [class][t][q][r] and toff [class][t] come from basis_kernel run on the times instead of the breakpoints;
// one thread per (problem, time, flag entry), consecutive threads = consecutive flag entries of one time.
// Per-problem grids: every problem has its own times, basis blocks and offsets (pp != 0: tblk [batch][t][q][r], toff [batch][t]; one class).
__global__ void interp_kernel(NtgDims D, int batch, int ntimes, const double *__restrict__ x, const double *__restrict__ tblk,
                              const int *__restrict__ toff, const int *__restrict__ tblk_base, double *__restrict__ z, int pp)
{
	const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	const long long total = (long long)batch * ntimes * D.nz;
	if (idx >= total) return;
	const int v = (int)(idx % D.nz), t = (int)((idx / D.nz) % ntimes), b = (int)(idx / ((long long)D.nz * ntimes));
	int o = 0;
	while (o + 1 < D.nout && D.iz[o + 1] <= v) o++;
	const int r = v - D.iz[o], c = D.cls[o], kk = D.order[o], d = D.d[o];
	const double *blk = tblk + (pp ? (size_t)b * ntimes * kk * d : (size_t)tblk_base[c]) + ((size_t)t * kk) * d;
	const double *cx = x + (size_t)b * D.nC + D.iC[o] + toff[(pp ? (size_t)b : (size_t)c) * ntimes + t];
	double acc = 0.0;
	for (int q = 0; q < kk; q++) acc += blk[q * d + r] * cx[q];   // colloc.c:476-481
	z[idx] = acc;
}

hipError_t ntg_launch_interp(const NtgDims &D, int batch, int ntimes, const double *x, const double *tblk, const int *toff,
                             const int *tblk_base, double *z, hipStream_t st, int pp)
{
	const long long total = (long long)batch * ntimes * D.nz;
	if (total == 0) return hipSuccess;
	hipLaunchKernelGGL(interp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, D, batch, ntimes, x, tblk, toff, tblk_base, z, pp);
	return hipGetLastError();
}

// Flat flag -> state and input of the kinematic car, the map of examples/kincar.c:68-92 (kincar_flat_reverse), for every car
// of every problem at every time of an ntg_batch_interp result: z [n][nz] with the flag of car c at entries 6 c .. 6 c + 5
// (x, x', x'', y, y', y'') -> out [n][ncars][5] = x, y, theta, v, delta.  One thread per (sample, car).
__global__ void kincar_reverse_kernel(long long nsamp, int nz, int ncars, double wheelbase, int reverse_gear,
                                      const double *__restrict__ z, double *__restrict__ out)
{
	const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= nsamp * ncars) return;
	const long long smp = idx / ncars; const int c = (int)(idx - smp * ncars);
	const double *f = z + smp * nz + 6 * c;
	const double xd = f[1], xdd = f[2], yd = f[4], ydd = f[5];
	const double th = reverse_gear ? atan2(-yd, -xd) : atan2(yd, xd);   // kincar.c:78-86
	double sn, cs;
	sincos(th, &sn, &cs);
	const double thdot_v = ydd * cs - xdd * sn, v = xd * cs + yd * sn;   // kincar.c:89-90
	double *o = out + idx * 5;
	o[0] = f[0]; o[1] = f[3]; o[2] = th; o[3] = v; o[4] = atan2(thdot_v, v * v / wheelbase);   // kincar.c:91 (pow(u[0], 2.0) / b)
}
hipError_t ntg_launch_kincar_reverse(long long nsamp, int nz, int ncars, double wheelbase, int reverse_gear, const double *z, double *out, hipStream_t st)
{
	const long long total = nsamp * ncars;
	if (total == 0) return hipSuccess;
	hipLaunchKernelGGL(kincar_reverse_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, nsamp, nz, ncars, wheelbase, reverse_gear, z, out);
	return hipGetLastError();
}

// receding-horizon bookkeeping: how many problems of the last re-solve did not end with inform 0
__global__ void count_notconv_kernel(int batch, const int *__restrict__ inform, int *__restrict__ count)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	const int bad = (i < batch && inform[i] != 0) ? 1 : 0;
	const unsigned long long m = __ballot(bad);
	if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, __popcll(m));
}
hipError_t ntg_launch_count_notconv(int batch, const int *inform, int *count, hipStream_t st)
{
	hipLaunchKernelGGL(count_notconv_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, batch, inform, count);
	return hipGetLastError();
}

hipError_t ntg_launch_linrows(const NtgDims &D, const NtgTables &T, const double *lic, const double *ltc,
                              const double *lfc, double *aband, int *rbp, hipStream_t st)
{
	const int total = D.nclin * D.sumk;
	if (total == 0) return hipSuccess;
	hipLaunchKernelGGL(linrows_kernel, dim3((total + 127) / 128), dim3(128), 0, st, D, T, lic, ltc, lfc, aband, rbp);
	return hipGetLastError();
}

hipError_t ntg_launch_bounds(const NtgDims &D, int batch, const double *lo, const double *up, double *bl, double *bu,
                             hipStream_t st)
{
	const long long total = (long long)batch * (D.nC + D.nclin + D.ncnln);
	if (total == 0) return hipSuccess;
	hipLaunchKernelGGL(bounds_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, D, batch, lo, up, bl, bu,
	                   1.7976931348623157e308);
	return hipGetLastError();
}


