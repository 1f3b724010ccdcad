// grids.hip -- per-problem grids on the device: the setup phase of ntg() (ntg.c:114-229) for every problem of a batch.
//
// ntg() builds its collocation per call (CollocMatrix per output, colloc.c:57-117; LinearConstraintsMatrix, constraints.c:198-261).
// With per-problem grids the plan's combinatorial structure is shared and the VALUES are per problem (ntg_plan_set_grids, plan.cpp):
// basis_kernel evaluates the basis blocks of every grid; the two kernels here derive everything else from them without leaving the
// device (round 2 did this algebra on host threads: 0.2 - 1.5 s for 16 384 grids):
//   grid_rows_kernel  channel rows rowv[chrow[r] + q P + i] = D^r B_{off+q}(bps_i) (the layout eval_kernel / sqp_kernel stage), and
//                     the check that every breakpoint lies in the plan's knot interval (block[i].offset, colloc.c:104-111)
//   grid_lin_kernel   one wavefront per problem: the rows of A_E (LinearConstraintsMatrix: the user's lic / ltc / lfc rows through
//                     the basis blocks of their breakpoint, constraints.c:225-261) as values of the plan's CSR / CSC patterns, a test
//                     that nothing of weight falls outside the pattern, S = A A' in LDS, its Cholesky factor, (A A')^-1 on the plan's
//                     pattern, and the projector Q = A'(A A')^-1 A on the plan's ELL pattern.
// A row of A_E has support nout * k (the k basis functions of its breakpoint, for every output): the kernel keeps the m supports in
// LDS and never forms the dense m x nC matrix.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "ntg_dev.hpp"
#include "plan.hpp"

// lanes of one wave hand data over through LDS: the LDS queue of a wave is processed in order, the compiler must not reorder or forward
__device__ __forceinline__ void nwt_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__global__ void grid_rows_kernel(NtgDims D, int batch, const double *__restrict__ blk, const int *__restrict__ off, const int *__restrict__ plan_off,
                                 double *__restrict__ rowv, int *__restrict__ err)
{
	const int b = blockIdx.x, P = D.P, k = D.cls_k[0], d = D.cls_d[0];
	if (b >= batch) return;
	const double *bk = blk + (size_t)b * P * k * d;
	double *rv = rowv + (size_t)b * D.row_total;
	for (int idx = threadIdx.x; idx < d * k * P; idx += blockDim.x) {
		const int i = idx % P, q = (idx / P) % k, r = idx / (P * k);
		const int ch = D.ch_row0[r];
		if (ch >= 0) rv[ch + q * P + i] = bk[((size_t)i * k + q) * d + r];   // (the one trailing zero per channel stays from the memset)
	}
	for (int i = threadIdx.x; i < P; i += blockDim.x)
		if (off[(size_t)b * P + i] != plan_off[i]) { if (atomicCAS(&err[0], 0, 1) == 0) { err[1] = b; err[2] = i; } }
}

struct GridLinArgs {
	const double *blk;        // [batch][P][k][d]
	const double *linrows;    // [nclin][nz] the user's lic / ltc / lfc rows, stacked
	const int *plan_off;      // [P]
	const int *erow, *csr_ptr, *csr_col, *csc_ptr, *csc_row, *sinv_ptr, *sinv_col, *q_col, *q_row2coef;
	const unsigned char *q_pad;   // [q_nt][q_w] 1: padding entry of the ELL pattern (stays 0)
	double *csr_val, *csc_val, *sinv_val, *q_val;   // per problem, strides lin_nnz / lin_nnz / sinv_nnz / q_nt q_w
	int *err;
};

// value of A_E(i, c) from the row supports in LDS: c = o nco + cl, the row's block starts at offe[i]
__device__ __forceinline__ double grid_getA(const double *sup, const int *offe, int sw, int k, int nco, int i, int c)
{
	const int o = c / nco, q = c - o * nco - offe[i];
	return (q >= 0 && q < k) ? sup[i * sw + o * k + q] : 0.0;
}

__global__ void __launch_bounds__(64)
grid_lin_kernel(NtgDims D, int batch, GridLinArgs A)
{
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	const int b = blockIdx.x, lane = threadIdx.x, m = D.mE, P = D.P, k = D.cls_k[0], d = D.cls_d[0], nout = D.nout, nco = D.ncoef[0], nz = D.nz;
	if (b >= batch) return;
	const int sw = nout * k, mp = m + 1;
	double *sup = (double *)smem_raw;          // [m][sw] supports of the rows
	double *S = sup + (size_t)m * sw;           // [m][m+1] A A', then its Cholesky factor (lower)
	double *Si = S + (size_t)m * mp;            // [m][m+1] (A A')^-1
	int *offe = (int *)(Si + (size_t)m * mp);   // [m] block offset of the row's breakpoint
	const double *bk = A.blk + (size_t)b * P * k * d;
	const int lnz = D.lin_nnz > 0 ? D.lin_nnz : 1, snz = D.sinv_nnz > 0 ? D.sinv_nnz : 1, qn = D.q_use ? D.q_nt * D.q_w : 0;
	// 1. supports: sup[e][o k + q] = sum_l row_e[iz[o] + l] * D^l B_{off+q}(bp_e)     (dIdz2dIdZ* + CollocConcatMult*, constraints.c:225-261)
	for (int idx = lane; idx < m * sw; idx += 64) {
		const int e = idx / sw, oq = idx - e * sw, o = oq / k, q = oq - o * k;
		const int r = A.erow[e];
		const double *row; int bp;
		if (r < D.nlic) { row = A.linrows + (size_t)r * nz; bp = 0; }
		else if (r < D.nlic + D.nltc * P) { const int rr = r - D.nlic; row = A.linrows + (size_t)(D.nlic + rr / P) * nz; bp = rr % P; }
		else { row = A.linrows + (size_t)(D.nlic + D.nltc + (r - D.nlic - D.nltc * P)) * nz; bp = P - 1; }
		double acc = 0.0;
		for (int l = 0; l < d; l++) acc += row[D.iz[o] + l] * bk[((size_t)bp * k + q) * d + l];
		sup[idx] = acc;
		if (oq == 0) offe[e] = A.plan_off[bp];
	}
	nwt_wave_sync();
	// 2. values on the plan's patterns; nothing of weight outside them (relative threshold as in build_newton_tables(): a final breakpoint
	//    one ulp past the last knot leaves ~1e-16 basis values where the plan has exact zeros)
	for (int i = lane; i < m; i += 64) {
		double rmax = 0.0;
		for (int oq = 0; oq < sw; oq++) rmax = fmax(rmax, fabs(sup[i * sw + oq]));
		for (int oq = 0; oq < sw; oq++) {
			const int o = oq / k, c = o * nco + offe[i] + (oq - o * k);
			bool in = false;
			for (int e = A.csr_ptr[i]; e < A.csr_ptr[i + 1]; e++) if (A.csr_col[e] == c) { in = true; break; }
			if (!in && fabs(sup[i * sw + oq]) > 1e-10 * rmax) { if (atomicCAS(&A.err[0], 0, 2) == 0) { A.err[1] = b; A.err[2] = i; } }
		}
		for (int e = A.csr_ptr[i]; e < A.csr_ptr[i + 1]; e++) A.csr_val[(size_t)b * lnz + e] = grid_getA(sup, offe, sw, k, nco, i, A.csr_col[e]);
	}
	for (int c = lane; c < D.nC; c += 64)
		for (int e = A.csc_ptr[c]; e < A.csc_ptr[c + 1]; e++) A.csc_val[(size_t)b * lnz + e] = grid_getA(sup, offe, sw, k, nco, A.csc_row[e], c);
	// 3. S = A A'
	for (int idx = lane; idx < m * m; idx += 64) {
		const int i = idx / m, j = idx - i * m;
		if (j > i) continue;
		const int sh = offe[i] - offe[j];
		double acc = 0.0;
		for (int o = 0; o < nout; o++)
			for (int q = 0; q < k; q++) { const int qj = q + sh; if (qj >= 0 && qj < k) acc += sup[i * sw + o * k + q] * sup[j * sw + o * k + qj]; }
		S[i * mp + j] = acc; S[j * mp + i] = acc;
	}
	nwt_wave_sync();
	// 4. Cholesky S = L L' in place (lane = row), right-looking
	bool bad = false;
	for (int j = 0; j < m; j++) {
		const double dj = S[j * mp + j];
		if (!(dj > 0.0)) { bad = true; break; }
		const double ld = sqrt(dj);
		nwt_wave_sync();
		for (int i = lane; i < m; i += 64) {
			if (i == j) S[i * mp + j] = ld;
			else if (i > j) S[i * mp + j] = S[i * mp + j] / ld;
		}
		nwt_wave_sync();
		for (int i = lane; i < m; i += 64)
			if (i > j) { const double lij = S[i * mp + j]; for (int k2 = j + 1; k2 <= i; k2++) S[i * mp + k2] -= lij * S[k2 * mp + j]; }
		nwt_wave_sync();
	}
	if (bad) { if (lane == 0 && atomicCAS(&A.err[0], 0, 3) == 0) { A.err[1] = b; A.err[2] = 0; } return; }
	// 5. (A A')^-1 column by column (lane = column): L y = e_c, L' x = y
	for (int c = lane; c < m; c += 64) {
		for (int i = 0; i < m; i++) {
			double s = (i == c) ? 1.0 : 0.0;
			for (int k2 = 0; k2 < i; k2++) s -= S[i * mp + k2] * Si[k2 * mp + c];
			Si[i * mp + c] = s / S[i * mp + i];
		}
		for (int i = m - 1; i >= 0; i--) {
			double s = Si[i * mp + c];
			for (int k2 = i + 1; k2 < m; k2++) s -= S[k2 * mp + i] * Si[k2 * mp + c];
			Si[i * mp + c] = s / S[i * mp + i];
		}
	}
	nwt_wave_sync();
	for (int idx = lane; idx < m * m; idx += 64) {   // symmetrise (as the host setup does)
		const int i = idx / m, j = idx - i * m;
		if (j < i) { const double a = 0.5 * (Si[i * mp + j] + Si[j * mp + i]); S[i * mp + j] = a; S[j * mp + i] = a; }
		else if (j == i) S[i * mp + i] = Si[i * mp + i];
	}
	nwt_wave_sync();   // S now holds the symmetrised inverse
	for (int i = lane; i < m; i += 64)
		for (int e = A.sinv_ptr[i]; e < A.sinv_ptr[i + 1]; e++) A.sinv_val[(size_t)b * snz + e] = S[i * mp + A.sinv_col[e]];
	// 6. projector Q = A'(A A')^-1 A on the plan's ELL pattern (its padding entries stay 0)
	for (int idx = lane; idx < qn; idx += 64) {
		double acc = 0.0;
		if (!A.q_pad[idx]) {
			const int t = idx / D.q_w, a = A.q_row2coef[t], c = A.q_col[idx];
			for (int i = 0; i < m; i++) {
				const double aia = grid_getA(sup, offe, sw, k, nco, i, a);
				if (aia == 0.0) continue;
				double yj = 0.0;
				for (int j = 0; j < m; j++) yj += S[i * mp + j] * grid_getA(sup, offe, sw, k, nco, j, c);
				acc += aia * yj;
			}
		}
		A.q_val[(size_t)b * qn + idx] = acc;
	}
}

hipError_t ntg_launch_grid_rows(const NtgDims &D, int batch, const double *blk, const int *off, const int *plan_off, double *rowv, int *err, hipStream_t st)
{
	hipLaunchKernelGGL(grid_rows_kernel, dim3(batch), dim3(256), 0, st, D, batch, blk, off, plan_off, rowv, err);
	return hipGetLastError();
}

hipError_t ntg_launch_grid_lin(const NtgDims &D, int batch, const NtgGridLin &g, hipStream_t st)
{
	GridLinArgs A;
	A.blk = g.blk; A.linrows = g.linrows; A.plan_off = g.plan_off; A.erow = g.erow; A.csr_ptr = g.csr_ptr; A.csr_col = g.csr_col;
	A.csc_ptr = g.csc_ptr; A.csc_row = g.csc_row; A.sinv_ptr = g.sinv_ptr; A.sinv_col = g.sinv_col; A.q_col = g.q_col; A.q_row2coef = g.q_row2coef;
	A.q_pad = g.q_pad; A.csr_val = g.csr_val; A.csc_val = g.csc_val; A.sinv_val = g.sinv_val; A.q_val = g.q_val; A.err = g.err;
	const int m = D.mE, sw = D.nout * D.cls_k[0];
	const size_t lds = ((size_t)m * sw + 2 * (size_t)m * (m + 1)) * 8 + (size_t)((m + 3) & ~3) * 4;
	if (lds > 160 * 1024) return hipErrorInvalidValue;
	if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)grid_lin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipLaunchKernelGGL(grid_lin_kernel, dim3(batch), dim3(64), lds, st, D, batch, A);
	return hipGetLastError();
}

// ---- preconditioner blocks of every grid (hessian = 1 on per-problem grids) ----
// W0 = Z (Z' H0 Z)^-1 Z' per distinct block (build_precond / precond_block of plan.cpp).  When the equality rows that touch a block pin whole
// coefficients (a square invertible system: the usual end conditions), null(A) is the coordinate subspace of the free coefficients, Z can
// be taken as their unit vectors -- an orthonormal basis like the Householder one of the host routine, and W0 with its regularisation
// (c I in an orthonormal basis) does not depend on which -- so W0 is the inverse of the principal submatrix of H0 over the free
// coefficients, embedded in zeros.  One wavefront per (problem, block): H0 on the free coefficients from the problem's basis blocks and
// node weights (lane = row), the host's regularisation and its retry, Cholesky and inverse in LDS (as grid_lin_kernel does for A A'), the
// host's drop rule for entries at rounding level.  err: 3 = not positive definite (host: NTG_E_UNSUPPORTED), as from the host path.
struct GridPrecArgs {
	const double *blk, *bps;   // [batch][P][k][d], [batch][P]
	const int *plan_off;       // [P]
	const int *fidx;           // [nblk][nb] index of a coefficient among the block's free ones, -1: pinned
	const int *binfo;          // [nblk][4]: free count, derivative mask of the running / initial / final cost of the block's outputs
	double *n0b;               // [batch][n0b_sz]
	int *err;
	int nblk, nb, spad, n0b_sz;
};

__global__ void __launch_bounds__(64)
grid_prec_kernel(NtgDims D, int batch, GridPrecArgs A)
{
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	const int b = blockIdx.x / A.nblk, q = blockIdx.x - b * A.nblk, lane = threadIdx.x;
	if (b >= batch) return;
	const int P = D.P, k = D.cls_k[0], d = D.cls_d[0], nb = A.nb;
	const int nr = A.binfo[4 * q], tmask = A.binfo[4 * q + 1], imask = A.binfo[4 * q + 2], fmask = A.binfo[4 * q + 3], mp = nr + 1;
	double *H = (double *)smem_raw;      // [nr][nr+1]
	double *G = H + (size_t)nr * mp;      // [nr][nr+1] copy of H (retry), then the inverse
	const int *fx = A.fidx + (size_t)q * nb;
	const double *bk = A.blk + (size_t)b * P * k * d, *tb = A.bps + (size_t)b * P;
	double *out = A.n0b + (size_t)b * A.n0b_sz + (size_t)q * A.spad * nb;
	// 1. H0 on the free coefficients: lane = row a (coefficient c with fx[c] == a)
	for (int idx = lane; idx < nr * mp; idx += 64) H[idx] = 0.0;
	nwt_wave_sync();
	for (int c = 0; c < nb; c++) {
		const int a = fx[c];
		if (a < 0 || (a & 63) != lane) continue;   // rows a, a + 64, ... belong to lane a % 64
		for (int i = 0; i < P; i++) {
			const int base = A.plan_off[i], q1 = c - base;
			if (q1 < 0 || q1 >= k) continue;
			double w = 0.0;
			if (i > 0) w += (tb[i] - tb[i - 1]) / 2;
			if (i < P - 1) w += (tb[i + 1] - tb[i]) / 2;
			const double *bb = bk + (size_t)i * k * d;
			for (int q2 = 0; q2 < k; q2++) {
				const int a2 = fx[base + q2];
				if (a2 < 0) continue;
				double s = 0.0;
				for (int r = 0; r < d; r++) {
					const double pr = bb[q1 * d + r] * bb[q2 * d + r];
					if ((tmask >> r) & 1) s += w * pr;
					if (i == 0 && ((imask >> r) & 1)) s += pr;
					if (i == P - 1 && ((fmask >> r) & 1)) s += pr;
				}
				H[a * mp + a2] += s;
			}
		}
	}
	nwt_wave_sync();
	double tr = 0.0;
	for (int a = 0; a < nr; a++) tr += H[a * mp + a];   // (every lane: nr reads of LDS broadcasts)
	for (int idx = lane; idx < nr * mp; idx += 64) G[idx] = H[idx];
	nwt_wave_sync();
	bool ok = false;
	for (int attempt = 0; attempt < 2 && !ok; attempt++) {
		const double reg = attempt == 0 ? 1e-12 : 1e-6;
		for (int idx = lane; idx < nr * mp; idx += 64) H[idx] = G[idx];
		nwt_wave_sync();
		for (int a = lane; a < nr; a += 64) H[a * mp + a] += reg * tr / nr + 1e-300;
		nwt_wave_sync();
		// Cholesky H = L L' in place (lane = row), right-looking
		bool bad = false;
		for (int j = 0; j < nr; j++) {
			const double dj = H[j * mp + j];
			if (!(dj > 0.0)) { bad = true; break; }
			const double ld = sqrt(dj);
			nwt_wave_sync();
			for (int i = lane; i < nr; i += 64) {
				if (i == j) H[i * mp + j] = ld;
				else if (i > j) H[i * mp + j] = H[i * mp + j] / ld;
			}
			nwt_wave_sync();
			for (int i = lane; i < nr; i += 64)
				if (i > j) { const double lij = H[i * mp + j]; for (int k2 = j + 1; k2 <= i; k2++) H[i * mp + k2] -= lij * H[k2 * mp + j]; }
			nwt_wave_sync();
		}
		ok = !bad;
	}
	if (ok) {   // H0 singular on null(A): the regularised inverse would scale those directions by 1e12 -- the host refuses, so do we
		double lo = 1e300, hi = 0.0;
		for (int a = 0; a < nr; a++) { const double dd = H[a * mp + a] * H[a * mp + a]; lo = fmin(lo, dd); hi = fmax(hi, dd); }
		if (lo < 1e-9 * hi) ok = false;
	}
	if (!ok) { if (lane == 0 && atomicCAS(&A.err[0], 0, 3) == 0) { A.err[1] = b; A.err[2] = q; } return; }
	// inverse column by column (lane = column): L y = e_c, L' x = y
	for (int c = lane; c < nr; c += 64) {
		for (int i = 0; i < nr; i++) {
			double s = (i == c) ? 1.0 : 0.0;
			for (int k2 = 0; k2 < i; k2++) s -= H[i * mp + k2] * G[k2 * mp + c];
			G[i * mp + c] = s / H[i * mp + i];
		}
		for (int i = nr - 1; i >= 0; i--) {
			double s = G[i * mp + c];
			for (int k2 = i + 1; k2 < nr; k2++) s -= H[k2 * mp + i] * G[k2 * mp + c];
			G[i * mp + c] = s / H[i * mp + i];
		}
	}
	nwt_wave_sync();
	// W0 [spad][nb]: the symmetrised inverse on the free coefficients, zeros elsewhere; entries at rounding level relative to the diagonal dropped
	for (int idx = lane; idx < A.spad * nb; idx += 64) {
		const int i = idx / nb, j = idx - i * nb;
		double v = 0.0;
		if (i < nb) {
			const int a = fx[i], a2 = fx[j];
			if (a >= 0 && a2 >= 0) {
				v = 0.5 * (G[a * mp + a2] + G[a2 * mp + a]);
				if (a != a2 && fabs(v) <= 1e-13 * sqrt(fabs(G[a * mp + a] * G[a2 * mp + a2]))) v = 0.0;
			}
		}
		out[idx] = v;
	}
}

hipError_t ntg_launch_grid_prec(const NtgDims &D, int batch, const NtgGridPrec &g, hipStream_t st)
{
	GridPrecArgs A;
	A.blk = g.blk; A.bps = g.bps; A.plan_off = g.plan_off; A.fidx = g.fidx; A.binfo = g.binfo; A.n0b = g.n0b; A.err = g.err;
	A.nblk = g.nblk; A.nb = g.nb; A.spad = g.spad; A.n0b_sz = g.n0b_sz;
	const size_t lds = 2 * (size_t)g.nrmax * (g.nrmax + 1) * 8;
	if (lds > 160 * 1024) return hipErrorInvalidValue;
	if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)grid_prec_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipLaunchKernelGGL(grid_prec_kernel, dim3(batch * g.nblk), dim3(64), lds, st, D, batch, A);
	return hipGetLastError();
}


// ---- structured Newton mode (hessian = 2 / 3) on per-problem grids: the cost model of every grid ----
// What build_newton_tables() (plan.cpp) does once for the plan's grid, per problem: K0 = sum_i 2 w_i sum_{cost variables (o, r)} m_i m_i'
// on the compact lower band of every coupling group (free coefficients interleaved by output, p = (cl - clo) go + ov; two-sided plans:
// the rows below the separator in the reversed array), and the band Cholesky factor (diagonal inverted: the layout nwt_solve_wave reads)
// of every free output's block.  One workgroup per problem; an entry sums over the breakpoints in whose block both coefficients lie
// (nwt_lo / nwt_hi), reading the problem's channel rows.  err[0] = 3: a free output's cost model is not positive definite.
__global__ void __launch_bounds__(128)
grid_nwt_kernel(NtgDims D, int batch, NtgGridNwt A)
{
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	const int b = blockIdx.x, tid = threadIdx.x, P = D.P, k = D.cls_k[0];
	if (b >= batch) return;
	const double *rv = A.rowv + (size_t)b * D.row_total, *bps = A.bps + (size_t)b * P;
	double *k0 = A.k0 + (size_t)b * A.k0_sz, *lf = A.lf + (size_t)b * A.lf_sz;
	const int ngp = D.nwt_ngrp, ng = D.nwt_ng, hb = D.nwt_hb, ld = hb + 1, go = D.nwt_go, clo = D.nwt_clo;
	const bool tw = D.nwt_tw != 0;
	const int brows = 16 * D.nwt_jb + 48, ngt = tw ? ng - 16 * D.nwt_jb : ng;
	auto wgt = [&](int i) { double w = 0.0; if (i > 0) w += (bps[i] - bps[i - 1]) / 2; if (i < P - 1) w += (bps[i + 1] - bps[i]) / 2; return w; };
	// sum over the breakpoints and cost variables of output o for the local coefficients cl1 >= cl2
	auto entry = [&](int o, int cl1, int cl2) {
		const int i0 = max((int)A.lo[cl1], (int)A.lo[cl2]), i1 = min((int)A.hi[cl1], (int)A.hi[cl2]);
		double v = 0.0;
		for (int i = i0; i < i1; i++) {
			const int q1 = cl1 - A.plan_off[i], q2 = cl2 - A.plan_off[i];
			if (q1 < 0 || q1 >= k || q2 < 0 || q2 >= k) continue;
			const double w = wgt(i);
			for (int r = 0; r < D.d[o] && r < NTG_MAX_ORDER; r++) {
				const int ch = D.ch_row0[r];
				if (ch < 0) continue;
				const double pr = rv[ch + q1 * P + i] * rv[ch + q2 * P + i];
				const int fl = D.iz[o] + r;
				if (D.nucf && ((D.tcost_mask >> fl) & 1ull)) v += 2.0 * w * pr;
				if (D.nicf && i == 0 && ((D.icost_mask >> fl) & 1ull)) v += 2.0 * pr;
				if (D.nfcf && i == P - 1 && ((D.fcost_mask >> fl) & 1ull)) v += 2.0 * pr;
			}
		}
		return v;
	};
	for (int e = tid; e < A.k0_sz; e += 128) k0[e] = 0.0;
	__syncthreads();
	for (int idx = tid; idx < ngp * ng * ld; idx += 128) {
		const int g = idx / (ng * ld), rem = idx - g * ng * ld, p1 = rem / ld, e = rem - p1 * ld, p2 = p1 - hb + e;
		if (p2 < 0) continue;
		const int ov1 = p1 % go, ov2 = p2 % go;
		if (ov1 != ov2) continue;
		const double v = entry(g * go + ov1, clo + p1 / go, clo + p2 / go);
		if (tw && p1 >= ngt) k0[(size_t)ngp * ng * ld + ((size_t)g * brows + (ng - 1 - p2)) * ld + e] = v;   // entry (i, j) -> row n - 1 - j of the reversed array, same band offset
		else k0[idx] = v;
	}
	// free outputs: band of half width k - 1 in LDS, factored by one thread each (a few thousand operations), stored with the diagonal inverted
	const int nfo = D.nwt_nfo, ngf = D.nwt_ngf, hbf = D.nwt_hbf, ldf = hbf + 1;
	double *kf = (double *)smem_raw;
	for (int idx = tid; idx < nfo * ngf * ldf; idx += 128) {
		const int f = idx / (ngf * ldf), rem = idx - f * ngf * ldf, p1 = rem / ldf, e = rem - p1 * ldf, p2 = p1 - hbf + e;
		kf[idx] = p2 >= 0 ? entry(ngp * go + f, clo + p1, clo + p2) : 0.0;
	}
	__syncthreads();
	if (tid < nfo) {
		double *a = kf + (size_t)tid * ngf * ldf;
		auto L = [&](int i, int j) -> double & { return a[(size_t)i * ldf + (j - i + hbf)]; };
		bool ok = true;
		for (int j = 0; j < ngf && ok; j++) {
			double dd = L(j, j);
			for (int t = max(0, j - hbf); t < j; t++) dd -= L(j, t) * L(j, t);
			if (!(dd > 0.0)) { ok = false; break; }
			dd = sqrt(dd); L(j, j) = dd;
			for (int i = j + 1; i <= min(ngf - 1, j + hbf); i++) {
				double sv = L(i, j);
				for (int t = max(0, i - hbf); t < j; t++) sv -= L(i, t) * L(j, t);
				L(i, j) = sv / dd;
			}
		}
		if (!ok) { if (atomicCAS(&A.err[0], 0, 3) == 0) A.err[1] = b; }
		for (int i = 0; i < ngf; i++) for (int e = 0; e <= hbf; e++) { const int j = i - hbf + e; lf[((size_t)tid * ngf + i) * ldf + e] = j < 0 ? 0.0 : (j == i ? 1.0 / L(i, i) : L(i, j)); }
	}
}

hipError_t ntg_launch_grid_nwt(const NtgDims &D, int batch, const NtgGridNwt &g, hipStream_t st)
{
	const size_t lds = (size_t)std::max(1, D.nwt_nfo * D.nwt_ngf * (D.nwt_hbf + 1)) * 8;
	if (lds > 64 * 1024) return hipErrorInvalidValue;
	hipLaunchKernelGGL(grid_nwt_kernel, dim3(batch), dim3(128), lds, st, D, batch, g);
	return hipGetLastError();
}
