// eval_fast.hpp -- NPfunobj (ntg.c:274-335) for the shape the headline workload has, with nothing else in the kernel:
// one basis class, a running cost only, its active variables = the derivative channels CHM of every output
// (kincar: the second derivatives, examples/kincar.c:133-137), no constraints.  Same tables, same arithmetic and the
// same summation orders as eval_kernel's general path (solve_impl.hpp) -- Z = M C (colloc.c:318-367), the cost functor,
// the banded gradient (cost.c:117-134 regrouped by coefficient) and the trapezoid rule (integrator.c:21-24) -- but every
// decision the general kernel takes at run time (masks, channels, rows, widths) is a template constant here, the
// gradient goes straight from registers to HBM, and a problem costs two workgroup barriers.
#pragma once
#include "solve_impl.hpp"

struct FastEvalDims {
	int P, nC, nco, W;          // breakpoints, coefficients, coefficients per output, column width (entries)
	int chrow[5], chcol[5];     // channel offsets into rowv / colv per derivative (only the CHM ones are used)
	int row_total, col_total;   // doubles in rowv / colv
};

// OPG = outputs per lane in the gradient pass (compile time: NOUT / number of lane groups, see launch_eval_fast)
template <int FAM, int NOUT, int K, int CHM, int W, int OPG, int NT>
__global__ void __launch_bounds__(NT, 4)
eval_fast_kernel(FastEvalDims D, NtgTables T, int batch, int mode, const double *__restrict__ x, double *__restrict__ f,
                 double *__restrict__ g)
{
	using Fam = Family<FAM>;
	constexpr int DM = Fam::DM, NCH = chm_count(CHM), NZ = NOUT * DM, NW = NT / 64, XE = 4, WP = W + 2;
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	const int P = D.P, nC = D.nC, nco = D.nco, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	// LDS carve-up (all sizes multiples of 16 bytes)
	double *s_rowv = (double *)smem_raw;
	double *s_colv = s_rowv + ((D.row_total + 1) & ~1);   // 16-byte aligned: row_total is rounded up to even
	int *s_off = (int *)(s_colv + ((D.col_total + 1) & ~1));
	double *s_dt = (double *)(s_off + ((P + 3) & ~3));            // bps[i+1] - bps[i]
	double *s_wts = s_dt + ((P + 1) & ~1);                         // trapezoid weight of node i
	double *s_x = s_wts + ((P + 1) & ~1);
	double *s_f = s_x + ((nC + 1) & ~1);
	double *s_dfz = s_f + ((P + 1) & ~1);                          // [NOUT*NCH][P+1] + W zeros
	double *s_red = s_dfz + ((NOUT * NCH * (P + 1) + W + 1) & ~1); // [2][NW] wave partial sums of F, double buffered
	for (int i = tid; i < D.row_total; i += NT) s_rowv[i] = T.rowv[i];
	for (int i = tid; i < D.col_total; i += NT) s_colv[i] = T.colv[i];
	for (int i = tid; i < P; i += NT) {
		s_off[i] = T.off[i];
		double w = 0.0;
		if (i > 0) w += (T.bps[i] - T.bps[i - 1]) / 2;
		if (i < P - 1) w += (T.bps[i + 1] - T.bps[i]) / 2;
		s_wts[i] = w;
		s_dt[i] = i < P - 1 ? T.bps[i + 1] - T.bps[i] : 0.0;
	}
	for (int i = tid; i < NOUT * NCH * (P + 1) + W; i += NT) s_dfz[i] = 0.0;   // row tails and the overrun area stay 0

	// gradient pass: lane -> (column cl, group of outputs); see cost_phase2's shared gather
	constexpr int G = (NOUT + OPG - 1) / OPG;   // lane groups; the host checked G * nco <= NT
	const int grp = tid / nco, cl = tid - grp * nco, o0 = grp * OPG;
	const bool gat = tid < G * nco;

	double xn[XE];
	if ((int)blockIdx.x < batch) {
#pragma unroll
		for (int e = 0; e < XE; e++) { const int i = tid + e * NT; xn[e] = i < nC ? x[(size_t)blockIdx.x * nC + i] : 0.0; }
	}
	int bprev = -1, par = 0;
	for (int b = blockIdx.x; b < batch; b += gridDim.x) {
#pragma unroll
		for (int e = 0; e < XE; e++) { const int i = tid + e * NT; if (i < nC) s_x[i] = xn[e]; }
		{
			const int bn = b + gridDim.x;
			if (bn < batch) {
#pragma unroll
				for (int e = 0; e < XE; e++) { const int i = tid + e * NT; xn[e] = i < nC ? x[(size_t)bn * nC + i] : 0.0; }
			}
		}
		lds_sync();   // (1) x of this problem complete; last problem's gradient pass and wave sums done
		if (bprev >= 0 && tid == 0 && f && mode != 1) {
			double F = s_red[(par ^ 1) * NW];
#pragma unroll
			for (int w2 = 1; w2 < NW; w2++) F += s_red[(par ^ 1) * NW + w2];
			f[bprev] = F;
		}
		// ---- functor pass: one breakpoint per lane ----
		double Fpart = 0.0;
		for (int i = tid; i < P; i += NT) {
			double bb[NCH][K];
#pragma unroll
			for (int r = 0; r < DM; r++) {
				if (!((CHM >> r) & 1)) continue;
#pragma unroll
				for (int q = 0; q < K; q++) bb[chm_rank(CHM, r)][q] = s_rowv[D.chrow[r] + q * P + i];
			}
			const int ofs = s_off[i];
			double z[NZ], df[NZ], fv = 0.0;
#pragma unroll
			for (int o = 0; o < NOUT; o++) {
				const double *cx = s_x + o * nco + ofs;
				double xv[K];
#pragma unroll
				for (int q = 0; q < K; q++) xv[q] = cx[q];
#pragma unroll
				for (int r = 0; r < DM; r++) {
					double acc = 0.0;
					if ((CHM >> r) & 1) {
#pragma unroll
						for (int q = 0; q < K; q++) acc += bb[chm_rank(CHM, r)][q] * xv[q];
					}
					z[DM * o + r] = acc;
				}
			}
			Fam::ucf(NOUT, i, z, fv, df);
			s_f[i] = fv;
			const double w = s_wts[i];
#pragma unroll
			for (int o = 0; o < NOUT; o++)
#pragma unroll
				for (int r = 0; r < DM; r++) { if ((CHM >> r) & 1) s_dfz[(o * NCH + chm_rank(CHM, r)) * (P + 1) + i] = df[DM * o + r] * w; }
		}
		lds_sync();   // (2) fvals and weighted gradient rows complete
		// trapezoid terms of the running cost (integrator.c:21-24), one interval per lane
		for (int i = tid; i < P - 1; i += NT) Fpart += s_dt[i] * (s_f[i + 1] + s_f[i]) / 2;
		Fpart = wave_sum(Fpart);
		if (lane == 0) s_red[par * NW + wave] = Fpart;
		// ---- gradient pass ----
		if (gat && g && mode != 0) {
			double a[OPG];
#pragma unroll
			for (int j = 0; j < OPG; j++) a[j] = 0.0;
#pragma unroll
			for (int r = 0; r < DM; r++) {
				if (!((CHM >> r) & 1)) continue;
				// the column's W basis values and its first breakpoint: WP/2 aligned 16-byte reads, 16 lanes = 16 distinct bank groups
				double vv[W];
				const double2 *cv = (const double2 *)(s_colv + D.chcol[r] + cl * WP);
#pragma unroll
				for (int s2 = 0; s2 < W / 2; s2++) { const double2 t2 = cv[s2]; vv[2 * s2] = t2.x; vv[2 * s2 + 1] = t2.y; }
				const int i0 = (int)cv[W / 2].x;
#pragma unroll
				for (int j = 0; j < OPG; j++) {
					const int o = o0 + j;
					if (o >= NOUT) break;
					const double *wdf = s_dfz + (o * NCH + chm_rank(CHM, r)) * (P + 1) + i0;
					double ww[W];
#pragma unroll
					for (int s2 = 0; s2 < W; s2++) ww[s2] = wdf[s2];
#pragma unroll
					for (int s2 = 0; s2 < W; s2++) a[j] += vv[s2] * ww[s2];
				}
			}
#pragma unroll
			for (int j = 0; j < OPG; j++) {
				const int o = o0 + j;
				if (o >= NOUT) break;
				g[(size_t)b * nC + o * nco + cl] = a[j];
			}
		}
		bprev = b; par ^= 1;
	}
	lds_sync();
	if (bprev >= 0 && tid == 0 && f && mode != 1) {
		double F = s_red[(par ^ 1) * NW];
#pragma unroll
		for (int w2 = 1; w2 < NW; w2++) F += s_red[(par ^ 1) * NW + w2];
		f[bprev] = F;
	}
}

static inline size_t eval_fast_lds(const FastEvalDims &D, int nout, int nch, int nt)
{
	size_t n = 0;
	n += (size_t)((D.row_total + 1) & ~1) * 8;
	n += (size_t)((D.col_total + 1) & ~1) * 8;
	n += (size_t)((D.P + 3) & ~3) * 4;
	n += 2 * (size_t)((D.P + 1) & ~1) * 8;
	n += (size_t)((D.nC + 1) & ~1) * 8;
	n += (size_t)((D.P + 1) & ~1) * 8;
	n += (size_t)((nout * nch * (D.P + 1) + D.W + 1) & ~1) * 8;
	n += (size_t)2 * (nt / 64) * 8;
	return (n + 15) & ~(size_t)15;
}

// true when (D, a) fit the fast kernel; fills the small dimension block
static inline bool eval_fast_match(const NtgDims &D, int chm, int dm, int nt, FastEvalDims *F)
{
	if (!ntg_chm_match(D, chm, dm) || D.ncnln || !D.uniform || D.nI) return false;
	const int W = D.cls_W[0];
	if (W != 8 && W != 12 && W != 16) return false;
	if (D.ncoef[0] > nt || D.nC > 4 * nt || D.P > 4 * nt) return false;
	F->P = D.P; F->nC = D.nC; F->nco = D.ncoef[0]; F->W = W; F->row_total = D.row_total; F->col_total = D.colv_total;
	return true;
}

template <int FAM, int NOUT, int K, int CHM, int NT>
static hipError_t launch_eval_fast(const NtgDims &D, const NtgTables &T, FastEvalDims F, const EvalArgs &a)
{
	const int ncu = a.ncu > 0 ? a.ncu : 256;
	for (int r = 0; r < 5; r++) { F.chrow[r] = D.ch_row0[r]; F.chcol[r] = D.ch_colv0[r]; }
	const size_t lds = eval_fast_lds(F, NOUT, chm_count(CHM), NT);
	const int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>(std::min(8, 32 / (NT / 64)), (160 * 1024) / lds));
	const int grid = std::min(a.batch, ncu * wg_per_cu);
	// lane groups of the gradient pass: as many as fit the workgroup (each lane then owns fewer outputs), else one
	constexpr int GMAX = NOUT % 2 == 0 ? 2 : 1, OPG2 = NOUT / GMAX;
	const bool two = GMAX == 2 && 2 * F.nco <= NT;
#define NTG_FAST(WV, OPGV)                                                                                            \
	{                                                                                                                   \
		auto kfn = eval_fast_kernel<FAM, NOUT, K, CHM, WV, OPGV, NT>;                                                   \
		if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
		hipLaunchKernelGGL(kfn, dim3(grid), dim3(NT), lds, a.st, F, T, a.batch, a.mode, a.x, a.f, a.g);               \
	}
	if (two) { if (F.W == 8) NTG_FAST(8, OPG2) else if (F.W == 12) NTG_FAST(12, OPG2) else NTG_FAST(16, OPG2) }
	else { if (F.W == 8) NTG_FAST(8, NOUT) else if (F.W == 12) NTG_FAST(12, NOUT) else NTG_FAST(16, NOUT) }
#undef NTG_FAST
	return hipGetLastError();
}
