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

// ------------------------------------------------------------------------------------------------------------------
// The same evaluation with one LANE PER (KNOT INTERVAL, GROUP OF OUTPUTS).  The kernel above has one lane per breakpoint,
// then one per coefficient, and pays for the hand-over in LDS traffic: ~95 KB through LDS per evaluation, the pipe it is
// bound by (128 B per clock and CU).  The breakpoints of an interval share their block of K coefficients
// (colloc.c:104-111), so a lane that owns an interval
//   reads its K coefficients of its outputs ONCE (the other kernel: once per breakpoint),
//   evaluates the flag and the cost functor at its <= 6 breakpoints,
//   and accumulates the interval's contribution to the gradient of those K coefficients in registers
//   (the other kernel: weighted gradients to LDS, then a gather per coefficient).
// With m = K/2 an interior coefficient belongs to exactly two neighbouring intervals: the two partial sums meet through one
// lane shuffle.  OPL outputs per lane: all of them in general (the functor sees the whole flag); families whose running
// cost is a sum of identical terms per output (Family::PER_OUTPUT_COST, kincar: examples/kincar.c:105-117) may split the
// outputs over lanes -- less state per lane, more waves per SIMD.  A problem takes LP = nint * NOUT / OPL lanes, a
// wavefront floor(64 / LP) problems at a time; wavefronts are independent (no workgroup barrier in the loop), share the
// read-only tables in LDS, and move the coefficient vectors and gradients of their problems through a private LDS buffer
// so that every access to HBM is a full 512-byte wavefront transaction.  ~36 KB through LDS per evaluation.
// Summation order of the gradient: per interval over its breakpoints, then the two intervals -- the reference integrates
// column by column (cost.c:117-134, integrator.c:44-48); results agree to rounding (tests: 1e-12 relative).
// ------------------------------------------------------------------------------------------------------------------
struct IntervalEvalDims {
	int P, nC, nco, nint, s;      // breakpoints, coefficients, coefficients per output, knot intervals, coefficients a block advances per interval
	int chrow[5];                 // channel offsets into rowv per derivative (only the CHM ones are used)
	unsigned short igb[66];       // first breakpoint of every interval, igb[nint] = P
};

// NINT: the number of knot intervals as a compile-time constant (every LDS offset of the loop then folds into the instruction).
// WANT_G: the instance stores gradients (modes 1, 2); a values-only instance has no store in its loop.
// PPG: per-problem grids (ntg_plan_set_grids): the interval tables are wave private and restaged for every problem from the problem's own
// channel rows and breakpoints (T.rowv + b pp_rowv, T.bps + b pp_bps); one problem per wave at a time (the host checks LP > 32)
template <int FAM, int NOUT, int OPL, int K, int CHM, int NT, int MINW, int NINT, bool WANT_G, bool PPG = false>
__global__ void __launch_bounds__(NT, MINW)
eval_interval_kernel(IntervalEvalDims D, NtgTables T, int batch, int mode, const double *__restrict__ x, double *__restrict__ f,
                     double *__restrict__ g)
{
	using Fam = Family<FAM>;
	constexpr int DM = Fam::DM, NCH = chm_count(CHM), NZL = OPL * DM, NW = NT / 64, SMAX = 6, S = K / 2, NG = NOUT / OPL, XE = 6;   // XE: the host checked PW nC <= 64 XE
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	const int P = D.P, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	constexpr int nint = NINT, nco = S * NINT + S, nC = NOUT * nco;   // (the host checked them against the plan)
	constexpr int LP = nint * NG, PW = 64 / LP;                     // lanes per problem, problems a wavefront takes at a time
	// LDS: tables [NCH][SMAX][K][nint] basis values, [SMAX][nint] trapezoid node weights and interval lengths; per wave: coefficient /
	// gradient staging [PW][nC] and one partial cost per lane
	constexpr int TABD = NCH * SMAX * K * nint + 2 * SMAX * nint;   // doubles of one set of interval tables
	double *s_bt = (double *)smem_raw + (PPG ? (size_t)wave * TABD : 0);
	double *s_wt = s_bt + NCH * SMAX * K * nint;
	double *s_dt = s_wt + SMAX * nint;
	double *s_xs = (double *)smem_raw + (PPG ? (size_t)NW * TABD : TABD) + (size_t)wave * (PW * nC + 64);
	double *s_fs = s_xs + PW * nC;
	// tables of grid `gb` into (s_bt, s_wt, s_dt): by the whole workgroup (shared grid) or by one wave (its problem's grid)
	auto stage = [&](int gb, int t0, int tstep) {
		const double *rowv = T.rowv + (size_t)gb * T.pp_rowv, *bps = T.bps + (size_t)gb * T.pp_bps;
		for (int e = t0; e < NCH * SMAX * K * nint; e += tstep) {
			const int t = e % nint, q = (e / nint) % K, s2 = (e / (nint * K)) % SMAX, ch = e / (nint * K * SMAX);
			int r = 0, seen = -1;
			for (int rr = 0; rr < DM; rr++) if ((CHM >> rr) & 1) { seen++; if (seen == ch) r = rr; }
			const int i = D.igb[t] + s2;
			s_bt[e] = i < D.igb[t + 1] ? rowv[D.chrow[r] + q * P + i] : 0.0;
		}
		for (int e = t0; e < SMAX * nint; e += tstep) {
			const int t = e % nint, s2 = e / nint, i = D.igb[t] + s2;
			double w = 0.0, dt = 0.0;
			if (i < D.igb[t + 1]) {
				if (i > 0) w += (bps[i] - bps[i - 1]) / 2;
				if (i < P - 1) { w += (bps[i + 1] - bps[i]) / 2; dt = bps[i + 1] - bps[i]; }
			}
			s_wt[e] = w; s_dt[e] = dt;
		}
	};
	if (!PPG) stage(0, tid, NT);
	__syncthreads();
	const int pl = lane / LP, rl = lane - pl * LP, t = rl / NG, og = rl - t * NG, o0 = og * OPL;
	const bool lane_on = pl < PW;
	const int cnt = lane_on ? D.igb[t + 1] - D.igb[t] : 0, i0 = lane_on ? D.igb[t] : 0;
	const int nwaves = gridDim.x * NW, wid = blockIdx.x * NW + wave;
	// Software pipeline over this wave's problems: while a group is evaluated, the coefficient vectors of the next one are in flight
	// (registers).  Every load and store of the loop is UNCONDITIONAL and sits at a fixed place of the loop body -- [evaluate]
	// [gradients -> staging -> HBM] [wait for the prefetched coefficients -> staging] [prefetch the group after] -- so that the wait can
	// be a counted s_waitcnt vmcnt(6) that does not also wait for the six younger gradient stores (gfx9 counts loads and stores in
	// one in-order counter; behind a branch the compiler has to fall back to vmcnt(0), which stalls every iteration for the stores).
	// The 6 x 64 slots cover the PW nC staged doubles; a slot past them repeats the slot 64 before it, a problem past the end of the
	// batch repeats the last problem (identical values land on identical addresses).
	int sidx[XE], spl[XE], sof[XE];
#pragma unroll
	for (int e = 0; e < XE; e++) {
		int i = lane + 64 * e;
		while (i >= PW * nC) i -= 64;   // (4 outputs: 252 staged doubles, the last two slots repeat earlier ones)
		sidx[e] = i; spl[e] = i / nC; sof[e] = i - spl[e] * nC;
	}
	double xn[XE];
	if (wid * PW >= batch) return;
#pragma unroll
	for (int e = 0; e < XE; e++) s_xs[sidx[e]] = x[(size_t)min(wid * PW + spl[e], batch - 1) * nC + sof[e]];
#pragma unroll
	for (int e = 0; e < XE; e++) xn[e] = x[(size_t)min(wid * PW + nwaves * PW + spl[e], batch - 1) * nC + sof[e]];
	for (int b0 = wid * PW; b0 < batch; b0 += nwaves * PW) {
		nwt_wave_sync();
		if (PPG) { stage(b0, lane, 64); nwt_wave_sync(); }   // this problem's grid (PW == 1)
		const bool on = lane_on && b0 + pl < batch;
		// the interval's K coefficients of this lane's outputs
		double xb[OPL][K];
#pragma unroll
		for (int o = 0; o < OPL; o++)
#pragma unroll
			for (int q = 0; q < K; q++) xb[o][q] = lane_on ? s_xs[pl * nC + (o0 + o) * nco + S * t + q] : 0.0;
		double pg[OPL][K];   // the interval's share of the gradient of those coefficients
#pragma unroll
		for (int o = 0; o < OPL; o++)
#pragma unroll
			for (int q = 0; q < K; q++) pg[o][q] = 0.0;
		// One breakpoint slot: basis values, flag, cost functor, the slot's share of the gradient; returns the cost value.
		// (A breakpoint on a knot may fall to either side of it by rounding, ntg.c:385-388: intervals hold 4, 5 or 6 breakpoints.  Slots an
		// interval does not fill carry zero basis values, weights and lengths in the tables and are evaluated like the others: no
		// divergence inside the wave.)  The slot loop is NOT unrolled: unrolled, the basis loads of all six slots are hoisted and the
		// kernel spills at 128 registers (measured: 1.7 ms instead of 0.4 ms per 2^18 evaluations).
		auto slot = [&](int s2) -> double {
			double bb[NCH][K];
#pragma unroll
			for (int ch = 0; ch < NCH; ch++)
#pragma unroll
				for (int q = 0; q < K; q++) bb[ch][q] = s_bt[((ch * SMAX + s2) * K + q) * nint + t];
			double z[NZL], df[NZL], fval = 0.0;
#pragma unroll
			for (int o = 0; o < OPL; o++)
#pragma unroll
				for (int r = 0; r < DM; r++) {
					double acc = 0.0;
					if ((CHM >> r) & 1) {
#pragma unroll
						for (int q = 0; q < K; q++) acc += bb[chm_rank(CHM, r)][q] * xb[o][q];
					}
					z[DM * o + r] = acc;
				}
			Fam::ucf(OPL, i0 + s2, z, fval, df);   // OPL < NOUT: the cost is a sum over the outputs (PER_OUTPUT_COST), this is the lane's share
			const double w = s_wt[s2 * nint + t];
#pragma unroll
			for (int o = 0; o < OPL; o++)
#pragma unroll
				for (int r = 0; r < DM; r++) {
					if ((CHM >> r) & 1) {
						const double wd = w * df[DM * o + r];
#pragma unroll
						for (int q = 0; q < K; q++) pg[o][q] += wd * bb[chm_rank(CHM, r)][q];
					}
				}
			return fval;
		};
		// trapezoid rule (integrator.c:21-24): the term of a breakpoint needs the next cost value; across the interval boundary it
		// comes from the lane of the next interval (the end point's length is 0, and so is the length of an empty slot)
		double Fp = 0.0;
		{
			double fprev = slot(0), dtprev = s_dt[t];
			double fnext = __shfl_down(fprev, NG, 64);
			if (t == nint - 1) fnext = 0.0;
#pragma unroll 1
			for (int s2 = 1; s2 < SMAX; s2++) {
				const double fval = slot(s2);
				Fp += dtprev * ((s2 < cnt ? fval : fnext) + fprev) / 2;
				fprev = fval; dtprev = s_dt[s2 * nint + t];
			}
			Fp += dtprev * (fnext + fprev) / 2;
		}
		if (PW > 1) s_fs[lane] = Fp;
		// gradient: coefficient S t + j (j < S) = this interval's pg[j] + the previous interval's pg[S + j]; the last interval also owns
		// the coefficients S t + S + j.  Through the staging buffer (x is no longer needed) so that the stores to HBM are contiguous.
		nwt_wave_sync();
#pragma unroll
		for (int o = 0; o < OPL; o++)
#pragma unroll
			for (int j = 0; j < S; j++) {
				double up = __shfl_up(pg[o][S + j], NG, 64);
				if (t == 0) up = 0.0;
				if (lane_on) {
					s_xs[pl * nC + (o0 + o) * nco + S * t + j] = pg[o][j] + up;
					if (t == nint - 1) s_xs[pl * nC + (o0 + o) * nco + S * t + S + j] = pg[o][S + j];
				}
			}
		nwt_wave_sync();
		if (PW == 1) {
			const double F = wave_sum(lane_on ? Fp : 0.0);
			if (f && mode != 1 && lane == 0 && b0 < batch) f[b0] = F;
		} else if (f && mode != 1 && on && rl == 0) {
			double F = 0.0;
			for (int u = 0; u < LP; u++) F += s_fs[pl * LP + u];
			f[b0 + pl] = F;
		}
		if (WANT_G) {
#pragma unroll
			for (int e = 0; e < XE; e++) g[(size_t)min(b0 + spl[e], batch - 1) * nC + sof[e]] = s_xs[sidx[e]];
		}
		nwt_wave_sync();   // the staging buffer is free again
#pragma unroll
		for (int e = 0; e < XE; e++) s_xs[sidx[e]] = xn[e];
#pragma unroll
		for (int e = 0; e < XE; e++) xn[e] = x[(size_t)min(b0 + 2 * nwaves * PW + spl[e], batch - 1) * nC + sof[e]];
	}
}

// true when the plan fits the interval kernel: the lean kernel's shape, K even with half of it advancing per interval
// (mult = K/2: every interior coefficient lies in exactly two neighbouring intervals), at most 6 breakpoints per interval
static inline bool eval_interval_match(const NtgDims &D, int chm, int dm, int K, int opl, IntervalEvalDims *F)
{
	if (!ntg_chm_match(D, chm, dm) || D.ncnln || !D.uniform || D.nI) return false;
	if (D.order[0] != K || (K & 1) || D.mult[0] != K / 2 || D.ig_n < 1 || D.nout % opl) return false;
	if (D.ncoef[0] != (K / 2) * D.ig_n + K / 2) return false;
	const int LP = D.ig_n * (D.nout / opl);
	if (LP > 64 || (64 / LP) * D.nC > 6 * 64) return false;
	F->P = D.P; F->nC = D.nC; F->nco = D.ncoef[0]; F->nint = D.ig_n; F->s = K / 2;
	for (int r = 0; r < 5; r++) F->chrow[r] = D.ch_row0[r];
	for (int t = 0; t <= D.ig_n; t++) F->igb[t] = D.igb[t];
	return true;
}

template <int FAM, int NOUT, int OPL, int K, int CHM, int MINW, int NINT>
static hipError_t launch_eval_interval(const NtgTables &T, const IntervalEvalDims &F, const EvalArgs &a)
{
	if (F.nint != NINT || F.nco != (K / 2) * NINT + K / 2 || F.nC != NOUT * F.nco) return hipErrorInvalidValue;
	static_assert(OPL == NOUT || Family<FAM>::PER_OUTPUT_COST, "outputs may be split over lanes only when the cost is a sum over the outputs");
#ifndef NTG_EVI_NT
#define NTG_EVI_NT 256
#endif
	constexpr int NT = NTG_EVI_NT, NW = NT / 64, NCH = chm_count(CHM);
	const int ncu = a.ncu > 0 ? a.ncu : 256, PW = 64 / (F.nint * (NOUT / OPL));
	const bool ppg = T.pp_rowv != 0;
	if (ppg && PW != 1) return hipErrorInvalidValue;   // (the caller checks: wave-private tables hold one grid)
	const size_t lds = ((size_t)(ppg ? NW : 1) * (NCH * 6 * K * F.nint + 2 * 6 * F.nint) + (size_t)NW * (PW * F.nC + 64)) * 8;
	const int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>(MINW * 4 / NW, (160 * 1024) / lds));   // MINW waves per SIMD = 4 MINW waves per CU
	const int need = (a.batch + NW * PW - 1) / (NW * PW);
	const int grid = std::max(1, std::min(need, ncu * wg_per_cu));
	if (ppg) {
		auto kfn = eval_interval_kernel<FAM, NOUT, OPL, K, CHM, NT, MINW, NINT, true, true>;
		if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		hipLaunchKernelGGL(kfn, dim3(grid), dim3(NT), lds, a.st, F, T, a.batch, a.mode, a.x, a.f, a.g);
		return hipGetLastError();
	}
	if (a.g && a.mode != 0) {
		auto kfn = eval_interval_kernel<FAM, NOUT, OPL, K, CHM, NT, MINW, NINT, true>;
		if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		hipLaunchKernelGGL(kfn, dim3(grid), dim3(NT), lds, a.st, F, T, a.batch, a.mode, a.x, a.f, a.g);
	} else {
		auto kfn = eval_interval_kernel<FAM, NOUT, OPL, K, CHM, NT, MINW, NINT, false>;
		if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		hipLaunchKernelGGL(kfn, dim3(grid), dim3(NT), lds, a.st, F, T, a.batch, a.mode, a.x, a.f, a.g);
	}
	return hipGetLastError();
}
