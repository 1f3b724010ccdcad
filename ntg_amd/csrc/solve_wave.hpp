// solve_wave.hpp -- the npsol_ call (ntg.c:250) for the headline problem class with ONE WAVEFRONT PER PROBLEM.
//
// Class (same as eval_interval_kernel, eval_fast.hpp): one basis class, order K with multiplicity K/2, a running cost only whose
// active variables are the derivative channels CHM of every output (kincar: z'', examples/kincar.c:105-117,133-137), linear
// equality rows kept by projection (constraints.c:198-261), no nonlinear rows.  The iteration is DESIGN.md section 4a word for
// word (oracle/sqp.c, sqp_kernel): feasibility step, projected gradient, strong-Wolfe line search, inverse BFGS kept as the
// chain of search directions (section 4a.4 "direction form").
//
// Why a second kernel: sqp_kernel spreads one problem over a workgroup, so every dot product of the iteration is a DPP reduction,
// an LDS hand-over and an s_barrier, and a problem holds 40 KB of LDS for vectors that are read across lanes.  Here a lane owns
// S = K/2 consecutive coefficients of OPL outputs (lane = (output group, knot interval), nint + 1 lanes per group: 63 of 64 lanes for
// 6 outputs on 20 intervals) and keeps its share of x, g_p, g_p+, d, t in registers:
//   * Z = M C and the gradient need the neighbouring lane's coefficients only (an interval's K coefficients belong to lanes t and
//     t + 1): one DPP wave shift each way, no LDS;
//   * every reduction is a wave reduction (DPP butterfly + v_readlane): no barrier anywhere in the iteration;
//   * the quasi-Newton chain d_0 .. d_k -- the bytes that bound sqp_kernel (16 GB per headline launch from HBM) -- stays ON CHIP:
//     the kernel is compiled for one wave per SIMD and keeps NREG vectors in the accumulator half of the unified register file
//     (v_accvgpr_write / _read, 252 of the 256 AGPRs), NLDS more in LDS, and only the tail of a long chain in HBM;
//   * waves are persistent and take problems from an atomic queue (to-convergence batches have ragged iteration counts).
// The read-only tables (per-interval basis values, trapezoid weights, the projector Q) are shared by the waves of a workgroup.
#pragma once
#include "solve_impl.hpp"

namespace ntgw {

template <int J, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
	if constexpr (J < N) { f(std::integral_constant<int, J>{}); static_for<J + 1, N>(f); }
}

// value of the previous / next lane of the wavefront (DPP wave_shr:1 / wave_shl:1; lane 0 / lane 63 receive 0)
__device__ __forceinline__ double from_prev(double v)
{
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, false);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, false);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_next(double v)
{
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, false);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, false);
	return __hiloint2double(hi, lo);
}

// The register allocator hands out accumulator registers from a0 upwards when the 256 architectural VGPRs run short (AV-class values:
// load results, copies).  It cannot be told to keep out, so the chain starts at a[ABASE] -- a template parameter of the kernel, visible in
// its mangled name: a0 .. a[ABASE - 1] are the compiler's, and ntg_amd/isa_audit.py rejects an object in which compiler-generated code
// touches anything from a[ABASE] up.  ntg_amd/build.py then RAISES the base of the class of instances that failed (NTGW_ABASE: 20 knot
// intervals on the plan's grid; NTGW_ABASE_ALT: the 16-interval and per-problem-grid instances) and compiles again -- fewer chain slots in
// registers, never a broken build; at 256 the instance keeps no slot in registers at all (NREG = 0: nothing hand-managed is left).
#ifndef NTGW_ABASE
#define NTGW_ABASE 16
#endif
#ifndef NTGW_ABASE_ALT
#define NTGW_ABASE_ALT 16
#endif
// accumulator registers a[ABASE + 2 IDX], a[ABASE + 2 IDX + 1] as one double.  The kernel lists a0..a255 as clobbers once (which makes
// the kernel descriptor allocate them all).
template <int ABASE, int IDX>
__device__ __forceinline__ void areg_write(double v)
{
	asm volatile("v_accvgpr_write_b32 a[%c2], %0\n\tv_accvgpr_write_b32 a[%c3], %1" ::"v"(__double2loint(v)), "v"(__double2hiint(v)), "i"(ABASE + 2 * IDX), "i"(ABASE + 2 * IDX + 1));
}
template <int ABASE, int IDX>
__device__ __forceinline__ double areg_read()
{
	int lo, hi;
	asm volatile("v_accvgpr_read_b32 %0, a[%c2]\n\tv_accvgpr_read_b32 %1, a[%c3]" : "=v"(lo), "=v"(hi) : "i"(ABASE + 2 * IDX), "i"(ABASE + 2 * IDX + 1));
	return __hiloint2double(hi, lo);
}
#define NTGW_C10(p) "a" #p "0", "a" #p "1", "a" #p "2", "a" #p "3", "a" #p "4", "a" #p "5", "a" #p "6", "a" #p "7", "a" #p "8", "a" #p "9"
#define NTGW_CLAIM_AGPRS()                                                                                                         \
	asm volatile("" ::: NTGW_C10(), NTGW_C10(1), NTGW_C10(2), NTGW_C10(3), NTGW_C10(4), NTGW_C10(5), NTGW_C10(6), NTGW_C10(7),     \
	             NTGW_C10(8), NTGW_C10(9), NTGW_C10(10), NTGW_C10(11), NTGW_C10(12), NTGW_C10(13), NTGW_C10(14), NTGW_C10(15),    \
	             NTGW_C10(16), NTGW_C10(17), NTGW_C10(18), NTGW_C10(19), NTGW_C10(20), NTGW_C10(21), NTGW_C10(22), NTGW_C10(23),  \
	             NTGW_C10(24), "a250", "a251", "a252", "a253", "a254", "a255")

__device__ __forceinline__ double bcast(double v, int lane)
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
	return __hiloint2double(hi, lo);
}
// sums of KV per-lane values over the wavefront, every lane receives all of them
// FEW: the interleaved rotation / permlane form for one to three values (solve_impl.hpp: wave_sums_few).  Measured: it pays on the instances
// with one wave per SIMD (headline -0.5 %) and costs the two-waves-per-SIMD instances dearly (preconditioned solves 0.151 -> 0.178 ms per
// 4096), so those keep the row_shr / row_bcast chains.
template <int KV, bool FEW = false>
__device__ __forceinline__ void wave_sums(double (&v)[KV], int lane)
{
	if constexpr (KV < 4) {
		if constexpr (FEW) wave_sums_few<KV>(v);
		else {
#pragma unroll
			for (int k = 0; k < KV; k++) v[k] = wave_sum(v[k]);
		}
	} else {
		double w[16];
#pragma unroll
		for (int k = 0; k < 16; k++) w[k] = k < KV ? v[k] : 0.0;
		const double t = wave_sum_many<KV>(w, lane);   // lane L: total of value L & 15
#pragma unroll
		for (int k = 0; k < KV; k++) v[k] = bcast(t, k);
	}
}

struct WaveArgs {
	int batch, cap;   // problems; slots of the direction chain a wave may hold (registers + LDS + HBM)
	const double *lower, *upper;
	double *xio, *objective;
	int *inform, *iters, *nfev;
	double *clambda, *hist;   // hist: HBM tier of the chain, [wave][slot - on-chip slots][EPL][64]
	unsigned int *counter;    // problem queue (zeroed before the launch)
	int hbm_slots;            // slots per wave in hist
};

// LDS (doubles) of one workgroup: tables, then per wave [stage | tmp | delta | kappa | links | line search | chain tier]
template <int NCH, int K, int NINT>
__host__ __device__ constexpr int wave_tab_doubles() { return NCH * 6 * K * (NINT + 1) + 2 * 6 * (NINT + 1); }
__host__ __device__ inline int wave_priv_doubles(int nC, int cap, int nlds, int epl)
{
	const int capp = (cap + 3) & ~3;
	return ((nC + 3) & ~3) + (4 * capp > 128 ? 4 * capp : 128) + 48 + nlds * epl * 64;   // (the feasibility / multiplier scratch shares the delta / kappa / link area)
}

// XLDS: the preconditioner blocks W_b (zero padded to 64 x 64) and the sparse operator of the linear rows (A as CSR and CSC, (A A')^-1 as
// CSR) are staged in the workgroup's LDS: the instance of short solves, where their L2 latency is a visible share of a problem
// PPG: per-problem grids (ntg_plan_set_grids) -- every wave keeps its OWN copy of the value tables (basis values, node weights, projector
// values) and restages it for each problem it takes; the index tables stay shared.  Without the preconditioner only (hessian != 1).
template <int FAM, int NOUT, int OPL, int K, int CHM, int NINT, int NWV, int MINW, int NREG, int NLDS, bool HESS, bool XLDS = false, bool PPG = false, int ABASE = NTGW_ABASE>
__global__ void __launch_bounds__(64 * NWV, MINW)
sqp_wave_kernel(NtgDims D, NtgTables T, SolveParams sp, WaveArgs A)
{
	using Fam = Family<FAM>;
	if constexpr (NREG > 0) NTGW_CLAIM_AGPRS();
	constexpr int DM = Fam::DM, NCH = chm_count(CHM), SMAX = 6, S = K / 2, NG = NOUT / OPL, EPL = OPL * S, NL = NINT + 1, LP = NL * NG;
	constexpr int nco = S * NINT + S, nC = NOUT * nco, NZL = OPL * DM;
	static_assert(LP <= 64 && (NREG == 0 || ABASE + NREG * 2 * EPL <= 256), "lanes / accumulator registers");
	static_assert(OPL == NOUT || Fam::PER_OUTPUT_COST, "outputs may be split over lanes only when the cost is a sum over the outputs");
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int P = D.P, m = D.mE, cap = A.cap, capp = (cap + 3) & ~3, qw = D.q_w;
	// ---- shared tables ----
	static_assert(!(PPG && XLDS), "per-problem grids: the linear operator and the preconditioner blocks stay in HBM (they are per problem)");
	const int tabw = NCH * SMAX * K * NL + 2 * SMAX * NL + D.q_nt * 6;   // doubles of one copy of the value tables (one per workgroup; PPG: one per wave)
	double *s_bt = (double *)smem_raw + (PPG ? wave * tabw : 0);   // [NCH][SMAX][K][NL] basis values of the CHM channels per interval slot; column NINT = 0
	double *s_wt = s_bt + NCH * SMAX * K * NL;              // [SMAX][NL] trapezoid node weights
	double *s_dt = s_wt + SMAX * NL;                        // [SMAX][NL] interval lengths
	double *s_qv = s_dt + SMAX * NL;                        // [q_nt][6] projector rows (ELL)
	int *s_qc = (int *)((double *)smem_raw + (PPG ? NWV : 1) * tabw);   // [q_nt][8]
	// XLDS extras: [nblk][64][64] preconditioner blocks; csr_ptr[m+1] csr_col[nnz] csc_ptr[nC+1] csc_row[nnz] sinv_ptr[m+1] sinv_col[snz] (ints),
	// csr_val[nnz] csc_val[nnz] sinv_val[snz] (doubles)
	double *s_w0 = (double *)(s_qc + D.q_nt * 8);
	const int w0n = (XLDS && HESS && sp.hessian == 1) ? T.n0b_nblk * 64 * 64 : 0, lnz = D.lin_nnz > 0 ? D.lin_nnz : 1, snz = D.sinv_nnz > 0 ? D.sinv_nnz : 1;
	double *s_lv = s_w0 + w0n;                                     // csr_val | csc_val | sinv_val
	int *s_li = (int *)(s_lv + (XLDS ? 2 * lnz + snz : 0));          // csr_ptr | csr_col | csc_ptr | csc_row | sinv_ptr | sinv_col
	const int lin_ints = XLDS ? ((2 * (m + 1) + (nC + 1) + 2 * lnz + snz + 3) & ~3) : 0;
	double *s_priv0 = (double *)(s_li + lin_ints);
	const double *l_csr_val = XLDS ? s_lv : T.csr_val, *l_csc_val = XLDS ? s_lv + lnz : T.csc_val, *l_sinv_val = XLDS ? s_lv + 2 * lnz : T.sinv_val;
	const int *l_csr_ptr = XLDS ? s_li : T.csr_ptr, *l_csr_col = XLDS ? s_li + (m + 1) : T.csr_col;
	const int *l_csc_ptr = XLDS ? s_li + (m + 1) + lnz : T.csc_ptr, *l_csc_row = XLDS ? s_li + (m + 1) + lnz + (nC + 1) : T.csc_row;
	const int *l_sinv_ptr = XLDS ? s_li + (m + 1) + 2 * lnz + (nC + 1) : T.sinv_ptr, *l_sinv_col = XLDS ? s_li + 2 * (m + 1) + 2 * lnz + (nC + 1) : T.sinv_col;
	if (XLDS) {
		for (int e = tid; e < w0n; e += 64 * NWV) {
			const int row = e & 63, k = (e >> 6) & 63, b2 = e >> 12;
			s_w0[e] = (k < T.n0b_sp && row < nco) ? T.n0b[((size_t)b2 * T.n0b_sp + k) * nco + row] : 0.0;
		}
		if (m > 0) {
			for (int e = tid; e < lnz; e += 64 * NWV) { s_lv[e] = T.csr_val[e]; s_lv[lnz + e] = T.csc_val[e]; s_li[(m + 1) + e] = T.csr_col[e]; s_li[(m + 1) + lnz + (nC + 1) + e] = T.csc_row[e]; }
			for (int e = tid; e < snz; e += 64 * NWV) { s_lv[2 * lnz + e] = T.sinv_val[e]; s_li[2 * (m + 1) + 2 * lnz + (nC + 1) + e] = T.sinv_col[e]; }
			for (int e = tid; e <= m; e += 64 * NWV) { s_li[e] = T.csr_ptr[e]; s_li[(m + 1) + 2 * lnz + (nC + 1) + e] = T.sinv_ptr[e]; }
			for (int e = tid; e <= nC; e += 64 * NWV) s_li[(m + 1) + lnz + e] = T.csc_ptr[e];
		}
	}
	// the value tables of one grid: by the whole workgroup once (shared grid), or by a wave for the problem it is about to solve (PPG)
	auto stage_values = [&](const double *rowv, const double *bps, const double *qval, int t0, int nth) {
		for (int e = t0; e < NCH * SMAX * K * NL; e += nth) {
			const int t = e % NL, q = (e / NL) % K, s2 = (e / (NL * K)) % SMAX, ch = e / (NL * K * SMAX);
			int r = 0, seen = -1;
			for (int rr = 0; rr < DM; rr++) if ((CHM >> rr) & 1) { seen++; if (seen == ch) r = rr; }
			const int i = t < NINT ? D.igb[t] + s2 : P;
			s_bt[e] = (t < NINT && i < D.igb[t + 1]) ? rowv[D.ch_row0[r] + q * P + i] : 0.0;
		}
		for (int e = t0; e < SMAX * NL; e += nth) {
			const int t = e % NL, s2 = e / NL, i = t < NINT ? D.igb[t] + s2 : P;
			double w = 0.0, dt = 0.0;
			if (t < NINT && i < D.igb[t + 1]) {   // trapezoid weight of node i: integrator.c:21-24 regrouped per node
				if (i > 0) w += (bps[i] - bps[i - 1]) / 2;
				if (i < P - 1) { w += (bps[i + 1] - bps[i]) / 2; dt = bps[i + 1] - bps[i]; }
			}
			s_wt[e] = w; s_dt[e] = dt;
		}
		for (int e = t0; e < D.q_nt * 6; e += nth) { const int r = e / 6, w2 = e - 6 * r; s_qv[e] = w2 < qw ? qval[r * qw + w2] : 0.0; }   // rows padded to 6 entries
	};
	if (!PPG) stage_values(T.rowv, T.bps, T.q_val, tid, 64 * NWV);
	for (int e = tid; e < D.q_nt * 8; e += 64 * NWV) { const int r = e / 8, w2 = e - 8 * r; s_qc[e] = w2 < qw ? T.q_col[r * qw + w2] : 0; }     // ... and to 8 indices
	__syncthreads();   // the only workgroup barrier: from here on the waves are independent
	// ---- this wave's private LDS ----
	double *s_st = s_priv0 + (size_t)wave * wave_priv_doubles(nC, cap, NLDS, EPL);   // [nC] staging in the natural layout (output-major)
	double *s_tmp = s_st + ((nC + 3) & ~3);    // [128] scratch of the feasibility step and of the final multipliers: used before / after the chain exists,
	double *s_dl = s_tmp;                      // [capp] d_j . v     ... so it shares the chain's scalar area
	double *s_kp = s_dl + capp;                // [capp] kappa_j
	double *s_lk = s_kp + capp;                // [capp][2] links (e_i, f_i)
	const int scal = 4 * capp > 128 ? 4 * capp : 128;
	double *s_hl = s_tmp + scal + 48;          // [NLDS][lane][EPL]
	// ---- lane roles ----
	const int og = lane / NL, t = lane - og * NL, o0 = og * OPL;
	const bool lane_on = lane < LP, has_int = lane_on && t < NINT;
	const int tt = has_int ? t : NINT;   // column of the interval tables (NINT: the all-zero column)
	const int cnt = has_int ? D.igb[t + 1] - D.igb[t] : 0, i0 = has_int ? D.igb[t] : 0;
	const int cbase = lane_on ? o0 * nco + S * t : 0;   // owned coefficient (o, q): cbase + o nco + q
	int qrow[EPL];                                      // row of the owned coefficient in the compact projector, or -1
#pragma unroll
	for (int o = 0; o < OPL; o++)
#pragma unroll
		for (int q = 0; q < S; q++) qrow[o * S + q] = (lane_on && m > 0) ? (int)T.q_idx[cbase + o * nco + q] : -1;
	constexpr int QW = 6;   // entries per projector row this kernel handles (wave_match checks q_w <= 6)
	// NtgDims::q_pin: Q is the identity on the pinned coefficients and zero elsewhere -- which of this lane's coefficients are pinned (the
	// rows of Q with a unit diagonal; per-problem grids share the pattern), one bit each
	unsigned pinmask = 0;
	if (D.q_pin && lane_on) {
#pragma unroll
		for (int e = 0; e < EPL; e++) pinmask |= (unsigned)T.q_pinned[cbase + (e / S) * nco + (e % S)] << e;
	}
	const int wgid = blockIdx.x * NWV + wave;
	double *hbm = A.hist + (size_t)wgid * A.hbm_slots * EPL * 64;

	const double *w0_base = T.n0b;   // preconditioner blocks (PPG: of the problem being solved)
	// ================= building blocks (all wave-uniform control flow) =================
	// NPfunobj (ntg.c:274-335) at xt: gradient into g, returns this lane's shares of the quadrature and of |g|^2
	auto evaluate = [&](const double (&xt)[EPL], double (&g)[EPL], double &Fq, double &g2) {
		double xb[OPL][K], pg[OPL][K];
#pragma unroll
		for (int o = 0; o < OPL; o++)
#pragma unroll
			for (int q = 0; q < S; q++) { xb[o][q] = xt[o * S + q]; xb[o][S + q] = from_next(xt[o * S + q]); pg[o][q] = 0.0; pg[o][S + q] = 0.0; }
		// One breakpoint slot of the lane's interval: basis values (and weight, interval length) from the shared tables, flag, cost
		// functor, the slot's share of the gradient.  The loads of slot s + 1 are issued before slot s is computed (two register
		// buffers, the loop fully unrolled): with one wave per SIMD nothing else hides the LDS latency.  Lanes without an interval read
		// the tables' all-zero column: no exec masking anywhere.
		auto slot_load = [&](int s2, double (&bb)[NCH][K], double &w, double &dt) __attribute__((always_inline)) {
#pragma unroll
			for (int ch = 0; ch < NCH; ch++)
#pragma unroll
				for (int q = 0; q < K; q++) bb[ch][q] = s_bt[((ch * SMAX + s2) * K + q) * NL + tt];
			w = s_wt[s2 * NL + tt]; dt = s_dt[s2 * NL + tt];
		};
		auto slot_compute = [&](int s2, const double (&bb)[NCH][K], double w) __attribute__((always_inline)) -> double {
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
			Fam::ucf(OPL, i0 + s2, z, fval, df);
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
		double Fp = 0.0;
		{
			double bbA[NCH][K], bbB[NCH][K], wA, wB, dtA, dtB;
			slot_load(0, bbA, wA, dtA);
			slot_load(1, bbB, wB, dtB);
			double fprev = slot_compute(0, bbA, wA), dtprev = dtA;
			double fnext = from_next(fprev);
			if (t >= NINT - 1) fnext = 0.0;
			static_for<1, SMAX>([&](auto Sc) __attribute__((always_inline)) {
				constexpr int s2 = decltype(Sc)::value;
				double fval;
				if constexpr (s2 & 1) {
					if constexpr (s2 + 1 < SMAX) slot_load(s2 + 1, bbA, wA, dtA);
					fval = slot_compute(s2, bbB, wB);
				} else {
					if constexpr (s2 + 1 < SMAX) slot_load(s2 + 1, bbB, wB, dtB);
					fval = slot_compute(s2, bbA, wA);
				}
				Fp += dtprev * ((s2 < cnt ? fval : fnext) + fprev) / 2;
				fprev = fval; dtprev = (s2 & 1) ? dtB : dtA;
			});
			Fp += dtprev * (fnext + fprev) / 2;
		}
		Fq = Fp; g2 = 0.0;
		// coefficient S t + q: this interval's share + the previous interval's (lane t - 1; the lane before t = 0 is an interval-less lane
		// of the previous group or nothing: zero either way)
#pragma unroll
		for (int o = 0; o < OPL; o++)
#pragma unroll
			for (int q = 0; q < S; q++) {
				const double gv = pg[o][q] + from_prev(pg[o][S + q]);
				g[o * S + q] = gv; g2 += gv * gv;
			}
	};
	auto stage_put = [&](const double (&v)[EPL]) {
		if (lane_on) {
#pragma unroll
			for (int o = 0; o < OPL; o++)
#pragma unroll
				for (int q = 0; q < S; q++) s_st[cbase + o * nco + q] = v[o * S + q];
		}
		nwt_wave_sync();
	};
	auto stage_get = [&](double (&v)[EPL]) {
		nwt_wave_sync();
#pragma unroll
		for (int o = 0; o < OPL; o++)
#pragma unroll
			for (int q = 0; q < S; q++) v[o * S + q] = lane_on ? s_st[cbase + o * nco + q] : 0.0;
	};
	// gp = g - Q g, Q = A'(AA')^-1 A as ELL over its non-zero rows (sqp_kernel: project).  The column indices of a lane's rows sit in
	// registers (rows of at most QW entries): the reads of g and of the values are then independent of each other -- one LDS round trip
	// per projection instead of a dependent chain of 2 q_w EPL of them (measured: 6.2 k -> cycles per projection on one wave per SIMD).
	auto project = [&](const double (&g)[EPL], double (&gp)[EPL]) {
		if (m == 0) {
#pragma unroll
			for (int e = 0; e < EPL; e++) gp[e] = g[e];
			return;
		}
		if (D.q_pin) {   // the rows pin whole coefficients: Q is the identity on them (NtgDims::q_pin) -- no staging, no LDS round trips
#pragma unroll
			for (int e = 0; e < EPL; e++) gp[e] = ((pinmask >> e) & 1u) ? 0.0 : g[e];
			return;
		}
		stage_put(g);
		// every load is unconditional (a lane without a row reads row 0: one broadcast address) and the result is selected afterwards.
		// Two LDS round trips: the rows' column indices (16-byte reads), then g at those columns and the rows' values.
		// (the row numbers pass through an opaque copy: their LDS addresses are then formed here, not once per kernel and kept in a dozen
		// registers across everything else)
		int col[EPL][8], qr[EPL];
#pragma unroll
		for (int e = 0; e < EPL; e++) { qr[e] = qrow[e]; asm volatile("" : "+v"(qr[e])); }
#pragma unroll
		for (int e = 0; e < EPL; e++) {
			const int4 *cp = (const int4 *)(s_qc + (qr[e] >= 0 ? qr[e] : 0) * 8);
			const int4 c0 = cp[0], c1 = cp[1];
			col[e][0] = c0.x; col[e][1] = c0.y; col[e][2] = c0.z; col[e][3] = c0.w; col[e][4] = c1.x; col[e][5] = c1.y; col[e][6] = 0; col[e][7] = 0;
		}
#pragma unroll
		for (int e = 0; e < EPL; e++) {
			const double2 *vp = (const double2 *)(s_qv + (qr[e] >= 0 ? qr[e] : 0) * 6);
			const double2 v0 = vp[0], v1 = vp[1], v2 = vp[2];
			const double s = ((v0.x * s_st[col[e][0]] + v0.y * s_st[col[e][1]]) + (v1.x * s_st[col[e][2]] + v1.y * s_st[col[e][3]])) +
			                 (v2.x * s_st[col[e][4]] + v2.y * s_st[col[e][5]]);
			gp[e] = g[e] - (qr[e] >= 0 ? s : 0.0);
		}
		nwt_wave_sync();
	};
	// out = W0 v: identity (NPSOL cold start) or the collocation preconditioner, block diagonal by output: the one GEMM-shaped piece,
	// Out(nco x NOUT) = W_b V on v_mfma_f64_16x16x4_f64 (apply_n0_block of sqp_kernel, for one wave)
	auto apply_w0 = [&](const double (&v)[EPL], double (&out)[EPL]) {
		if constexpr (!HESS) {
#pragma unroll
			for (int e = 0; e < EPL; e++) out[e] = v[e];
		} else {
			if (sp.hessian != 1) {
#pragma unroll
				for (int e = 0; e < EPL; e++) out[e] = v[e];
				return;
			}
			constexpr int TM = (nco + 15) / 16;
			stage_put(v);
			const int li = lane & 15, lk = lane >> 4, spad = T.n0b_sp;
			const int myblk = li < NOUT ? D.n0_blk[li] : -1;
			const double *bcol = s_st + (li < NOUT ? li : 0) * nco;
			ntg_d4 acc[TM];
#pragma unroll
			for (int tl = 0; tl < TM; tl++) acc[tl] = ntg_d4{0.0, 0.0, 0.0, 0.0};
			for (int b = 0; b < T.n0b_nblk; b++) {
				const double *wbb = w0_base + (size_t)b * spad * nco + li;
				const double *wlb = s_w0 + (size_t)b * 64 * 64 + li;   // staged copy: rows and columns zero padded to 64
				const bool mine = myblk == b;
				for (int k0 = 0; k0 < spad; k0 += 16) {
					double av[4][TM], bv[4];
#pragma unroll
					for (int u = 0; u < 4; u++) {
						const int k = k0 + 4 * u + lk;
						bv[u] = mine ? bcol[min(k, nco - 1)] : 0.0;   // k >= nco: W's row is zero padding
						const double *wk = wbb + (size_t)k * nco;
#pragma unroll
						for (int tl = 0; tl < TM; tl++) {
							if constexpr (XLDS) av[u][tl] = wlb[k * 64 + 16 * tl];
							else av[u][tl] = (16 * tl + li < nco) ? wk[16 * tl] : 0.0;
						}
					}
#pragma unroll
					for (int u = 0; u < 4; u++)
#pragma unroll
						for (int tl = 0; tl < TM; tl++) acc[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][tl], bv[u], acc[tl], 0, 0, 0);
				}
			}
			nwt_wave_sync();   // every read of the staged v is done
			if (li < NOUT) {
#pragma unroll
				for (int tl = 0; tl < TM; tl++)
#pragma unroll
					for (int r = 0; r < 4; r++) { const int row = tl * 16 + 4 * r + lk; if (row < nco) s_st[li * nco + row] = acc[tl][r]; }
			}
			stage_get(out);
			nwt_wave_sync();
		}
	};
	// ---- the direction chain: slot j lives in the accumulator registers (j < NREG), in LDS (next NLDS) or in HBM ----
	// (Tried: the register tier as a plain local array with static indices, left to the register allocator.  It first sank the NREG
	// conditional stores into one store through a pointer phi -- the whole tier went to scratch -- and, with that prevented, still spilled
	// 150-390 registers to scratch at 16-21 slots.  Hence the accumulator registers by hand, and an ISA audit in the build.)
	auto slot_store = [&](int j, const double (&v)[EPL]) {
		if (j < NREG) {
			static_for<0, NREG>([&](auto Jc) __attribute__((always_inline)) {
				constexpr int J = decltype(Jc)::value;
				if (j == J) {
					static_for<0, EPL>([&](auto Ec) __attribute__((always_inline)) { constexpr int E = decltype(Ec)::value; areg_write<ABASE, J * EPL + E>(v[E]); });
				}
			});
		} else if (j < NREG + NLDS) {
#pragma unroll
			for (int e = 0; e < EPL; e++) s_hl[((size_t)(j - NREG) * 64 + lane) * EPL + e] = v[e];
		} else {
#pragma unroll
			for (int e = 0; e < EPL; e++) hbm[((size_t)(j - NREG - NLDS) * EPL + e) * 64 + lane] = v[e];
		}
	};
	// tv += sum_j kappa_j d_j, kappa = (symmetric tridiagonal of the links) (d_j . v): DESIGN.md 4a.4, apply_dform of sqp_kernel
	// LDS tier: [slot][lane][EPL] -- a lane's EPL doubles are contiguous (16-byte reads, conflict free: lane stride 12 dwords)
	auto lds_get = [&](int j, double (&h)[EPL]) {
		const double *p = s_hl + ((size_t)(j - NREG) * 64 + lane) * EPL;
		if constexpr (EPL % 2 == 0) {
#pragma unroll
			for (int e = 0; e < EPL / 2; e++) { const double2 t2 = ((const double2 *)p)[e]; h[2 * e] = t2.x; h[2 * e + 1] = t2.y; }
		} else {
#pragma unroll
			for (int e = 0; e < EPL; e++) h[e] = p[e];
		}
	};
	auto dot = [&](const double (&h)[EPL], const double (&v)[EPL]) -> double {   // two chains: half the dependent latency
		double a0 = 0.0, a1 = 0.0;
#pragma unroll
		for (int e = 0; e < EPL; e++) { if (e & 1) a1 += h[e] * v[e]; else a0 += h[e] * v[e]; }
		return a0 + a1;
	};
	// tv += sum_j kappa_j d_j, kappa = (symmetric tridiagonal of the links) (d_j . v): DESIGN.md 4a.4, apply_dform of sqp_kernel.
	// On-chip slots: pass 1 (all dot products, 16 per butterfly), the kappa of every slot by one lane each (kept in that lane: pass 2
	// broadcasts it from there, no LDS read per slot), pass 2 (axpys).  Slots in HBM are read ONCE: rounds of HG vectors through two register
	// buffers, link by link (t += d_i (e delta_{i+1} + f delta_i) + d_{i+1} e delta_i), the first round requested before the on-chip
	// passes start -- its latency hides behind them.  (Round 4 tried the two-pass form for the HBM tier as well -- one FMA per element
	// instead of two, 16 dot products per butterfly instead of a reduction per round: fewer instructions, but the tier is bound by its
	// loads, and reading it twice made the headline 7 % and the cold start to convergence 32 % slower.  Measured, dropped.)
	constexpr int H0 = NREG + NLDS, HG = 2;   // chain slots per round of the HBM tier (two register buffers of HG vectors; other round sizes were never
	                                          // validated -- variant builds with 3 and 4 faulted on the GPU, profiles/r04_incidents -- so this is not a knob)
	auto sweep = [&](int ns, const double (&v)[EPL], double (&tv)[EPL]) {
		if (ns < 2) return;   // a chain of one vector carries no update yet
		const int nso = min(ns, H0);   // on-chip slots; the links nso-1 .. ns-2 belong to the HBM rounds
		double hA[HG][EPL], hB[HG][EPL];
		auto hload = [&](int base, double (&h)[HG][EPL]) {   // unconditional: past the end the newest vector again (masked where it is used)
#pragma unroll
			for (int g2 = 0; g2 < HG; g2++) {
				const double *p = hbm + (size_t)(min(base + g2, ns - 1) - H0) * EPL * 64 + lane;
#pragma unroll
				for (int e = 0; e < EPL; e++) h[g2][e] = p[e * 64];
			}
		};
		if (ns > H0) hload(H0, hA);
		double acc[16];
#pragma unroll
		for (int k = 0; k < 16; k++) acc[k] = 0.0;
		auto flush = [&](int base) __attribute__((always_inline)) {   // slots base .. base+15 -> delta
			const double tot = wave_sum_many<16>(acc, lane);
			if (lane < 16 && base + lane < nso) s_dl[base + lane] = tot;
#pragma unroll
			for (int k = 0; k < 16; k++) acc[k] = 0.0;
		};
		// pass 1: delta_j = d_j . v
		if constexpr (NREG > 0) {
			static_for<0, NREG>([&](auto Jc) __attribute__((always_inline)) {
				constexpr int J = decltype(Jc)::value;
				if (J < nso) {
					double h[EPL];
					static_for<0, EPL>([&](auto Ec) __attribute__((always_inline)) { constexpr int E = decltype(Ec)::value; h[E] = areg_read<ABASE, J * EPL + E>(); });
					acc[J & 15] = dot(h, v);
				}
				if ((J & 15) == 15 || J == NREG - 1) { if (J - (J & 15) < nso) flush(J - (J & 15)); }
			});
		}
		if constexpr (NLDS > 0) {
			for (int base = NREG; base < nso; base += 16) {
#pragma unroll
				for (int u = 0; u < 16; u++) {
					const int j = base + u;
					if (u < NLDS && j < nso) { double h[EPL]; lds_get(j, h); acc[u] = dot(h, v); }
				}
				flush(base);
			}
		}
		nwt_wave_sync();
		// kappa_j = f_j delta_j + e_j delta_{j+1} + e_{j-1} delta_{j-1}; link i = (e_i, f_i) joins slots i and i + 1.  Lane j keeps kappa_j
		// for the broadcasts of pass 2 (the on-chip tiers hold fewer than 64 slots)
		static_assert(H0 <= 64, "kappa of the on-chip slots lives in one register");
		double kap0 = 0.0;
		for (int j = lane; j < nso; j += 64) {
			const double dj = s_dl[j];
			double k = 0.0;
			if (j < nso - 1) k += s_lk[2 * j + 1] * dj + s_lk[2 * j] * s_dl[j + 1];
			if (j > 0) k += s_lk[2 * j - 2] * s_dl[j - 1];
			if (j == lane) kap0 = k;
		}
		auto kappa_of = [&](int j) -> double {   // uniform j < 64
			return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(kap0), j), __builtin_amdgcn_readlane(__double2loint(kap0), j));
		};
		// pass 2: tv += kappa_j d_j
		if constexpr (NREG > 0) {
			static_for<0, NREG>([&](auto Jc) __attribute__((always_inline)) {
				constexpr int J = decltype(Jc)::value;
				if (J < nso) {
					const double kj = bcast(kap0, J);
					static_for<0, EPL>([&](auto Ec) __attribute__((always_inline)) { constexpr int E = decltype(Ec)::value; tv[E] += kj * areg_read<ABASE, J * EPL + E>(); });
				}
			});
		}
		if constexpr (NLDS > 0) {
			for (int j = NREG; j < nso; j++) {
				const double kj = kappa_of(j);
				double h[EPL]; lds_get(j, h);
#pragma unroll
				for (int e = 0; e < EPL; e++) tv[e] += kj * h[e];
			}
		}
		if (ns > H0) {
			// the chain's tail in HBM, one pass; carried in: the last on-chip slot and its dot product
			double dc[EPL], dcl = 0.0;
#pragma unroll
			for (int e = 0; e < EPL; e++) dc[e] = 0.0;
			if constexpr (H0 > 0) {
				if constexpr (NLDS > 0) lds_get(H0 - 1, dc);
				else static_for<0, EPL>([&](auto Ec) __attribute__((always_inline)) { constexpr int E = decltype(Ec)::value; dc[E] = areg_read<ABASE, (NREG > 0 ? NREG - 1 : 0) * EPL + E>(); });
				dcl = s_dl[H0 - 1];
			}
			auto consume = [&](int base, double (&h)[HG][EPL]) {
				double ac[HG];
#pragma unroll
				for (int g2 = 0; g2 < HG; g2++) ac[g2] = dot(h[g2], v);
				wave_sums<HG, MINW == 1>(ac, lane);
#pragma unroll
				for (int g2 = 0; g2 < HG; g2++) {
					const int i = base - 1 + g2;   // link i joins slot i (left) and slot i + 1 = base + g2 (right)
					const bool on = i >= 0 && i < ns - 1;
					const int ii = on ? i : 0;
					const double le = on ? s_lk[2 * ii] : 0.0, lf = on ? s_lk[2 * ii + 1] : 0.0;
					const double dl = g2 == 0 ? dcl : ac[g2 > 0 ? g2 - 1 : 0], dr = ac[g2];
					const double ca = le * dr + lf * dl, cb = le * dl;
#pragma unroll
					for (int e = 0; e < EPL; e++) tv[e] += (g2 == 0 ? dc[e] : h[g2 > 0 ? g2 - 1 : 0][e]) * ca + h[g2][e] * cb;
				}
#pragma unroll
				for (int e = 0; e < EPL; e++) dc[e] = h[HG - 1][e];
				dcl = ac[HG - 1];
			};
			for (int base = H0; base < ns; base += 2 * HG) {
				hload(base + HG, hB);
				consume(base, hA);
				if (base + HG >= ns) break;
				hload(base + 2 * HG, hA);
				consume(base + HG, hB);
			}
		}
	};

	// ================= persistent loop over problems =================
	for (;;) {
		// (the wave barriers pin the queue pop between the end of one problem and the start of the next: without them the compiler
		// merged this lane-0 region with the lane-0 stores at the end of the loop body and sent the other 63 lanes round the loop on
		// their own, past the readfirstlane -- they then re-solved problem 0 for ever)
		__builtin_amdgcn_wave_barrier();
		int b = 0;
		if (lane == 0) b = (int)atomicAdd(A.counter, 1u);
		__builtin_amdgcn_wave_barrier();
		b = __builtin_amdgcn_readfirstlane(b);
		if (b >= A.batch) break;
		const double *lo = A.lower + (size_t)b * D.nbounds, *up = A.upper + (size_t)b * D.nbounds;
		if constexpr (PPG) {   // this problem's grid: values into the wave's own tables, the linear operator's values by pointer
			stage_values(T.rowv + (size_t)b * T.pp_rowv, T.bps + (size_t)b * T.pp_bps, T.q_val + (size_t)b * T.pp_q, lane, 64);
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			__builtin_amdgcn_wave_barrier();
			l_csr_val = T.csr_val + (size_t)b * T.pp_lin; l_csc_val = T.csc_val + (size_t)b * T.pp_lin; l_sinv_val = T.sinv_val + (size_t)b * T.pp_sinv;
			if (HESS && T.n0b) w0_base = T.n0b + (size_t)b * T.pp_n0b;
		}
#ifdef NTGW_DEBUG
		if (lane == 0) printf("wave %d takes problem %d of %d (cap %d)\n", wgid, b, A.batch, cap);
#endif
		double *xrow = A.xio + (size_t)b * nC;
		// (per-problem copy of the lane's coefficient base, opaque to the optimiser: with the plain kernel-lifetime constant it hoisted the six
		// CSC column pointers, the x addresses and more out of this loop and parked ~30 of them in accumulator registers for the whole
		// kernel -- inside the range the chain's register tier needs)
		int cb = cbase, ln = lane;
		asm volatile("" : "+v"(cb), "+v"(ln));
		double x[EPL], gp[EPL], gpt[EPL], d[EPL], g[EPL];
#pragma unroll
		for (int o = 0; o < OPL; o++)
#pragma unroll
			for (int q = 0; q < S; q++) x[o * S + q] = lane_on ? xrow[cb + o * nco + q] : 0.0;
#pragma unroll
		for (int e = 0; e < EPL; e++) { d[e] = 0.0; gp[e] = 0.0; gpt[e] = 0.0; g[e] = 0.0; }
		enum { ST_INIT = 0, ST_LS = 1, ST_FORCE = 2, ST_FINAL = 3 };
		int inform = 4, iter = 0, nfev = 0, nupd = 0, ns = 0, state = ST_INIT;
		bool headpair = false;   // the quasi-Newton memory was restarted at an accepted step: its first pair is not in the span of the chain
		// phase clock: variant builds only (-DNTGW_STAMPS, tools/mkvariant.sh + tests/tools_wave_stamps.py).  Compiled in, its eight 64-bit
		// accumulators and the time base were 18 scalar registers live across the whole kernel -- the shipped FAT instance carried ~930
		// v_writelane / v_readlane spill instructions in its major-iteration loop, most of them for these.
#ifdef NTGW_STAMPS
		unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
		// phase clock: the value a phase ends on is made opaque first, so that the phase's arithmetic cannot sink below the clock read
#define NTGW_STAMPV(slot_, val_) do { if (sp.stamps) { asm volatile("" ::"v"(val_)); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tk[slot_] += now_ - tlast; tlast = now_; } } while (0)
#else
#define NTGW_STAMPV(slot_, val_) do { } while (0)
#endif
#define NTGW_STAMP(slot_) NTGW_STAMPV(slot_, x[0])
#ifdef NTGW_STAMPS
		if (sp.stamps) tlast = __builtin_amdgcn_s_memtime();
#endif
		// ---- scope check: every linear row is an equality (lower == upper), see sqp_kernel ----
		{
			double bad[1] = {0.0};
			for (int s = ln; s < D.nlic + D.nltc + D.nlfc; s += 64) if (lo[s] != up[s]) bad[0] += 1.0;
			wave_sums<1, MINW == 1>(bad, lane);
			if (bad[0] != 0.0) inform = 9;
		}
		double F = 0.0, gn2 = 0.0, alpha = 0.0, pnorm = 0.0;
		int inner_inform = 4;
		bool weak = false, finished = false;
		if (inform != 9) {
			// ---- linear feasibility: x += A'(AA')^-1 (b - A x) ----
			if (m > 0) {
				stage_put(x);
				for (int r = ln; r < m; r += 64) {
					double a = 0.0;
					for (int e = l_csr_ptr[r]; e < l_csr_ptr[r + 1]; e++) a += l_csr_val[e] * s_st[l_csr_col[e]];
					s_tmp[r] = lo[lin_slot(D, r)] - a;
				}
				nwt_wave_sync();
				for (int r = ln; r < m; r += 64) {
					double a = 0.0;
					for (int e = l_sinv_ptr[r]; e < l_sinv_ptr[r + 1]; e++) a += l_sinv_val[e] * s_tmp[l_sinv_col[e]];
					s_tmp[64 + r] = a;
				}
				nwt_wave_sync();
				if (lane_on) {
#pragma unroll
					for (int o = 0; o < OPL; o++)
#pragma unroll
						for (int q = 0; q < S; q++) {
							const int c = cb + o * nco + q;
							double s = 0.0;
							for (int e = l_csc_ptr[c]; e < l_csc_ptr[c + 1]; e++) s += l_csc_val[e] * s_tmp[64 + l_csc_row[e]];
							x[o * S + q] += s;
						}
				}
				nwt_wave_sync();
			}
			NTGW_STAMP(0);
			// Line-search state.  One wave per SIMD (MINW == 1): in registers, uniform (linesearch.hpp: make_uniform) -- scalar branches, no LDS
			// round trips.  Two waves per SIMD: the 256-register budget has no room for it (measured: 16-32 registers spilled to scratch,
			// preconditioned solves 12 % slower), so those instances keep it in the wave's LDS, lane 0 writing it back.
			constexpr bool LSREG = (MINW == 1);
			LineSearch ls;
			LineSearch *lsm = (LineSearch *)(s_tmp + scal);   // (48 doubles reserved in front of the chain's LDS tier)
			ls.init(0.0, 0.0, 0.0, 0.0, sp.ls_mu, sp.ls_eta, sp.ls_maxfev);
			double ls_a = 0.0;
			double tstep = 0.0;            // the point to evaluate is x + tstep (-d): a line-search trial, or x itself (first and final evaluation)
			double r4[4] = {0, 0, 0, 0};   // gp.d, d.d, x.x, gp.gp of the current iterate
			for (;;) {
				// ================= the one evaluation site =================
				double part[3] = {0.0, 0.0, 0.0};
				{
					// the trial point is formed here and dies inside the evaluation (it is not kept across the line-search step: an accepted
					// step recomputes x + alpha (-d), the same operation on the same operands)
					double xt[EPL];
#pragma unroll
					for (int e = 0; e < EPL; e++) xt[e] = x[e] + tstep * (-d[e]);
					evaluate(xt, g, part[0], part[1]);
				}
				NTGW_STAMPV(1, g[0] + part[0] + part[1]);
				if (state != ST_FINAL) {
					project(g, gpt);
#pragma unroll
					for (int e = 0; e < EPL; e++) part[2] += gpt[e] * (-d[e]);
					NTGW_STAMPV(2, part[2]);
				}
#ifdef NTGW_DBL_RED3
				{ double q_[3] = {part[0], part[1], part[2]}; asm volatile("" : "+v"(q_[0]), "+v"(q_[1]), "+v"(q_[2])); wave_sums<3, MINW == 1>(q_, lane); asm volatile("" ::"s"(q_[0]), "s"(q_[1]), "s"(q_[2])); }
#endif
				wave_sums<3, MINW == 1>(part, lane);
				NTGW_STAMPV(6, part[0] + part[1] + part[2]);
				const double Fn = part[0], gn2n = part[1];
#ifdef NTGW_DEBUG
				if (lane == 0 && nfev < 40) printf("b %d state %d iter %d nfev %d ns %d F %.10g g2 %.6g slope %.6g a %.6g\n", b, state, iter, nfev, ns, Fn, gn2n, part[2], ls_a);
#endif
				if (state == ST_FINAL) {
					// multipliers estimate lam = (AA')^-1 A g at the final point
					stage_put(g);
					for (int r = ln; r < m; r += 64) {
						double a = 0.0;
						for (int e = l_csr_ptr[r]; e < l_csr_ptr[r + 1]; e++) a += l_csr_val[e] * s_st[l_csr_col[e]];
						s_tmp[r] = a;
					}
					nwt_wave_sync();
					for (int r = ln; r < m; r += 64) {
						double a = 0.0;
						for (int e = l_sinv_ptr[r]; e < l_sinv_ptr[r + 1]; e++) a += l_sinv_val[e] * s_tmp[l_sinv_col[e]];
						s_tmp[64 + r] = a;
					}
					nwt_wave_sync();
#ifdef NTGW_DEBUG
					if (lane == 0) printf("b %d multipliers done\n", b);
#endif
					break;
				}
				nfev++;
				bool new_major = false;
				if (state == ST_INIT) {
					F = Fn; gn2 = gn2n;
#pragma unroll
					for (int e = 0; e < EPL; e++) gp[e] = gpt[e];
					apply_w0(gp, d);
#pragma unroll
					for (int k = 0; k < 4; k++) r4[k] = 0.0;
#pragma unroll
					for (int e = 0; e < EPL; e++) { r4[0] += gp[e] * d[e]; r4[1] += d[e] * d[e]; r4[2] += x[e] * x[e]; r4[3] += gp[e] * gp[e]; }
					wave_sums<4, MINW == 1>(r4, lane);
					NTGW_STAMP(4);
					new_major = true;
				} else {
					int rc = 1;
					if (state == ST_LS) {
#ifdef NTGW_DBL_LS
						{ LineSearch l2_ = ls; double fq_ = Fn, sq_ = part[2]; asm volatile("" : "+v"(fq_), "+v"(sq_)); fq_ = LineSearch::uni_(fq_); sq_ = LineSearch::uni_(sq_);
						  int rq_ = __builtin_amdgcn_readfirstlane(l2_.step(fq_, sq_)); l2_.make_uniform(); asm volatile("" ::"s"(rq_), "s"(l2_.a), "s"(l2_.a_lo), "s"(l2_.a_hi), "s"(l2_.phi_lo), "s"(l2_.phi_hi), "s"(l2_.a_prev)); }
#endif
						if constexpr (LSREG) {
							rc = __builtin_amdgcn_readfirstlane(ls.step(Fn, part[2]));
							ls.make_uniform();
							ls_a = ls.a;
						} else {
							LineSearch lsr = *lsm;
							rc = lsr.step(Fn, part[2]);
							ls_a = lsr.a;
							nwt_wave_sync();   // every lane has read the state before lane 0 rewrites it
							if (lane == 0) *lsm = lsr;
							nwt_wave_sync();
						}
						NTGW_STAMPV(7, ls_a);
					}
					if (rc == 0 || rc == 2) {
						if (rc == 2) state = ST_FORCE;
						tstep = ls_a;
						NTGW_STAMP(5);
						continue;
					}
					if (rc != 1) {
						const double tolg = sp.sr * (1.0 + fmax(1.0 + fabs(F), sqrt(gn2)));
						if (nupd > 0 && sqrt(r4[3]) > tolg) {
							// line search failed with a non-trivial W: drop the updates and retry from the same point with W0
							nupd = 0; ns = 0; headpair = false;
							apply_w0(gp, d);
							double r2[2] = {0, 0};
#pragma unroll
							for (int e = 0; e < EPL; e++) { r2[0] += gp[e] * d[e]; r2[1] += d[e] * d[e]; }
							wave_sums<2, MINW == 1>(r2, lane);
							r4[0] = r2[0]; r4[1] = r2[1];
							new_major = true;
						} else {
							const double gpn0 = sqrt(r4[3]);
							if (gpn0 <= tolg) inner_inform = 0;
							else if (gpn0 <= 1e3 * tolg) { inner_inform = 0; weak = true; }
							else inner_inform = 6;
							finished = true;
						}
					} else {
						alpha = ls_a;
						// accept: s = alpha p, x = xt; t = W gp+ (W0, then the chain), u = t - d, BFGS update on the inverse
						double sv[EPL], tv[EPL];
#pragma unroll
						for (int e = 0; e < EPL; e++) { sv[e] = alpha * (-d[e]); x[e] = x[e] + tstep * (-d[e]); }   // (tstep == alpha: the point just evaluated)
						// memory full: restart the approximation from W0.  oracle/sqp.c restarts when the number of UPDATES reaches the memory; here a
						// skipped update (s'y <= 1e-12 |s||y|: a null link) still takes a chain slot, so after skipped updates the slots can run out
						// before the update count does and this kernel restarts a few majors EARLIER than the oracle and sqp_kernel -- a different (still
						// valid) quasi-Newton operator from there on.  On the strictly convex problems of this kernel's class s'y > 0 at every accepted
						// step (no update is ever skipped: the fixed-iteration parity tests compare evaluation counts), so the two conditions coincide.
						if (nupd == sp.memcap || ns + 3 > cap) {
							nupd = 0; ns = 0; headpair = true;
							apply_w0(gp, d);
						}
						if (ns == 0 && !headpair) { slot_store(0, d); ns = 1; }   // a chain starts: d_0 = the direction of this step
						NTGW_STAMP(5);
						apply_w0(gpt, tv);
						NTGW_STAMPV(4, tv[0]);
#ifdef NTGW_DBL_SWEEP
						{ double t2_[EPL], v2_[EPL];
#pragma unroll
						  for (int e = 0; e < EPL; e++) { t2_[e] = tv[e]; v2_[e] = gpt[e]; asm volatile("" : "+v"(v2_[e])); }
						  sweep(ns, v2_, t2_);
#pragma unroll
						  for (int e = 0; e < EPL; e++) asm volatile("" ::"v"(t2_[e])); }
#endif
						sweep(ns, gpt, tv);
						NTGW_STAMPV(3, tv[0]);
						// one butterfly for everything that does not depend on the update's scalars: the six products of the BFGS formulas and the
						// two norms of the next major's tests (|x|^2, |gp+|^2); the two that involve the new direction follow below
						double r6[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
						for (int e = 0; e < EPL; e++) {
							const double s = sv[e], y = gpt[e] - gp[e], u = tv[e] - d[e], gq = gpt[e];
							r6[0] += s * y; r6[1] += y * u; r6[2] += s * gq; r6[3] += u * gq; r6[4] += s * s; r6[5] += y * y;
							r6[6] += x[e] * x[e]; r6[7] += gq * gq;
						}
						wave_sums<8, MINW == 1>(r6, lane);
						const bool upd = r6[0] > 1e-12 * sqrt(r6[4]) * sqrt(r6[5]);
						const double rho = upd ? 1.0 / r6[0] : 0.0, c2 = upd ? rho * (1.0 + rho * r6[1]) : 0.0;
						double dn[EPL];
#pragma unroll
						for (int e = 0; e < EPL; e++) {
							const double s = sv[e], u = tv[e] - d[e];
							dn[e] = upd ? tv[e] - rho * (s * r6[3] + u * r6[2]) + c2 * s * r6[2] : tv[e];
						}
						if (headpair && upd) {
							// first update after a restart at an accepted step: s is the step actually taken (along the OLD direction), not
							// -alpha W0 gp: the pair (s, u) is stored as two slots joined by the link (-rho, c2) -- exactly the pair term
							// -rho (s u' + u s') + c2 s s' -- followed by a null link to the chain that starts with d+
							double uv[EPL];
#pragma unroll
							for (int e = 0; e < EPL; e++) uv[e] = tv[e] - d[e];
							slot_store(0, sv); slot_store(1, uv); slot_store(2, dn);
							if (lane == 0) { s_lk[0] = -rho; s_lk[1] = c2; s_lk[2] = 0.0; s_lk[3] = 0.0; }
							ns = 3; nupd = 1;
						} else if (headpair) {
							slot_store(0, dn); ns = 1;   // no update: W stays W0, a clean chain starts at d+
						} else {
							// link of this major: s = -alpha d_k, u = beta d_{k+1} + gamma d_k (omega = -(s.g)/(s.y) > 0 whenever the update is taken)
							double le = 0.0, lf = 0.0;
							if (upd) {
								const double omega = 1.0 - rho * r6[2], theta = rho * r6[2] - alpha * c2 * r6[2] + alpha * rho * r6[3];
								const double beta = 1.0 / omega, gamma = -theta * beta - 1.0;
								le = rho * alpha * beta; lf = 2.0 * rho * alpha * gamma + c2 * alpha * alpha;
								nupd++;
							}
							slot_store(ns, dn);
							if (lane == 0) { s_lk[2 * (ns - 1)] = le; s_lk[2 * (ns - 1) + 1] = lf; }
							ns++;
						}
						headpair = false;
						nwt_wave_sync();
						double r2[2] = {0.0, 0.0};
#pragma unroll
						for (int e = 0; e < EPL; e++) {
							gp[e] = gpt[e]; d[e] = dn[e];
							r2[0] += gp[e] * d[e]; r2[1] += d[e] * d[e];
						}
						wave_sums<2, MINW == 1>(r2, lane);
						r4[0] = r2[0]; r4[1] = r2[1]; r4[2] = r6[6]; r4[3] = r6[7];
						F = Fn; gn2 = gn2n;
						iter++;
						if (!sp.fixed_iters && alpha * pnorm <= sp.sr * (1.0 + sqrt(r4[2])) &&
						    sqrt(r4[3]) <= sp.sr * (1.0 + fmax(1.0 + fabs(F), sqrt(gn2)))) { inner_inform = 0; finished = true; }
						else new_major = true;
					}
				}
				if (new_major) {
					// ---- start of a major iteration at (x, gp, d) ----
					if (iter >= sp.itlim) { inner_inform = 4; finished = true; }
					else {
						double dphi0 = -r4[0];
						pnorm = sqrt(r4[1]);
						const double xnorm = sqrt(r4[2]), gpnorm = sqrt(r4[3]);
						const double tolg = sp.sr * (1.0 + fmax(1.0 + fabs(F), sqrt(gn2)));
						if (pnorm == 0.0 || !(dphi0 < 0.0)) {
							if (pnorm != 0.0) {   // W lost definiteness numerically: restart from W0 once
								nupd = 0; ns = 0; headpair = false;
								apply_w0(gp, d);
								double r2[2] = {0, 0};
#pragma unroll
								for (int e = 0; e < EPL; e++) { r2[0] += gp[e] * d[e]; r2[1] += d[e] * d[e]; }
								wave_sums<2, MINW == 1>(r2, lane);
								r4[0] = r2[0]; r4[1] = r2[1];
								dphi0 = -r2[0]; pnorm = sqrt(r2[1]);
							}
							if (pnorm == 0.0 || !(dphi0 < 0.0)) { inner_inform = (gpnorm <= tolg) ? 0 : 6; finished = true; }
						}
						if (!finished && !sp.fixed_iters && gpnorm <= 1e-3 * tolg) { inner_inform = 0; finished = true; }
						if (!finished) {
							const double amax = sp.steplimit * (1.0 + xnorm) / pnorm;
							const double a = amax < 1.0 ? amax : 1.0;
							if constexpr (LSREG) ls.init(F, dphi0, a, amax, sp.ls_mu, sp.ls_eta, sp.ls_maxfev);
							else {
								if (lane == 0) lsm->init(F, dphi0, a, amax, sp.ls_mu, sp.ls_eta, sp.ls_maxfev);
								nwt_wave_sync();
							}
							ls_a = a; tstep = a;
							state = ST_LS;
						}
					}
				}
				NTGW_STAMP(5);
				if (finished) {
					inform = (inner_inform == 0 && weak) ? 1 : inner_inform;
					if (A.clambda && m > 0) {   // one more pass at x for the multipliers (the Q form does not produce them)
						state = ST_FINAL; tstep = 0.0;
						continue;
					}
					break;
				}
			}
		}
#ifdef NTGW_DEBUG
		if (lane == 0) printf("b %d left the iteration: inform %d iter %d\n", b, inform, iter);
#endif
		if (inform != 9 && lane_on) {
#pragma unroll
			for (int o = 0; o < OPL; o++)
#pragma unroll
				for (int q = 0; q < S; q++) xrow[cb + o * nco + q] = x[o * S + q];
		}
#ifdef NTGW_DEBUG
		if (lane == 0) printf("b %d x written\n", b);
#endif
		NTGW_STAMP(5);
		if (A.clambda) {
			const int ntot = nC + D.nclin;   // NPSOL's layout: coefficients, linear rows (no nonlinear rows in this class)
			int ln2 = lane;
			asm volatile("" : "+v"(ln2));
			for (int i = ln2; i < ntot; i += 64) {
				double v = 0.0;
				if (inform != 9 && i >= nC && m > 0) v = s_tmp[64 + (i - nC)];
				A.clambda[(size_t)b * ntot + i] = v;
			}
#ifdef NTGW_STAMPS
			if (sp.stamps && lane == 0) for (int i = 0; i < 8; i++) A.clambda[(size_t)b * ntot + i] = (double)tk[i];
#endif
			nwt_wave_sync();
		}
#undef NTGW_STAMP
#undef NTGW_STAMPV
		if (lane == 0) {
			if (A.objective) A.objective[b] = F;
			if (A.inform) A.inform[b] = inform;
			if (A.iters) A.iters[b] = iter;
			if (A.nfev) A.nfev[b] = nfev;
		}
#ifdef NTGW_DEBUG
		if (lane == 0) printf("b %d done\n", b);
#endif
		__builtin_amdgcn_wave_barrier();
	}
#ifdef NTGW_DEBUG
	if (lane == 0) printf("wave %d exits\n", wgid);
#endif
}

// does the plan fit the wave kernel?  (The dispatcher falls back to sqp_kernel otherwise.)
static inline bool wave_match(const NtgDims &D, const NtgTables &T, const SolveParams &sp, int chm, int dm, int K, int opl, int nint)
{
	if (!ntg_chm_match(D, chm, dm) || D.ncnln || !D.uniform || D.nI) return false;
	if (D.order[0] != K || (K & 1) || D.mult[0] != K / 2 || D.ig_n != nint || D.nout % opl) return false;
	if (D.ncoef[0] != (K / 2) * D.ig_n + K / 2) return false;
	if ((D.ig_n + 1) * (D.nout / opl) > 64) return false;
	if (D.mE != D.nclin || D.mE > 64) return false;
	if (D.mE > 0 && (!D.q_use || D.q_w > 6)) return false;
	if ((T.pp_rowv || T.pp_bps || T.pp_q) && nint != 20) return false;   // per-problem grids: the PPG instances (20 intervals), else sqp_kernel
	if ((T.pp_rowv || T.pp_bps || T.pp_q) && sp.hessian == 1 && !T.pp_n0b) return false;
	if (sp.hessian == 1 && !(T.n0b && T.n0b_n == D.ncoef[0])) return false;
	if (sp.hessian == 2) return false;
	return true;
}

struct WaveLaunch {
	int nwv, grid, cap, hbm_slots;
	size_t lds;
};

}   // namespace ntgw
