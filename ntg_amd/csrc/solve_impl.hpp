// solve_impl.hpp -- templated device code of eval_kernel and sqp_kernel (gfx950).  Included by the per-family
// translation units (fam_*.hip: one per problem family, so that they compile in parallel) and by kernels.hip
// (host-callback kernels share the quadrature / gradient assembly).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>
#include "ntg_dev.hpp"
#include "families.hpp"
#include "linesearch.hpp"
#include "qpdual.hpp"

// tuning knobs: minimum waves per SIMD the register allocator must leave room
// for, and how many quasi-Newton pairs one reduction round of apply_history covers
#ifndef NTG_SQP_WAVES
#define NTG_SQP_WAVES 2
#endif
#ifndef NTG_EVAL_WAVES
#define NTG_EVAL_WAVES 2
#endif
// phase clock of eval_kernel (variant builds only, tests/tools_jac_clock.py): s_memtime ticks of workgroup 0's first lane per phase
#ifdef NTG_EVAL_CLOCK
// accumulators in LDS: a global read-modify-write here would wait for vmcnt(0), i.e. for the wave's outstanding stores -- the clock would
// charge the drain of the emission's stores to the emission
#define EVCLK(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); ntg_evclk_s[slot] += t_ - ntg_evclk_s[15]; ntg_evclk_s[15] = t_; } } while (0)
#define EVCLK0() do { if (blockIdx.x == 0 && threadIdx.x == 0) { for (int i_ = 0; i_ < 15; i_++) ntg_evclk_s[i_] = 0; ntg_evclk_s[15] = __builtin_readcyclecounter(); } } while (0)
#define EVCLK_DECL __shared__ unsigned long long ntg_evclk_s[16];
#define EVCLK_TK (blockIdx.x == 0 ? ntg_evclk_s + 4 : nullptr)
#define EVCLK_ARG , unsigned long long *ntg_evclk_s
#define EVCLK_PASS , ntg_evclk_s
#else
#define EVCLK(slot) do { } while (0)
#define EVCLK0() do { } while (0)
#define EVCLK_DECL
#define EVCLK_TK nullptr
#define EVCLK_ARG
#define EVCLK_PASS
#endif
#ifndef NTG_HIST_G
#define NTG_HIST_G 6
#endif
#ifndef NTG_HIST_PIPE
#define NTG_HIST_PIPE 2   // pairs per round of the double-buffered sweep (pair scalars in LDS, 3 coefficients per lane); 0 = off
#endif
#ifndef NTG_DFORM
#define NTG_DFORM 1
#endif
#ifndef NTG_DF_G
#define NTG_DF_G 4
#endif
#ifndef NTG_HIST_ZIGZAG
#define NTG_HIST_ZIGZAG 0
#endif

// ------------------------------------------------------------------------------------------
// reductions: sum K values over the workgroup, result broadcast to every lane
// ------------------------------------------------------------------------------------------
// Workgroup barrier for LDS-only hand-offs.  __syncthreads() also drains vmcnt(0), which would
// serialise every in-flight HBM load (history pairs, prefetched coefficient vectors) behind each of
// the many barriers of this kernel; LDS visibility only needs lgkmcnt(0) on both sides.
__device__ __forceinline__ void lds_sync()
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

// 64-lane wavefront sum with DPP (row shifts + row broadcasts stay in the VALU; __shfl_down
// would go through ds_bpermute, i.e. the LDS crossbar, ~100 cycles per hop).  The total lands
// in lane 63 and is broadcast from there through an SGPR.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
	return v + __hiloint2double(hi, lo);   // disabled / out-of-row lanes contribute +0.0
}
__device__ __forceinline__ double wave_sum(double v)
{
	v = dpp_add<0x111, 0xf>(v);   // row_shr:1
	v = dpp_add<0x112, 0xf>(v);   // row_shr:2
	v = dpp_add<0x114, 0xf>(v);   // row_shr:4
	v = dpp_add<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every row holds its row total
	v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
	v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
	const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
	return __hiloint2double(hi, lo);
}

// exchange a double with the lane whose index differs in one bit (1, 2, 4, 8: DPP inside a row of
// 16; 16: ds_swizzle; 32: bpermute)
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_mov(double v, double old)
{
	const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xf, BANK, false);
	const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xf, BANK, false);
	return __hiloint2double(hi, lo);
}
template <int BIT>
__device__ __forceinline__ double lane_xchg(double v)
{
	if (BIT == 1) return dpp_mov<0xB1, 0xf>(v, 0.0);                       // quad_perm [1,0,3,2]
	if (BIT == 2) return dpp_mov<0x4E, 0xf>(v, 0.0);                       // quad_perm [2,3,0,1]
	if (BIT == 4) return dpp_mov<0x114, 0xA>(v, dpp_mov<0x104, 0x5>(v, 0.0)); // row_shl:4 into banks 0,2 ; row_shr:4 into banks 1,3
	if (BIT == 8) return dpp_mov<0x128, 0xf>(v, 0.0);                      // row_ror:8
	if (BIT == 16) {
		const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F), hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F);
		return __hiloint2double(hi, lo);
	}
	return __shfl_xor(v, 32, 64);
}

// t(lane) + t(lane ^ 16) / t(lane) + t(lane ^ 32) in every lane, in the VALU: gfx950's v_permlane16_swap / v_permlane32_swap exchange
// the odd rows (the upper half) of one register with the even rows (the lower half) of another -- applied to two copies of t they leave
// "my half's value" in one and "the other half's" in the other, and the sum of the two is the same number in both partners (the earlier
// form went through the LDS crossbar: ds_swizzle + ds_bpermute, two ~100-cycle round trips at the end of every butterfly).
template <int BIT>
__device__ __forceinline__ double xsum_rows(double t)
{
	const int lo = __double2loint(t), hi = __double2hiint(t);
	if constexpr (BIT == 16) {
		const auto r = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), s = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
		return __hiloint2double(s[0], r[0]) + __hiloint2double(s[1], r[1]);
	} else {
		const auto r = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), s = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
		return __hiloint2double(s[0], r[0]) + __hiloint2double(s[1], r[1]);
	}
}

// Sums of a FEW (1..3) per-lane values over the wavefront, every lane receives all of them.  Row stage by rotation (DPP row_ror 1, 2, 4,
// 8: every lane of a row ends up with its row's total, no masked-out lanes and therefore no zero-initialised "old" operands), the two
// cross-row stages on the permlane swaps, the stages of the KV values interleaved (independent chains: with one wave per SIMD nothing
// else hides the DPP latency; the earlier form ran KV complete 6-stage row_shr / row_bcast chains one after the other, 36 instructions
// each).  Lanes of a row add in different orders, so the result is taken from ONE lane (readfirstlane) -- a scalar, known uniform.
template <int CTRL>
__device__ __forceinline__ double dpp_rot(double v)
{
	const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
	const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double uniform_of(double v)   // lane 0's value, as a scalar
{
	return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <int KV>
__device__ __forceinline__ void wave_sums_few(double (&v)[KV])
{
#pragma unroll
	for (int k = 0; k < KV; k++) v[k] += dpp_rot<0x121>(v[k]);   // row_ror:1
#pragma unroll
	for (int k = 0; k < KV; k++) v[k] += dpp_rot<0x122>(v[k]);   // row_ror:2
#pragma unroll
	for (int k = 0; k < KV; k++) v[k] += dpp_rot<0x124>(v[k]);   // row_ror:4
#pragma unroll
	for (int k = 0; k < KV; k++) v[k] += dpp_rot<0x128>(v[k]);   // row_ror:8
#pragma unroll
	for (int k = 0; k < KV; k++) v[k] = xsum_rows<16>(v[k]);
#pragma unroll
	for (int k = 0; k < KV; k++) v[k] = xsum_rows<32>(v[k]);
#pragma unroll
	for (int k = 0; k < KV; k++) v[k] = uniform_of(v[k]);
}

// Sum KV (<= 16) per-lane values over the 64 lanes with the "halving" butterfly: at the step of
// bit b a lane keeps the half of its values whose index has bit b equal to its own lane bit and
// hands the other half to its partner, so the number of live values halves every step (15 exchanges
// for 16 values instead of 16 x 6).  Afterwards lane L holds the wave total of value L & 15.
template <int KV, int N, int BIT>
__device__ __forceinline__ void halve_step(const double *in, double *out, int lane)
{
	const bool hi = (lane & BIT) != 0;
#pragma unroll
	for (int j = 0; j < N / 2; j++) {
		// original indices covered by in[2j] / in[2j+1] start at (2j)*BIT and (2j+1)*BIT: skip all-padding pairs
		if ((2 * j) * BIT >= KV) { out[j] = 0.0; continue; }
		const double a = in[2 * j], b = ((2 * j + 1) * BIT < KV) ? in[2 * j + 1] : 0.0;
		const double keep = hi ? b : a, send = hi ? a : b;
		out[j] = keep + lane_xchg<BIT>(send);
	}
}
template <int KV>
__device__ __forceinline__ double wave_sum_many(const double *v, int lane)   // v has 16 slots, KV valid
{
	double a[8], b[4], c[2], d[1];
	halve_step<KV, 16, 1>(v, a, lane);
	halve_step<KV, 8, 2>(a, b, lane);
	halve_step<KV, 4, 4>(b, c, lane);
	halve_step<KV, 2, 8>(c, d, lane);
	double t = d[0];
	t = xsum_rows<16>(t);
	t = xsum_rows<32>(t);
	return t;
}

// The cross-wave scratch is double buffered (S.red_sel alternates between two halves): a buffer is rewritten two calls after it was
// read, and the barrier of the call in between already orders those, so ONE barrier per reduction is enough (a single buffer needs a
// second one in front of the writes).
template <int NT, int K, class SM>
__device__ __forceinline__ void block_sum(double (&v)[K], const SM &S)
{
	constexpr int NW = NT / 64;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (K >= 4) {
		double w[16];
#pragma unroll
		for (int k = 0; k < 16; k++) w[k] = k < K ? v[k] : 0.0;
		const double t = wave_sum_many<K>(w, lane);   // lane L: wave total of value L & 15
		double *red = S.red + (S.red_sel ? 16 * NW : 0);
		S.red_sel ^= 1;
		if (lane < K) red[lane * NW + wave] = t;
		lds_sync();
#pragma unroll
		for (int k = 0; k < K; k++) {
			double s2 = red[k * NW];
#pragma unroll
			for (int w2 = 1; w2 < NW; w2++) s2 += red[k * NW + w2];
			v[k] = s2;
		}
		return;
	}
#pragma unroll
	for (int k = 0; k < K; k++) v[k] = wave_sum(v[k]);
	if (NW == 1) return;
	double *red = S.red + (S.red_sel ? 16 * NW : 0);
	S.red_sel ^= 1;
	if (lane == 0) {
#pragma unroll
		for (int k = 0; k < K; k++) red[k * NW + wave] = v[k];
	}
	lds_sync();
#pragma unroll
	for (int k = 0; k < K; k++) {
		double s = red[k * NW];
#pragma unroll
		for (int w = 1; w < NW; w++) s += red[k * NW + w];
		v[k] = s;
	}
}

// every lane owns the vector elements c = tid + e*NT.  For n <= 3*NT the three slots are unrolled
// so that their (independent) LDS dependency chains overlap; longer vectors take the plain loop.
template <int NT, class F>
__device__ __forceinline__ void for_vec(int n, F f)
{
	if (n <= 3 * NT) {
#pragma unroll
		for (int e = 0; e < 3; e++) { const int c = threadIdx.x + e * NT; if (c < n) f(c); }
	} else {
		for (int c = threadIdx.x; c < n; c += NT) f(c);
	}
}

#include "newton.hpp"

// ------------------------------------------------------------------------------------------
// LDS carve-up shared by eval_kernel and sqp_kernel
// ------------------------------------------------------------------------------------------
struct Smem {
	double *rowv; unsigned int *colp; int *chrow, *chcol;
	int *off; double *bps, *wts;
	double *x, *dfz, *fvals, *red, *dfi, *dff, *vecs, *lam, *rho, *c2;
	// sparse linear-constraint operator: LDS copies when they fit, HBM/L2 otherwise
	const int *csr_ptr, *csr_col, *csc_ptr, *csc_row, *sinv_ptr, *sinv_col; const double *csr_val, *csc_val, *sinv_val;
	int *oinfo, *tavrow, *tcomp;   // per-output scalars, flag->row map, flag->compact trajectory-constraint index, in LDS
	short *q_idx; int *q_col; double *q_val;
	int tav_rows;   // SmemLayout::tav_rows
	mutable int red_sel;   // which half of the reduction scratch the next block_sum writes (wave uniform)
	__device__ __forceinline__ Smem(char *base, const SmemLayout &L, const NtgDims &D, const NtgTables &T, int b = 0)
	{
		rowv = (double *)(base + L.rowv); colp = (unsigned int *)(base + L.colp);
		chrow = (int *)(base + L.chrow); chcol = (int *)(base + L.chcol);
		off = (int *)(base + L.off); bps = (double *)(base + L.bps);
		wts = (double *)(base + L.wts);
		x = (double *)(base + L.x); dfz = (double *)(base + L.dfz); fvals = (double *)(base + L.fvals);
		red = (double *)(base + L.red); dfi = (double *)(base + L.dfi); dff = (double *)(base + L.dff);
		vecs = (double *)(base + L.vecs); lam = (double *)(base + L.lam); rho = (double *)(base + L.rho);
		c2 = (double *)(base + L.c2);
		if (D.lin_lds && L.with_lin) {
			csr_ptr = (const int *)(base + L.csr_ptr); csr_col = (const int *)(base + L.csr_col); csr_val = (const double *)(base + L.csr_val);
			csc_ptr = (const int *)(base + L.csc_ptr); csc_row = (const int *)(base + L.csc_row); csc_val = (const double *)(base + L.csc_val);
			sinv_ptr = (const int *)(base + L.sinv_ptr); sinv_col = (const int *)(base + L.sinv_col); sinv_val = (const double *)(base + L.sinv_val);
		} else {
			csr_ptr = T.csr_ptr; csr_col = T.csr_col; csr_val = T.csr_val + (size_t)b * T.pp_lin;
			csc_ptr = T.csc_ptr; csc_row = T.csc_row; csc_val = T.csc_val + (size_t)b * T.pp_lin;
			sinv_ptr = T.sinv_ptr; sinv_col = T.sinv_col; sinv_val = T.sinv_val + (size_t)b * T.pp_sinv;
		}
		oinfo = (int *)(base + L.oinfo); tavrow = (int *)(base + L.tavrow); tcomp = (int *)(base + L.tcomp);
		q_idx = (short *)(base + L.q_idx); q_col = (int *)(base + L.q_col); q_val = (double *)(base + L.q_val);
		tav_rows = L.tav_rows; red_sel = 0;
	}
};

// b: the problem whose grid is staged (matters only with per-problem grids, NtgTables::pp_*)
template <int NT>
__device__ __forceinline__ void stage_tables(const NtgDims &D, const NtgTables &T0, const Smem &S, char *base, const SmemLayout &L, int b = 0)
{
	const int tid = threadIdx.x;
	NtgTables T = T0;
	T.rowv += (size_t)b * T0.pp_rowv; T.bps += (size_t)b * T0.pp_bps;
	T.csr_val += (size_t)b * T0.pp_lin; T.csc_val += (size_t)b * T0.pp_lin; T.sinv_val += (size_t)b * T0.pp_sinv; T.q_val += (size_t)b * T0.pp_q;
	for (int i = tid; i < D.row_total; i += NT) S.rowv[i] = T.rowv[i];
	for (int i = tid; i < D.col_total; i += NT) S.colp[i] = T.colp[i];
	for (int i = tid; i < D.nclass * NTG_MAX_ORDER; i += NT) { S.chrow[i] = T.chrow[i]; S.chcol[i] = T.chcol[i]; }
	for (int i = tid; i < D.nclass * D.P; i += NT) S.off[i] = T.off[i];
	for (int i = tid; i < D.P; i += NT) {
		S.bps[i] = T.bps[i];
		// trapezoid weight of breakpoint i: integrator.c:21-24 regrouped per node
		double w = 0.0;
		if (i > 0) w += (T.bps[i] - T.bps[i - 1]) / 2;
		if (i < D.P - 1) w += (T.bps[i + 1] - T.bps[i]) / 2;
		S.wts[i] = w;
	}
	if (D.lin_lds && L.with_lin) {
		int *rp = (int *)(base + L.csr_ptr), *rc = (int *)(base + L.csr_col), *cp = (int *)(base + L.csc_ptr), *cr = (int *)(base + L.csc_row);
		double *rv = (double *)(base + L.csr_val), *cv = (double *)(base + L.csc_val), *sv = (double *)(base + L.sinv_val);
		int *sp_ = (int *)(base + L.sinv_ptr), *sc = (int *)(base + L.sinv_col);
		for (int i = tid; i <= D.mE; i += NT) rp[i] = T.csr_ptr[i];
		for (int i = tid; i <= D.nC; i += NT) cp[i] = T.csc_ptr[i];
		for (int i = tid; i < D.lin_nnz; i += NT) { rc[i] = T.csr_col[i]; rv[i] = T.csr_val[i]; cr[i] = T.csc_row[i]; cv[i] = T.csc_val[i]; }
		for (int i = tid; i <= D.mE; i += NT) sp_[i] = T.sinv_ptr[i];
		for (int i = tid; i < D.sinv_nnz; i += NT) { sc[i] = T.sinv_col[i]; sv[i] = T.sinv_val[i]; }
	}
	// per-output scalars: k, m, preconditioner block, d, iC, iz, channel base, offset-table base, column-form width, ncoef
	for (int o = tid; o < D.nout; o += NT) {
		int *q = S.oinfo + o * 10;
		q[0] = D.order[o]; q[1] = D.mult[o]; q[2] = D.n0_blk[o]; q[3] = D.d[o]; q[4] = D.iC[o]; q[5] = D.iz[o];
		q[6] = D.cls[o] * NTG_MAX_ORDER; q[7] = D.cls[o] * D.P; q[8] = D.cls_W[D.cls[o]]; q[9] = D.ncoef[o];
	}
	for (int v = tid; v < D.nz; v += NT) {
		S.tavrow[v] = D.tav_row[v] < L.tav_rows ? D.tav_row[v] : -1;
		S.tcomp[v] = ((D.tcon_mask >> v) & 1ull) ? __popcll(D.tcon_mask & ((1ull << v) - 1ull)) : -1;   // see eval_constraints
	}
	for (int r = tid; r < L.dfz_rows; r += NT) S.dfz[r * (D.P + 1) + D.P] = 0.0;   // the row's extra element stays 0
	for (int i = tid; i < ntg_dfz_tail(D); i += NT) S.dfz[L.dfz_rows * (D.P + 1) + i] = 0.0;   // overrun of the last columns' reads (times 0)
	if (D.q_use && L.with_lin) {
		for (int i = tid; i < D.nC; i += NT) S.q_idx[i] = T.q_idx[i];
		for (int i = tid; i < D.q_nt * D.q_w; i += NT) { S.q_col[i] = T.q_col[i]; S.q_val[i] = T.q_val[i]; }
	}
}

// Z = M C at one breakpoint for the declared active variables (colloc.c:318-326,344-367);
// entries that are not active stay 0 like the reference's calloc'd GZ (ntg.c:119).
// CHM (channel mask, compile time): the instance was picked because the trajectory-cost active variables are exactly
// "derivative r of EVERY output, for the r in CHM" (kincar: CHM = 4, the second derivatives).  Rows, masks and channel
// tests that the general code looks up at run time are then constants.
__host__ __device__ constexpr int chm_count(int chm) { return (chm & 1) + ((chm >> 1) & 1) + ((chm >> 2) & 1) + ((chm >> 3) & 1) + ((chm >> 4) & 1); }
__host__ __device__ constexpr int chm_rank(int chm, int r) { return chm_count(chm & ((1 << r) - 1)); }
__host__ __device__ constexpr u64 chm_full_mask(int chm, int nout, int dm)
{
	u64 m = 0;
	for (int o = 0; o < nout; o++) for (int r = 0; r < dm; r++) if ((chm >> r) & 1) m |= 1ull << (dm * o + r);
	return m;
}

template <int NOUT, int K, int DM, int CHM = 0>
__device__ __forceinline__ void compute_z(const NtgDims &D, const Smem &S, const double *sx, int bp,
                                          u64 mask, double *z, bool chm_ok = false)
{
	const int nout = NOUT > 0 ? NOUT : D.nout, P = D.P;
	if (NOUT > 0 && K > 0 && CHM != 0 && chm_ok) {
		// every output, the derivative channels of CHM, nothing else: no masks, no channel tests
		constexpr int KK = K > 0 ? K : 1, NCH = chm_count(CHM);
		double b[NCH > 0 ? NCH : 1][KK];
#pragma unroll
		for (int r = 0; r < DM; r++) {
			if (!((CHM >> r) & 1)) continue;
			const int ch = S.chrow[r];
#pragma unroll
			for (int q = 0; q < KK; q++) b[chm_rank(CHM, r)][q] = S.rowv[ch + q * P + bp];
		}
		const int ofs = S.off[bp], nco = D.ncoef[0];
#pragma unroll
		for (int o = 0; o < (NOUT > 0 ? NOUT : 1); o++) {
			const double *cx = sx + o * nco + ofs;
			double xv[KK];
#pragma unroll
			for (int q = 0; q < KK; q++) xv[q] = cx[q];
#pragma unroll
			for (int r = 0; r < DM; r++) {
				double acc = 0.0;
				if ((CHM >> r) & 1) {
#pragma unroll
					for (int q = 0; q < KK; q++) acc += b[chm_rank(CHM, r)][q] * xv[q];
				}
				z[DM * o + r] = acc;
			}
		}
		return;
	}
	if (NOUT > 0 && K > 0 && DM > 3 && D.uniform) {
		// one basis class, compile-time order, many derivatives: the coefficients of one output stay in
		// registers while its active derivative rows stream from LDS (keeping all DM rows would not fit)
		constexpr int KK = K > 0 ? K : 1;
		const int ofs = S.off[bp], nco = D.ncoef[0];
#pragma unroll
		for (int o = 0; o < (NOUT > 0 ? NOUT : 1); o++) {
			const double *cx = sx + o * nco + ofs;
			double xv[KK];
#pragma unroll
			for (int q = 0; q < KK; q++) xv[q] = cx[q];
#pragma unroll
			for (int r = 0; r < DM; r++) {
				double acc = 0.0;
				const int ch = S.chrow[r];
				if (((mask >> (DM * o + r)) & 1ull) && ch >= 0) {   // wave-uniform
#pragma unroll
					for (int q = 0; q < KK; q++) acc += S.rowv[ch + q * P + bp] * xv[q];
				}
				z[DM * o + r] = acc;
			}
		}
		return;
	}
	if (NOUT > 0 && K > 0 && D.uniform) {
		// one basis class, compile-time order, maxderiv == 3: the rows of the active derivative
		// channels at this breakpoint are read once into registers and reused by every output
		constexpr int KK = K > 0 ? K : 1;
		double b[3][KK];
#pragma unroll
		for (int r = 0; r < 3; r++) {
			const int ch = S.chrow[r];
			if (ch >= 0) {                                   // wave-uniform
#pragma unroll
				for (int q = 0; q < KK; q++) b[r][q] = S.rowv[ch + q * P + bp];
			} else {
#pragma unroll
				for (int q = 0; q < KK; q++) b[r][q] = 0.0;
			}
		}
		const int ofs = S.off[bp], nco = D.ncoef[0];
#pragma unroll
		for (int o = 0; o < (NOUT > 0 ? NOUT : 1); o++) {
			const double *cx = sx + o * nco + ofs;
			double xv[KK];
#pragma unroll
			for (int q = 0; q < KK; q++) xv[q] = cx[q];
#pragma unroll
			for (int r = 0; r < 3; r++) {
				double acc = 0.0;
				if ((mask >> (3 * o + r)) & 1ull) {           // wave-uniform
#pragma unroll
					for (int q = 0; q < KK; q++) acc += b[r][q] * xv[q];
				}
				z[3 * o + r] = acc;
			}
		}
		return;
	}
#pragma unroll
	for (int o = 0; o < nout; o++) {
		const int k = D.order[o], c = D.cls[o];
		const int d = NOUT > 0 ? DM : D.d[o], iz = NOUT > 0 ? DM * o : D.iz[o];
		const double *cx = sx + D.iC[o] + S.off[c * P + bp];
#pragma unroll
		for (int r = 0; r < (NOUT > 0 ? DM : NTG_MAX_ORDER); r++) {
			if (r >= d) break;
			double acc = 0.0;
			if ((mask >> (iz + r)) & 1ull) {
				const double *rv = S.rowv + S.chrow[c * NTG_MAX_ORDER + r] + bp;
				for (int q = 0; q < k; q++) acc += rv[q * P] * cx[q];
			}
			z[iz + r] = acc;
		}
	}
}

// column form of one collocation-matrix column (W entries, W a multiple of 4), see NtgTables::colp:
// word 0 = first breakpoint i0 of the column (its breakpoints are consecutive), then W/2 words of two 16-bit value
// indices each; a column occupies colp_words(W) words (multiple of 4: read as uint4)
template <int W>
__device__ __forceinline__ void colp_load(const unsigned int *cp, int &i0, unsigned int (&idx)[W])
{
	constexpr int WW = colp_words(W);
	unsigned int w[WW];
	const uint4 *cp4 = (const uint4 *)cp;
#pragma unroll
	for (int s4 = 0; s4 < WW / 4; s4++) { const uint4 t4 = cp4[s4]; w[4 * s4] = t4.x; w[4 * s4 + 1] = t4.y; w[4 * s4 + 2] = t4.z; w[4 * s4 + 3] = t4.w; }
	i0 = (int)w[0];
#pragma unroll
	for (int s = 0; s < W; s++) idx[s] = (w[1 + s / 2] >> (16 * (s & 1))) & 0xffffu;
}

// which coefficients a lane owns (c = tid + e*NT) and where their column-form data sit: decoded
// once per kernel, kept in registers for every evaluation of the solve
template <int EPT>
struct CoefMap {
	int o[EPT], cl[EPT];   // output and local coefficient index of slot e (c = tid + e*NT), o = -1: no coefficient
};
template <int NT, int EPT>
__device__ __forceinline__ void make_coefmap(const NtgDims &D, const Smem &S, CoefMap<EPT> &cm)
{
#pragma unroll
	for (int e = 0; e < EPT; e++) {
		const int c = threadIdx.x + e * NT;
		cm.o[e] = -1; cm.cl[e] = 0;
		if (c < D.nC) {
			int o = 0;
			while (o + 1 < D.nout && D.iC[o + 1] <= c) o++;
			cm.o[e] = o; cm.cl[e] = c - D.iC[o];
		}
	}
}

// Augmented-Lagrangian state of one problem: with mu > 0 the evaluation returns
//   F_A = F + sum_j (t_j^2 - lam_j^2)/(2 mu),  t_j = mu (v_j - clamp(v_j, bl_j, bu_j)),  v_j = c_j + lam_j/mu
// and its gradient g + J' t; t (the next multiplier estimate) is written to tnew.  mu == 0: plain F.
struct ALState {
	double mu;
	const double *lam;   // [ncnln] current multipliers (HBM)
	double *tnew;        // [ncnln] multiplier estimates of the last evaluation (HBM)
	const double *lo, *up; // this problem's rows of lowerb/upperb [nbounds]
	// QP-based SQP step (ntg_solve_opts.hessian = 3, qpdual.hpp): 1 = the evaluation returns the l1 merit function F + mu sum_j viol_j with the
	// gradient of F alone, and tnew receives the row VALUES c_j (what the QP subproblem linearises); 2 = the gradient of the Lagrangian
	// with the multipliers lam (the final pass that recovers the linear rows' multipliers), tnew receives lam.  0 = augmented Lagrangian.
	int qp;
};

// one constraint value -> multiplier estimate t; adds the row's augmented-Lagrangian term and squared scaled residual
__device__ __forceinline__ double al_row(double mu, double lamj, double l, double u, double cj, double &psi, double &rv2, int qp = 0)
{
	if (qp) {   // l1 merit: weight mu on the violation of the row's bounds; the same scaled residual as below for a feasible row with lam = 0
		const double pc = cj < l ? l : (cj > u ? u : cj), rc = (cj - pc) / (1.0 + fabs(cj));
		psi += mu * fabs(cj - pc);
		rv2 += rc * rc;
		return qp == 2 ? lamj : 0.0;
	}
	const double v = cj + lamj / mu;
	const double pj = v < l ? l : (v > u ? u : v);
	// c - p: |c - b| for an active row, min(slack, lam/mu) for a feasible one -- zero only when feasibility AND
	// complementarity hold (the measure of LANCELOT / ALGENCAN)
	const double t = mu * (v - pj), rj = (cj - pj) / (1.0 + fabs(cj));
	psi += (t - lamj) * (t + lamj) / (2.0 * mu);   // factored: no cancellation when c is tiny
	rv2 += rj * rj;
	return t;
}

// bound slot (index into lowerb/upperb) of linear row r: [lic; ltc constraint-major x breakpoint; lfc] (constraints.c:5-33)
__device__ __forceinline__ int lin_slot(const NtgDims &D, int r)
{
	if (r < D.nlic) return r;
	if (r < D.nlic + D.nltc * D.P) return D.nlic + (r - D.nlic) / D.P;
	return D.nlic + D.nltc + (r - D.nlic - D.nltc * D.P);
}

// Linear rows declared as inequalities (ntg_spec.lin_ineq): c_j = A_j x enters the augmented Lagrangian like a
// nonlinear row with a constant Jacobian.  Lanes take rows; t_j goes to LDS (tI) for the gradient pass and to HBM
// (the next multiplier estimate).  Runs between the functor pass and the quadrature / gather pass.
struct LinIneq {
	int nI; const int *irow, *rptr, *rcol, *cptr, *crow; const double *rval, *cval; double *tI;
};
template <int NT>
__device__ __forceinline__ void lin_ineq_phase(const NtgDims &D, const LinIneq &li, const double *sx, const ALState &al, double &psi, double &rv2)
{
	for (int j = threadIdx.x; j < li.nI; j += NT) {
		double cj = 0.0;
		for (int e = li.rptr[j]; e < li.rptr[j + 1]; e++) cj += li.rval[e] * sx[li.rcol[e]];
		const int slot = lin_slot(D, li.irow[j]), row = D.ncnln + j;
		const double t = al_row(al.mu, al.lam[row], al.lo[slot], al.up[slot], cj, psi, rv2);
		al.tnew[row] = t;
		li.tI[j] = t;
	}
}

// per-breakpoint cost functor pass: Z = M C, then ucf/icf/fcf -> fvals, dfz, dfi, dff in LDS
template <int FAM, int NOUT, int K, int NT, int CHM = 0>
__device__ __forceinline__ void cost_phase1(const NtgDims &D, const Smem &S, const double *sx, const ALState &al,
                                            double &psi, double &rv2, bool chm_ok = false)
{
	using Fam = Family<FAM>;
	constexpr int DM = Fam::DM, NZ = NOUT > 0 ? DM * NOUT : NTG_MAX_NZ;
	const int P = D.P, nout = NOUT > 0 ? NOUT : D.nout, nz = D.nz;
	const int tid = threadIdx.x;
	constexpr bool HASCON = Fam::NNLIC + Fam::NNLTC + Fam::NNLFC > 0;
	constexpr int NI = Fam::NNLIC > 0 ? Fam::NNLIC : 1, NTc = Fam::NNLTC > 0 ? Fam::NNLTC : 1, NF = Fam::NNLFC > 0 ? Fam::NNLFC : 1;
	const bool alon = HASCON && al.mu > 0.0;
	const int b0 = D.nlic + D.nltc + D.nlfc;   // first nonlinear slot of lowerb/upperb
	psi = 0.0; rv2 = 0.0;
	// one constraint value -> AL term, violation, multiplier estimate; returns t
	auto al_term = [&](double cj, int row, int slot) -> double {
		const double t = al_row(al.mu, al.lam[row], al.lo[slot], al.up[slot], cj, psi, rv2, al.qp);
		al.tnew[row] = al.qp == 1 ? cj : t;
		return t;
	};
	lds_sync(); // sx complete, previous users of dfz/fvals done
	if ((D.nucf || (alon && D.nnltc))) {
		const u64 zmask = alon ? (D.tcost_mask | D.tcon_mask) : D.tcost_mask;
		// the channel-mask shortcuts apply when this pass uses exactly "channels CHM of every output" (wave-uniform)
		const bool chm_now = NOUT > 0 && CHM != 0 && chm_ok && zmask == chm_full_mask(CHM, NOUT > 0 ? NOUT : 1, DM);
		for (int i = tid; i < P; i += NT) {                       // cost.c:103-109
			double z[NZ], df[NZ], f = 0.0;
			compute_z<NOUT, K, DM, CHM>(D, S, sx, i, zmask, z, chm_now);
			if (D.nucf) Fam::ucf(nout, i, z, f, df);
			else {
#pragma unroll
				for (int v = 0; v < NZ; v++) df[v] = 0.0;
			}
			S.fvals[i] = f;
			const double w = S.wts[i];
#pragma unroll
			for (int v = 0; v < NZ; v++) df[v] *= w;
			if (HASCON && alon && D.nnltc) {                      // constraints.c:148-155 folded into the same pass
				double c[NTc], t[NTc];
				double tape[Fam::TAPE];
				Fam::template nltc_val<NZ>(nout, i, z, c, tape);
#pragma unroll
				for (int j = 0; j < NTc; j++) t[j] = j < D.nnltc ? al_term(c[j], D.nnlic + j * P + i, b0 + D.nnlic + j) : 0.0;
				Fam::template nltc_vjp<NZ>(nout, nz, i, z, t, df, tape);   // df += J' t, constraint-major like the dense loop
			}
			if (chm_now) {
				// the weighted-gradient rows are (output, channel of CHM) in flag order: row = o NCH + rank(r)
				constexpr int NCH = chm_count(CHM);
#pragma unroll
				for (int o = 0; o < (NOUT > 0 ? NOUT : 1); o++)
#pragma unroll
					for (int r = 0; r < DM; r++) { if ((CHM >> r) & 1) S.dfz[(o * NCH + chm_rank(CHM, r)) * (P + 1) + i] = df[DM * o + r]; }
			} else {
#pragma unroll
				for (int v = 0; v < NZ; v++) { if (v < nz && D.tav_row[v] >= 0 && D.tav_row[v] < S.tav_rows) S.dfz[D.tav_row[v] * (P + 1) + i] = df[v]; }
			}
		}
	}
	if ((D.nicf || (alon && D.nnlic)) && tid == 0) {              // cost.c:4-36, constraints.c:88-117
		double z[NZ], df[NZ], f = 0.0;
		compute_z<NOUT, K, DM>(D, S, sx, 0, alon ? (D.icost_mask | D.icon_mask) : D.icost_mask, z);
		if (D.nicf) Fam::icf(nout, z, f, df); else for (int v = 0; v < nz; v++) df[v] = 0.0;
		if (HASCON && alon && D.nnlic) {
			double c[NI], dc[NI * NZ];
			Fam::nlicf(nout, z, c, dc);
			for (int j = 0; j < D.nnlic; j++) { const double t = al_term(c[j], j, b0 + j); for (int v = 0; v < nz; v++) df[v] += t * dc[j * nz + v]; }
		}
		for (int v = 0; v < nz; v++) S.dfi[v] = df[v];
		S.dfi[nz] = f;
	}
	if ((D.nfcf || (alon && D.nnlfc)) && tid == (NT > 64 ? 64 : 0)) {   // cost.c:141-174, constraints.c:165-195
		double z[NZ], df[NZ], f = 0.0;
		compute_z<NOUT, K, DM>(D, S, sx, P - 1, alon ? (D.fcost_mask | D.fcon_mask) : D.fcost_mask, z);
		if (D.nfcf) Fam::fcf(nout, z, f, df); else for (int v = 0; v < nz; v++) df[v] = 0.0;
		if (HASCON && alon && D.nnlfc) {
			double c[NF], dc[NF * NZ];
			Fam::nlfcf(nout, z, c, dc);
			for (int j = 0; j < D.nnlfc; j++) { const double t = al_term(c[j], D.nnlic + D.nnltc * P + j, b0 + D.nnlic + D.nnltc + j); for (int v = 0; v < nz; v++) df[v] += t * dc[j * nz + v]; }
		}
		for (int v = 0; v < nz; v++) S.dff[v] = df[v];
		S.dff[nz] = f;
	}
}

// quadrature + banded gradient assembly from the per-breakpoint values in LDS (fvals, dfz,
// dfi, dff); shared by the device-functor path and the host-callback path of ntg()
template <int NOUT, int K, int NT, int DM, int EPT, bool SHARED = false, int CHM = 0>
__device__ __forceinline__ double cost_phase2(const NtgDims &D, const Smem &S, double *sg, double *gnorm2, const CoefMap<EPT> &cm,
                                              bool hasI, bool hasF, double psi, double rv2, double *Fpure, double *rv2_out,
                                              const LinIneq *li = nullptr, bool chm_ok = false, double *defer = nullptr)
{
	const int P = D.P, nout = NOUT > 0 ? NOUT : D.nout, nz = D.nz;
	const int tid = threadIdx.x;
	lds_sync();
	// trapezoid of the running cost (integrator.c:21-24); per-interval terms across the lanes,
	// wavefront reduction
	double acc[4] = {0.0, 0.0, psi, rv2};
	if (D.nucf)
		for (int i = tid; i < P - 1; i += NT)
			acc[0] += (S.bps[i + 1] - S.bps[i]) * (S.fvals[i + 1] + S.fvals[i]) / 2;
	// gradient: the reference integrates the dense nbps x nC matrix column by column
	// (cost.c:117-134, integrator.c:44-48); here every lane owns a coefficient and takes the same
	// sum node-wise, g[c] = sum_s colv[c][s] * (w_i df_i)[coli[c][s]], over the non-zeros of column c
	// of the collocation matrix only.  The column form is s-major ([s][cl]): lanes with consecutive
	// coefficients read consecutive LDS words.
	const int W4u = (SHARED && NOUT > 0 && D.uniform) ? D.cls_W[0] : 0;
	if (NOUT > 0 && !hasI && !hasF && (W4u == 8 || W4u == 12 || W4u == 16)) {
		// (SHARED: chosen per kernel where it measured faster -- the evaluation kernel with 3 or more outputs, the solve
		// of the large configs.)  One basis class: column cl of the collocation matrix is the same for every output.  A lane takes one
		// local coefficient cl for a share of the outputs: the packed (value index, breakpoint) entries and the
		// basis values of the column are read ONCE and reused for each of its outputs; only the weighted gradient
		// rows differ.  Per coefficient the sum runs over the derivative channels, then the column entries --
		// the same order as the per-coefficient form below, so the two are bit-identical.
		auto gather = [&](auto Wtag) {
			constexpr int W = decltype(Wtag)::value;
			constexpr int NO = NOUT > 0 ? NOUT : 1;
			const int nco = D.ncoef[0];
			const int G = max(1, min(NT / nco, NO)), OPG = (NO + G - 1) / G;   // lane groups, outputs per group
			for (int i = tid; i < G * nco; i += NT) {
				const int grp = i / nco, cl = i - grp * nco, o0 = grp * OPG;
				double a[NO];
#pragma unroll
				for (int j = 0; j < NO; j++) a[j] = 0.0;
#pragma unroll
				for (int r = 0; r < DM; r++) {
					const bool fixed = CHM != 0 && chm_ok;               // wave-uniform; with CHM the channel set is a constant
					if (fixed ? !((CHM >> r) & 1) : !((D.tav_rmask >> r) & 1)) continue;
					const int chc = S.chcol[r];
					if (!fixed && chc < 0) continue;
					const double *rv = S.rowv + S.chrow[r];
					int i0; unsigned int pe[W];
					colp_load<W>(S.colp + chc + cl * colp_words(W), i0, pe);
					double vv[W];
#pragma unroll
					for (int s2 = 0; s2 < W; s2++) vv[s2] = rv[pe[s2]];
#pragma unroll
					for (int j = 0; j < NO; j++) {
						const int o = o0 + j;
						if (j >= OPG || o >= NO) break;
						const int row = fixed ? o * chm_count(CHM) + chm_rank(CHM, r) : S.tavrow[DM * o + r];
						if (row < 0) continue;
						const double *wdf = S.dfz + row * (P + 1) + i0;   // consecutive breakpoints: constant offsets
						double ww[W];
#pragma unroll
						for (int s2 = 0; s2 < W; s2++) ww[s2] = wdf[s2];
#pragma unroll
						for (int s2 = 0; s2 < W; s2++) a[j] += vv[s2] * ww[s2];
					}
				}
#pragma unroll
				for (int j = 0; j < NO; j++) {
					const int o = o0 + j;
					if (j >= OPG || o >= NO) break;
					sg[o * nco + cl] = a[j];
					acc[1] += a[j] * a[j];
				}
			}
		};
		if (W4u == 8) gather(std::integral_constant<int, 8>());
		else if (W4u == 12) gather(std::integral_constant<int, 12>());
		else gather(std::integral_constant<int, 16>());
	} else if (NOUT > 0 && !hasI && !hasF) {
		// all (breakpoint, block column) pairs of a column are read first, then all values and
		// weighted gradients, then the FMAs: two LDS round trips per coefficient, independent of W
		auto gather = [&](auto Wtag) {
			constexpr int W = decltype(Wtag)::value;   // 0: run-time width
#pragma unroll
			for (int e = 0; e < EPT; e++) {
				const int c = tid + e * NT;
				if (c >= D.nC) break;
				double dIn = 0.0;
				const int o = cm.o[e], cl = cm.cl[e];
				const int *oi = S.oinfo + o * 10;
				const int chb = oi[6], nc = oi[9], Wr = oi[8];
#pragma unroll
				for (int r = 0; r < DM; r++) {
					const bool fixed = CHM != 0 && chm_ok;               // wave-uniform; with CHM the channel set is a constant
					if (fixed ? !((CHM >> r) & 1) : !((D.tav_rmask >> r) & 1)) continue;
					const int row = fixed ? o * chm_count(CHM) + chm_rank(CHM, r) : S.tavrow[DM * o + r], chc = S.chcol[chb + r];
					if (!fixed && (row < 0 || chc < 0)) continue;
					const double *rv = S.rowv + S.chrow[chb + r]; const double *wdf = S.dfz + row * (P + 1);
					if (W > 0) {
						constexpr int WC = W > 0 ? W : 4;
						int i0; unsigned int pe[WC];
						colp_load<WC>(S.colp + chc + cl * colp_words(WC), i0, pe);
						double vv[WC], ww[WC];
#pragma unroll
						for (int s2 = 0; s2 < WC; s2++) { vv[s2] = rv[pe[s2]]; ww[s2] = wdf[i0 + s2]; }
#pragma unroll
						for (int s2 = 0; s2 < WC; s2++) dIn += vv[s2] * ww[s2];
					} else {
						const unsigned int *cp = S.colp + chc + cl * colp_words(Wr);
						const int i0 = (int)cp[0];
						for (int s2 = 0; s2 < Wr; s2++) dIn += rv[(cp[1 + s2 / 2] >> (16 * (s2 & 1))) & 0xffffu] * wdf[i0 + s2];
					}
				}
				sg[c] = dIn;
				acc[1] += dIn * dIn;
				__builtin_amdgcn_sched_barrier(0);   // keep the slots sequential: their temporaries share registers
			}
		};
		const int W4 = D.uniform ? D.cls_W[0] : 1000;
		if (W4 == 8) gather(std::integral_constant<int, 8>());
		else if (W4 == 12) gather(std::integral_constant<int, 12>());
		else if (W4 == 16) gather(std::integral_constant<int, 16>());
		else gather(std::integral_constant<int, 0>());
	} else {
		for (int c = tid; c < D.nC; c += NT) {
			int o = 0;
			while (o + 1 < nout && S.oinfo[(o + 1) * 10 + 4] <= c) o++;
			const int *oi = S.oinfo + o * 10;
			const int k = oi[0], d = oi[3], iz = oi[5], cl = c - oi[4], chb = oi[6], W4 = oi[8], nc = oi[9];
			const int *coff = S.off + oi[7];
			double dI = 0.0, dIn = 0.0, dF = 0.0;
			for (int r = 0; r < d; r++) {
				const int row = S.tavrow[iz + r], chc = S.chcol[chb + r], chr = S.chrow[chb + r];
				if (row >= 0 && chc >= 0) {
					const unsigned int *cp = S.colp + chc + cl * colp_words(W4); const double *wdf = S.dfz + row * (P + 1) + (int)cp[0];
					for (int s = 0; s < W4; s++) dIn += S.rowv[chr + ((cp[1 + s / 2] >> (16 * (s & 1))) & 0xffffu)] * wdf[s];
				}
				if (chr >= 0) {
					if (hasI && cl < k) dI += S.dfi[iz + r] * S.rowv[chr + cl * P];                // colloc.c:243-260 (block 0)
					const int ol = coff[P - 1];
					if (hasF && cl >= ol && cl < ol + k) dF += S.dff[iz + r] * S.rowv[chr + (cl - ol) * P + P - 1];   // colloc.c:287-316
				}
			}
			double g = dI + dIn + dF;                             // Vector3Add (matrix.c:177)
			if (NOUT == 0 && li) {                                // + A_I' t of the linear inequality rows (generic instances only)
				double a = 0.0;
				for (int e = li->cptr[c]; e < li->cptr[c + 1]; e++) a += li->cval[e] * li->tI[li->crow[e]];
				g += a;
			}
			sg[c] = g;
			acc[1] += g * g;
		}
	}
	if (defer) {   // the caller folds these four partial sums into a reduction of its own (sqp_kernel: with the line search's slope)
		defer[0] = acc[0]; defer[1] = acc[1]; defer[2] = acc[2]; defer[3] = acc[3];
		return 0.0;
	}
	block_sum<NT, 4>(acc, S);
	const double I = D.nicf ? S.dfi[nz] : 0.0, Ff = D.nfcf ? S.dff[nz] : 0.0;
	*gnorm2 = acc[1];
	const double Fp = I + acc[0] + Ff;                            // ntg.c:328
	if (Fpure) *Fpure = Fp;
	if (rv2_out) *rv2_out = acc[3];
	return Fp + acc[2];
}

// NPfunobj (ntg.c:274-335): F and the full gradient into LDS vector sg.  Returns F; *gnorm2
// receives |g|^2.  Every lane of the workgroup must call it (it contains barriers).
template <int FAM, int NOUT, int K, int NT, int EPT, bool SHARED = false, int CHM = 0>
__device__ __forceinline__ double eval_cost(const NtgDims &D, const Smem &S, const double *sx, double *sg, double *gnorm2,
                                            const CoefMap<EPT> &cm, const ALState &al, double *Fpure = nullptr,
                                            double *rv2_out = nullptr, unsigned long long *tk = nullptr, const LinIneq *li = nullptr,
                                            bool chm_ok = false, double *defer = nullptr)
{
	using Fam = Family<FAM>;
	constexpr bool HASCON = Fam::NNLIC + Fam::NNLTC + Fam::NNLFC > 0;
	const bool alon = HASCON && al.mu > 0.0;
	unsigned long long t0 = 0;
	if (tk) t0 = __builtin_amdgcn_s_memtime();
	double psi, rv2;
	cost_phase1<FAM, NOUT, K, NT, CHM>(D, S, sx, al, psi, rv2, chm_ok);
	const bool lin_on = NOUT == 0 && li && li->nI > 0 && al.mu > 0.0;
	if (lin_on) lin_ineq_phase<NT>(D, *li, sx, al, psi, rv2);   // sx was complete before the functor pass
	if (tk) { lds_sync(); const unsigned long long t1 = __builtin_amdgcn_s_memtime(); tk[6] += t1 - t0; t0 = t1; }
	const double F = cost_phase2<NOUT, K, NT, Fam::DM, EPT, SHARED, CHM>(D, S, sg, gnorm2, cm, D.nicf || (alon && D.nnlic), D.nfcf || (alon && D.nnlfc),
	                                          psi, rv2, Fpure, rv2_out, lin_on ? li : nullptr,
	                                          chm_ok && (alon ? (D.tcost_mask | D.tcon_mask) : D.tcost_mask) == chm_full_mask(CHM, NOUT > 0 ? NOUT : 1, Fam::DM),
	                                          defer);
	if (tk) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); tk[7] += t1 - t0; }
	return F;
}

// Per-lane decode of the banded Jacobian's row emission (eval_constraints, the pair form), once per workgroup: lane ln of a wave owns the
// pair of entries (2 e2, 2 e2 + 1) of row slot rsub; tab[i][ln] = (scratch offset or -1, table offset) of the i-th derivative channel that
// contributes to ANY entry.  Returns false (nothing written) when the pair form does not apply: generic instance, odd order, a row wider
// than a wave's 64 pairs.
template <int NOUT, int K, int DM>
__device__ __forceinline__ bool emit_decode(const NtgDims &D, const Smem &S, int *emit_tab)
{
	if (!(NOUT > 0 && K > 0 && K % 2 == 0 && DM <= NTG_MAX_ORDER) || !emit_tab) return false;
	const int s2 = D.sumk >> 1, P = D.P;
	if (s2 > 64 || (D.sumk & 1)) return false;
	if (threadIdx.x >= 64) return true;
	const int ln = threadIdx.x, rpi = 64 / s2, rsub = ln / s2, e2 = ln - rsub * s2;
	const bool live = rsub < rpi;
	const int e = 2 * e2, o = min(e / (K > 0 ? K : 1), NOUT > 0 ? NOUT - 1 : 0), q = e - o * K, cc = D.cls[o], iz = DM * o;
	unsigned um = 0;
#pragma unroll
	for (int r = 0; r < DM; r++) {
		const bool on = live && S.tcomp[iz + r] >= 0 && S.chrow[cc * NTG_MAX_ORDER + r] >= 0;
		if (__ballot(on) != 0ull) um |= 1u << r;
	}
	um = __builtin_amdgcn_readfirstlane(um);
	const int na = __popc(um);
	int2 *tab = reinterpret_cast<int2 *>(emit_tab + 68);
#pragma unroll
	for (int i = 0; i < DM; i++) {
		if (i >= na) break;   // the layout holds the listed channels only
		const int r = um ? __ffs(um) - 1 : 0;
		const bool have = um != 0;
		um &= um - 1;
		const int comp = S.tcomp[iz + r], chr = S.chrow[cc * NTG_MAX_ORDER + r];
		const bool on = live && have && comp >= 0 && chr >= 0;
		tab[i * 64 + ln] = make_int2(on ? comp * P : -1, on ? chr + q * P : 0);
	}
	emit_tab[ln] = live ? (e2 | (rsub << 16)) : -1;
	if (ln == 0) emit_tab[64] = na;
	return true;
}

// NPfuncon (ntg.c:337-371, constraints.c:36-195): residuals and banded Jacobian rows straight
// to HBM.  Row order [initial; trajectory constraint-major x breakpoint; final].
template <int FAM, int NOUT, int K, int NT>
__device__ __forceinline__ void eval_constraints(const NtgDims &D, const Smem &S, const double *sx, int mode,
                                 double *c_out, double *jband, double *cjac, double *scratch, int scratch_cap, const int *emit_tab EVCLK_ARG)
{
	using Fam = Family<FAM>;
	constexpr int DM = Fam::DM, NZ = NOUT > 0 ? DM * NOUT : NTG_MAX_NZ;
	constexpr int NI = Fam::NNLIC > 0 ? Fam::NNLIC : 1, NTc = Fam::NNLTC > 0 ? Fam::NNLTC : 1, NF = Fam::NNLFC > 0 ? Fam::NNLFC : 1;
	const int P = D.P, nout = NOUT > 0 ? NOUT : D.nout, nz = D.nz, tid = threadIdx.x;
	auto emit_row = [&](int row, int bp, const double *dcrow, double *jb) {
		for (int o = 0; o < nout; o++) {
			const int k = D.order[o], cc = D.cls[o], d = D.d[o];
			const int col0 = D.iC[o] + S.off[cc * P + bp];
			for (int q = 0; q < k; q++) {
				double a = 0.0;
				for (int r = 0; r < d; r++) {
					const int chr = S.chrow[cc * NTG_MAX_ORDER + r];
					if (chr >= 0) a += dcrow[D.iz[o] + r] * S.rowv[chr + q * P + bp];
				}
				if (jb) jb[(size_t)row * D.sumk + D.koff[o] + q] = a;
				if (cjac) cjac[(size_t)(col0 + q) * D.ncnln + row] = a;
			}
		}
	};
	if (Fam::NNLIC > 0 && D.nnlic && tid == 0) {
		double z[NZ], c[NI], dc[NI * NZ];
		compute_z<NOUT, K, DM>(D, S, sx, 0, D.icon_mask, z);
		Fam::nlicf(nout, z, c, dc);
		for (int j = 0; j < D.nnlic; j++) {
			if (c_out && mode != 1) c_out[j] = c[j];
			if (mode != 0) emit_row(j, 0, dc + j * nz, jband);
		}
	}
	// Trajectory rows.  The banded Jacobian of one constraint is P consecutive rows of sum(k) entries: a contiguous block
	// of memory.  With one lane per breakpoint each store instruction would scatter 8-byte words sum(k)*8 bytes apart;
	// instead the lanes first leave the functor's derivatives dc[j][v] (v in the trajectory-constraint active variables --
	// a constraint depends on nothing else, ntg.h:57-60) in LDS, by breakpoint, and then walk the block in memory order:
	// consecutive lanes, consecutive words.  The LDS area is the weighted-gradient rows of the cost pass, which runs after
	// this; constraints are processed in chunks that fit it.
	const int ncomp = __popcll(D.tcon_mask);
	const bool coalesced = Fam::NNLTC > 0 && D.nnltc && jband && mode != 0 && ncomp > 0 && ncomp * P <= scratch_cap;
	constexpr u64 TV = Fam::TCON_VARS;
	const bool sparse_vars = TV != ~0ull && (D.tcon_mask & ~TV) == 0ull;   // every flagged entry is one the family's rows can touch
	if (Fam::NNLTC > 0 && D.nnltc) {
		const int chunk = coalesced ? max(1, min(D.nnltc, scratch_cap / (ncomp * P))) : D.nnltc;
		// with one breakpoint per lane (P <= NT) the flag, the values and the functor's tape are computed once and kept in
		// registers across the chunks; otherwise they are recomputed per chunk
		const bool keep = P <= NT;
		double z[NZ], c[NTc], tape[Fam::TAPE];
		for (int j0 = 0; j0 < D.nnltc; j0 += chunk) {
			const int jn = min(chunk, D.nnltc - j0);
			for (int i = tid; i < P; i += NT) {
				if (!keep || j0 == 0) {
					compute_z<NOUT, K, DM>(D, S, sx, i, D.tcon_mask, z);
					Fam::template nltc_val<NZ>(nout, i, z, c, tape);
					EVCLK(6);
				}
				for (int j = j0; j < j0 + jn; j++) {
					const int row = D.nnlic + j * P + i;              // constraints.c:139,153
					if (c_out && mode != 1) c_out[row] = c[j];
					if (mode == 0) continue;
					// row j of the functor's Jacobian = J' e_j: one (sparse, for the large families) vjp instead of a dense
					// [ncon][nz] array in registers
					double t[NTc], dcr[NZ];
#pragma unroll
					for (int jj = 0; jj < NTc; jj++) t[jj] = jj == j ? 1.0 : 0.0;
#pragma unroll
					for (int v = 0; v < NZ; v++) dcr[v] = 0.0;
					Fam::template nltc_vjp<NZ>(nout, nz, i, z, t, dcr, tape);
					if (cjac || !coalesced) emit_row(row, i, dcr, coalesced ? nullptr : jband);
					if (coalesced && sparse_vars) {
						// the family says which flag entries its rows can touch: only those are copied (12 of config E's 36, 6 of D's 20;
						// the walk over all NZ entries with a wave-uniform branch each was 5 k cycles per constraint)
#pragma unroll
						for (int v = 0; v < NZ; v++)
							if (((TV >> v) & 1ull) && ((D.tcon_mask >> v) & 1ull))   // compact index: scalar bit count, nothing kept in registers
								scratch[((j - j0) * ncomp + __popcll(D.tcon_mask & ((1ull << v) - 1ull))) * P + i] = dcr[v];
					} else if (coalesced) {
						int comp = 0;
#pragma unroll
						for (int v = 0; v < NZ; v++)
							if (v < nz && ((D.tcon_mask >> v) & 1ull)) { scratch[((j - j0) * ncomp + comp) * P + i] = dcr[v]; comp++; }
					}
				}
			}
			if (coalesced) {
				lds_sync();
				EVCLK(1);
				// a wave takes whole rows (stride NW); its lanes are the entries of the row, 64 at a time: a store
				// instruction writes 64 consecutive words, and what an entry needs to know about itself -- output, block
				// column, which derivative channels contribute -- is decoded once per lane, outside the loop over rows
				constexpr int NW = NT / 64, RMAX = NOUT > 0 ? DM : NTG_MAX_ORDER;
				const int sumk = D.sumk, nrows = jn * P, wv = tid >> 6, ln = tid & 63;
				double *dst = jband + (size_t)(D.nnlic + j0 * P) * sumk;
				// entries [e0, e0 + width) of 64 / width rows per step: a row narrower than a wave shares it with its neighbours
				// (sum(k) = 32: two rows per store instruction; the 8 left-over entries of a 72-entry row: eight rows)
				auto emit_pass = [&](int e0, int width) {
					const int rpi = 64 / width, rsub = ln / width, e = e0 + ln % width;
					if (rsub >= rpi) return;
					int o = 0;
					if (NOUT > 0 && K > 0) o = e / (K > 0 ? K : 1);    // one order for every output
					else while (o + 1 < nout && D.koff[o + 1] <= e) o++;
					const int q = e - (NOUT > 0 && K > 0 ? o * K : D.koff[o]), cc = D.cls[o], d = NOUT > 0 ? DM : D.d[o];
					const int iz = NOUT > 0 ? DM * o : D.iz[o];
					int cof[RMAX], rof[RMAX];                          // scratch / rowv offsets of the contributing channels, -1: none
#pragma unroll
					for (int r = 0; r < RMAX; r++) {
						const int comp = r < d ? S.tcomp[iz + r] : -1, chr = r < d ? S.chrow[cc * NTG_MAX_ORDER + r] : -1;
						const bool on = comp >= 0 && chr >= 0;
						cof[r] = on ? comp * P : -1; rof[r] = on ? chr + q * P : 0;
					}
					const int step = NW * rpi;
					int row = wv * rpi + rsub, jc = row / P, bp = row - jc * P;
					for (; row < nrows; row += step) {
						double a = 0.0;
#pragma unroll
						for (int r = 0; r < RMAX; r++)
							if (cof[r] >= 0) a += scratch[jc * ncomp * P + cof[r] + bp] * S.rowv[rof[r] + bp];
						dst[(size_t)row * sumk + e] = a;
						bp += step;
						while (bp >= P) { bp -= P; jc++; }
					}
				};
				// The tuned instances (one even order for every output, 16-byte aligned rows) emit PAIRS of adjacent entries: a lane
				// owns block columns (q, q + 1) of one output -- the functor's derivative is shared by the pair, the store is 16 bytes
				// per lane (1 KB per wave instruction).  The phase is bound by instruction issue (one or two waves per SIMD: ~8 cycles
				// per instruction, measured 760 cycles per 128 stored words before), so what counts is instructions per stored word:
				//  * the derivative channels that contribute to ANY entry of the wave (a wave-uniform set: the trajectory-constraint
				//    active variables name few derivatives -- 2 of config D's 6 channels, 1 of config E's 4) are compacted into a
				//    list once; the row loop is instantiated for the list's length NA and touches nothing else;
				//  * every load of a trip is unconditional (a lane whose entry lacks a listed channel multiplies by zero) and all of
				//    them -- UR rows x NA channels x 3 -- are in flight before the first is used: the empty asm makes every operand
				//    live at one point; under this kernel's register pressure the scheduler otherwise pairs each read with its wait;
				//  * the breakpoint index wraps branch-free.
				//  * what a lane needs to know about its pair is the same for every problem: it is decoded once per workgroup into LDS
				//    (emit_decode, called by eval_kernel before the problem loop) and costs 1 + NA reads here; decoding it per problem
				//    (two integer divisions, 4 DM table lookups, DM ballots) was HALF of the phase: 6.5 k of 13.5 k cycles, config D.
				auto emit_pairs = [&]() {
					const int s2 = sumk >> 1, rpi = 64 / s2;
					const int pk = emit_tab[ln];
					if (pk < 0) return;
					const int e2 = pk & 0xffff, rsub = pk >> 16;
					const int2 *tab = reinterpret_cast<const int2 *>(emit_tab + 68);
					const int na = __builtin_amdgcn_readfirstlane(emit_tab[64]);
					const int step = NW * rpi, cs = ncomp * P;
					double2 *dst2 = reinterpret_cast<double2 *>(dst);
					auto rows = [&](auto na_) {
						constexpr int NA = decltype(na_)::value, UR = NA == 1 ? 4 : (NA == 2 ? 4 : (NA <= 4 ? 2 : 1));
						int cofA[NA > 0 ? NA : 1], rofA[NA > 0 ? NA : 1]; double useA[NA > 0 ? NA : 1];
#pragma unroll
						for (int i = 0; i < NA; i++) {
							const int2 t = tab[i * 64 + ln];
							cofA[i] = max(t.x, 0); rofA[i] = t.y; useA[i] = t.x >= 0 ? 1.0 : 0.0;
						}
						int row = wv * rpi + rsub, jc = 0, bp = row;
						while (bp >= P) { bp -= P; jc++; }
						EVCLK(4);
						if (step <= P) {
							for (; row + (UR - 1) * step < nrows; row += UR * step) {
								int jcu[UR], bpu[UR];
								jcu[0] = jc; bpu[0] = bp;
#pragma unroll
								for (int u = 1; u < UR; u++) {
									const int bb = bpu[u - 1] + step; const bool w = bb >= P;
									bpu[u] = bb - (w ? P : 0); jcu[u] = jcu[u - 1] + (w ? 1 : 0);
								}
								double sa[UR][NA > 0 ? NA : 1], r0[UR][NA > 0 ? NA : 1], r1[UR][NA > 0 ? NA : 1];
#pragma unroll
								for (int u = 0; u < UR; u++)
#pragma unroll
									for (int i = 0; i < NA; i++) {
										sa[u][i] = scratch[jcu[u] * cs + cofA[i] + bpu[u]];
										r0[u][i] = S.rowv[rofA[i] + bpu[u]];
										r1[u][i] = S.rowv[rofA[i] + P + bpu[u]];
									}
#pragma unroll
								for (int u = 0; u < UR; u++)
#pragma unroll
									for (int i = 0; i < NA; i++) asm volatile("" : "+v"(sa[u][i]), "+v"(r0[u][i]), "+v"(r1[u][i]));
#pragma unroll
								for (int u = 0; u < UR; u++) {
									double a0 = 0.0, a1 = 0.0;
#pragma unroll
									for (int i = 0; i < NA; i++) { const double t = sa[u][i] * useA[i]; a0 += t * r0[u][i]; a1 += t * r1[u][i]; }
									dst2[(size_t)(row + u * step) * s2 + e2] = make_double2(a0, a1);
								}
								{ const int bb = bpu[UR - 1] + step; const bool w = bb >= P; bp = bb - (w ? P : 0); jc = jcu[UR - 1] + (w ? 1 : 0); }
							}
						}
						for (; row < nrows; row += step) {
							double a0 = 0.0, a1 = 0.0;
#pragma unroll
							for (int i = 0; i < NA; i++) {
								const double t = scratch[jc * cs + cofA[i] + bp] * useA[i];
								a0 += t * S.rowv[rofA[i] + bp]; a1 += t * S.rowv[rofA[i] + P + bp];
							}
							dst2[(size_t)row * s2 + e2] = make_double2(a0, a1);
							bp += step;
							while (bp >= P) { bp -= P; jc++; }
						}
					};
					if (na == 0) rows(std::integral_constant<int, 0>{});
					else if (na == 1) rows(std::integral_constant<int, 1>{});
					else if (na == 2) rows(std::integral_constant<int, 2>{});
					else if (na == 3) rows(std::integral_constant<int, 3>{});
					else if (na == 4) rows(std::integral_constant<int, 4>{});
					else if (na == 5) rows(std::integral_constant<int, (RMAX >= 5 ? 5 : 0)>{});
					else if (na == 6) rows(std::integral_constant<int, (RMAX >= 6 ? 6 : 0)>{});
					else if (na == 7) rows(std::integral_constant<int, (RMAX >= 7 ? 7 : 0)>{});
					else rows(std::integral_constant<int, (RMAX >= 8 ? 8 : 0)>{});
					EVCLK(5);
				};
				if (emit_tab && ((uintptr_t)dst & 15) == 0) emit_pairs();
				else
				for (int e0 = 0; e0 < sumk; e0 += 64) emit_pass(e0, min(64, sumk - e0));
				lds_sync();
				EVCLK(2);
			}
		}
	}
	if (Fam::NNLFC > 0 && D.nnlfc && tid == (NT > 64 ? 64 : 0)) {
		double z[NZ], c[NF], dc[NF * NZ];
		compute_z<NOUT, K, DM>(D, S, sx, P - 1, D.fcon_mask, z);
		Fam::nlfcf(nout, z, c, dc);
		for (int j = 0; j < D.nnlfc; j++) {
			const int row = D.nnlic + D.nnltc * P + j;
			if (c_out && mode != 1) c_out[row] = c[j];
			if (mode != 0) emit_row(row, P - 1, dc + j * nz, jband);
		}
	}
}

// Persistent workgroups stride over the batch; tables are staged once per workgroup.
// BANDONLY: the instance for the common full request (mode 2: values, gradient, residuals, banded Jacobian rows; no dense Jacobian) with
// that request as a compile-time fact -- the dense-row path, the per-entry scatter and the mode tests are not in its code at all (the
// general instance of config D is 26.6 k instructions with 3.1 k scalar-spill instructions among them).
template <int FAM, int NOUT, int K, int NT, int EPT, int CHM, bool BANDONLY = false>
__global__ void __launch_bounds__(NT, NTG_EVAL_WAVES)
eval_kernel(NtgDims D, NtgTables T, SmemLayout L, int batch, int mode, const double *__restrict__ x,
            double *__restrict__ f, double *__restrict__ g, double *__restrict__ c,
            double *__restrict__ jband, double *__restrict__ cjac)
{
	if (BANDONLY) { mode = 2; cjac = nullptr; }
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	EVCLK_DECL
	Smem S(smem_raw, L, D, T);
	const bool pp = T.pp_rowv != 0;   // per-problem grids: the value tables are staged again for every problem
	if (!pp) stage_tables<NT>(D, T, S, smem_raw, L);
	const int *emit_tab = nullptr;   // the row emission's decode depends on the plan only (not on the grid's values): once per workgroup
	if (L.emit >= 0 && jband && mode != 0) {
		if (pp) stage_tables<NT>(D, T, S, smem_raw, L, blockIdx.x < batch ? blockIdx.x : 0);
		lds_sync();
		if (emit_decode<NOUT, K, Family<FAM>::DM>(D, S, (int *)(smem_raw + L.emit))) emit_tab = (const int *)(smem_raw + L.emit);
	}
	double *sg = S.x;   // the gradient is assembled (owner lanes, after the last read of x) into the x buffer
	lds_sync();
	CoefMap<EPT> cm;
	if (NOUT > 0) make_coefmap<NT, EPT>(D, S, cm);   // the host only picks a NOUT > 0 instance when nC <= EPT NT
	// software pipeline over problems: the coefficient vector of the NEXT problem is already in
	// flight (registers) while the current one is evaluated
	constexpr int XE = EPT;
	const bool xreg = D.nC <= XE * NT;
	double xn[XE];
	if (xreg && (int)blockIdx.x < batch) {
#pragma unroll
		for (int e = 0; e < XE; e++) { const int i = threadIdx.x + e * NT; xn[e] = i < D.nC ? x[(size_t)blockIdx.x * D.nC + i] : 0.0; }
	}
	EVCLK0();
	for (int b = blockIdx.x; b < batch; b += gridDim.x) {
		lds_sync();
		if (pp) { stage_tables<NT>(D, T, S, smem_raw, L, b); lds_sync(); }
		if (xreg) {
#pragma unroll
			for (int e = 0; e < XE; e++) { const int i = threadIdx.x + e * NT; if (i < D.nC) S.x[i] = xn[e]; }
			const int bn = b + gridDim.x;
			if (bn < batch) {
#pragma unroll
				for (int e = 0; e < XE; e++) { const int i = threadIdx.x + e * NT; xn[e] = i < D.nC ? x[(size_t)bn * D.nC + i] : 0.0; }
			}
		} else {
			for (int i = threadIdx.x; i < D.nC; i += NT) S.x[i] = x[(size_t)b * D.nC + i];
		}
		const int P1 = D.P + 1;
		EVCLK(0);
		if (D.ncnln && (c || jband || cjac)) {   // constraints first: they still need x, the cost pass overwrites it with g
			lds_sync();
			eval_constraints<FAM, NOUT, K, NT>(D, S, S.x, mode, c ? c + (size_t)b * D.ncnln : nullptr,
			                                jband ? jband + (size_t)b * D.ncnln * D.sumk : nullptr,
			                                cjac ? cjac + (size_t)b * D.ncnln * D.nC : nullptr,
			                                S.dfz, L.dfz_rows * (P1) + ntg_dfz_tail(D), emit_tab EVCLK_PASS);
		}
		double gn2;
		const double F = eval_cost<FAM, NOUT, K, NT, EPT, (NOUT >= 3), CHM>(D, S, S.x, sg, &gn2, cm, ALState{0.0, nullptr, nullptr, nullptr, nullptr},
		                                                                   nullptr, nullptr, EVCLK_TK, nullptr, CHM != 0);
		if (f && mode != 1 && threadIdx.x == 0) f[b] = F;
		if (g && mode != 0)
			for (int i = threadIdx.x; i < D.nC; i += NT) g[(size_t)b * D.nC + i] = sg[i];
		EVCLK(3);
	}
#ifdef NTG_EVAL_CLOCK
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		printf("evclk: stage-x %llu functor %llu emission %llu cost+g %llu | emission: prologue %llu rows %llu | functor: z+val %llu | cost: phase1 %llu phase2 %llu\n", ntg_evclk_s[0], ntg_evclk_s[1], ntg_evclk_s[2], ntg_evclk_s[3], ntg_evclk_s[4], ntg_evclk_s[5], ntg_evclk_s[6], ntg_evclk_s[10], ntg_evclk_s[11]);
	}
#endif
}

// ------------------------------------------------------------------------------------------
// SQP pieces
// ------------------------------------------------------------------------------------------
// gp = g - A'(AA')^-1 A g  (projection onto null(A)); S.lam receives the multipliers estimate.
// A is kept sparse (CSR for A g, CSC for A' lam); (AA')^-1 is a small dense matrix.
// With dneg != nullptr the pass also accumulates this lane's share of gp . (-dneg) into *slope (the directional
// derivative the line search needs) and leaves the closing barrier to the reduction that follows.
template <int NT, bool BIG>
__device__ __forceinline__ void project(const NtgDims &D, const Smem &S, const double *sg, double *sgp, double *tmp /* LDS [nclin] */,
                                        const double *dneg = nullptr, double *slope = nullptr)
{
	const int m = D.mE, tid = threadIdx.x;
	if (BIG) __syncthreads();   // g lives in HBM/L2 and is read across lanes
	else lds_sync();
	if (m == 0) {
		for (int c = tid; c < D.nC; c += NT) { const double gq = sg[c]; sgp[c] = gq; if (dneg) *slope += gq * (-dneg[c]); }
		if (!dneg) lds_sync();
		return;
	}
	if (D.q_use) {
		// gp = g - Q g with Q = A'(AA')^-1 A stored as ELL over its non-zero rows: one pass, no
		// intermediate barrier; the padded entries (value 0, column 0) keep every load unconditional
		const int w = D.q_w;
		for_vec<NT>(D.nC, [&](int c) {
			const int t = S.q_idx[c];
			double s = 0.0;
			if (t >= 0) {
#pragma unroll 4
				for (int e = 0; e < w; e++) s += S.q_val[t * w + e] * sg[S.q_col[t * w + e]];
			}
			const double gq = sg[c] - s;
			sgp[c] = gq;
			if (dneg) *slope += gq * (-dneg[c]);
		});
		if (!dneg) lds_sync();
		return;
	}
	for (int r = tid; r < m; r += NT) {
		double a = 0.0;
		for (int e = S.csr_ptr[r]; e < S.csr_ptr[r + 1]; e++) a += S.csr_val[e] * sg[S.csr_col[e]];
		tmp[r] = a;
	}
	lds_sync();
	for (int r = tid; r < m; r += NT) {
		double a = 0.0;
		for (int e = S.sinv_ptr[r]; e < S.sinv_ptr[r + 1]; e++) a += S.sinv_val[e] * tmp[S.sinv_col[e]];
		S.lam[r] = a;
	}
	lds_sync();
	for (int c = tid; c < D.nC; c += NT) {
		double s = 0.0;
		for (int e = S.csc_ptr[c]; e < S.csc_ptr[c + 1]; e++) s += S.csc_val[e] * S.lam[S.csc_row[e]];
		const double gq = sg[c] - s;
		sgp[c] = gq;
		if (dneg) *slope += gq * (-dneg[c]);
	}
	if (!dneg) lds_sync();
}

typedef const __attribute__((address_space(3))) double *lds_cdp;
typedef const __attribute__((address_space(3))) int *lds_cip;
typedef __attribute__((address_space(3))) double *lds_dp;
typedef const __attribute__((address_space(1))) double *glb_cdp;
typedef __attribute__((address_space(1))) double *glb_dp;
typedef const __attribute__((address_space(1))) unsigned short *glb_cusp;
// a coefficient vector of the solve lives in LDS, or (BIG layouts) in the per-problem HBM workspace: the out-of-line routines below take
// it with its address space in the type (template flag G = "global"), so that they contain no FLAT instruction (ntg_amd/call_audit.py R2)
template <bool G> struct VecPtr { typedef lds_dp rw; typedef lds_cdp ro; };
template <> struct VecPtr<true> { typedef glb_dp rw; typedef glb_cdp ro; };
// out = W0 v for the collocation preconditioner (ELL, rows streamed from L2).  Kept out of line:
// it runs a handful of times per solve and must not add to the register pressure of the main loop.
template <int NT, bool VG, bool OG>
__device__ __attribute__((noinline)) void apply_n0(int n, int w, glb_cdp n0, glb_cusp n0c, typename VecPtr<VG>::ro v, typename VecPtr<OG>::rw out)
{
	for (int c = threadIdx.x; c < n; c += NT) {
		double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
		int s = 0;
		for (; s + 4 <= w; s += 4) {   // four independent chains: the L2 loads of a row overlap
			a0 += n0[(size_t)(s + 0) * n + c] * v[n0c[(size_t)(s + 0) * n + c]];
			a1 += n0[(size_t)(s + 1) * n + c] * v[n0c[(size_t)(s + 1) * n + c]];
			a2 += n0[(size_t)(s + 2) * n + c] * v[n0c[(size_t)(s + 2) * n + c]];
			a3 += n0[(size_t)(s + 3) * n + c] * v[n0c[(size_t)(s + 3) * n + c]];
		}
		for (; s < w; s++) a0 += n0[(size_t)s * n + c] * v[n0c[(size_t)s * n + c]];
		out[c] = (a0 + a1) + (a2 + a3);
	}
}

// out = W0 v when W0 is block diagonal by output with equally sized dense blocks, few of them distinct (one basis
// class, the same constraint pattern at the ends; outputs with the same cost derivative share a block): the one
// GEMM-shaped piece of the path,  Out(nco x nout) = sum_b W_b (nco x nco) V_b (nco x nout),  V_b = the columns of the
// outputs that use block b.  It runs on the matrix cores: v_mfma_f64_16x16x4_f64 tiles, a wave owns row tiles of 16
// coefficients and sweeps k in steps of 4.  Register layout of the instruction (probed on gfx950): lane l supplies
// A[i = l%16][k = l/16] and B[k = l/16][j = l%16] and receives D[i = 4r + l/16][j = l%16], r = 0..3.
//   A = W_b: s-major in HBM/L2 ([s][row], symmetric), so the 16 lanes of a k read 16 consecutive words;
//   B = V:   staged in LDS laid out like v ([output][coefficient]); column j >= nout or of another block is 0.
// Rows of W_b beyond nco are zero padding (spad rows, a multiple of 16); `out` may alias `stage`: the result tiles
// stay in registers until every wave has finished reading.
typedef double ntg_d4 __attribute__((ext_vector_type(4)));
template <int NT, bool VG, bool OG>
__device__ __attribute__((noinline)) void apply_n0_block(int nC, int nout, int nco, int spad, int nblk, glb_cdp wb,
                                                         lds_cip oinfo, typename VecPtr<VG>::ro v, lds_dp stage_w, typename VecPtr<OG>::rw out)
{
	constexpr int NW = NT / 64, TMAX = 4, U = 4;   // nco <= NT (checked by the caller): at most NT/16 = 4 NW row tiles
	lds_sync();   // previous readers of stage are done
	for (int c = threadIdx.x; c < nC; c += NT) stage_w[c] = v[c];   // owner lanes: v may live in HBM
	lds_sync();
	lds_cdp stage = (lds_cdp)stage_w;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
	const int ntile = (nco + 15) >> 4;
	const int myblk = li < nout ? oinfo[li * 10 + 2] : -1;
	lds_cdp bcol = stage + (li < nout ? li : 0) * nco;
	ntg_d4 acc[TMAX];
#pragma unroll
	for (int t = 0; t < TMAX; t++) acc[t] = ntg_d4{0.0, 0.0, 0.0, 0.0};
	for (int b = 0; b < nblk; b++) {
		glb_cdp wbb = wb + (size_t)b * spad * nco + li;
		const bool mine = myblk == b;
		for (int k0 = 0; k0 < spad; k0 += 4 * U) {   // spad is a multiple of 16 = 4 U
			// U k-steps of loads first (L2 latency overlaps), then their matrix instructions
			double av[U][TMAX], bv[U];
#pragma unroll
			for (int u = 0; u < U; u++) {
				const int k = k0 + 4 * u + lk;
				bv[u] = mine ? bcol[min(k, nco - 1)] : 0.0;   // k >= nco: W's row is zero padding
				glb_cdp wk = wbb + (size_t)k * nco;
#pragma unroll
				for (int t = 0; t < TMAX; t++) av[u][t] = (wave + t * NW < ntile) ? wk[(wave + t * NW) * 16] : 0.0;
			}
#pragma unroll
			for (int u = 0; u < U; u++)
#pragma unroll
				for (int t = 0; t < TMAX; t++)
					if (wave + t * NW < ntile) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][t], bv[u], acc[t], 0, 0, 0);
		}
	}
	lds_sync();   // every read of stage is done: out may be the same buffer
	if (li < nout) {
#pragma unroll
		for (int t = 0; t < TMAX; t++) {
			const int tile = wave + t * NW;
			if (tile < ntile) {
#pragma unroll
				for (int r = 0; r < 4; r++) { const int row = tile * 16 + 4 * r + lk; if (row < nco) out[li * nco + row] = acc[t][r]; }
			}
		}
	}
}

// out = W0 v : identity on null(A) (cold start) or the collocation preconditioner.  `stage` is an LDS buffer of
// nC doubles that is free at every call site (the trial point: it is rebuilt from x and d afterwards).
// HESS = false: the instance is only ever launched with the identity cold start (NPSOL's mode, the fixed-work
// benchmark): no preconditioner code, in particular no out-of-line calls, in its main loop.
// OLDS: `out` is an LDS vector even in the BIG layout (t = W gp+ lives in the trial-point buffer)
template <int NT, bool BIG, bool HESS, bool OLDS = false>
__device__ __forceinline__ void apply_w0(const NtgDims &D, const NtgTables &T, int hessian, const double *v, double *out, double *stage, const int *oinfo)
{
	constexpr bool VG = BIG, OG = BIG && !OLDS;
	if (!HESS) {
		lds_sync();
		for_vec<NT>(D.nC, [&](int c) { out[c] = v[c]; });
		lds_sync();
		return;
	}
	if (hessian == 1 && T.n0b && T.n0b_n <= NT) {
		apply_n0_block<NT, VG, OG>(D.nC, D.nout, T.n0b_n, T.n0b_sp, T.n0b_nblk, (glb_cdp)T.n0b, (lds_cip)oinfo, (typename VecPtr<VG>::ro)v, (lds_dp)stage, (typename VecPtr<OG>::rw)out);
		if (BIG) __syncthreads();   // out may live in HBM and was written by row, not by owner lane
		else lds_sync();
		return;
	}
	if (BIG) __syncthreads();   // v lives in HBM/L2 and apply_n0 reads it across lanes
	else lds_sync();
	if (hessian == 1 && T.n0) apply_n0<NT, VG, OG>(D.nC, T.n0_w, (glb_cdp)T.n0, (glb_cusp)T.n0c, (typename VecPtr<VG>::ro)v, (typename VecPtr<OG>::rw)out);
	else for_vec<NT>(D.nC, [&](int c) { out[c] = v[c]; });
	lds_sync();
}

// t += (sum of the stored rank-2 BFGS terms) v.  Pairs are streamed from HBM/L2 once, G at a
// time: 2G partial dots per lane, ONE workgroup reduction, then the axpys from the registers
// that still hold the pair elements.
template <int NT, int EPT = 3, int G = NTG_HIST_G>
__device__ __forceinline__ void apply_history(const NtgDims &D, const Smem &S, const double *hist, int npairs,
                              const double *v, double *t, const double *hrc = nullptr)
{
	const int n = D.nC, tid = threadIdx.x;
#if NTG_HIST_PIPE > 0
	if (EPT <= 3 && NT <= 128 && n <= EPT * NT && hrc) {   // (one workgroup per CU, NT >= 256: fewer, larger rounds are faster -- config D 162 vs 190 ms)
		constexpr int PG = NTG_HIST_PIPE;
		// Two register buffers of PG pairs: the loads of round r+1 are in flight while round r is reduced.  Every load is
		// unconditional (pair index clamped to the newest pair, column clamped to n-1) so that the waits are counted
		// (vmcnt(loads of the younger round)) instead of vmcnt(0); rounds past the end get rho = c2 = 0.  The pair scalars
		// come from LDS: nothing but the pair elements in the register buffers.
		if (npairs == 0) { lds_sync(); return; }
		double vv[EPT], tt[EPT];
		int cc[EPT];
#pragma unroll
		for (int e = 0; e < EPT; e++) { const int c = tid + e * NT; cc[e] = min(c, n - 1); vv[e] = c < n ? v[c] : 0.0; tt[e] = c < n ? t[c] : 0.0; }
		{
			// odd memories are swept newest-first, even ones oldest-first: the pairs a sweep ends on are the ones the next
			// sweep (one pair longer) starts on, while they are still in L2 / the memory-side cache
			const bool rev = NTG_HIST_ZIGZAG && (npairs & 1);
			auto pidx = [&](int i) { const int j = min(i, npairs - 1); return rev ? npairs - 1 - j : j; };
			double hsA[PG][EPT], huA[PG][EPT], hsB[PG][EPT], huB[PG][EPT];
			auto load = [&](int base, double (&hs)[PG][EPT], double (&hu)[PG][EPT]) {
#pragma unroll
				for (int g = 0; g < PG; g++) {
					const double *h = hist + (size_t)pidx(base + g) * (2 * n + 2);
#pragma unroll
					for (int e = 0; e < EPT; e++) { hs[g][e] = h[cc[e]]; hu[g][e] = h[n + cc[e]]; }
				}
			};
			auto consume = [&](int base, double (&hs)[PG][EPT], double (&hu)[PG][EPT]) {
				double acc[2 * PG];
#pragma unroll
				for (int g = 0; g < PG; g++) {
					acc[2 * g] = 0.0; acc[2 * g + 1] = 0.0;
#pragma unroll
					for (int e = 0; e < EPT; e++) { acc[2 * g] += hs[g][e] * vv[e]; acc[2 * g + 1] += hu[g][e] * vv[e]; }
				}
				block_sum<NT, 2 * PG>(acc, S);
#pragma unroll
				for (int g = 0; g < PG; g++) {
					const bool on = base + g < npairs;
					const int pi = pidx(base + g);
					const double live = on ? 1.0 : 0.0, rho = hrc[2 * pi] * live, c2 = hrc[2 * pi + 1] * live;   // unconditional LDS reads
#pragma unroll
					for (int e = 0; e < EPT; e++)
						tt[e] += -rho * (hs[g][e] * acc[2 * g + 1] + hu[g][e] * acc[2 * g]) + c2 * hs[g][e] * acc[2 * g];
				}
			};
			load(0, hsA, huA);
			for (int base = 0; base < npairs; base += 2 * PG) {
				load(base + PG, hsB, huB);
				consume(base, hsA, huA);
				if (base + PG >= npairs) break;
				load(base + 2 * PG, hsA, huA);
				consume(base + PG, hsB, huB);
			}
#pragma unroll
			for (int e = 0; e < EPT; e++) { const int c = tid + e * NT; if (c < n) t[c] = tt[e]; }
			lds_sync();
			return;
		}
	}
#endif
	if (n <= EPT * NT) {
		double vv[EPT], tt[EPT];
#pragma unroll
		for (int e = 0; e < EPT; e++) { const int c = tid + e * NT; vv[e] = c < n ? v[c] : 0.0; tt[e] = c < n ? t[c] : 0.0; }
		for (int base = 0; base < npairs; base += G) {
			const int cnt = min(G, npairs - base);
			double hs[G][EPT], hu[G][EPT], acc[2 * G];
#pragma unroll
			for (int g = 0; g < G; g++) {
				acc[2 * g] = 0.0; acc[2 * g + 1] = 0.0;
				const double *h = hist + (size_t)(base + g) * (2 * n + 2);
#pragma unroll
				for (int e = 0; e < EPT; e++) {
					const int c = tid + e * NT;
					const bool on = g < cnt && c < n;
					hs[g][e] = on ? h[c] : 0.0;
					hu[g][e] = on ? h[n + c] : 0.0;
				}
			}
#pragma unroll
			for (int g = 0; g < G; g++)
#pragma unroll
				for (int e = 0; e < EPT; e++) { acc[2 * g] += hs[g][e] * vv[e]; acc[2 * g + 1] += hu[g][e] * vv[e]; }
			block_sum<NT, 2 * G>(acc, S);
#pragma unroll
			for (int g = 0; g < G; g++) {
				if (g < cnt) {
					const double *hh = hrc ? hrc + 2 * (base + g) : hist + (size_t)(base + g) * (2 * n + 2) + 2 * n;
					const double rho = hh[0], c2 = hh[1];
#pragma unroll
					for (int e = 0; e < EPT; e++)
						tt[e] += -rho * (hs[g][e] * acc[2 * g + 1] + hu[g][e] * acc[2 * g]) + c2 * hs[g][e] * acc[2 * g];
				}
			}
		}
#pragma unroll
		for (int e = 0; e < EPT; e++) { const int c = tid + e * NT; if (c < n) t[c] = tt[e]; }
		lds_sync();
		return;
	}
	for (int base = 0; base < npairs; base += 4) {
		const int cnt = min(4, npairs - base);
		double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		for (int c = tid; c < n; c += NT) {
			const double vv = v[c];
#pragma unroll
			for (int g = 0; g < 4; g++) {
				if (g < cnt) {
					const double *h = hist + (size_t)(base + g) * (2 * n + 2);
					acc[2 * g] += h[c] * vv;
					acc[2 * g + 1] += h[n + c] * vv;
				}
			}
		}
		block_sum<NT, 8>(acc, S);
		for (int c = tid; c < n; c += NT) {
			double tt = t[c];
#pragma unroll
			for (int g = 0; g < 4; g++) {
				if (g < cnt) {
					const double *h = hist + (size_t)(base + g) * (2 * n + 2);
					const double s = h[c], u = h[n + c], rho = hrc ? hrc[2 * (base + g)] : h[2 * n], c2 = hrc ? hrc[2 * (base + g) + 1] : h[2 * n + 1];
					tt += -rho * (s * acc[2 * g + 1] + u * acc[2 * g]) + c2 * s * acc[2 * g];
				}
			}
			t[c] = tt;
		}
	}
	lds_sync();
}

// The same quasi-Newton operator from ONE stored vector per major iteration (half the history traffic of apply_history).
// With exact bookkeeping the pair (s_i, u_i) of major i lies in the span of two consecutive search directions: s_i = -alpha_i d_i and,
// because d_{i+1} = W_{i+1} g_{i+1} = omega_i t_i + theta_i d_i with t_i = W_i g_{i+1}, u_i = t_i - d_i = beta_i d_{i+1} + gamma_i d_i.
// Substituting into the rank-two terms gives  W_k v = W0 v + sum_j kappa_j d_j,  kappa = (symmetric tridiagonal) (d_j . v):
//   kappa_j = f_j delta_j + e_j delta_{j+1} + e_{j-1} delta_{j-1},   e_i = rho_i alpha_i beta_i,  f_i = 2 rho_i alpha_i gamma_i + c2_i alpha_i^2.
// HBM holds the chain d_0 .. d_{nv-1} (the last one is the current direction), LDS the link scalars (e_i, f_i); a skipped update is a
// null link (0, 0).  Rounds of G vectors through two register buffers like apply_history; the vector a round ends on is carried in
// registers into the next round (its link reaches across).
template <int NT, int EPT, int G>
__device__ __forceinline__ void apply_dform(const NtgDims &D, const Smem &S, const double *hist, int nv, const double *v, double *t, const double *links)
{
	const int n = D.nC, tid = threadIdx.x;
	double vv[EPT], tt[EPT], dc[EPT], dcl = 0.0;
	int cc[EPT];
#pragma unroll
	for (int e = 0; e < EPT; e++) { const int c = tid + e * NT; cc[e] = min(c, n - 1); vv[e] = c < n ? v[c] : 0.0; tt[e] = c < n ? t[c] : 0.0; dc[e] = 0.0; }
	double hA[G][EPT], hB[G][EPT];
	auto load = [&](int base, double (&h)[G][EPT]) {   // unconditional: past the end the newest vector again (its links are masked)
#pragma unroll
		for (int g = 0; g < G; g++) {
			const double *p = hist + (size_t)min(base + g, nv - 1) * n;
#pragma unroll
			for (int e = 0; e < EPT; e++) h[g][e] = p[cc[e]];
		}
	};
	auto consume = [&](int base, double (&h)[G][EPT]) {
		double acc[G];
#pragma unroll
		for (int g = 0; g < G; g++) {
			acc[g] = 0.0;
#pragma unroll
			for (int e = 0; e < EPT; e++) acc[g] += h[g][e] * vv[e];
		}
		block_sum<NT, G>(acc, S);
#pragma unroll
		for (int g = 0; g < G; g++) {
			const int i = base - 1 + g;   // link i joins vector i (left) and vector i+1 = base+g (right)
			const bool on = i >= 0 && i < nv - 1;
			const int ii = on ? i : 0;
			const double le = on ? links[2 * ii] : 0.0, lf = on ? links[2 * ii + 1] : 0.0;
			const double dl = g == 0 ? dcl : acc[g > 0 ? g - 1 : 0], dr = acc[g];
			const double a = le * dr + lf * dl, b = le * dl;
#pragma unroll
			for (int e = 0; e < EPT; e++) tt[e] += (g == 0 ? dc[e] : h[g > 0 ? g - 1 : 0][e]) * a + h[g][e] * b;
		}
#pragma unroll
		for (int e = 0; e < EPT; e++) dc[e] = h[G - 1][e];
		dcl = acc[G - 1];
	};
	load(0, hA);
	for (int base = 0; base < nv; base += 2 * G) {
		load(base + G, hB);
		consume(base, hA);
		if (base + G >= nv) break;
		load(base + 2 * G, hA);
		consume(base + G, hB);
	}
#pragma unroll
	for (int e = 0; e < EPT; e++) { const int c = tid + e * NT; if (c < n) t[c] = tt[e]; }
	lds_sync();
}

// One workgroup solves one problem from start to finish (ntg.c:250: the npsol_ call).
// The loop below has ONE evaluation site (funobj + projection at the trial point sxt); what the
// result means is decided by `state`: the first evaluation, a line-search trial, a forced
// acceptance, or the final multiplier estimate.  One site keeps the assembly code inlined once
// and the register state (coefficient map, line search) out of scratch.
// BIG: the coefficient vectors no longer fit in LDS next to the tables (config E: nC = 2196).  Only the trial point
// (read across lanes by Z = M C at every evaluation) stays in LDS; x, gp, gp+, d and g live in a per-problem HBM/L2
// workspace `vec_all`.  They are touched element-wise by their owner lane, except in the projection, the feasibility
// step and the preconditioner, which read a handful of entries across lanes behind a full barrier.
// NWT: the structured Newton mode (ntg_solve_opts.hessian = 2, newton.hpp): W is the inverse of the banded second-order
// model of the augmented Lagrangian, refactored at every major iteration; no quasi-Newton pairs.  The solve starts with
// a pass on the objective alone (mu = 0, "phase 0") before the augmented-Lagrangian passes.
// (The Newton instances of 256 lanes run one wave per SIMD anyway -- one workgroup per CU, by LDS -- and are compiled for it: the
// inlined factorisation and assembly then keep their window tiles and addresses in registers instead of scratch.)
// QPM (with NWT): the QP-based SQP step (ntg_solve_opts.hessian = 3, qpdual.hpp) replaces the augmented-Lagrangian
// passes after phase 0.
template <int FAM, int NOUT, int K, int NT, int EPT, bool BIG, bool HESS, int CHM, bool NWT = false, bool QPM = false>
__global__ void __launch_bounds__(NT, (NWT && NT <= 256) ? 1 : NTG_SQP_WAVES)
sqp_kernel(NtgDims D, NtgTables T, SmemLayout L, SolveParams sp, int batch,
           const double *__restrict__ lower, const double *__restrict__ upper, double *__restrict__ xio,
           double *__restrict__ objective, int *__restrict__ inform_out, int *__restrict__ iters_out,
           int *__restrict__ nfev_out, double *__restrict__ clambda, double *__restrict__ hist_all,
           double *__restrict__ al_all, double *__restrict__ vec_all, double *__restrict__ nwt_all)
{
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	const int b = blockIdx.x, tid = threadIdx.x, n = D.nC, m = D.mE /* rows kept by projection */, P = D.P;
	if (b >= batch) return;
	Smem S(smem_raw, L, D, T, b);
	const int npad = (n + 1) & ~1;
	double *gv = BIG ? vec_all + (size_t)b * 5 * npad : nullptr;
	double *sx = BIG ? gv : S.x, *sxt = S.vecs, *sgp = BIG ? gv + npad : S.vecs + npad,
	       *sgpt = BIG ? gv + 2 * npad : S.vecs + 2 * npad, *sd = BIG ? gv + 3 * npad : S.vecs + 3 * npad,
	       *st = sxt /* t = W gp+ lives in the trial-point buffer once x is committed */,
	       *sg = BIG ? gv + 4 * npad : S.vecs + 4 * npad, *tmp = S.vecs + (BIG ? 1 : 5) * npad;
	double *hist = hist_all + (size_t)b * sp.memcap * (2 * n + 2);   // pair i: [s (n) | u (n) | rho | c2]
	const double *hrc = L.hrc_n >= sp.memcap ? S.rho : nullptr;
	// one vector per major (apply_dform): problems without augmented-Lagrangian passes (the direction is never re-projected), a memory that
	// cannot fill up before the iteration limit, link scalars in LDS
	constexpr bool DF_OK = !NWT && !BIG && EPT <= 4 && Family<FAM>::NNLIC + Family<FAM>::NNLTC + Family<FAM>::NNLFC == 0;
	const bool dform = DF_OK && NTG_DFORM && D.nI == 0 && D.nC <= 3 * NT && hrc != nullptr && sp.memcap >= sp.itlim && L.hrc_n > sp.memcap;   // pair scalars (rho, c2): LDS for memories that fit, else with the pair in HBM
	stage_tables<NT>(D, T, S, smem_raw, L, b);
	NtgTables Tw = T;   // the preconditioner blocks of this problem (per-problem grids) or the shared ones (stride 0)
	if (HESS && T.n0b) Tw.n0b = T.n0b + (size_t)b * T.pp_n0b;
	if (NWT && T.nwt_k0) { Tw.nwt_k0 = T.nwt_k0 + (size_t)b * T.pp_k0; Tw.nwt_lf = T.nwt_lf + (size_t)b * T.pp_lf; }   // ... and the Newton mode's cost model / free-output factors
	for (int i = tid; i < n; i += NT) sx[i] = xio[(size_t)b * n + i];
	lds_sync();
	CoefMap<EPT> cm;
	if (NOUT > 0) make_coefmap<NT, EPT>(D, S, cm);

	enum { ST_INIT = 0, ST_LS = 1, ST_FORCE = 2, ST_FINAL = 3, ST_REEVAL = 4, ST_QP = 5 /* QP-based SQP step: evaluation of the l1 merit function */ };
	using Fam = Family<FAM>;
	constexpr bool HASCON = Fam::NNLIC + Fam::NNLTC + Fam::NNLFC > 0;
	const int ncn = D.ncnln;
	// linear rows declared as inequalities ride the same loop; only the generic instances carry that code
	constexpr bool LIN = NOUT == 0;
	const int nI = LIN ? D.nI : 0, nal = ncn + nI;
	// augmented-Lagrangian state (nonlinear rows, then linear inequality rows): multipliers and their estimates live in HBM
	double *al_lam = al_all + (size_t)b * 2 * (ncn + D.nI), *al_t = al_lam + ncn + D.nI;
	// diagnostic phase clock (sp.stamps): cycles spent in eval / project / history (structured Newton mode: model assembly) / W0 (Newton: factor + solve) / rest
	// Variant builds only (-DNTG_CLOCK: tools/mkvariant2.sh; tests/tools_newton.py, tests/tools_stamps.py): compiled in, the eight 64-bit
	// accumulators and the time base are 18 registers live across the whole kernel (the wave kernel gained 10 % when its clock went).
#ifdef NTG_CLOCK
	unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#define NTG_STAMP(slot) do { if (sp.stamps) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tk[slot] += now_ - tlast; tlast = now_; } } while (0)
#define NTG_CLOCK_TK(cond) ((cond) ? tk : nullptr)
	if (sp.stamps) tlast = __builtin_amdgcn_s_memtime();
#else
#define NTG_STAMP(slot) do { } while (0)
#define NTG_CLOCK_TK(cond) ((unsigned long long *)nullptr)
#endif
	const bool alprob = (HASCON && ncn > 0) || nI > 0;   // rows handled by the augmented-Lagrangian loop
	ALState al{(alprob && (!NWT || sp.warm)) ? 10.0 : 0.0, al_lam, al_t, lower + (size_t)b * D.nbounds, upper + (size_t)b * D.nbounds};
	// structured Newton mode: band matrix / factor and the per-breakpoint blocks of this problem (HBM), flags
	using FamN = Family<FAM>;
	constexpr int NWT_CG2 = FamN::CG * FamN::CG, NWT_CG2S = FamN::CG;
	// per problem: the groups' band arrays, [two-sided factorisation: their reversed arrays,] the per-breakpoint blocks
	const size_t nwt_ksz = (size_t)D.nwt_ngrp * D.nwt_ng * (D.nwt_hb + 1) + (D.nwt_tw ? (size_t)D.nwt_ngrp * (16 * D.nwt_jb + 48) * (D.nwt_hb + 1) : 0);
	// (QP-based SQP step: + the slots' columns U = W J' [NTG_QP_MAXA][npad] and the QP's multipliers of the previous major iteration [ncnln])
	const size_t ncq = (size_t)((D.ncnln + 1) & ~1);   // rows of the QP's per-row arrays: multipliers, c, J W g, derivative rows [CG], J U [NTG_QP_MAXA]
	const size_t qp_pp = QPM ? (size_t)ntg_qp_maxa(D.nwt_ngrp, NT) * npad + ncq * (3 + NWT_CG2S + ntg_qp_maxa(D.nwt_ngrp, NT)) : 0;
	double *nwt_K = NWT ? nwt_all + (size_t)b * (nwt_ksz + (size_t)D.nwt_ngrp * D.P * NWT_CG2 + qp_pp) : nullptr;
	double *nwt_B = NWT ? nwt_K + nwt_ksz : nullptr;
	double *qp_U = NWT ? nwt_B + (size_t)D.nwt_ngrp * D.P * NWT_CG2 : nullptr, *qp_lamq = NWT ? qp_U + (size_t)ntg_qp_maxa(D.nwt_ngrp, NT) * npad : nullptr;
	const NwtPair nwt_q{D.nwt_ng, D.nwt_hb, D.nwt_ja, D.nwt_jb};
	const int nwt_lena = 16 * (D.nwt_ja + 3) + 48, nwt_lenb = 16 * (D.nwt_jb + 3) + 48;
	// LDS of the mode (the area at L.nwt_y): the groups' solve vectors, the factorisation panels (one per factoring wave), the free outputs' vectors
	const int nwt_yall = D.nwt_ngrp * (D.nwt_tw ? nwt_lena + nwt_lenb : 16 * ((D.nwt_ng + 15) >> 4) + 48), nwt_npan = (D.nwt_tw ? 2 : 1) * D.nwt_ngrp;
	bool nwt_curv = false;        // the current factor includes the constraint curvature
	bool nwt_k0_only = false;     // the current factor is the cost model's alone (nwt_refresh_ex without blocks)
	int nwt_bad = 0;              // diagnostic: non-positive pivots replaced in Gauss-Newton factorisations (wave 0's count)
	int nwt_nfact = 0, nwt_nfail = 0, nwt_napply = 0;   // diagnostic (sp.stamps == 3): factorisations, of which not positive definite, solves
	bool phase0 = NWT && alprob && !sp.warm;  // the pass on the objective alone is still running (a warm start goes straight to the multipliers it was given)
	int *nwt_flag = (int *)(smem_raw + L.red) + 2 * (32 * (NT / 64) + 2) - 2;   // last word pair of the reduction scratch: "not positive definite"
	// K = model at the trial point buffer `xs` (must be the iterate x; needs the multiplier estimates al_t of the evaluation
	// at x), factored; then out = W v.  allow_curv = false: Gauss-Newton terms only.
	// tsrc / mu_gn: the multipliers of the curvature term and the weight of the Gauss-Newton term -- the estimates al_t and the penalty al.mu
	// for the augmented Lagrangian; the QP's multipliers al_lam and 0 for the QP-based SQP step (the Lagrangian's Hessian on the band)
	auto nwt_refresh_ex = [&](const double *xs, bool allow_curv, const double *tsrc, double mu_gn, bool blocks) __attribute__((always_inline)) {
		const int ngp = D.nwt_ngrp, ng = D.nwt_ng, hb = D.nwt_hb, P2 = D.P;
		const int wave = tid >> 6;
		// Without blocks the matrix is the cost model K0 -- a constant of the problem: while the factor in nwt_K is K0's (the majors of the pass
		// on the objective alone, the QP step's no-curvature regime) there is nothing to assemble or factor again.  Same matrix, same factor:
		// the iterates are bit for bit those of refactoring every time.
		if (!blocks && nwt_k0_only) return;
		unsigned long long *nwt_act = (unsigned long long *)((double *)(smem_raw + L.nwt_y) + (NT / 64) * 216);   // after the assembly's staging buffers
		double *panel = (double *)(smem_raw + L.nwt_y) + nwt_yall;
		for (int attempt = (allow_curv && blocks) ? 0 : 1; attempt < 2; attempt++) {
			const bool curv = attempt == 0;
			if (blocks) {
				constexpr int DM = FamN::DM, NZ = NOUT > 0 ? DM * NOUT : NTG_MAX_NZ, NTc = FamN::NNLTC > 0 ? FamN::NNLTC : 1;
				lds_sync();
				for (int i = tid; i < P2; i += NT) {
					double z[NZ], t[NTc];
					compute_z<NOUT, K, DM>(D, S, xs, i, D.tcon_mask, z);
#pragma unroll
					for (int j = 0; j < NTc; j++) t[j] = j < D.nnltc ? tsrc[D.nnlic + j * P2 + i] : 0.0;
					for (int g = 0; g < ngp; g++) {
						double Bk[NWT_CG2];
						FamN::template nltc_block<NZ>(NOUT > 0 ? NOUT : D.nout, g, z, t, mu_gn, curv, Bk);
						bool nzb = false;
#pragma unroll
						for (int e = 0; e < NWT_CG2; e++) { nwt_B[((size_t)g * P2 + i) * NWT_CG2 + e] = Bk[e]; nzb = nzb || Bk[e] != 0.0; }
						// which breakpoints carry a non-zero block: one 64-bit word per wave and sweep (a wave's lanes are 64 consecutive
						// breakpoints), read by the assembly to skip intervals that add nothing before it requests anything
						const unsigned long long mb = __ballot(nzb);
						if ((tid & 63) == 0) nwt_act[g * ((P2 + 63) >> 6) + (i >> 6)] = mb;
					}
				}
				__syncthreads();
			}
			NTG_STAMP(6);
			if (tid == 0) nwt_flag[0] = 0;
			nwt_assemble<NT, FamN::CG>(D, Tw, S.rowv, S.chrow, S.off, blocks ? nwt_B : nullptr, nwt_K, (double *)(smem_raw + L.nwt_y), NTG_CLOCK_TK(sp.stamps == 4));   // ends with a full barrier
			NTG_STAMP(3);
			if (D.nwt_tw) {   // two waves per group (wave uniform: every wave takes this branch)
				const int f = nwt_factor_pairs(nwt_K, nwt_K + (size_t)ngp * ng * (hb + 1), ngp, nwt_q, panel, curv ? 1 : 0, nwt_flag);
				if (f && !curv) nwt_bad += f;
			} else if (wave < ngp) {
				const int f = nwt_factor_wave((nwt_glb_dp)(nwt_K + (size_t)wave * ng * (hb + 1)), ng, hb, (nwt_lds_dp)(panel + (size_t)wave * NWT_PANEL), curv ? 1 : 0, 1 << 20);
				if (f && curv && (tid & 63) == 0) nwt_flag[0] = 1;
				if (f && !curv) nwt_bad += f;
			}
			__syncthreads();
			NTG_STAMP(7);
			nwt_curv = curv;
			nwt_nfact++;
			if (nwt_flag[0] == 0) break;
			nwt_nfail++;
		}
		nwt_k0_only = !blocks;
	};
	auto nwt_refresh = [&](const double *xs, bool allow_curv) __attribute__((always_inline)) { nwt_refresh_ex(xs, allow_curv, al_t, al.mu, al.mu > 0.0); };
	// restore = false: the caller keeps using the borrowed area (the QP-based SQP step's slots live behind the solve vectors) and puts
	// the zero padding of the weighted-gradient rows back itself (nwt_restore)
	auto nwt_restore = [&]() __attribute__((always_inline)) {
		for (int r = tid; r < L.dfz_rows; r += NT) S.dfz[r * (D.P + 1) + D.P] = 0.0;
		for (int i = tid; i < ntg_dfz_tail(D); i += NT) S.dfz[L.dfz_rows * (D.P + 1) + i] = 0.0;
		lds_sync();
	};
	auto nwt_apply = [&](const double *v, double *out, bool restore = true) __attribute__((always_inline)) {
		const int ngp = D.nwt_ngrp, ng = D.nwt_ng, hb = D.nwt_hb, ylen = 16 * ((ng + 15) >> 4) + 48;
		const int wave = tid >> 6;
		double *yv = (double *)(smem_raw + L.nwt_y);
		nwt_napply++;
		if (BIG) __syncthreads(); else lds_sync();
		const bool tw = D.nwt_tw != 0;
		const int ngt = ng - 16 * D.nwt_jb;   // two-sided: entries of the top part and the separator (vector a), the rest reversed (vector b)
		double *yb = yv + (size_t)ngp * nwt_lena;
		if (tw) {
			for (int i = tid; i < ngp * nwt_lena; i += NT) { const int g = i / nwt_lena, pp = i - g * nwt_lena; yv[i] = pp < ngt ? v[T.nwt_map[g * ng + pp]] : 0.0; }
			for (int i = tid; i < ngp * nwt_lenb; i += NT) { const int g = i / nwt_lenb, pp = i - g * nwt_lenb; yb[i] = pp < 16 * D.nwt_jb ? v[T.nwt_map[g * ng + ng - 1 - pp]] : 0.0; }
		} else
		for (int i = tid; i < ngp * ylen; i += NT) {
			const int g = i / ylen, pp = i - g * ylen;
			yv[i] = pp < ng ? v[T.nwt_map[g * ng + pp]] : 0.0;
		}
		// free outputs (in no nonlinear row): their vectors sit behind the groups' vectors and panels; the factor is the plan's
		const int nfo = D.nwt_nfo, ngf = D.nwt_ngf, hbf = D.nwt_hbf, ylenf = 16 * ((ngf + 15) >> 4) + 48;
		double *yvf = yv + (size_t)nwt_yall + (size_t)nwt_npan * NWT_PANEL;
		const int wfree = tw ? 2 * ngp : ngp;   // first wave that takes a free output
		for (int i = tid; i < nfo * ylenf; i += NT) {
			const int f = i / ylenf, pp = i - f * ylenf;
			yvf[i] = pp < ngf ? v[T.nwt_map[ngp * ng + f * ngf + pp]] : 0.0;
		}
		auto free_solves = [&] {
			if (wave >= wfree && wave < wfree + nfo) nwt_solve_wave((nwt_glb_cdp)(Tw.nwt_lf + (size_t)(wave - wfree) * ngf * (hbf + 1)), ngf, hbf, (nwt_lds_dp)(yvf + (size_t)(wave - wfree) * ylenf));
		};
		if (tw) {
			__syncthreads();
			nwt_solve_pairs(nwt_K, nwt_K + (size_t)ngp * ng * (hb + 1), ngp, nwt_q, yv, yb, free_solves);   // ends with a workgroup barrier
		} else {
			lds_sync();
			if (wave < ngp) nwt_solve_wave((nwt_glb_cdp)(nwt_K + (size_t)wave * ng * (hb + 1)), ng, hb, (nwt_lds_dp)(yv + (size_t)wave * ylen));
			else free_solves();
			lds_sync();
		}
		for (int c = tid; c < n; c += NT) {
			const int pos = T.nwt_pos[c];
			double o = 0.0;
			if (pos >= ngp * ng) { const int pf = pos - ngp * ng; o = yvf[(pf / ngf) * ylenf + (pf % ngf)]; }
			else if (pos >= 0) {
				const int g = pos / ng, pp = pos - g * ng;
				o = !tw ? yv[g * ylen + pp] : (pp < ngt ? yv[g * nwt_lena + pp] : yb[g * nwt_lenb + (ng - 1 - pp)]);
			}
			out[c] = o;
		}
		if (BIG) __syncthreads(); else lds_sync();
		// the area borrowed from the weighted-gradient rows goes back with its zero padding restored (stage_tables)
		if (restore) nwt_restore();
	};

	// =====================================================================================================================
	// QP-based SQP step on the band model (QPM; ntg_solve_opts.hessian = 3; qpdual.hpp).  Per major iteration:
	//   K = cost model + sum_j lam_j d2c_j/dz2 on the band (nwt_refresh_ex with the QP's multipliers), factored;
	//   the dual active-set QP of every coupling group on its SLOTS (rows in the working set): a row enters when the model step violates
	//   its linearised bound most; its column U = W J' is ONE band solve (the groups' entering rows share the solve: W is block diagonal
	//   by group), its entries of S = J W J' are products of derivative rows with Z = M U at the slots' breakpoints; the passive-set
	//   solves are scalar code on the group's slots in LDS (qp_passive_solve, one lane per group);
	//   p = -W g - sum_a lam_a U_a.
	// The line search backtracks on the l1 merit function, which the one evaluation site returns in this mode (ALState::qp).
	// =====================================================================================================================
	const int QA = ntg_qp_maxa(D.nwt_ngrp, NT), QPD = ntg_qp_doubles(QA);   // slots per coupling group, LDS doubles of a group's slot state
	// (the slots are addressed through LDS-typed pointers: ds_read / ds_write instead of FLAT accesses, which would also count in vmcnt and wait
	// for whatever the wave has in flight to HBM)
	typedef __attribute__((address_space(3))) int *lds_ip;
	typedef QpSlotsT<lds_dp, lds_ip> QpSlots;
	const int qp_ylenf = 16 * ((D.nwt_ngf + 15) >> 4) + 48;
	lds_dp qpbase = (lds_dp)((double *)(smem_raw + L.nwt_y) + nwt_yall + nwt_npan * NWT_PANEL + D.nwt_nfo * qp_ylenf);
	lds_dp qpred = qpbase + D.nwt_ngrp * QPD;   // [waves][rows per breakpoint][2] scratch of the entering-row search
	double qp_rho = 1.0, qp_phi0 = 0.0, qp_D = 0.0, qp_alpha = 1.0, qp_viol1 = 0.0, qp_pn = 0.0, qp_xn = 0.0, qp_lmax = 0.0, qp_gl = 0.0;
	int qp_k = 0, qp_over = 0, qp_nsolve = 0, qp_ncol = 0, qp_nocurv = 0, qp_fell = 0, qp_ntab = 0;
	bool qp_first = true, qp_full = false, qp_last = false;   // qp_last: the final pass evaluates at the point the last step led to   // qp_full: a group's working set did not hold every row the last QP wanted
	(void)qp_over; (void)qp_nsolve; (void)qp_ncol; (void)qp_fell; (void)qp_ntab;
	// flag entry (index into z) of variable u of group g
	auto qp_flag = [&](int g, int u) __attribute__((always_inline)) { return FamN::DM * (g * D.nwt_go + (int)((D.nwt_upack >> (8 * u + 4)) & 15u)) + (int)((D.nwt_upack >> (8 * u)) & 15u); };
	// The passive-set solve of qpdual.hpp (qp_passive_solve: same matrices, same ratio test, same drop rule) by ALL lanes of the group's wave:
	// lane i owns row i of the passive matrix in LDS -- right-looking Cholesky with one rank-one update per pivot, substitutions with the
	// pivot's unknown broadcast by __shfl.  The scalar routine's O(np^3) dependent LDS accesses were the tail of config D (working sets of
	// 20 - 30 rows: a millisecond per solve); this is O(np^2) per lane.
	auto qp_passive_solve_wave = [&](QpSlots &q, int lane) __attribute__((always_inline)) -> int {
		int solves = 0;
		// (every round that does not end the loop removes at least one slot in exact arithmetic; the count is bounded all the same -- a wave
		// that never leaves this loop would take the GPU with it)
		for (int round = 0; round < 3 * QA + 8; round++) {
			const int ns = *q.ns;
			int np = 0;
			for (int a = 0; a < ns; a++) if (q.inP[a]) { if (lane == 0) q.pl[np] = a; np++; }   // (uniform count; lane 0 writes the list)
			nwt_wave_sync();
			if (np == 0) break;
			const int mya = lane < np ? q.pl[lane] : 0;
			const double mys = q.sgn[mya] < 0 ? -1.0 : 1.0;
			bool ok = false;
			for (int attempt = 0; attempt < 2 && !ok; attempt++) {
				const double shift = attempt ? 1e4 : 1.0;
				if (lane < np)
					for (int l = 0; l <= lane; l++) {
						const int b2 = q.pl[l]; const double sb = q.sgn[b2] < 0 ? -1.0 : 1.0;
						double h = mys * sb * q.S[mya >= b2 ? NTG_QP_TR(mya, b2) : NTG_QP_TR(b2, mya)];
						if (l == lane) h += shift * 1e-10 * q.S[NTG_QP_TR(mya, mya)];
						q.H[NTG_QP_TR(lane, l)] = h;
					}
				nwt_wave_sync();
				ok = true;
				for (int k = 0; k < np; k++) {
					const double dkk = q.H[NTG_QP_TR(k, k)];
					if (!(dkk > 0.0)) { ok = false; break; }   // (uniform: every lane reads the same word)
					const double r = sqrt(dkk);
					double lik = 0.0;
					if (lane > k && lane < np) { lik = q.H[NTG_QP_TR(lane, k)] / r; q.H[NTG_QP_TR(lane, k)] = lik; }
					nwt_wave_sync();
					if (lane == k) q.H[NTG_QP_TR(k, k)] = r;
					if (lane > k && lane < np) for (int j = k + 1; j <= lane; j++) q.H[NTG_QP_TR(lane, j)] -= lik * q.H[NTG_QP_TR(j, k)];
					nwt_wave_sync();
				}
			}
			if (!ok) {   // not factorable even with the larger shift: the working set is given up
				if (lane < np) { q.nu[mya] = 0.0; q.inP[mya] = 0; }
				nwt_wave_sync();
				break;
			}
			// L z = -q, L' x = z
			double zi = lane < np ? -(mys * (q.jwg[mya] + q.rr[mya]) - 1e-10 * q.S[NTG_QP_TR(mya, mya)] * q.nu0[mya]) : 0.0;
			for (int k = 0; k < np; k++) {
				const double lkk = q.H[NTG_QP_TR(k, k)];
				if (lane == k) zi /= lkk;
				const double zk = __shfl(zi, k);
				if (lane > k && lane < np) zi -= q.H[NTG_QP_TR(lane, k)] * zk;
			}
			for (int k = np - 1; k >= 0; k--) {
				const double lkk = q.H[NTG_QP_TR(k, k)];
				if (lane == k) zi /= lkk;
				const double xk = __shfl(zi, k);
				if (lane < k) zi -= q.H[NTG_QP_TR(k, lane)] * xk;
			}
			solves++;
			// ratio test: back to the first sign change
			const double nua = lane < np ? q.nu[mya] : 0.0;
			const bool bad = lane < np && q.sgn[mya] != 0 && !(zi > 0.0);
			double al = bad ? nua / (nua - zi) : 1.0;
#pragma unroll
			for (int sh = 1; sh < 64; sh <<= 1) al = fmin(al, __shfl_xor(al, sh));
			if (__ballot(bad) == 0ull) { if (lane < np) q.nu[mya] = zi; nwt_wave_sync(); break; }
			if (!(al >= 0.0)) al = 0.0;
			if (lane < np) {
				const double nn = nua + al * (zi - nua);
				if (q.sgn[mya] != 0 && !(nn > 1e-14 * (1.0 + fabs(zi)))) { q.nu[mya] = 0.0; q.inP[mya] = 0; }
				else q.nu[mya] = nn;
			}
			nwt_wave_sync();
		}
		return solves;
	};
	// Row caches of one major iteration (HBM, per problem): for every trajectory row (constraint-major, like c) its value c, J W g, its
	// derivative row on the group's CG flag entries, and -- per slot index -- J U (the row times the slot's column): the entering-row
	// search, the entries of S and the right-hand sides are then reads and short sums, no evaluation-shaped pass per QP iteration.
	double *qp_c = qp_lamq + ncq, *qp_jwg = qp_c + ncq, *qp_a = qp_jwg + ncq, *qp_JU = qp_a + (size_t)ncq * FamN::CG;
	// one pass over the breakpoints: lane = breakpoint.  what = 0: c, derivative rows (at x = sx) and J W g (W g = sd); what = 1: J U of
	// the columns just formed (U = sxt; slot index flag[0] of the row's group)
	auto qp_rows_pass = [&](int what) __attribute__((always_inline)) {
		constexpr int DM = FamN::DM, NZ = NOUT > 0 ? DM * NOUT : NTG_MAX_NZ, NTc = FamN::NNLTC > 0 ? FamN::NNLTC : 1;
		for (int i = tid; i < P; i += NT) {
			double zv[NZ];
			compute_z<NOUT, K, DM>(D, S, what == 0 ? sd : sxt, i, D.tcon_mask, zv);
			if (what == 0) {
				double z[NZ], c[NTc], tape[FamN::TAPE];
				compute_z<NOUT, K, DM>(D, S, sxt, i, D.tcon_mask, z);   // (the trial-point buffer holds x here -- in LDS also for the BIG layouts, whose x lives in HBM)
				FamN::template nltc_val<NZ>(NOUT > 0 ? NOUT : D.nout, i, z, c, tape);
#pragma unroll
				for (int j = 0; j < NTc; j++) {
					if (j >= D.nnltc) continue;
					double df[NZ], t[NTc];
#pragma unroll
					for (int v = 0; v < NZ; v++) df[v] = 0.0;
#pragma unroll
					for (int jj = 0; jj < NTc; jj++) t[jj] = jj == j ? 1.0 : 0.0;
					FamN::template nltc_vjp<NZ>(NOUT > 0 ? NOUT : D.nout, D.nz, i, z, t, df, tape);
					const int g = FamN::row_group(j), row = j * P + i;
					double jw = 0.0;
					for (int u = 0; u < FamN::CG; u++) {
						const int fl = qp_flag(g, u);
						double av = 0.0, zz = 0.0;
#pragma unroll
						for (int v = 0; v < NZ; v++) if (v == fl) { av = df[v]; zz = zv[v]; }
						qp_a[(size_t)row * FamN::CG + u] = av;
						jw += av * zz;
					}
					qp_c[row] = c[j]; qp_jwg[row] = jw;
				}
			} else {
#pragma unroll
				for (int j = 0; j < NTc; j++) {
					if (j >= D.nnltc) continue;
					const int g = FamN::row_group(j), row = j * P + i;
					QpSlots q(qpbase + g * QPD, QA);
					const int acol = q.flag[0];
					if (acol < 0) continue;
					double ju = 0.0;
					for (int u = 0; u < FamN::CG; u++) {
						const int fl = qp_flag(g, u);
						double zz = 0.0;
#pragma unroll
						for (int v = 0; v < NZ; v++) if (v == fl) zz = zv[v];
						ju += qp_a[(size_t)row * FamN::CG + u] * zz;
					}
					qp_JU[(size_t)acol * ncq + row] = ju;
				}
			}
		}
	};
	// derivative rows (for the scatter of J'), J W g and r = bound - c of the slots of every group (one lane per slot): reads of the caches
	auto qp_slot_rows = [&](bool only_new) __attribute__((always_inline)) {
		const int ngp = D.nwt_ngrp, b0 = D.nlic + D.nltc + D.nlfc + D.nnlic;
		for (int e = tid; e < ngp * QA; e += NT) {
			const int g = e / QA, a = e - g * QA;
			QpSlots q(qpbase + g * QPD, QA);
			if (a >= *q.ns || (only_new && a != q.flag[0])) continue;
			const int row = q.row[a], j = row / P;
			for (int u = 0; u < FamN::CG; u++) q.ar[a * NTG_QP_MAXCG + u] = qp_a[(size_t)row * FamN::CG + u];
			q.jwg[a] = qp_jwg[row];
			q.rr[a] = (q.sgn[a] < 0 ? al.lo[b0 + j] : al.up[b0 + j]) - qp_c[row];
		}
	};
	// the columns U = W J' of the slots named by flag[0] of every group (-1: none), their J U over all rows and their entries of S
	auto qp_column = [&]() __attribute__((always_inline)) {
		const int ngp = D.nwt_ngrp, ng = D.nwt_ng, go = D.nwt_go, kk = K > 0 ? K : D.order[0];
		__syncthreads();
		if (D.nwt_tab && !nwt_curv && !T.pp_k0) {   // (the current factor is the cost model's: K0 -- of the plan's grid, whose tables these are)
			// The model is the cost model K0 (no constraint curvature any more in this solve): W M_i' per breakpoint and M_k W M_i' per pair
			// of breakpoints are the plan's tables (NtgTables::nwt_tu, nwt_g) -- the column and its J U are short combinations of table
			// rows, no band solve, no breakpoint pass.
			// (the column itself is not stored: the step's p = -W g - sum_a lam_a U_a combines the table rows of the slots with a multiplier, once)
			constexpr int CGc = FamN::CG, NTc2 = FamN::NNLTC > 0 ? FamN::NNLTC : 1;
			if (tid < ngp) { QpSlots q(qpbase + tid * QPD, QA); const int a = q.flag[0]; if (a >= 0) q.tab[a] = 1; }
			for (int i2 = tid; i2 < P; i2 += NT) {
#pragma unroll
				for (int j = 0; j < NTc2; j++) {
					if (j >= D.nnltc) continue;
					const int g = FamN::row_group(j), row = j * P + i2;
					QpSlots q(qpbase + g * QPD, QA);
					const int a = q.flag[0];
					if (a < 0) continue;
					const int i = q.row[a] % P;
					const double *gp2 = T.nwt_g + ((size_t)i2 * P + i) * CGc * CGc;
					double ju = 0.0;
#pragma unroll
					for (int v = 0; v < CGc; v++) {
						double sv = 0.0;
#pragma unroll
						for (int u = 0; u < CGc; u++) sv += gp2[v * CGc + u] * q.ar[a * NTG_QP_MAXCG + u];
						ju += qp_a[(size_t)row * CGc + v] * sv;
					}
					qp_JU[(size_t)a * ncq + row] = ju;
				}
			}
			__syncthreads();
			for (int e = tid; e < ngp * QA; e += NT) {
				const int g = e / QA, a2 = e - g * QA;
				QpSlots q(qpbase + g * QPD, QA);
				const int bcol = q.flag[0];
				if (bcol < 0 || a2 >= *q.ns) continue;
				q.S[a2 <= bcol ? NTG_QP_TR(bcol, a2) : NTG_QP_TR(a2, bcol)] = qp_JU[(size_t)bcol * ncq + q.row[a2]];
			}
			qp_ntab++;
			__syncthreads();
			return;
		}
		for (int c = tid; c < n; c += NT) sgpt[c] = 0.0;
		__syncthreads();
		for (int e = tid; e < ngp * go * kk; e += NT) {
			const int g = e / (go * kk), r = e - g * go * kk, ov = r / kk, qq = r - ov * kk;
			QpSlots q(qpbase + g * QPD, QA);
			const int a = q.flag[0];
			if (a < 0) continue;
			const int i = q.row[a] % P;
			double val = 0.0;
			for (int u = 0; u < FamN::CG; u++)
				if ((int)((D.nwt_upack >> (8 * u + 4)) & 15u) == ov) val += q.ar[a * NTG_QP_MAXCG + u] * S.rowv[S.chrow[(int)((D.nwt_upack >> (8 * u)) & 15u)] + qq * P + i];
			sgpt[D.iC[g * go + ov] + S.off[i] + qq] = val;
		}
		__syncthreads();
		nwt_apply(sgpt, sxt, false);
		__syncthreads();
		for (int c = tid; c < n; c += NT) {
			const int pos = T.nwt_pos[c];
			if (pos >= 0 && pos < ngp * ng) {
				QpSlots q(qpbase + (pos / ng) * QPD, QA);
				const int a = q.flag[0];
				if (a >= 0) qp_U[(size_t)a * npad + c] = sxt[c];
			}
		}
		if (tid < ngp) { QpSlots q(qpbase + tid * QPD, QA); const int a = q.flag[0]; if (a >= 0) q.tab[a] = 0; }
		qp_rows_pass(1);
		__syncthreads();
		for (int e = tid; e < ngp * QA; e += NT) {
			const int g = e / QA, a2 = e - g * QA;
			QpSlots q(qpbase + g * QPD, QA);
			const int bcol = q.flag[0];
			if (bcol < 0 || a2 >= *q.ns) continue;   // (the whole row and column of the slot: a reused slot sits in the middle)
			q.S[a2 <= bcol ? NTG_QP_TR(bcol, a2) : NTG_QP_TR(a2, bcol)] = qp_JU[(size_t)bcol * ncq + q.row[a2]];
		}
		qp_ncol++;
		__syncthreads();
	};
	// one major iteration's QP at x (sx = sxt), g (sg), multipliers al_lam, previous QP multipliers qp_lamq: leaves p in sgp, the QP's
	// multipliers in qp_lamq, |p|, |x|, g.p and the largest multiplier in the qp_* scalars
	double qp_gp = 0.0;
	auto qp_major = [&]() __attribute__((always_inline)) {
		const int ngp = D.nwt_ngrp, ng = D.nwt_ng, wave = tid >> 6, lane = tid & 63, b0 = D.nlic + D.nltc + D.nlfc + D.nnlic;
		constexpr int NTc = FamN::NNLTC > 0 ? FamN::NNLTC : 1;
		double qp_bl[NTc], qp_bu[NTc];   // the rows' bounds (one pair per row function), read once per major
#pragma unroll
		for (int j = 0; j < NTc; j++) { qp_bl[j] = j < D.nnltc ? al.lo[b0 + j] : 0.0; qp_bu[j] = j < D.nnltc ? al.up[b0 + j] : 0.0; }
		__syncthreads();   // the multipliers crossed lanes through HBM
		// After the model with the constraint curvature was not positive definite at two major iterations in a row the curvature is not tried
		// again in this solve: the model is then the cost model, the same matrix at every later major -- its factor is kept (no block pass, no
		// assembly, no factorisation; oracle/sqp.c sqpqp_run has the same rule and the measurements behind it)
		if (qp_nocurv < 2) { nwt_refresh_ex(sxt, true, al_lam, 0.0, true); qp_nocurv = nwt_curv ? 0 : qp_nocurv + 1; }
		nwt_apply(sg, sd, false);   // W g
		NTG_STAMP(4);
		__syncthreads();
		qp_rows_pass(0);
		// slots of the previous major's working set, in row order (a wave per group: ballot + prefix count)
		if (wave < ngp) {
			QpSlots q(qpbase + wave * QPD, QA);
			int cnt = 0;
			for (int j = 0; j < D.nnltc; j++) {
				if (FamN::row_group(j) != wave) continue;
				for (int i0 = 0; i0 < P; i0 += 64) {
					const int i = i0 + lane;
					const double lq = i < P ? qp_lamq[D.nnlic + j * P + i] : 0.0;
					const bool nz = lq != 0.0;
					const unsigned long long mk = __ballot(nz);
					const int slot = cnt + __popcll(mk & ((1ull << lane) - 1ull));
					if (nz && slot < QA) { q.row[slot] = j * P + i; q.sgn[slot] = lq > 0.0 ? 1 : -1; q.inP[slot] = 1; q.nu[slot] = q.nu0[slot] = fabs(lq); }
					cnt += __popcll(mk);
				}
			}
			if (lane == 0) { *q.ns = cnt < QA ? cnt : QA; q.flag[0] = -1; q.flag[1] = 0; q.flag[2] = cnt > QA ? 1 : 0; }
		}
		__syncthreads();
		qp_slot_rows(false);
		__syncthreads();
		NTG_STAMP(5);   // (variant builds: row caches + slots of the carried-over working set)
		int nswarm = 0;   // the first iterations form the columns of the slots carried over (one band solve per slot index, all groups at once)
		for (int g = 0; g < ngp; g++) { QpSlots q(qpbase + g * QPD, QA); nswarm = max(nswarm, *q.ns); }
		for (int it = 0; it < 5 * QA + 8; it++) {
			__syncthreads();
			bool docol = false;
			if (it < nswarm) {
				if (tid < ngp) { QpSlots q(qpbase + tid * QPD, QA); q.flag[0] = it < *q.ns ? it : -1; }
				docol = true;
			} else {
			if (wave < ngp) { QpSlots q(qpbase + wave * QPD, QA); qp_nsolve += qp_passive_solve_wave(q, lane); }
			__syncthreads();
			// most violated linearised bound of every trajectory row function among the rows outside the passive set:
			// c + J p = c - J W g - sum_a lam_a (J U_a)
			double bw[NTc]; int bk[NTc];
#pragma unroll
			for (int j = 0; j < NTc; j++) { bw[j] = 0.0; bk[j] = 0x7fffffff; }
			for (int i = tid; i < P; i += NT) {
#pragma unroll
				for (int j = 0; j < NTc; j++) {
					if (j >= D.nnltc) continue;
					const int row = j * P + i;
					QpSlots q(qpbase + FamN::row_group(j) * QPD, QA);
					const int ns = *q.ns;
					double lin = qp_c[row] - qp_jwg[row];
					int pas = 0;   // bit 0 / 1: the row's upper / lower side is passive
					for (int a = 0; a < ns; a++) {
						const double nua = q.nu[a];
						if (nua != 0.0) lin -= (q.sgn[a] < 0 ? -nua : nua) * qp_JU[(size_t)a * ncq + row];
						if (q.inP[a] && q.row[a] == row) pas |= q.sgn[a] < 0 ? 2 : 1;
					}
					const double bl = qp_bl[j], bu = qp_bu[j];
					const double wu = (bu < 1e19 && !(pas & 1)) ? lin - bu : -1.0, wl = (bl > -1e19 && !(pas & 2)) ? bl - lin : -1.0;
					const bool up = wu >= wl;
					const double w = up ? wu : wl, bound = up ? bu : bl;
					if (!(w > 1e-9 * (1.0 + fabs(bound)))) continue;
					const int key = 2 * row + (up ? 0 : 1);
					if (w > bw[j] || (w == bw[j] && key < bk[j])) { bw[j] = w; bk[j] = key; }
				}
			}
#pragma unroll
			for (int j = 0; j < NTc; j++) {
#pragma unroll
				for (int sh = 1; sh < 64; sh <<= 1) {
					const double ow = __shfl_xor(bw[j], sh); const int ok = __shfl_xor(bk[j], sh);
					if (ow > bw[j] || (ow == bw[j] && ok < bk[j])) { bw[j] = ow; bk[j] = ok; }
				}
				if (lane == 0) { qpred[(wave * NTc + j) * 2] = bw[j]; qpred[(wave * NTc + j) * 2 + 1] = (double)bk[j]; }
			}
			__syncthreads();
			if (wave < ngp && lane == 0) {
				QpSlots q(qpbase + wave * QPD, QA);
				double w = 0.0; int key = 0x7fffffff;
				for (int wv = 0; wv < NT / 64; wv++)
					for (int j = 0; j < D.nnltc; j++) {
						if (FamN::row_group(j) != wave) continue;
						const double ow = qpred[(wv * NTc + j) * 2]; const int ok = (int)qpred[(wv * NTc + j) * 2 + 1];
						if (ow > w || (ow == w && ok < key)) { w = ow; key = ok; }
					}
				q.flag[0] = -1; q.flag[1] = 0;
				if (w > 0.0) {
					const int row = key >> 1, sg = (key & 1) ? -1 : 1, ns = *q.ns;
					int a = -1;
					for (int a2 = 0; a2 < ns; a2++) if (q.row[a2] == row && q.sgn[a2] == sg) a = a2;
					if (a >= 0) { q.inP[a] = 1; q.flag[1] = 1; }   // a slot that left comes back: its column and entries of S are there
					else if (ns < QA) { q.row[ns] = row; q.sgn[ns] = sg; q.inP[ns] = 1; q.nu[ns] = 0.0; q.nu0[ns] = 0.0; q.flag[0] = ns; q.flag[1] = 1; *q.ns = ns + 1; }
					else {
						// every slot taken: a slot whose row left the passive set is given to the new row (its column, its J U and its row AND
						// column of S are formed anew by qp_column); none: the working set of this major is full
						int fr = -1;
						for (int a2 = 0; a2 < ns; a2++) if (!q.inP[a2] && fr < 0) fr = a2;
						if (fr >= 0) { q.row[fr] = row; q.sgn[fr] = sg; q.inP[fr] = 1; q.nu[fr] = 0.0; q.nu0[fr] = 0.0; q.flag[0] = fr; q.flag[1] = 1; }
						else q.flag[2] = 1;
					}
				}
			}
			__syncthreads();
			bool anycol = false, any = false;
			for (int g = 0; g < ngp; g++) { QpSlots q(qpbase + g * QPD, QA); anycol = anycol || q.flag[0] >= 0; any = any || q.flag[1] != 0; }
			if (!any) break;
			if (anycol) { qp_slot_rows(true); docol = true; }
			}
			NTG_STAMP(2);
			if (docol) { qp_column(); NTG_STAMP(0); }   // (the one call site: the band solve inside is inlined)
		}
		__syncthreads();
		// the QP's multipliers (signed: > 0 at an upper bound), the step p = -W g - sum_a lam_a U_a, the scalars of the exit test and of the
		// merit function
		NTG_STAMP(2);
		for (int j = tid; j < D.ncnln; j += NT) qp_lamq[j] = 0.0;
		__syncthreads();
		double lm = 0.0;
		for (int g = 0; g < ngp; g++) {
			QpSlots q(qpbase + g * QPD, QA);
			const int ns = *q.ns;
			for (int a = 0; a < ns; a++) { const double nua = q.nu[a]; lm = fmax(lm, fabs(nua)); if (tid == 0 && nua != 0.0) qp_lamq[D.nnlic + q.row[a]] = q.sgn[a] < 0 ? -nua : nua; }
			if (q.flag[2]) { qp_over++; qp_full = true; }
		}
		qp_lmax = lm;
		// ... and |Z'(g + J'lam)|, the reduced gradient of the Lagrangian with the QP's multipliers (the free coefficients span null(A_E)):
		// the optimality measure of the exit test, the same one the other modes use
		double r3[4] = {0.0, 0.0, 0.0, 0.0};
		const int go2 = D.nwt_go, kk2 = K > 0 ? K : D.order[0];
		for (int c = tid; c < n; c += NT) {
			const int pos = T.nwt_pos[c];
			double pc = -sd[c], gl = sg[c];
			if (pos >= 0 && pos < ngp * ng) {
				const int g = pos / ng, pp = pos - g * ng, ov = pp % go2, cl = D.nwt_clo + pp / go2;
				QpSlots q(qpbase + g * QPD, QA);
				const int ns = *q.ns;
				for (int a = 0; a < ns; a++) {
					const double nua = q.nu[a];
					if (nua == 0.0) continue;
					const double la = q.sgn[a] < 0 ? -nua : nua;
					const int i = q.row[a] % P, qq = cl - S.off[i];
					if (q.tab[a]) {   // the slot's column from the plan's table: U = sum_u a_u K0^-1 M_i' e_u
						double uv = 0.0;
						for (int u = 0; u < FamN::CG; u++) uv += q.ar[a * NTG_QP_MAXCG + u] * T.nwt_tu[((size_t)i * FamN::CG + u) * ng + pp];
						pc -= la * uv;
					} else pc -= la * qp_U[(size_t)a * npad + c];
					if (qq >= 0 && qq < kk2)
						for (int u = 0; u < FamN::CG; u++)
							if ((int)((D.nwt_upack >> (8 * u + 4)) & 15u) == ov) gl += la * q.ar[a * NTG_QP_MAXCG + u] * S.rowv[S.chrow[(int)((D.nwt_upack >> (8 * u)) & 15u)] + qq * P + i];
				}
			}
			const double xc = sx[c];
			sgp[c] = pc; r3[0] += pc * pc; r3[1] += xc * xc; r3[2] += sg[c] * pc;
			if (pos >= 0) r3[3] += gl * gl;
		}
		block_sum<NT, 4>(r3, S);
		qp_pn = sqrt(r3[0]); qp_xn = sqrt(r3[1]); qp_gp = r3[2]; qp_gl = sqrt(r3[3]);
		nwt_restore();
		NTG_STAMP(3);   // (variant builds: the step and its scalars -- shares the slot with the model assembly, which the QP step seldom runs)
		__syncthreads();
	};
	const LinIneq lin{nI, T.irow, T.icsr_ptr, T.icsr_col, T.icsc_ptr, T.icsc_row, T.icsr_val, T.icsc_val, (double *)(smem_raw + L.tI)};
	int inform = 4, iter = 0, nfev = 0, npairs = 0, state = ST_INIT;
	// ---- scope check (uniform): linear rows are equalities unless the plan declared them inequalities ----
	{
		double bad[1] = {0.0};
		for (int s = tid; s < D.nlic + D.nltc + D.nlfc; s += NT) {
			const double l = lower[(size_t)b * D.nbounds + s], u = upper[(size_t)b * D.nbounds + s];
			const bool ineq = D.nI > 0 && T.linflag[s] != 0;
			if (ineq ? !(l <= u) : l != u) bad[0] += 1.0;
		}
		block_sum<NT, 1>(bad, S);
		if (bad[0] != 0.0 || (ncn > 0 && !HASCON) || (D.nI > 0 && !LIN) || (nal > 0 && sp.fixed_iters)) inform = 9;
	}
	double F = 0.0, Fp = 0.0, gn2 = 0.0, rv2 = 0.0, alpha = 0.0, pnorm = 0.0;
	double sri = sp.sr, rvprev = HUGE_VAL;   // inner tolerance and best violation so far (AL outer loop)
	int outer = 0, inner_inform = 4;
	bool at_x = true, weak = false;
	double mfres = 0.0;   // diagnostic: linear residual seen by the last feasibility step
	bool final_pass = false;   // the multiplier pass ran: S.lam belongs to the gradient with the estimates al_t, report those
	if (inform != 9) {
		// ---- linear feasibility: x += A'(AA')^-1 (b - A x) ----
		auto make_feasible = [&]() {
			if (m <= 0) return;
			if (BIG) __syncthreads();   // x lives in HBM/L2 and the rows of A read it across lanes
			else lds_sync();
			for (int r = tid; r < m; r += NT) {
				const int s = lin_slot(D, D.nI > 0 ? T.erow[r] : r);
				double a = 0.0;
				for (int e = S.csr_ptr[r]; e < S.csr_ptr[r + 1]; e++) a += S.csr_val[e] * sx[S.csr_col[e]];
				tmp[r] = lower[(size_t)b * D.nbounds + s] - a;
			}
			lds_sync();
			if (sp.stamps == 2) { mfres = 0.0; for (int r = 0; r < m; r++) mfres = fmax(mfres, fabs(tmp[r])); }
			for (int r = tid; r < m; r += NT) {
				double a = 0.0;
				for (int e = S.sinv_ptr[r]; e < S.sinv_ptr[r + 1]; e++) a += S.sinv_val[e] * tmp[S.sinv_col[e]];
				tmp[m + r] = a;   // not S.lam: that holds the multiplier estimate reported in clambda
			}
			lds_sync();
			for (int c = tid; c < n; c += NT) {
				double s = 0.0;
				for (int e = S.csc_ptr[c]; e < S.csc_ptr[c + 1]; e++) s += S.csc_val[e] * tmp[m + S.csc_row[e]];
				sx[c] += s;
			}
			lds_sync();
		};
		make_feasible();
		for_vec<NT>(n, [&](int c) { sxt[c] = sx[c]; });
		if (alprob) {
			// cold: multipliers 0.  Warm (ntg_solve_opts.warm_start): the estimates the previous solve of this batch left in the workspace
			// (shifted with the horizon by ntg_batch_mpc_shift_multipliers) are the starting multipliers
			for (int j = tid; j < nal; j += NT) {
				if (sp.warm) al_lam[j] = al_t[j];
				else { al_lam[j] = 0.0; if (NWT) al_t[j] = 0.0; }
			}
			sri = fmax(sp.sr, 1e-3);
			__syncthreads();   // multipliers cross lanes through HBM: full barrier
			if (QPM && sp.warm) {
				// warm start of the QP-based SQP step (receding horizon): no pass on the objective alone -- the QP's first working set is the rows
				// the shifted multipliers of the previous solve name (ntg.h:64-68: what istate / clambda were meant to carry)
				for (int j = tid; j < ncn; j += NT) qp_lamq[j] = al_lam[j];
				al.mu = qp_rho = 1.0; al.qp = 1; state = ST_QP; qp_first = true;
				__syncthreads();
			}
		}
		NTG_STAMP(0);

		// line-search state lives in LDS (17 doubles would otherwise sit in every lane's registers);
		// each step works on a register copy and lane 0 publishes it back between two barriers
		// two copies of the line-search state: every lane advances its own register copy, lane 0 stores the result into the
		// OTHER copy, which is first read after the barriers of the next evaluation -- no barrier for the hand-over
		LineSearch *lsb = (LineSearch *)(smem_raw + L.ls);
		int lsi = 0;
		double ls_a = 0.0;   // current trial step (every lane)
		double r4[4] = {0, 0, 0, 0};   // gp.d, d.d, x.x, gp.gp of the current iterate
		bool finished = false;
		for (;;) {
			// ================= the one evaluation site =================
			// The evaluation leaves its four sums (quadrature, |g|^2, penalty, violation) as per-lane partials; the
			// projection pass adds the slope of the line search, and ONE workgroup reduction serves all five.
			double part[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, gdummy;
			(void)eval_cost<FAM, NOUT, K, NT, EPT, (NT >= 256), CHM>(D, S, sxt, sg, &gdummy, cm, al, nullptr, nullptr, NTG_CLOCK_TK(sp.stamps && !NWT),
			                                                        LIN ? &lin : nullptr, CHM != 0, part);
			NTG_STAMP(1);
			if (state != ST_FINAL && state != ST_REEVAL && !(QPM && state == ST_QP)) {
				project<NT, BIG>(D, S, sg, sgpt, tmp, sd, &part[4]);
				NTG_STAMP(2);
				block_sum<NT, 5>(part, S);
			} else {
				double p4[4] = {part[0], part[1], part[2], part[3]};
				block_sum<NT, 4>(p4, S);
				part[0] = p4[0]; part[1] = p4[1]; part[2] = p4[2]; part[3] = p4[3];
			}
			const double Fpn = (D.nicf ? S.dfi[D.nz] : 0.0) + part[0] + (D.nfcf ? S.dff[D.nz] : 0.0);   // ntg.c:328
			const double Fn = Fpn + part[2], gn2n = part[1], rv2n = part[3];
			if (state == ST_FINAL) {
				if (QPM && qp_last) { Fp = Fpn; F = Fn; rv2 = rv2n; nfev++; }
				// multipliers estimate lam = (AA')^-1 A g at the final point
				if (BIG) __syncthreads(); else lds_sync();
				for (int r = tid; r < m; r += NT) {
					double a = 0.0;
					for (int e = S.csr_ptr[r]; e < S.csr_ptr[r + 1]; e++) a += S.csr_val[e] * sg[S.csr_col[e]];
					tmp[r] = a;
				}
				lds_sync();
				for (int r = tid; r < m; r += NT) {
					double a = 0.0;
					for (int e = S.sinv_ptr[r]; e < S.sinv_ptr[r + 1]; e++) a += S.sinv_val[e] * tmp[S.sinv_col[e]];
					S.lam[r] = a;
				}
				lds_sync();
				final_pass = true;
				break;
			}
			nfev++;
			if (QPM && state == ST_QP) {
				// ---- QP-based SQP step: Fn is the l1 merit function F + rho sum_j viol_j at the trial point (at x itself the first time) ----
				bool qp_exit = false, qp_to_al = false;
				if (!qp_first) {
					if (!(Fn <= qp_phi0 + 1e-4 * qp_alpha * fmin(qp_D, 0.0) + 1e-14 * fabs(qp_phi0))) {
						if (++qp_k >= 25) qp_to_al = true;   // no acceptable step along the QP's direction (x and the multipliers are those of the last accepted point)
						else {
							qp_alpha *= 0.5;
							const double a = qp_alpha;
							for_vec<NT>(n, [&](int c) { sxt[c] = sx[c] + a * sgp[c]; });
							continue;
						}
					} else {
						const double a = qp_alpha;
						for_vec<NT>(n, [&](int c) { sx[c] = sxt[c]; });
						for (int j = tid; j < ncn; j += NT) al_lam[j] += a * (qp_lamq[j] - al_lam[j]);
						iter++;
					}
				}
				if (!qp_exit && !qp_to_al) {
					qp_first = false;
					F = Fn; Fp = Fpn; gn2 = gn2n; rv2 = rv2n; at_x = true;
					qp_viol1 = part[2] / qp_rho;
					if (iter >= sp.itlim) { inform = 4; qp_exit = true; }
				}
				if (!qp_exit && !qp_to_al) {
					qp_major();
					if (qp_full) qp_to_al = true;
				}
				if (qp_to_al) {
					// The working set of a group is full (NTG_QP_MAXA slots; rows active along a whole arc of the trajectory: the QP was not
					// solved, and its step would not restore feasibility), or the l1 merit function found no acceptable step along the QP's
					// direction.  This problem continues with the augmented-Lagrangian passes of the structured Newton mode from the current
					// point and the current multipliers (the warm-start path of hessian = 2): the mode is never less robust than that one.
					{
						al.qp = 0; al.mu = 10.0; qp_fell++;
						sri = fmax(sp.sr, 1e-3); rvprev = HUGE_VAL; outer = 0;
						npairs = 0; finished = false; inner_inform = 4; state = ST_INIT; weak = false; at_x = true;
						__syncthreads();
						for_vec<NT>(n, [&](int c) { sxt[c] = sx[c]; });
						continue;
					}
				}
				if (!qp_exit) {
					// exit: the step is small, the rows are feasible, and the reduced gradient of the Lagrangian meets NPSOL's optimality tolerance
					if (qp_pn <= 1e-2 * sp.sr * (1.0 + qp_xn) && sqrt(rv2) <= 1e-8 && qp_gl <= 0.1 * sp.sr * (1.0 + fmax(1.0 + fabs(Fp), sqrt(gn2)))) {
						// the last step is taken (it is below the exit tolerance, but K is large: the point it leads to is stationary to rounding --
						// a solve that ends right after phase 0 would otherwise keep that pass's looser tolerance); the final pass below
						// evaluates there: objective, multipliers of the linear rows
						for (int j = tid; j < ncn; j += NT) al_lam[j] = qp_lamq[j];
						for_vec<NT>(n, [&](int c) { sx[c] += sgp[c]; });
						inform = 0; qp_exit = true; qp_last = true;
					} else {
						if (qp_rho < 1.5 * qp_lmax + 1e-3) qp_rho = 2.0 * qp_lmax + 1e-3;
						al.mu = qp_rho;
						qp_D = qp_gp - qp_rho * qp_viol1; qp_phi0 = Fp + qp_rho * qp_viol1; qp_alpha = 1.0; qp_k = 0;
						for_vec<NT>(n, [&](int c) { sxt[c] = sx[c] + sgp[c]; });
						continue;
					}
				}
				// exit: one more pass for the linear rows' multipliers, with the gradient of the Lagrangian (ALState::qp = 2)
				__syncthreads();
				if (m > 0) {   // (always: the pass also leaves the multipliers in the workspace, where a warm start of the next solve finds them)
					state = ST_FINAL; al.qp = 2;
					make_feasible();   // before the pass that evaluates the multipliers: the x that is reported is the x they belong to
					lds_sync();
					for_vec<NT>(n, [&](int c) { sxt[c] = sx[c]; });
					continue;
				}
				break;
			}
			bool new_major = false;
			if (state == ST_REEVAL) {   // constraint values and multiplier estimates refreshed at x
				F = Fn; Fp = Fpn; gn2 = gn2n; rv2 = rv2n; at_x = true;
			} else {
			if (state == ST_INIT) {
				F = Fn; Fp = Fpn; gn2 = gn2n; rv2 = rv2n; at_x = true;
				for_vec<NT>(n, [&](int c) { sgp[c] = sgpt[c]; });
				if (NWT) { nwt_refresh(sxt, true); nwt_apply(sgp, sd); }
				else apply_w0<NT, BIG, HESS>(D, Tw, sp.hessian, sgp, sd, sxt, S.oinfo);
				r4[0] = r4[1] = r4[2] = r4[3] = 0.0;
				for_vec<NT>(n, [&](int c) { r4[0] += sgp[c] * sd[c]; r4[1] += sd[c] * sd[c]; r4[2] += sx[c] * sx[c]; r4[3] += sgp[c] * sgp[c]; });
				block_sum<NT, 4>(r4, S);
				NTG_STAMP(4);
				new_major = true;
			} else {
				int rc = 1;
				if (state == ST_LS) {
					const double dd[1] = {part[4]};   // gp+ . (-d), reduced together with the evaluation's sums
					LineSearch lsr = lsb[lsi];
					rc = lsr.step(Fn, dd[0]);
					ls_a = lsr.a;
					if (tid == 0) lsb[lsi ^ 1] = lsr;
					lsi ^= 1;
				}
				if (rc == 0 || rc == 2) {
					if (rc == 2) state = ST_FORCE;
					const double a = ls_a;
					for_vec<NT>(n, [&](int c) { sxt[c] = sx[c] + a * (-sd[c]); });
					NTG_STAMP(5);
					continue;
				}
				if (rc != 1) {
					const double tolg = sri * (1.0 + fmax(1.0 + fabs(F), sqrt(gn2)));
					if ((npairs > 0 || (NWT && nwt_curv)) && sqrt(r4[3]) > tolg) {
						// line search failed with a non-trivial W: drop the pairs and retry from the same point with W0
						// (structured Newton mode: with the Gauss-Newton factor at x; sxt is rebuilt from x first)
						npairs = 0;
						if (NWT) { lds_sync(); for_vec<NT>(n, [&](int c) { sxt[c] = sx[c]; }); nwt_refresh(sxt, false); nwt_apply(sgp, sd); }
						else apply_w0<NT, BIG, HESS>(D, Tw, sp.hessian, sgp, sd, sxt, S.oinfo);
						double r2[2] = {0, 0};
						for_vec<NT>(n, [&](int c) { r2[0] += sgp[c] * sd[c]; r2[1] += sd[c] * sd[c]; });
						block_sum<NT, 2>(r2, S);
						r4[0] = r2[0]; r4[1] = r2[1];
						new_major = true;
					} else {
						// no further decrease obtainable: converged within tolerance, or "optimal but not to the
						// requested accuracy" (NPSOL inform 1) within 10^3 of it, else failure (6)
						const double gpn0 = sqrt(r4[3]);
						if (gpn0 <= tolg) inner_inform = 0;
						else if (gpn0 <= 1e3 * tolg) { inner_inform = 0; weak = true; }
						else inner_inform = 6;
						finished = true; at_x = false;
					}
				} else {
					alpha = ls_a;
					// accept: commit x (frees sxt, which then holds t), t = W gp+, u = t - d, pair (s, u) to HBM
					for_vec<NT>(n, [&](int c) { sg[c] = alpha * (-sd[c]); sx[c] = sxt[c]; });   // sg = the step s
					if (!NWT && npairs == sp.memcap) { npairs = 0; apply_w0<NT, BIG, HESS>(D, Tw, sp.hessian, sgp, sd, sxt, S.oinfo); } // memory full: restart
					NTG_STAMP(5);
					bool conv_now = false;
					if (NWT && !sp.fixed_iters) {
						// The exit test of this major needs |x| and |gp+| only.  Taken BEFORE the model is rebuilt: a pass that ends here
						// (every pass ends on an accepted step) would otherwise assemble and factor a matrix nobody uses -- one of the
						// 2.7 (config D) refreshes per pass.
						double r2[2] = {0, 0};
						for_vec<NT>(n, [&](int c) { const double xn = sx[c], gq = sgpt[c]; r2[0] += xn * xn; r2[1] += gq * gq; });
						block_sum<NT, 2>(r2, S);
						conv_now = alpha * pnorm <= sri * (1.0 + sqrt(r2[0])) && sqrt(r2[1]) <= sri * (1.0 + fmax(1.0 + fabs(Fn), sqrt(gn2n)));
						if (conv_now) {
							for_vec<NT>(n, [&](int c) { sgp[c] = sgpt[c]; });
							r4[2] = r2[0]; r4[3] = r2[1];
							F = Fn; Fp = Fpn; gn2 = gn2n; rv2 = rv2n;
							iter++;
							inner_inform = 0; finished = true;
							lds_sync();
						}
					}
					if (!conv_now) {
					if (NWT) { nwt_refresh(sxt, true); nwt_apply(sgpt, st); }   // sxt still holds the accepted point; st (= sxt) is written last
					else apply_w0<NT, BIG, HESS, true>(D, Tw, sp.hessian, sgpt, st, sxt, S.oinfo);
					NTG_STAMP(4);
					// register-resident pairs: 3 coefficients per lane, 6 pairs per round; the 5-coefficient instances (config E)
					// take 3 pairs per round
					if (DF_OK && dform) {
						if (npairs == 0) for_vec<NT>(n, [&](int c) { hist[c] = sd[c]; });   // a chain starts: d_0 (read back by the owner lanes only)
						apply_dform<NT, 3, NTG_DF_G>(D, S, hist, npairs + 1, sgpt, st, hrc);
					} else
					apply_history<NT, (EPT > 4 ? 5 : 3), (EPT > 4 ? 3 : NTG_HIST_G)>(D, S, hist, npairs, sgpt, st, hrc);
					NTG_STAMP(3);
					double r6[6] = {0, 0, 0, 0, 0, 0};
					for_vec<NT>(n, [&](int c) {
						const double s = sg[c], y = sgpt[c] - sgp[c], u = st[c] - sd[c], gpn = sgpt[c];
						r6[0] += s * y; r6[1] += y * u; r6[2] += s * gpn; r6[3] += u * gpn; r6[4] += s * s; r6[5] += y * y;
					});
					block_sum<NT, 6>(r6, S);
					const bool upd = !NWT && r6[0] > 1e-12 * sqrt(r6[4]) * sqrt(r6[5]);
					const double rho = upd ? 1.0 / r6[0] : 0.0, c2 = upd ? rho * (1.0 + rho * r6[1]) : 0.0;
					r4[0] = r4[1] = r4[2] = r4[3] = 0.0;
					const bool df = DF_OK && dform;
					for_vec<NT>(n, [&](int c) {
						const double s = sg[c], u = st[c] - sd[c];
						if (upd && !df) { hist[(size_t)npairs * (2 * n + 2) + c] = s; hist[(size_t)npairs * (2 * n + 2) + n + c] = u; }
						const double dn = upd ? st[c] - rho * (s * r6[3] + u * r6[2]) + c2 * s * r6[2] : st[c];
						if (df) hist[(size_t)(npairs + 1) * n + c] = dn;   // the chain's next vector
						const double xn = sx[c], gq = sgpt[c];
						sgp[c] = gq;
						sd[c] = dn;
						r4[0] += gq * dn; r4[1] += dn * dn; r4[2] += xn * xn; r4[3] += gq * gq;
					});
					if (df) {
						// link scalars of this major (see apply_dform); omega = -(s.g)/(s.y) > 0 whenever the update is taken
						double le = 0.0, lf = 0.0;
						if (upd) {
							const double omega = 1.0 - rho * r6[2], theta = rho * r6[2] - alpha * c2 * r6[2] + alpha * rho * r6[3];
							const double beta = 1.0 / omega, gamma = -theta * beta - 1.0;
							le = rho * alpha * beta; lf = 2.0 * rho * alpha * gamma + c2 * alpha * alpha;
						}
						if (tid == 0) { S.rho[2 * npairs] = le; S.rho[2 * npairs + 1] = lf; }
						npairs++;
					} else if (upd) {
						if (tid == 0) {
							if (hrc) { S.rho[2 * npairs] = rho; S.rho[2 * npairs + 1] = c2; }   // read after the barriers of the block_sum below
							else { hist[(size_t)npairs * (2 * n + 2) + 2 * n] = rho; hist[(size_t)npairs * (2 * n + 2) + 2 * n + 1] = c2; }
						}
						npairs++;
					}
					if (!hrc) {
						__threadfence_block();
						__syncthreads();   // rho/c2 of the new pair go through HBM/L2: needs the full barrier (vmcnt(0))
					}
					block_sum<NT, 4>(r4, S);
					F = Fn; Fp = Fpn; gn2 = gn2n; rv2 = rv2n;
					iter++;
					if (!sp.fixed_iters && alpha * pnorm <= sri * (1.0 + sqrt(r4[2])) &&
					    sqrt(r4[3]) <= sri * (1.0 + fmax(1.0 + fabs(F), sqrt(gn2)))) { inner_inform = 0; finished = true; }
					else new_major = true;
					}
				}
			}
			}
			if (new_major) {
				// ---- start of a major iteration at (sx, sgp, sd) ----
				if (iter >= sp.itlim) { inner_inform = 4; finished = true; }
				else {
					if (al.mu > 0.0 && m > 0) {
						// Under a large penalty |g| >> |Z'g|: the rounding error of the projection, relative to |g|, is then
						// a visible fraction of gp and of d = W gp, and x would creep off A x = b along the path.
						// Projecting the direction itself leaves an error relative to |d| only.  (g's buffer is free here.)
						project<NT, BIG>(D, S, sd, sg, tmp);
						double r2[2] = {0, 0};
						for_vec<NT>(n, [&](int c) { const double dn = sg[c]; sd[c] = dn; r2[0] += sgp[c] * dn; r2[1] += dn * dn; });
						block_sum<NT, 2>(r2, S);
						r4[0] = r2[0]; r4[1] = r2[1];
					}
					double dphi0 = -r4[0];
					pnorm = sqrt(r4[1]);
					const double xnorm = sqrt(r4[2]), gpnorm = sqrt(r4[3]);
					const double tolg = sri * (1.0 + fmax(1.0 + fabs(F), sqrt(gn2)));
					if (pnorm == 0.0 || !(dphi0 < 0.0)) {
						if (pnorm != 0.0) { // W lost definiteness numerically: restart from W0 once
							npairs = 0;
							if (NWT) { lds_sync(); for_vec<NT>(n, [&](int c) { sxt[c] = sx[c]; }); nwt_refresh(sxt, false); nwt_apply(sgp, sd); }
							else apply_w0<NT, BIG, HESS>(D, Tw, sp.hessian, sgp, sd, sxt, S.oinfo);
							double r2[2] = {0, 0};
							for_vec<NT>(n, [&](int c) { r2[0] += sgp[c] * sd[c]; r2[1] += sd[c] * sd[c]; });
							block_sum<NT, 2>(r2, S);
							r4[0] = r2[0]; r4[1] = r2[1];
							dphi0 = -r2[0]; pnorm = sqrt(r2[1]);
						}
						if (pnorm == 0.0 || !(dphi0 < 0.0)) { inner_inform = (gpnorm <= tolg) ? 0 : 6; finished = true; }
					}
					if (!finished && !sp.fixed_iters && gpnorm <= 1e-3 * tolg) { inner_inform = 0; finished = true; }
					if (!finished) {
						const double amax = sp.steplimit * (1.0 + xnorm) / pnorm;
						const double a = amax < 1.0 ? amax : 1.0;
						if (tid == 0) lsb[lsi ^ 1].init(F, dphi0, a, amax, sp.ls_mu, sp.ls_eta, sp.ls_maxfev);
						lsi ^= 1; ls_a = a;
						state = ST_LS;
						for_vec<NT>(n, [&](int c) { sxt[c] = sx[c] + a * (-sd[c]); });
					}
				}
			}
			NTG_STAMP(5);
			if (finished && phase0) {
				// the pass on the objective alone is over: switch the augmented Lagrangian on (multipliers 0) and start its first pass
				phase0 = false;
				if (inner_inform == 4) { inform = 4; break; }
				if (QPM) {   // QP-based SQP from the unconstrained optimum: multipliers 0, merit weight 1
					for (int j = tid; j < ncn; j += NT) { al_lam[j] = 0.0; qp_lamq[j] = 0.0; }
					al.mu = qp_rho = 1.0; al.qp = 1;
					state = ST_QP; qp_first = true; finished = false;
					__syncthreads();
					make_feasible();   // like every further pass of the other modes: the first projection leaves a residual of rounding x cond(A A')
					for_vec<NT>(n, [&](int c) { sxt[c] = sx[c]; });
					continue;
				}
				al.mu = 10.0;
				npairs = 0; finished = false; inner_inform = 4; state = ST_INIT; weak = false; at_x = true;
				lds_sync();
				for_vec<NT>(n, [&](int c) { sxt[c] = sx[c]; });
				continue;
			}
			if (finished) {
				if (al.mu > 0.0) {
					// ---- multiplier / penalty update of the augmented Lagrangian ----
					if (!at_x) {   // inner solve ended on a rejected trial: refresh c, t at x first
						state = ST_REEVAL;
						lds_sync();
						for_vec<NT>(n, [&](int c) { sxt[c] = sx[c]; });
						continue;
					}
					const double rv = sqrt(rv2);
					bool done_al = false;
					bool take = false;
					if (inner_inform == 6) { inform = 6; done_al = true; }
					else if (rv <= 1e-8 && sri <= sp.sr && inner_inform == 0) { take = true; inform = weak ? 1 : 0; done_al = true; }
					else if (inner_inform == 4) { inform = 4; done_al = true; }
					else {
						if (rv <= 0.25 * rvprev) { take = true; rvprev = rv; }
						else al.mu *= 10.0;
						outer++;
						if (outer >= 30) { inform = 3; done_al = true; }
					}
					// on exit the multipliers stay as they are: x is stationary for the augmented Lagrangian of the CURRENT
					// multipliers, whose estimates t (al_t) are the ones consistent with it and the ones reported
					// the estimates al_t were stored by other lanes / waves during the last evaluation (row -> lane of its
					// breakpoint); only LDS barriers followed on some paths: drain the stores before any lane reads them
					__syncthreads();
					if (take && !done_al) for (int j = tid; j < nal; j += NT) al_lam[j] = al_t[j];
					__syncthreads();   // multipliers cross lanes through HBM: full barrier
					if (!done_al) {
						sri = fmax(sp.sr, fmin(1e-3, 0.1 * rvprev));
						npairs = 0; finished = false; inner_inform = 4; state = ST_INIT; weak = false;
						make_feasible();   // steps stay in null(A) only to rounding: re-project before every further pass
						for_vec<NT>(n, [&](int c) { sxt[c] = sx[c]; });
						continue;
					}
				} else inform = (inner_inform == 0 && weak) ? 1 : inner_inform;
				if (clambda && m > 0 && (D.q_use || al.mu > 0.0 || !at_x)) {   // one more pass for the multipliers (the Q form does not
				                                                       // produce them; projecting directions overwrote the estimate; a solve that
				                                                       // ended on a rejected line-search trial projected last at that trial, not at x)
					state = ST_FINAL;
					lds_sync();
					for_vec<NT>(n, [&](int c) { sxt[c] = sx[c]; });
					continue;
				}
				break;
			}
		}
		// many hundreds of majors under a large penalty let x drift off A x = b by rounding: restore it
		if (al.mu > 0.0) make_feasible();
	}
	lds_sync();
	for (int i = tid; i < n; i += NT) xio[(size_t)b * n + i] = sx[i];
	NTG_STAMP(5);
	if (clambda) {
		const int mall = D.nclin, ntot = n + mall + D.ncnln;   // NPSOL's layout: coefficients, all linear rows, nonlinear rows
		const double *al_rep = al_t;     // estimates of the last evaluation at x: g = A_E' lam_E - [A_I; J]' t
		(void)final_pass;
		__syncthreads();                 // al_t was written by other lanes
		for (int i = tid; i < ntot; i += NT) {
			double v = 0.0;
			if (inform != 9 && i >= n) {
				if (i < n + mall) {
					const int e = D.nI > 0 ? T.rowmap[i - n] : i - n;   // equality index, or -(j+1) for inequality row j
					v = e >= 0 ? S.lam[e] : (al.mu > 0.0 ? -al_rep[ncn - e - 1] : 0.0);
				} else if (al.mu > 0.0) v = -al_rep[i - n - mall];
			}
			clambda[(size_t)b * ntot + i] = v;
		}
#ifdef NTG_CLOCK
		if ((sp.stamps == 1 || sp.stamps == 4) && tid == 0) for (int i = 0; i < 8; i++) clambda[(size_t)b * ntot + i] = (double)tk[i];
#endif
		if (sp.stamps == 3 && tid == 0) {   // diagnostic: work counters of the structured Newton mode
			double *o = clambda + (size_t)b * ntot;
			o[0] = nwt_nfact; o[1] = nwt_nfail; o[2] = nwt_napply; o[3] = outer; o[4] = iter; o[5] = nfev; o[6] = qp_nsolve; o[7] = qp_ncol; o[8] = qp_over; o[9] = qp_fell; o[10] = qp_ntab;
		}
		if (sp.stamps == 2 && tid == 0) {   // diagnostic: state of the augmented-Lagrangian loop at exit
			double *o = clambda + (size_t)b * ntot;
			o[0] = sqrt(rv2); o[1] = al.mu; o[2] = outer; o[3] = sri; o[4] = rvprev; o[5] = inner_inform; o[6] = NWT ? (double)nwt_bad : mfres; o[7] = F;
		}
	}
#undef NTG_STAMP
#undef NTG_CLOCK_TK
	if (tid == 0) {
		if (objective) objective[b] = Fp;
		if (inform_out) inform_out[b] = inform;
		if (iters_out) iters_out[b] = iter;
		if (nfev_out) nfev_out[b] = nfev;
	}
}

// ------------------------------------------------------------------------------------------
// launchers: one (family, nout, order) instance at the workgroup size the host picked
// ------------------------------------------------------------------------------------------
template <int FAM, int NOUT, int K, int NT, int EPT, int CHM = 0>
static hipError_t launch_eval_one(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const EvalArgs &a)
{
	if (NOUT > 0 && D.ncnln > 0 && a.mode == 2 && !a.cj && a.f && a.g && a.c && a.jb && !getenv("NTG_AMD_NO_BANDONLY")) {
		auto kfb = eval_kernel<FAM, NOUT, K, NT, EPT, CHM, (NOUT > 0)>;
		if (L.total > 64 * 1024) (void)hipFuncSetAttribute((const void *)kfb, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
		hipLaunchKernelGGL(kfb, dim3(a.grid), dim3(NT), L.total, a.st, D, T, L, a.batch, a.mode, a.x, a.f, a.g, a.c, a.jb, a.cj);
		return hipGetLastError();
	}
	auto kfn = eval_kernel<FAM, NOUT, K, NT, EPT, CHM>;
	if (L.total > 64 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
	hipLaunchKernelGGL(kfn, dim3(a.grid), dim3(NT), L.total, a.st, D, T, L, a.batch, a.mode, a.x, a.f, a.g, a.c, a.jb, a.cj);
	return hipGetLastError();
}
template <int FAM, int NOUT, int K, int NT, int EPT, bool BIG, bool HESS = true, int CHM = 0, bool NWT = false, bool QPM = false>
static hipError_t launch_sqp_one(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const SolveParams &sp, const SqpArgs &a)
{
	auto kfn = sqp_kernel<FAM, NOUT, K, NT, EPT, BIG, HESS, CHM, NWT, QPM>;
	if (QPM != (sp.hessian == 3)) return hipErrorInvalidValue;
	if (NWT && (!D.nwt_on || !a.nwtw || D.nwt_cg != Family<FAM>::CG || D.nwt_go != Family<FAM>::COUPLE || ((D.nwt_tw ? 2 : 1) * D.nwt_ngrp + D.nwt_nfo) * 64 > NT)) return hipErrorInvalidValue;
	if (L.total > 64 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
	hipLaunchKernelGGL(kfn, dim3(a.batch), dim3(NT), L.total, a.st, D, T, L, sp, a.batch, a.lo, a.up, a.x, a.obj, a.inf, a.it, a.nf,
	                   a.cl, a.hist, a.alw, a.vecw, a.nwtw);
	return hipGetLastError();
}
// the small-problem instances: 128 or 256 lanes, all vectors in LDS
template <int FAM, int NOUT, int K, int CHM = 0>
static hipError_t launch_eval_small(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const EvalArgs &a)
{
	if (a.nt == 128) return launch_eval_one<FAM, NOUT, K, 128, 4, CHM>(D, T, L, a);
	if (a.nt == 256) return launch_eval_one<FAM, NOUT, K, 256, 4, CHM>(D, T, L, a);
	return hipErrorInvalidValue;
}
template <int FAM, int NOUT, int K, int CHM = 0>
static hipError_t launch_sqp_small(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const SolveParams &sp, const SqpArgs &a)
{
	if (a.big) return hipErrorInvalidValue;
	if (sp.hessian != 1) {   // identity cold start: the instance without any preconditioner code
		if (a.nt == 128) return launch_sqp_one<FAM, NOUT, K, 128, 4, false, false, CHM>(D, T, L, sp, a);
		if (a.nt == 256) return launch_sqp_one<FAM, NOUT, K, 256, 4, false, false, CHM>(D, T, L, sp, a);
		return hipErrorInvalidValue;
	}
	if (a.nt == 128) return launch_sqp_one<FAM, NOUT, K, 128, 4, false, true, CHM>(D, T, L, sp, a);
	if (a.nt == 256) return launch_sqp_one<FAM, NOUT, K, 256, 4, false, true, CHM>(D, T, L, sp, a);
	return hipErrorInvalidValue;
}
// does the plan match a channel-mask instance?  Running cost only, its active variables = derivative channels CHM of every
// output, no other weighted-gradient rows
static inline bool ntg_chm_match(const NtgDims &D, int chm, int dm)
{
	// (with trajectory constraints the mask is the one of the augmented-Lagrangian evaluation: cost and constraint variables)
	return D.nucf && !D.nicf && !D.nfcf && D.nnlic == 0 && D.nnlfc == 0 && (D.tcost_mask | D.tcon_mask) == chm_full_mask(chm, D.nout, dm) &&
	       D.ntav == D.nout * chm_count(chm);
}
// the generic instance of a family (run-time nout / order) at every workgroup size, LDS-resident or BIG
template <int FAM>
static hipError_t launch_eval_generic(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const EvalArgs &a)
{
	if (a.nt == 128) return launch_eval_one<FAM, 0, 0, 128, 4>(D, T, L, a);
	if (a.nt == 256) return launch_eval_one<FAM, 0, 0, 256, 4>(D, T, L, a);
	if (a.nt == 512) return launch_eval_one<FAM, 0, 0, 512, 4>(D, T, L, a);
	return hipErrorInvalidValue;
}
// structured Newton mode, generic instances (families that offer it: Family::COUPLE > 0)
template <int FAM>
static hipError_t launch_sqp_newton_generic(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const SolveParams &sp, const SqpArgs &a)
{
	if constexpr (Family<FAM>::COUPLE > 0) {
		if (sp.hessian == 3) {   // QP-based SQP step: generic instances
			if (a.big) {
				if (a.nt == 512) return launch_sqp_one<FAM, 0, 0, 512, 4, true, true, 0, true, true>(D, T, L, sp, a);
				return hipErrorInvalidValue;
			}
			if (a.nt == 128) return launch_sqp_one<FAM, 0, 0, 128, 4, false, true, 0, true, true>(D, T, L, sp, a);
			if (a.nt == 256) return launch_sqp_one<FAM, 0, 0, 256, 4, false, true, 0, true, true>(D, T, L, sp, a);
			if (a.nt == 512) return launch_sqp_one<FAM, 0, 0, 512, 4, false, true, 0, true, true>(D, T, L, sp, a);
			return hipErrorInvalidValue;
		}
		if (a.big) {
			if (a.nt == 256) return launch_sqp_one<FAM, 0, 0, 256, 4, true, true, 0, true>(D, T, L, sp, a);
			if (a.nt == 512) return launch_sqp_one<FAM, 0, 0, 512, 4, true, true, 0, true>(D, T, L, sp, a);
			return hipErrorInvalidValue;
		}
		if (a.nt == 128) return launch_sqp_one<FAM, 0, 0, 128, 4, false, true, 0, true>(D, T, L, sp, a);
		if (a.nt == 256) return launch_sqp_one<FAM, 0, 0, 256, 4, false, true, 0, true>(D, T, L, sp, a);
		if (a.nt == 512) return launch_sqp_one<FAM, 0, 0, 512, 4, false, true, 0, true>(D, T, L, sp, a);
	}
	return hipErrorInvalidValue;
}
template <int FAM>
static hipError_t launch_sqp_generic(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const SolveParams &sp, const SqpArgs &a)
{
	if (sp.hessian == 2 || sp.hessian == 3) return launch_sqp_newton_generic<FAM>(D, T, L, sp, a);
	if (a.big) {
		if (a.nt == 256) return launch_sqp_one<FAM, 0, 0, 256, 4, true>(D, T, L, sp, a);
		if (a.nt == 512) return launch_sqp_one<FAM, 0, 0, 512, 4, true>(D, T, L, sp, a);
		return hipErrorInvalidValue;
	}
	if (a.nt == 128) return launch_sqp_one<FAM, 0, 0, 128, 4, false>(D, T, L, sp, a);
	if (a.nt == 256) return launch_sqp_one<FAM, 0, 0, 256, 4, false>(D, T, L, sp, a);
	if (a.nt == 512) return launch_sqp_one<FAM, 0, 0, 512, 4, false>(D, T, L, sp, a);
	return hipErrorInvalidValue;
}
// shape tests shared by the per-family dispatchers
static inline bool ntg_all_d(const NtgDims &D, int d) { for (int o = 0; o < D.nout; o++) if (D.d[o] != d) return false; return true; }
// 0 unless a tuned instance may run: one basis class, nC within the lanes' slots, no linear inequality rows (those
// are handled by the generic instances only)
static inline int ntg_uniform_order(const NtgDims &D, int nt, int ept) { return (D.uniform && D.nC <= ept * nt && D.nI == 0) ? D.order[0] : 0; }
