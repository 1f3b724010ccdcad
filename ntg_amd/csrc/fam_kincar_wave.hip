// fam_kincar_wave.hip -- instances of sqp_wave_kernel (solve_wave.hpp: one wavefront per problem) for the kincar shapes of
// BASELINE.json: 2, 4 and 6 flat outputs, order 6, 20 knot intervals (configs B / C, two cars, config M).  Own translation unit
// so that it compiles next to the other families.
//   FAT   one wave per SIMD (512 registers): the chain of search directions lives in the accumulator registers and LDS; the
//         instance of long chains -- the identity cold start (NPSOL's mode, the fixed-work benchmark)
//   LEAN  several waves per SIMD, chain in LDS / HBM: the instance of short solves -- the collocation preconditioner
//         (3 majors to convergence), receding-horizon re-solves
#include "solve_wave.hpp"
#include "plan.hpp"
#include <cstdlib>

using namespace ntgw;

namespace {

// base of the hand-managed accumulator range of an instance (solve_wave.hpp): its class's macro, raised by ntg_amd/build.py when the
// ISA audit finds compiler-generated code in the range; instances without register slots carry 0 (nothing to audit)
constexpr int abase_of(int nint, bool ppg, int nreg) { return nreg == 0 ? 0 : ((nint != 20 || ppg) ? NTGW_ABASE_ALT : NTGW_ABASE); }
constexpr int nreg_of(int nint, bool ppg, int epl) { return (256 - ((nint != 20 || ppg) ? NTGW_ABASE_ALT : NTGW_ABASE)) / (2 * epl); }

template <int NOUT, int OPL, int NWV, int MINW, int NREG, int NLDS, bool HESS, bool XLDS = false, int NINT = 20, bool PPG = false>
hipError_t launch_one(const NtgDims &D, const NtgTables &T, const SolveParams &sp, const SqpArgs &a, const NtgWavePlan &w)
{
	auto kfn = sqp_wave_kernel<NTG_FAM_KINCAR, NOUT, OPL, 6, 4, NINT, NWV, MINW, NREG, NLDS, HESS, XLDS, PPG, abase_of(NINT, PPG, NREG)>;
	WaveArgs A;
	A.batch = a.batch; A.cap = w.cap; A.lower = a.lo; A.upper = a.up; A.xio = a.x; A.objective = a.obj; A.inform = a.inf; A.iters = a.it;
	A.nfev = a.nf; A.clambda = a.cl; A.hist = a.hist; A.counter = a.counter; A.hbm_slots = w.hbm_slots;
	if (w.lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w.lds);
	hipError_t e = hipMemsetAsync(a.counter, 0, sizeof(unsigned int), a.st);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(kfn, dim3(w.grid), dim3(64 * NWV), w.lds, a.st, D, T, sp, A);
	return hipGetLastError();
}

// LDS bytes of a workgroup (xlds: with the preconditioner blocks and the sparse linear operator staged)
size_t wave_lds(const NtgDims &D, const NtgTables &T, int hessian, int nwv, int cap, int nlds, int epl, bool xlds = false, bool ppg = false)
{
	// value tables (basis, weights, projector values): one copy, or one per wave with per-problem grids; the projector's indices once
	size_t tab = (size_t)(ppg ? nwv : 1) * ((size_t)(D.ig_n == 16 ? wave_tab_doubles<1, 6, 16>() : wave_tab_doubles<1, 6, 20>()) * 8 + (size_t)D.q_nt * 6 * 8) + (size_t)D.q_nt * 8 * 4;
	if (xlds) {
		const int lnz = std::max(D.lin_nnz, 1), snz = std::max(D.sinv_nnz, 1);
		tab += (size_t)(hessian == 1 ? T.n0b_nblk * 64 * 64 : 0) * 8 + (size_t)(2 * lnz + snz) * 8 + (size_t)((2 * (D.mE + 1) + (D.nC + 1) + 2 * lnz + snz + 3) & ~3) * 4;
	}
	return tab + (size_t)nwv * wave_priv_doubles(D.nC, cap, nlds, epl) * 8;
}

#ifndef NTGW_LEAN_NLDS
#define NTGW_LEAN_NLDS 2
#endif
#ifndef NTGW_LEAN_MINW
#define NTGW_LEAN_MINW 2
#endif
constexpr int FAT_NLDS = 10, FAT_NLDS2 = 6, PPG_NLDS = 2, LEAN_NLDS = NTGW_LEAN_NLDS, LEAN_MINW = NTGW_LEAN_MINW;   // PPG_NLDS: four copies of the value tables leave room for two chain slots per wave next to a full-length memory's scalars   // FAT_NLDS2: the instance for long chains (their scalars take more of the LDS)

}   // namespace

// Does the wave kernel take this solve, and with what launch shape / workspace?  (plan.cpp sizes the workspace with it.)
bool ntg_wave_plan(const NtgDims &D, const NtgTables &T, const SolveParams &sp, int batch, int ncu, NtgWavePlan *w)
{
	if (getenv("NTG_AMD_NOWAVE")) return false;
	if (D.family != NTG_FAM_KINCAR || !(D.nout == 2 || D.nout == 4 || D.nout == 6)) return false;
	const int opl = D.nout == 2 ? 1 : 2;
	// 20 knot intervals (BASELINE's kincar configs), and 16 for the four-output shape (bench.py `generic_instances`)
	if (!wave_match(D, T, sp, 4, 3, 6, opl, 20) && !(D.nout == 4 && wave_match(D, T, sp, 4, 3, 6, opl, 16))) return false;
	const int epl = opl * 3;
	w->cap = std::min(sp.memcap, sp.itlim) + 4;
	w->ppg = (T.pp_rowv || T.pp_bps || T.pp_q) ? 1 : 0;   // per-problem grids: the FAT instance with wave-private tables (wave_match: no preconditioner)
	w->fat = (sp.hessian != 1 && (!getenv("NTG_AMD_WAVE_LEAN") || w->ppg)) ? 1 : 0;
	w->nwv = 4; w->noagpr = 0;
	if (w->fat) {
		// NTG_AMD_WAVE_NOAGPR=1: the fallback instance that keeps no chain slot in registers (nothing hand-managed: what a toolchain that
		// breaks the accumulator scheme would still build and run) -- chain in LDS and HBM only
		w->noagpr = (getenv("NTG_AMD_WAVE_NOAGPR") && !w->ppg && D.ig_n == 20) ? 1 : 0;
		const int nreg = w->noagpr ? 0 : nreg_of(D.ig_n, w->ppg != 0, epl);
		w->nlds = w->ppg ? PPG_NLDS : FAT_NLDS;
		w->lds = wave_lds(D, T, sp.hessian, 4, w->cap, w->nlds, epl, false, w->ppg != 0);
		if (w->lds > 160 * 1024 && !w->ppg) { w->nlds = FAT_NLDS2; w->lds = wave_lds(D, T, sp.hessian, 4, w->cap, w->nlds, epl); }
		if (w->lds > 160 * 1024 && w->ppg) return false;
		if (w->lds > 160 * 1024) w->fat = 0;
		else {
			w->hbm_slots = std::max(0, w->cap - nreg - w->nlds);
			w->grid = std::max(1, std::min((batch + 3) / 4, ncu));   // one workgroup per CU: four waves, one per SIMD
		}
	}
	if (!w->fat && w->ppg) {
		// per-problem grids with the preconditioner: four waves, each with its own value tables; linear operator and preconditioner blocks
		// (per problem) read from HBM / L2
		w->nlds = LEAN_NLDS; w->nwv = 4;
		w->lds = wave_lds(D, T, sp.hessian, 4, w->cap, LEAN_NLDS, epl, false, true);
		if (w->lds > 160 * 1024) return false;
		w->hbm_slots = std::max(0, w->cap - LEAN_NLDS);
		const int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>((4 * LEAN_MINW) / w->nwv, (160 * 1024) / w->lds));
		w->grid = std::max(1, std::min((batch + w->nwv - 1) / w->nwv, ncu * wg_per_cu));
	} else if (!w->fat) {
		// eight waves sharing one copy of the tables, the preconditioner block and the linear operator in LDS; four when the scalars of
		// a long quasi-Newton memory leave no room for eight
		w->nlds = LEAN_NLDS; w->nwv = 8;
		w->lds = wave_lds(D, T, sp.hessian, 8, w->cap, LEAN_NLDS, epl, true);
		if (w->lds > 160 * 1024 || (sp.hessian == 1 && T.n0b_sp > 64)) { w->nwv = 4; w->lds = wave_lds(D, T, sp.hessian, 4, w->cap, LEAN_NLDS, epl, sp.hessian != 1 || T.n0b_sp <= 64); }
		if (w->lds > 160 * 1024) return false;
		w->hbm_slots = std::max(0, w->cap - LEAN_NLDS);
		const int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>((4 * LEAN_MINW) / w->nwv, (160 * 1024) / w->lds));
		w->grid = std::max(1, std::min((batch + w->nwv - 1) / w->nwv, ncu * wg_per_cu));
	}
	w->hist_doubles = (size_t)w->grid * w->nwv * w->hbm_slots * epl * 64;
	return true;
}

hipError_t ntg_launch_sqp_wave(const NtgDims &D, const NtgTables &T, const SolveParams &sp, const SqpArgs &a, const NtgWavePlan &w)
{
	if (!a.counter) return hipErrorInvalidValue;
	constexpr int R3 = nreg_of(20, false, 3), R6 = nreg_of(20, false, 6), R3A = nreg_of(20, true, 3), R6A = nreg_of(20, true, 6);
	if (D.ig_n == 16) {   // four outputs on 16 knot intervals
		if (w.fat && w.nlds == FAT_NLDS) return launch_one<4, 2, 4, 1, R6A, FAT_NLDS, false, false, 16>(D, T, sp, a, w);
		if (w.fat) return launch_one<4, 2, 4, 1, R6A, FAT_NLDS2, false, false, 16>(D, T, sp, a, w);
		const bool xl16 = sp.hessian != 1 || T.n0b_sp <= 64;
		if (w.nwv == 8 && xl16) return launch_one<4, 2, 8, LEAN_MINW, 0, LEAN_NLDS, true, true, 16>(D, T, sp, a, w);
		if (xl16) return launch_one<4, 2, 4, LEAN_MINW, 0, LEAN_NLDS, true, true, 16>(D, T, sp, a, w);
		return launch_one<4, 2, 4, LEAN_MINW, 0, LEAN_NLDS, true, false, 16>(D, T, sp, a, w);
	}
	if (w.ppg) {   // per-problem grids
		if (!w.fat) {   // with the preconditioner
			if (D.nout == 2) return launch_one<2, 1, 4, LEAN_MINW, 0, LEAN_NLDS, true, false, 20, true>(D, T, sp, a, w);
			if (D.nout == 4) return launch_one<4, 2, 4, LEAN_MINW, 0, LEAN_NLDS, true, false, 20, true>(D, T, sp, a, w);
			return launch_one<6, 2, 4, LEAN_MINW, 0, LEAN_NLDS, true, false, 20, true>(D, T, sp, a, w);
		}
		if (w.nlds != PPG_NLDS) return hipErrorInvalidValue;
		if (D.nout == 2) return launch_one<2, 1, 4, 1, R3A, PPG_NLDS, false, false, 20, true>(D, T, sp, a, w);
		if (D.nout == 4) return launch_one<4, 2, 4, 1, R6A, PPG_NLDS, false, false, 20, true>(D, T, sp, a, w);
		return launch_one<6, 2, 4, 1, R6A, PPG_NLDS, false, false, 20, true>(D, T, sp, a, w);
	}
	if (w.fat && w.noagpr) {   // no chain slot in registers (NTG_AMD_WAVE_NOAGPR, or a build whose accumulator base was raised to 256)
		if (D.nout == 2) return w.nlds == FAT_NLDS ? launch_one<2, 1, 4, 1, 0, FAT_NLDS, false>(D, T, sp, a, w) : launch_one<2, 1, 4, 1, 0, FAT_NLDS2, false>(D, T, sp, a, w);
		if (D.nout == 4) return w.nlds == FAT_NLDS ? launch_one<4, 2, 4, 1, 0, FAT_NLDS, false>(D, T, sp, a, w) : launch_one<4, 2, 4, 1, 0, FAT_NLDS2, false>(D, T, sp, a, w);
		return w.nlds == FAT_NLDS ? launch_one<6, 2, 4, 1, 0, FAT_NLDS, false>(D, T, sp, a, w) : launch_one<6, 2, 4, 1, 0, FAT_NLDS2, false>(D, T, sp, a, w);
	}
	if (w.fat && w.nlds == FAT_NLDS) {
		if (D.nout == 2) return launch_one<2, 1, 4, 1, R3, FAT_NLDS, false>(D, T, sp, a, w);
		if (D.nout == 4) return launch_one<4, 2, 4, 1, R6, FAT_NLDS, false>(D, T, sp, a, w);
		return launch_one<6, 2, 4, 1, R6, FAT_NLDS, false>(D, T, sp, a, w);
	}
	if (w.fat) {
		if (D.nout == 2) return launch_one<2, 1, 4, 1, R3, FAT_NLDS2, false>(D, T, sp, a, w);
		if (D.nout == 4) return launch_one<4, 2, 4, 1, R6, FAT_NLDS2, false>(D, T, sp, a, w);
		return launch_one<6, 2, 4, 1, R6, FAT_NLDS2, false>(D, T, sp, a, w);
	}
	const bool xl = sp.hessian != 1 || T.n0b_sp <= 64;
	if (w.nwv == 8 && xl) {
		if (D.nout == 2) return launch_one<2, 1, 8, LEAN_MINW, 0, LEAN_NLDS, true, true>(D, T, sp, a, w);
		if (D.nout == 4) return launch_one<4, 2, 8, LEAN_MINW, 0, LEAN_NLDS, true, true>(D, T, sp, a, w);
		return launch_one<6, 2, 8, LEAN_MINW, 0, LEAN_NLDS, true, true>(D, T, sp, a, w);
	}
	if (xl) {
		if (D.nout == 2) return launch_one<2, 1, 4, LEAN_MINW, 0, LEAN_NLDS, true, true>(D, T, sp, a, w);
		if (D.nout == 4) return launch_one<4, 2, 4, LEAN_MINW, 0, LEAN_NLDS, true, true>(D, T, sp, a, w);
		return launch_one<6, 2, 4, LEAN_MINW, 0, LEAN_NLDS, true, true>(D, T, sp, a, w);
	}
	if (D.nout == 2) return launch_one<2, 1, 4, LEAN_MINW, 0, LEAN_NLDS, true>(D, T, sp, a, w);
	if (D.nout == 4) return launch_one<4, 2, 4, LEAN_MINW, 0, LEAN_NLDS, true>(D, T, sp, a, w);
	return launch_one<6, 2, 4, LEAN_MINW, 0, LEAN_NLDS, true>(D, T, sp, a, w);
}
