// fam_manip.hip -- eval_kernel / sqp_kernel instances of one problem family (own translation unit: the
// families compile in parallel).  Tuned instances fix nout and the spline order at compile time.
#include "solve_impl.hpp"

// config E: 12 outputs, order 6 (2196 coefficients, 301 breakpoints): 512 lanes, five coefficients per lane
hipError_t ntg_launch_eval_manip(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const EvalArgs &a)
{
	if (a.nt == 512 && ntg_all_d(D, 3) && D.nout == 12 && ntg_uniform_order(D, 512, 5) == 6)
		return launch_eval_one<NTG_FAM_MANIP, 12, 6, 512, 5>(D, T, L, a);
	return launch_eval_generic<NTG_FAM_MANIP>(D, T, L, a);
}

hipError_t ntg_launch_sqp_manip(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const SolveParams &sp, const SqpArgs &a)
{
#ifdef NTG_SLIM   // experiments (tools/mkvariant2.sh ... -DNTG_SLIM): config E's Newton-mode instances only -- a fifth of the compile time
	if (a.nt == 512 && a.big && sp.hessian == 3) return launch_sqp_one<NTG_FAM_MANIP, 12, 6, 512, 5, true, true, 5, true, true>(D, T, L, sp, a);
	if (a.nt == 512 && a.big && sp.hessian == 2) return launch_sqp_one<NTG_FAM_MANIP, 12, 6, 512, 5, true, true, 5, true>(D, T, L, sp, a);
	return hipErrorInvalidValue;
#else
	if (a.nt == 512 && ntg_all_d(D, 3) && D.nout == 12 && ntg_uniform_order(D, 512, 5) == 6) {
		if (sp.hessian == 3) {   // QP-based SQP step on the band model (qpdual.hpp)
			if (a.big && ntg_chm_match(D, 5, 3)) return launch_sqp_one<NTG_FAM_MANIP, 12, 6, 512, 5, true, true, 5, true, true>(D, T, L, sp, a);
			return launch_sqp_generic<NTG_FAM_MANIP>(D, T, L, sp, a);
		}
		if (sp.hessian == 2) {   // structured Newton mode (newton.hpp)
			if (a.big && ntg_chm_match(D, 5, 3)) return launch_sqp_one<NTG_FAM_MANIP, 12, 6, 512, 5, true, true, 5, true>(D, T, L, sp, a);
			if (a.big) return launch_sqp_one<NTG_FAM_MANIP, 12, 6, 512, 5, true, true, 0, true>(D, T, L, sp, a);
			return launch_sqp_one<NTG_FAM_MANIP, 12, 6, 512, 5, false, true, 0, true>(D, T, L, sp, a);
		}
		// channel mask 5: joint angles (constraints) and joint accelerations (cost) of every joint
		if (a.big && ntg_chm_match(D, 5, 3)) return launch_sqp_one<NTG_FAM_MANIP, 12, 6, 512, 5, true, true, 5>(D, T, L, sp, a);
		if (a.big) return launch_sqp_one<NTG_FAM_MANIP, 12, 6, 512, 5, true>(D, T, L, sp, a);
		return launch_sqp_one<NTG_FAM_MANIP, 12, 6, 512, 5, false>(D, T, L, sp, a);
	}
	return launch_sqp_generic<NTG_FAM_MANIP>(D, T, L, sp, a);
#endif
}
