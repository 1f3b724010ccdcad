// fam_obstacle.hip -- eval_kernel / sqp_kernel instances of one problem family (own translation unit: the
// families compile in parallel).  Tuned instances fix nout and the spline order at compile time.
#include "solve_impl.hpp"

hipError_t ntg_launch_eval_obstacle(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const EvalArgs &a)
{
	const bool small = (a.nt == 128 || a.nt == 256) && ntg_all_d(D, 3);
	const int ku = ntg_uniform_order(D, a.nt, 4);
	(void)ku;
	if (small && D.nout == 2 && ku == 6) return launch_eval_small<NTG_FAM_OBSTACLE, 2, 6>(D, T, L, a);
	return launch_eval_generic<NTG_FAM_OBSTACLE>(D, T, L, a);
}

hipError_t ntg_launch_sqp_obstacle(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const SolveParams &sp, const SqpArgs &a)
{
	const bool small = (a.nt == 128 || a.nt == 256) && ntg_all_d(D, 3);
	const int ku = ntg_uniform_order(D, a.nt, 4);
	(void)ku;
	if (small && !a.big && D.nout == 2 && ku == 6 && sp.hessian == 3) {   // QP-based SQP step on the band model (qpdual.hpp)
		if (a.nt == 128) return launch_sqp_one<NTG_FAM_OBSTACLE, 2, 6, 128, 4, false, true, 0, true, true>(D, T, L, sp, a);
		return launch_sqp_one<NTG_FAM_OBSTACLE, 2, 6, 256, 4, false, true, 0, true, true>(D, T, L, sp, a);
	}
	if (small && !a.big && D.nout == 2 && ku == 6 && sp.hessian == 2) {   // structured Newton mode (newton.hpp)
		if (a.nt == 128) return launch_sqp_one<NTG_FAM_OBSTACLE, 2, 6, 128, 4, false, true, 0, true>(D, T, L, sp, a);
		return launch_sqp_one<NTG_FAM_OBSTACLE, 2, 6, 256, 4, false, true, 0, true>(D, T, L, sp, a);
	}
	if (small && !a.big && D.nout == 2 && ku == 6) return launch_sqp_small<NTG_FAM_OBSTACLE, 2, 6>(D, T, L, sp, a);
	return launch_sqp_generic<NTG_FAM_OBSTACLE>(D, T, L, sp, a);
}
