// fam_quadrotor.hip -- eval_kernel / sqp_kernel instances of one problem family (own translation unit: the
// families compile in parallel).  Tuned instances fix nout and the spline order at compile time.
#include "solve_impl.hpp"

// config D: 4 outputs, order 8, maxderiv 5 (656 coefficients, 201 breakpoints): 256 lanes, three coefficients per lane
hipError_t ntg_launch_eval_quadrotor(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const EvalArgs &a)
{
	if (a.nt == 256 && ntg_all_d(D, 5) && D.nout == 4 && ntg_uniform_order(D, 256, 4) == 8)
		return launch_eval_one<NTG_FAM_QUADROTOR, 4, 8, 256, 4>(D, T, L, a);
	if (a.nt == 512 && ntg_all_d(D, 5) && D.nout == 4 && ntg_uniform_order(D, 512, 4) == 8)
		return launch_eval_one<NTG_FAM_QUADROTOR, 4, 8, 512, 4>(D, T, L, a);   // one workgroup per CU: more waves per workgroup
	return launch_eval_generic<NTG_FAM_QUADROTOR>(D, T, L, a);
}

hipError_t ntg_launch_sqp_quadrotor(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const SolveParams &sp, const SqpArgs &a)
{
	if (a.nt == 256 && ntg_all_d(D, 5) && D.nout == 4 && ntg_uniform_order(D, 256, 4) == 8) {
		if (sp.hessian == 3) {   // QP-based SQP step on the band model (qpdual.hpp)
			if (!a.big) return launch_sqp_one<NTG_FAM_QUADROTOR, 4, 8, 256, 4, false, true, 0, true, true>(D, T, L, sp, a);
			return launch_sqp_generic<NTG_FAM_QUADROTOR>(D, T, L, sp, a);
		}
		if (sp.hessian == 2) {   // structured Newton mode (newton.hpp)
			if (a.big) return launch_sqp_one<NTG_FAM_QUADROTOR, 4, 8, 256, 4, true, true, 0, true>(D, T, L, sp, a);
			return launch_sqp_one<NTG_FAM_QUADROTOR, 4, 8, 256, 4, false, true, 0, true>(D, T, L, sp, a);
		}
		if (a.big) return launch_sqp_one<NTG_FAM_QUADROTOR, 4, 8, 256, 4, true>(D, T, L, sp, a);
		return launch_sqp_one<NTG_FAM_QUADROTOR, 4, 8, 256, 4, false>(D, T, L, sp, a);
	}
	return launch_sqp_generic<NTG_FAM_QUADROTOR>(D, T, L, sp, a);
}
