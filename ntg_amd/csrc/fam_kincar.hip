// fam_kincar.hip -- eval_kernel / sqp_kernel instances of one problem family (own translation unit: the
// families compile in parallel).  Tuned instances fix nout and the spline order at compile time.
#include "solve_impl.hpp"

// fam_kincar_chm.hip: 2 or 6 outputs of order 6 whose cost active variables are the second derivative of every output
hipError_t ntg_launch_eval_kincar_chm(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const EvalArgs &a);
hipError_t ntg_launch_sqp_kincar_chm(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const SolveParams &sp, const SqpArgs &a);

hipError_t ntg_launch_eval_kincar(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const EvalArgs &a)
{
	const bool small = (a.nt == 128 || a.nt == 256) && ntg_all_d(D, 3);
	const int ku = ntg_uniform_order(D, a.nt, 4);
	const bool chm = ntg_chm_match(D, 4, 3);
	(void)ku;
	// channel-mask instances: the cost's active variables are the second derivative of every output (kincar.c:133-137)
	if (small && chm && ku == 6 && (D.nout == 2 || D.nout == 4 || D.nout == 6)) return ntg_launch_eval_kincar_chm(D, T, L, a);
	if (small && D.nout == 2 && ku == 6) return launch_eval_small<NTG_FAM_KINCAR, 2, 6>(D, T, L, a);
	if (small && D.nout == 6 && ku == 6) return launch_eval_small<NTG_FAM_KINCAR, 6, 6>(D, T, L, a);
	if (small && D.nout == 2 && ku == 5) return launch_eval_small<NTG_FAM_KINCAR, 2, 5>(D, T, L, a);
	return launch_eval_generic<NTG_FAM_KINCAR>(D, T, L, a);
}

hipError_t ntg_launch_sqp_kincar(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const SolveParams &sp, const SqpArgs &a)
{
	const bool small = (a.nt == 128 || a.nt == 256) && ntg_all_d(D, 3);
	const int ku = ntg_uniform_order(D, a.nt, 4);
	const bool chm = ntg_chm_match(D, 4, 3);
	(void)ku;
	if (small && !a.big && chm && ku == 6 && (D.nout == 2 || D.nout == 4 || D.nout == 6)) return ntg_launch_sqp_kincar_chm(D, T, L, sp, a);
	if (small && !a.big && D.nout == 2 && ku == 6) return launch_sqp_small<NTG_FAM_KINCAR, 2, 6>(D, T, L, sp, a);
	if (small && !a.big && D.nout == 6 && ku == 6) return launch_sqp_small<NTG_FAM_KINCAR, 6, 6>(D, T, L, sp, a);
	if (small && !a.big && D.nout == 2 && ku == 5) return launch_sqp_small<NTG_FAM_KINCAR, 2, 5>(D, T, L, sp, a);
	return launch_sqp_generic<NTG_FAM_KINCAR>(D, T, L, sp, a);
}
