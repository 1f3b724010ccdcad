// fam_testfam.hip -- eval_kernel / sqp_kernel instances of one problem family (own translation unit: the
// families compile in parallel).  Tuned instances fix nout and the spline order at compile time.
#include "solve_impl.hpp"

hipError_t ntg_launch_eval_testfam(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const EvalArgs &a)
{
	const bool small = (a.nt == 128 || a.nt == 256) && ntg_all_d(D, 3);
	const int ku = ntg_uniform_order(D, a.nt, 4);
	(void)ku;
	if (small && D.nout == 3 && D.nC <= 4 * a.nt && D.nI == 0) return launch_eval_small<NTG_FAM_TESTFAM, 3, 0>(D, T, L, a);
	return launch_eval_generic<NTG_FAM_TESTFAM>(D, T, L, a);
}

hipError_t ntg_launch_sqp_testfam(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const SolveParams &sp, const SqpArgs &a)
{
	const bool small = (a.nt == 128 || a.nt == 256) && ntg_all_d(D, 3);
	const int ku = ntg_uniform_order(D, a.nt, 4);
	(void)ku;
	if (small && !a.big && D.nout == 3 && D.nC <= 4 * a.nt && D.nI == 0) return launch_sqp_small<NTG_FAM_TESTFAM, 3, 0>(D, T, L, sp, a);
	return launch_sqp_generic<NTG_FAM_TESTFAM>(D, T, L, sp, a);
}
