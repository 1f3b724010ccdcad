// fam_kincar_chm.hip -- kincar instances with a compile-time channel mask (CHM = 4: the trajectory-cost active
// variables are the second derivative of every output, examples/kincar.c:133-137): the shipped example's shape with
// order-6 splines (config B, 2 outputs), two cars (4 outputs) and the headline workload (config M, 6 outputs).  Own translation unit so that it
// compiles next to fam_kincar.hip.
#include "solve_impl.hpp"
#include "eval_fast.hpp"
#include <cstdlib>

hipError_t ntg_launch_eval_kincar_chm(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const EvalArgs &a)
{
	// values and gradient only (no Jacobian outputs: the family has no constraints): the lean kernel
	IntervalEvalDims FI;
	const bool shared_grid = T.pp_rowv == 0;   // the lean kernels stage one grid per workgroup: per-problem grids take eval_kernel
	// per-problem grids: the interval kernel with wave-private tables restaged per problem (shapes that fill more than half a wave: one
	// problem per wave at a time); needs the gradient output (its store slots are unconditional)
	if (!shared_grid && !a.c && !a.jb && !a.cj && a.g && a.mode != 0 && !getenv("NTG_AMD_EVAL_V1")) {
		if (D.nout == 4 && D.ig_n == 20 && eval_interval_match(D, 4, 3, 6, 2, &FI)) return launch_eval_interval<NTG_FAM_KINCAR, 4, 2, 6, 4, 4, 20>(T, FI, a);
		if (D.nout == 6 && D.ig_n == 20 && eval_interval_match(D, 4, 3, 6, 2, &FI)) return launch_eval_interval<NTG_FAM_KINCAR, 6, 2, 6, 4, 4, 20>(T, FI, a);
	}
	if (shared_grid && !a.c && !a.jb && !a.cj && !getenv("NTG_AMD_EVAL_V1")) {   // one lane per (knot interval, pair of outputs)
		// (instances for 20 knot intervals: BASELINE's kincar configs; 16 below; other grids take the breakpoint-lane kernel)
		if (D.nout == 2 && D.ig_n == 20 && eval_interval_match(D, 4, 3, 6, 2, &FI)) return launch_eval_interval<NTG_FAM_KINCAR, 2, 2, 6, 4, 4, 20>(T, FI, a);
		if (D.nout == 4 && D.ig_n == 20 && eval_interval_match(D, 4, 3, 6, 2, &FI)) return launch_eval_interval<NTG_FAM_KINCAR, 4, 2, 6, 4, 4, 20>(T, FI, a);
		if (D.nout == 6 && D.ig_n == 20 && eval_interval_match(D, 4, 3, 6, 2, &FI)) return launch_eval_interval<NTG_FAM_KINCAR, 6, 2, 6, 4, 4, 20>(T, FI, a);
		// ... and for 16 (the second interval count with tuned solve instances: `generic_instances` of bench.py)
		if (D.nout == 2 && D.ig_n == 16 && eval_interval_match(D, 4, 3, 6, 2, &FI)) return launch_eval_interval<NTG_FAM_KINCAR, 2, 2, 6, 4, 4, 16>(T, FI, a);
		if (D.nout == 4 && D.ig_n == 16 && eval_interval_match(D, 4, 3, 6, 2, &FI)) return launch_eval_interval<NTG_FAM_KINCAR, 4, 2, 6, 4, 4, 16>(T, FI, a);
		if (D.nout == 6 && D.ig_n == 16 && eval_interval_match(D, 4, 3, 6, 2, &FI)) return launch_eval_interval<NTG_FAM_KINCAR, 6, 2, 6, 4, 4, 16>(T, FI, a);
	}
	FastEvalDims F;
	if (shared_grid && a.nt == 128 && !a.c && !a.jb && !a.cj && eval_fast_match(D, 4, 3, 128, &F)) {
		if (D.nout == 2) return launch_eval_fast<NTG_FAM_KINCAR, 2, 6, 4, 128>(D, T, F, a);
		if (D.nout == 4) return launch_eval_fast<NTG_FAM_KINCAR, 4, 6, 4, 128>(D, T, F, a);
		if (D.nout == 6) return launch_eval_fast<NTG_FAM_KINCAR, 6, 6, 4, 128>(D, T, F, a);
	}
	if (D.nout == 2) return launch_eval_small<NTG_FAM_KINCAR, 2, 6, 4>(D, T, L, a);
	if (D.nout == 4) return launch_eval_small<NTG_FAM_KINCAR, 4, 6, 4>(D, T, L, a);
	if (D.nout == 6) return launch_eval_small<NTG_FAM_KINCAR, 6, 6, 4>(D, T, L, a);
	return hipErrorInvalidValue;
}

hipError_t ntg_launch_sqp_kincar_chm(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const SolveParams &sp, const SqpArgs &a)
{
	if (D.nout == 2) return launch_sqp_small<NTG_FAM_KINCAR, 2, 6, 4>(D, T, L, sp, a);
	if (D.nout == 4) return launch_sqp_small<NTG_FAM_KINCAR, 4, 6, 4>(D, T, L, sp, a);
	if (D.nout == 6) return launch_sqp_small<NTG_FAM_KINCAR, 6, 6, 4>(D, T, L, sp, a);
	return hipErrorInvalidValue;
}
