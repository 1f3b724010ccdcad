/*
 * include/ntg_amd.h -- C ABI of the MI355X-native batched NTG engine (libntg_amd.so).
 *
 * Plain pointers and sizes only.  Pointers named d_* are DEVICE pointers (HBM of the GPU the
 * plan was created on); everything else is host memory.  `stream` is a hipStream_t passed as
 * void* (NULL = default stream).  All entry points return 0 on success, a negative NTG_E_*
 * code otherwise; ntg_last_error() returns a message.  Nothing here falls back to the CPU:
 * without a usable gfx950 device every call fails with NTG_E_NODEVICE.
 *
 * Which reference interface each entry point replaces (files under /root/reference/src):
 *   ntg_plan_create      ntg.c:114-229   ConcatCollocMatrix + LinearConstraintsMatrix + the
 *                                        argument stash into file-scope globals (ntg.c:119-152)
 *                        colloc.c:57-117 CollocMatrix: knots_/interv_/bsplvd_ at every breakpoint
 *   ntg_plan_tables      colloc.h:42-71  read-back of Block.matrix / Block.offset / A
 *   ntg_basis_batch      colloc.c:92-111 the same basis evaluation for many grids at once
 *   ntg_plan_set_grids   ntg.c:114-229 per problem: own knots and breakpoints for every problem of a batch
 *   ntg_batch_eval       ntg.c:274-371   NPfunobj + NPfuncon (cost.c, constraints.c, integrator.c)
 *   ntg_batch_bounds     constraints.c:5-33 bounds()
 *   ntg_batch_solve      ntg.c:237-253   the npsol_() call, for `batch` problems at once
 *   npsolCostFunction / npsolConstraintFunction
 *                        ntg.c:274-280 / ntg.c:337-346  (static NPfunobj / NPfuncon; exported
 *                        under the names BASELINE.json uses, same Fortran-style signature)
 *   ntg_batch_interp     colloc.c:449-484 SplineInterp, for a batch
 *   ntg_batch_kincar_reverse  examples/kincar.c:68-92 kincar_flat_reverse (the example's flat-to-state map), for a batch
 *   ntg(), npsoloption(), linspace(), SplineInterp(), matrix helpers: see include/ntg.h
 */
#ifndef NTG_AMD_H
#define NTG_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define NTG_MAX_OUT 16     /* outputs per problem */
#define NTG_MAX_ORDER 10   /* spline order k */
#define NTG_MAX_NZ 64      /* sum of maxderiv (active-variable masks are 64-bit) */

#define NTG_E_NODEVICE (-1)
#define NTG_E_BADARG   (-2)
#define NTG_E_HIP      (-3)
#define NTG_E_UNSUPPORTED (-4)

#define NTG_INF_BOUND 1e20   /* NPSOL's "infinite bound": use for one-sided constraints */

/* problem families = device functors for the user callbacks of ntg.h:81-83,90-92 */
#define NTG_FAM_KINCAR 0     /* examples/kincar.c:105-117 generalised to nout outputs */
#define NTG_FAM_VANDERPOL 1  /* examples/vanderpol.c:206-241 */
#define NTG_FAM_TESTFAM 2    /* synthetic, all six callback slots */
#define NTG_FAM_OBSTACLE 3   /* kincar cost + circular-obstacle trajectory constraint (x-20)^2+(y-0.5)^2 >= r^2 */
#define NTG_FAM_QUADROTOR 4  /* 4 outputs x,y,z,yaw, maxderiv 5: snap^2 + yaw''^2; rows: thrust^2 = x''^2+y''^2+(z''+g)^2, speed^2 */
#define NTG_FAM_MANIP 5      /* 3 joints per planar arm, maxderiv 3: sum q''^2; one tip-height row sin(qa)+sin(qa+qb)+sin(qa+qb+qc) per arm */
#define NTG_FAM_HOST (-1)    /* host function pointers (ntg() drop-in path only) */

typedef struct { int output; int deriv; } ntg_av; /* == AV of av.h:22-26 */

/* Everything ntg() takes that is common to a batch (ntg.h:72-99). lic/ltc/lfc are row-major
 * [n][nz] with nz = sum(maxderiv) (the examples' DoubleMatrix layout). */
typedef struct {
	int nout, nbps;
	const double *bps;
	const int *kninterv;
	const double *const *knots;
	const int *order, *mult, *maxderiv;
	int family;
	int nlic, nltc, nlfc;
	const double *lic, *ltc, *lfc;
	int nnlic, nnltc, nnlfc;
	int nicav, ntcav, nfcav;
	const ntg_av *icav, *tcav, *fcav;
	int nicf, nucf, nfcf;
	int nicostav, ntcostav, nfcostav;
	const ntg_av *icostav, *tcostav, *fcostav;
	/* optional, may be NULL: [nlic+nltc+nlfc] flags, non-zero = this linear row is an INEQUALITY row (lower <= row <= upper)
	 * for every problem of the batch.  Rows not flagged must have lower == upper in every problem (they are kept satisfied
	 * by projection); a problem that breaks this returns inform 9.  ntg() itself needs no flags: it reads the bounds. */
	const int *lin_ineq;
} ntg_spec;

typedef struct {
	int itlim;          /* major iteration limit; <=0: max(50, 3(n+nclin)+10 ncnln) (NPSOL default) */
	double opttol;      /* optimality tolerance r; <=0: eps^0.8 */
	double steplimit;   /* NPSOL "step limit", 2.0 */
	double ls_mu, ls_eta; /* 1e-4, 0.9 (NPSOL "line search tolerance") */
	int ls_maxfev;      /* 20 */
	int hessian;        /* 0 identity cold start (NPSOL), 1 collocation preconditioner (ignored when the cost model is
	                     * singular on the null space of the equality rows, e.g. a plan without equality rows),
	                     * 2 structured Newton step for nonlinear trajectory rows: band model of the augmented Lagrangian's Hessian,
	                     *   assembled and factored on the matrix cores at every major (families with second-order blocks: obstacle,
	                     *   quadrotor, manipulator; acts as 1 where it does not apply, e.g. more coupling groups than a
	                     *   workgroup has wavefronts) -- the robust mode for BASELINE's configs D and E,
	                     * 3 QP-based SQP step on the same band model: per major iteration the inequality QP on the linearised rows (what NPSOL does with the
	                     *   Jacobian ntg() hands it, ntg.c:217-220,250-253), solved through its dual by an active-set method on at most 16 (32: single-group plans
	                     *   on large workgroups) rows per coupling group, l1 merit function; a problem whose working set does not fit continues in mode 2 by itself.  Config E:
	                     *   17 majors instead of 60, 2.9 x mode 2's rate.  With warm_start the QP's first working set is the rows the carried-over
	                     *   multipliers name (no pass on the objective alone).  Acts as 1 where the band model does not apply. */
	int fixed_iters;    /* 1: exactly itlim majors, no convergence exit */
	int block_threads;  /* 0 = auto (128/256/512) */
	int qn_memory;      /* quasi-Newton updates kept before the approximation restarts from W0; <= 0: 256 */
	int warm_start;     /* plans with nonlinear / inequality rows: 1 = start the augmented-Lagrangian loop from the multiplier estimates the
	                     * previous ntg_batch_solve of the same batch left in d_work (the use NPSOL's clambda was meant for, ntg.h:64-68:
	                     * receding-horizon re-solves; ntg_batch_mpc_run shifts them with the horizon), 0 = multipliers start at 0.
	                     * The structured Newton mode then skips its pass on the objective alone. */
} ntg_solve_opts;

typedef struct ntg_plan ntg_plan;

int ntg_device_count(void);
const char *ntg_last_error(void);
void ntg_default_opts(ntg_solve_opts *o);

/* Build the device-resident, batch-shared part of a problem: basis blocks and offsets (HIP
 * basis kernel), banded linear-constraint rows A, (A A')^-1, optional preconditioner. */
int ntg_plan_create(const ntg_spec *spec, int device, ntg_plan **out);
void ntg_plan_destroy(ntg_plan *p);

/* sizes: nC, nz, nclin, ncnln, nbounds, sumk, njrows (banded Jacobian rows), nblk (doubles in blk) */
int ntg_plan_dims(const ntg_plan *p, int *nC, int *nz, int *nclin, int *ncnln, int *nbounds,
                  int *sumk, int *nblk);
/* read the setup tables back to the host (any pointer may be NULL):
 *   blk  outputs concatenated, per output [bp][q][r]         (reference block[bp].matrix->elements[q][r])
 *   off  [nout][nbps]                                         (reference block[bp].offset)
 *   A    dense column-major nclin x nC, ld = nclin            (what ntg.c:206 hands to NPSOL) */
int ntg_plan_tables(const ntg_plan *p, double *blk, int *off, double *A);
/* workspace (bytes) ntg_batch_solve needs for `batch` problems with these options */
long long ntg_batch_workspace_bytes(const ntg_plan *p, int batch, const ntg_solve_opts *o);

/* bounds(): d_lower/d_upper [batch][nbounds] -> d_bl/d_bu [batch][nC+nclin+ncnln] */
int ntg_batch_bounds(const ntg_plan *p, int batch, const double *d_lower, const double *d_upper,
                     double *d_bl, double *d_bu, void *stream);

/* funobj+funcon for `batch` coefficient vectors d_x [batch][nC], mode as in NPSOL (0 values,
 * 1 gradients, 2 both).  Outputs (any may be NULL): d_f [batch], d_g [batch][nC],
 * d_c [batch][ncnln] (rows: initial; trajectory constraint-major x breakpoint; final),
 * d_jband [batch][ncnln][sumk] banded Jacobian rows (for output o the entries
 * koff[o]..koff[o]+k_o-1 sit in columns iC[o]+off[o][bp(row)]+q),
 * d_cjac [batch][nC][ncnln] = dense column-major (ld = ncnln) Jacobian as NPSOL sees it. */
int ntg_batch_eval(const ntg_plan *p, int batch, const double *d_x, int mode,
                   double *d_f, double *d_g, double *d_c, double *d_jband, double *d_cjac,
                   void *stream);

/* Solve `batch` problems: d_x [batch][nC] in/out (initial guess -> solution, ntg.c:109),
 * d_lower/d_upper [batch][nbounds].  Outputs (may be NULL): d_objective, d_inform, d_iters,
 * d_nfev [batch]; d_clambda [batch][nC+nclin+ncnln].  d_work: ntg_batch_workspace_bytes(). */
int ntg_batch_solve(const ntg_plan *p, int batch, const double *d_lower, const double *d_upper,
                    double *d_x, const ntg_solve_opts *o,
                    double *d_objective, int *d_inform, int *d_iters, int *d_nfev,
                    double *d_clambda, void *d_work, long long work_bytes, void *stream);
/* name of the solve kernel, for bench/roofline bookkeeping: the general one, and the one ntg_batch_solve launches for (plan, batch,
 * options) -- "sqp_wave_kernel" (one wavefront per problem: the kincar class of BASELINE's configs B, C, M) or "sqp_kernel" */
const char *ntg_solve_kernel_name(void);
const char *ntg_batch_solve_kernel(const ntg_plan *p, int batch, const ntg_solve_opts *o);

/* bsplvd at every collocation point for `ngrids` different grids of one spline spec
 * (per-problem horizons): d_knots [ngrids][ninterv+1], d_bps [ngrids][nbps] ->
 * d_blk [ngrids][nbps][order][maxderiv], d_off [ngrids][nbps]. */
int ntg_basis_batch(int ngrids, int ninterv, int order, int mult, int maxderiv, int nbps,
                    const double *d_knots, const double *d_bps, double *d_blk, int *d_off,
                    void *stream);

/* Per-problem grids (free final time / per-problem horizons): every problem of a batch gets its own break sequence and breakpoints --
 * the setup phase of ntg() (ntg.c:114-229: CollocMatrix per output, colloc.c:57-117; LinearConstraintsMatrix, constraints.c:198-261)
 * run per problem, as the reference runs it per call.  d_knots [batch][ninterv+1], d_bps [batch][nbps] (device).  The combinatorial
 * structure must be the plan's: one basis class, and every breakpoint in the same knot interval as in the plan's grid (checked;
 * NTG_E_BADARG otherwise) -- the index tables stay shared, the VALUES (basis blocks, trapezoid weights, linear-constraint rows,
 * (A A')^-1, projector, with_precond != 0: the preconditioner blocks) become per problem.  Nonlinear rows are allowed (free final time
 * with obstacle / thrust / speed rows: their evaluation and the augmented-Lagrangian solve read the same per-problem tables); linear
 * inequality rows are not (NTG_E_UNSUPPORTED).
 * Afterwards ntg_batch_eval / ntg_batch_solve / ntg_batch_interp of exactly `batch` problems use these grids (hessian = 2 / 3 included: the band model's cost part is built per grid;
 * ntg_batch_interp then takes d_times as [batch][ntimes]: every problem at its own times; ntg_batch_mpc_shift and ntg_batch_mpc_run
 * re-pin with every problem's own basis blocks) until ntg_plan_clear_grids(). */
int ntg_plan_set_grids(ntg_plan *p, int batch, const double *d_knots, const double *d_bps, int with_precond, void *stream);
void ntg_plan_clear_grids(ntg_plan *p);

/* SplineInterp (colloc.c:449-484) for a whole batch: the flat flag of every problem at ntimes points in time shared by
 * the batch (d_times [ntimes], inside the knot range of every output) -> d_z [batch][ntimes][nz], entry iz[o]+r =
 * D^r z_o(t).  This is the input of a flat-to-state map such as kincar_flat_reverse (kincar.c:68-92). */
int ntg_batch_interp(const ntg_plan *p, int batch, const double *d_x, int ntimes, const double *d_times, double *d_z,
                     void *stream);
/* The same with the layout of d_times stated by the caller instead of implied by the plan's state: problem b reads its ntimes points at
 * d_times + b * times_stride.  times_stride = 0: one time vector shared by the batch (allowed with and without per-problem grids);
 * times_stride >= ntimes: per-problem times, only with per-problem grids (NTG_E_BADARG otherwise).  ntg_batch_interp() is this call with
 * times_stride = 0 on the plan's grid and = ntimes after ntg_plan_set_grids(). */
int ntg_batch_interp_strided(const ntg_plan *p, int batch, const double *d_x, int ntimes, const double *d_times,
                             long long times_stride, double *d_z, void *stream);

/* The flat-to-state map of the kinematic car for a whole ntg_batch_interp result (examples/kincar.c:68-92 kincar_flat_reverse,
 * called per sample by the example's output loop, kincar.c:392-406): d_z [batch][ntimes][nz] -> d_state [batch][ntimes][ncars][5] =
 * x, y, theta, v, delta for every car (ncars = nout / 2; outputs 2c, 2c+1 are the rear-axle position of car c).  reverse_gear != 0
 * is the reference's dir == 'r'.  kincar-family plans only. */
int ntg_batch_kincar_reverse(const ntg_plan *p, int batch, int ntimes, const double *d_z, double wheelbase, int reverse_gear,
                             double *d_state, void *stream);

/* Receding-horizon step (the warm-start use NPSOL's istate/clambda/R were meant for, ntg.h:64-68):
 * re-pin the linear initial-constraint bounds of every problem to the flat flag of its current
 * solution at breakpoint shift_bp, and shift the coefficients by shift_knots knot intervals
 * (tail = last coefficient) as the next initial guess.  d_x, d_lower, d_upper are updated in place.  After ntg_plan_set_grids: with
 * every problem's own basis blocks (the batch must be the grids'). */
int ntg_batch_mpc_shift(const ntg_plan *p, int batch, int shift_bp, int shift_knots, double *d_x,
                        double *d_lower, double *d_upper, void *stream);

/* nsteps receding-horizon steps (solve, then ntg_batch_mpc_shift) without returning to the caller in between; after the
 * first step the (solve, shift) pair is replayed as a hipGraph.  d_inform [batch] receives the last step's inform,
 * d_notconv [1] (may be NULL; zero it first) accumulates the number of problems whose re-solve did not end with inform 0.
 * With stream == NULL the run uses a private stream and returns when it has finished. */
/* The multiplier part of the receding-horizon step: the estimates of the trajectory rows kept in d_work move shift_bp breakpoints towards the
 * start of the horizon (row (j, i) <- row (j, i + shift_bp), the tail starts at 0), ready for a solve with warm_start = 1.
 * ntg_batch_mpc_run does this itself when the options ask for a warm start; callers that run their own loop call it after
 * ntg_batch_mpc_shift.  No-op for plans without such rows. */
int ntg_batch_mpc_shift_multipliers(const ntg_plan *p, int batch, int shift_bp, const ntg_solve_opts *o, void *d_work, long long work_bytes,
                                    void *stream);

int ntg_batch_mpc_run(const ntg_plan *p, int batch, int nsteps, int shift_bp, int shift_knots, double *d_x,
                      double *d_lower, double *d_upper, const ntg_solve_opts *o, int *d_inform, int *d_notconv,
                      void *d_work, long long work_bytes, void *stream);

/* ntg_open(): everything ntg() does before it calls npsol_ (ntg.c:114-229), with ntg()'s own
 * argument list minus initialguess, the bounds and the outputs; the problem stays current until
 * ntg_close().  For external SQP/IPOPT drivers (Pending:9) that iterate on their own and only need
 * the two callbacks below. */
int ntg_open(
	int nout, double *bps, int nbps, int *kninterv, double **knots, int *order, int *mult, int *max_deriv,
	int nlic, double **lic, int nltc, double **ltc, int nlfc, double **lfc,
	int nnlic, void (*nlicf)(int *, int *, double *, double **, double **),
	int nnltc, void (*nltcf)(int *, int *, int *, double *, double **, double **),
	int nnlfc, void (*nlfcf)(int *, int *, double *, double **, double **),
	int ninitialconstrav, ntg_av *initialconstrav, int ntrajectoryconstrav, ntg_av *trajectoryconstrav,
	int nfinalconstrav, ntg_av *finalconstrav,
	int nicf, void (*icf)(int *, int *, double *, double *, double **),
	int nucf, void (*ucf)(int *, int *, int *, double *, double *, double **),
	int nfcf, void (*fcf)(int *, int *, double *, double *, double **),
	int ninitialcostav, ntg_av *initialcostav, int ntrajectorycostav, ntg_av *trajectorycostav,
	int nfinalcostav, ntg_av *finalcostav);
void ntg_close(void);

/* NPSOL-facing callbacks of the single-problem drop-in (valid while ntg() is running or between
 * ntg_open() and ntg_close()).  Host pointers, Fortran conventions (scalars by pointer),
 * cJac column-major ldJ x n. */
void npsolCostFunction(int *mode, int *n, double *x, double *f, double *g, int *nstate);
void npsolConstraintFunction(int *mode, int *ncnln, int *n, int *ldJ, int *needc, double *x,
                             double *c, double *cJac, int *nstate);

#ifdef __cplusplus
}
#endif
#endif
