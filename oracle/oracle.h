/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, gcc) of the reference hot path of murrayrm/ntg
 * (NTG 2.2.3): colloc.c / cost.c / constraints.c / integrator.c / ntg.c, plus the
 * two absent third-party engines it calls (PGS spline routines, NPSOL SQP).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (ntg_amd/, libntg_amd.so) never links or calls it.
 *
 * PARITY STATUS
 *   colloc/cost/constraints/integrator/bounds/linear-constraint restatement:
 *     pinned against the reference's own C code (oracle/_ref partial build,
 *     tests/golden fixtures).
 *   PGS (basis values) and NPSOL (SQP iterates): sources absent from
 *     /root/reference, reference holds no vectors -> "parity unpinned" by the
 *     reference; pinned by independent known answers.
 */
#ifndef NTG_ORACLE_H
#define NTG_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- PGS restatement (oracle/pgs.c) ---------------- */
typedef struct { int j; double deltal[20], deltar[20]; } orc_bsplvb_state;
void orc_knots(const double *brk, int l, int kpm, int m, double *t, int *n);
void orc_interv(const double *xt, int lxt, double x, int *left, int *mflag);
void orc_bsplvb(const double *t, int jhigh, int index, double x, int left,
                double *biatx, orc_bsplvb_state *st);
void orc_bsplvd(const double *t, int k, double x, int left, double *a,
                double *dbiatx, int nderiv);

/* ---------------- active variables (av.h:18-26) ---------------- */
typedef struct { int output; int deriv; } orc_AV;
#define ORC_AVINITIAL 0
#define ORC_AVTRAJECTORY 1
#define ORC_AVFINAL 2

/* ---------------- collocation structure (colloc.h:42-71) ----------------
 * blk[o][(bp*order[o] + q)*maxderiv[o] + r] = D^r B_{off+q}(bps[bp])
 *   == reference block[bp].matrix->elements[q][r]   (colloc.c:100-101) */
typedef struct {
	int nout, nbps, nz, nZ, nC;
	int *order, *mult, *ninterv, *maxderiv, *ncoef;
	int *iZ, *iz, *iC;
	double *bps;
	double **blk;
	int **off;
} orc_colloc;

orc_colloc *orc_colloc_make(int nout, double **knots, const int *ninterv,
                            const double *bps, int nbps, const int *maxderiv,
                            const int *order, const int *mult);
void orc_colloc_free(orc_colloc *cc);

/* callback types: ntg.h:81-83,90-92 */
typedef void (*orc_icf_t)(int *, int *, double *, double *, double **);
typedef void (*orc_ucf_t)(int *, int *, int *, double *, double *, double **);
typedef void (*orc_nlic_t)(int *, int *, double *, double **, double **);
typedef void (*orc_nltc_t)(int *, int *, int *, double *, double **, double **);
/* second derivatives of the nonlinear constraints (no reference counterpart: NPSOL never asks for them; used by the
 * structured Newton mode of oracle/sqp.c only): Hz (nz x nz, row-major) += sum_j t[j] d2 c_j / dz dz at one breakpoint */
typedef void (*orc_nlhess_t)(int *i, const double *t, double *Hz, double **zp);

/* one NTG problem = everything ntg() stashes in its globals (ntg.c:17-41,119-152) */
typedef struct {
	orc_colloc *cc;
	double *Z; /* calloc'd nZ, persists across calls (ntg.c:119) */
	int nlic, nltc, nlfc, nnlic, nnltc, nnlfc;
	orc_nlic_t nlicf; orc_nltc_t nltcf; orc_nlic_t nlfcf;
	int nicav, ntcav, nfcav; orc_AV *icav, *tcav, *fcav; /* constraint AVs */
	int nicf, nucf, nfcf;
	orc_icf_t icf; orc_ucf_t ucf; orc_icf_t fcf;
	int nicostav, ntcostav, nfcostav; orc_AV *icostav, *tcostav, *fcostav;
	int nclin, ncnln;
	double *A;    /* nclin x nC column-major, ld = max(nclin,1)  (ntg.c:162-206) */
	double *cJac; /* ncnln x nC column-major, persists (ntg.c:210-220) */
	double *bl, *bu; /* nC+nclin+ncnln (ntg.c:222-229) */
	orc_nlhess_t nlic_hess, nltc_hess, nlfc_hess; /* optional (NULL: Gauss-Newton terms only), see orc_nlhess_t */
	/* structured Newton mode (opts.hessian = 2): outputs per coupling group of the family and the constraint flag entries of
	 * group 0 as a bit mask (bit maxderiv*o + r); couple = 0: the family does not offer the mode (host callbacks never do) */
	int couple; unsigned long long group_mask;
	/* CPU baseline flavour (SURVEY 8d "cpu-opt"): 1 = banded, allocation-free evaluation of the running cost (no dense
	 * nbps x nC temporary, node-wise trapezoid weights); 0 = the reference's loops (cost.c:117-134).  Values agree to rounding. */
	int banded;
} orc_problem;

orc_problem *orc_problem_make(
	int nout, double *bps, int nbps, int *kninterv, double **knots, int *order,
	int *mult, int *maxderiv,
	int nlic, double **lic, int nltc, double **ltc, int nlfc, double **lfc,
	int nnlic, orc_nlic_t nlicf, int nnltc, orc_nltc_t nltcf, int nnlfc, orc_nlic_t nlfcf,
	int nicav, orc_AV *icav, int ntcav, orc_AV *tcav, int nfcav, orc_AV *fcav,
	double *lowerb, double *upperb,
	int nicf, orc_icf_t icf, int nucf, orc_ucf_t ucf, int nfcf, orc_icf_t fcf,
	int nicostav, orc_AV *icostav, int ntcostav, orc_AV *tcostav, int nfcostav, orc_AV *fcostav);
void orc_problem_free(orc_problem *p);

void orc_updateZ(double *Z, const orc_colloc *cc, const double *C, const orc_AV *av, int nav, int type);
void orc_integrate_vector(double *I, const double *f, const double *t, int n);
void orc_integrate_cols(double *I, const double *f, int rows, int cols, const double *t);
void orc_bounds(double *bbar, const double *b, int nc, int nlic, int nltc, int nlfc,
                int nnlic, int nnltc, int nnlfc, int nbps, double bigbnd);
/* NPSOL-facing callbacks (ntg.c:274-371) */
/* timed CPU baseline only: serve the per-call dense temporaries from a reused per-thread buffer (see ntg_oracle.c) */
void orc_set_scratch_reuse(int on);
int orc_scratch_reuse_on(void);
int orc_thread_cpus(int nthreads, int *cpus);
void orc_funobj(orc_problem *p, int *mode, const double *x, double *f, double *g, int *nstate);
void orc_funcon(orc_problem *p, int *mode, const double *x, double *c, double *cJac, int *nstate);
void orc_linspace(double *v, double d0, double d1, int n);
void orc_spline_interp(double *f, double x, double *knots, int ninterv, double *coefs,
                       int ncoefs, int order, int mult, int maxderiv);

/* ---------------- SQP (NPSOL replacement; oracle/sqp.c) ---------------- */
typedef struct {
	int itlim;           /* major iteration limit (<=0: NPSOL default max(50,3(n+nclin)+10 ncnln)) */
	double opttol;       /* optimality tolerance r (<=0: eps^0.8) */
	double steplimit;    /* NPSOL "step limit", default 2.0 */
	double ls_mu, ls_eta;/* sufficient decrease / curvature (line search tolerance), 1e-4 / 0.9 */
	int ls_maxfev;       /* 20 */
	int hessian;         /* 0: identity cold start (NPSOL), 1: collocation preconditioner, 2: structured Newton step for
	                      * problems with nonlinear rows (banded H0 + mu J'J + curvature, refreshed every major);
	                      * 2 behaves like 1 where it does not apply */
	int fixed_iters;     /* 1: run exactly itlim majors, no convergence exit */
	int verbose;
	int qn_memory;       /* BFGS updates kept before W restarts from W0 (the device's pair memory); <= 0: 256 */
	int banded;          /* 1: evaluate with the banded, allocation-free path (orc_problem.banded) -- timing flavour only */
	const double *warm_lam; /* NULL: multipliers of the augmented-Lagrangian rows start at 0.  Else [ncnln + nI] starting multipliers in
	                      * the internal sign (= -clambda of those rows): the warm start of a receding-horizon re-solve (ntg.h:64-68);
	                      * the structured Newton mode then skips its pass on the objective alone (mirrors ntg_solve_opts.warm_start) */
} orc_sqp_opts;
void orc_sqp_default_opts(orc_sqp_opts *o);

typedef struct {
	int inform, iters, nfev;
	double objective, pg_norm, feas;
} orc_sqp_result;

/* solve one problem in place (x: nC). clambda/istate may be NULL. */
void orc_sqp_solve(orc_problem *p, double *x, const orc_sqp_opts *o, orc_sqp_result *res,
                   double *clambda, int *istate, double *R,
                   double *trace, int trace_cap);

/* device-functor families restated as host callbacks (oracle/families.c) */
#define ORC_FAM_KINCAR 0
#define ORC_FAM_VANDERPOL 1
#define ORC_FAM_TESTFAM 2
#define ORC_FAM_OBSTACLE 3
#define ORC_FAM_QUADROTOR 4
#define ORC_FAM_MANIP 5
orc_ucf_t orc_family_ucf(int fam);
orc_icf_t orc_family_icf(int fam);
orc_icf_t orc_family_fcf(int fam);
orc_nlic_t orc_family_nlicf(int fam);
orc_nltc_t orc_family_nltcf(int fam);
orc_nlic_t orc_family_nlfcf(int fam);
orc_nlhess_t orc_family_nlic_hess(int fam);
orc_nlhess_t orc_family_nltc_hess(int fam);
orc_nlhess_t orc_family_nlfc_hess(int fam);
void orc_family_set_nout(int nout); /* thread-local nout for generic families */

/* batched CPU driver used by tests and bench.py's cpu_baseline leg */
typedef struct {
	int nout, nbps; const double *bps; const int *kninterv; const double *const *knots;
	const int *order, *mult, *maxderiv;
	int family;
	int nlic, nltc, nlfc; const double *lic, *ltc, *lfc; /* row-major [n][nz] */
	int nnlic, nnltc, nnlfc;
	int nicav, ntcav, nfcav; const orc_AV *icav, *tcav, *fcav;
	int nicf, nucf, nfcf;
	int nicostav, ntcostav, nfcostav; const orc_AV *icostav, *tcostav, *fcostav;
} orc_batch_spec;
int orc_solve_batch(const orc_batch_spec *s, int batch, const double *lowerb, const double *upperb,
                    double *x /* [batch][nC] in/out */, const orc_sqp_opts *o,
                    double *objective, int *inform, int *iters, int *nfev, int nthreads);
int orc_eval_batch(const orc_batch_spec *s, int batch, const double *x, int mode,
                   double *f, double *g, double *c, double *cJac, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
