/*
 * oracle/sqp.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, see oracle.h).
 *
 * The reference hands the assembled problem to NPSOL (ntg.c:237-253), a proprietary
 * dense SQP code whose source is absent from /root/reference (npsol/README:1-13,
 * version unstated).  This file restates the published structure of NPSOL's
 * algorithm (Gill, Murray, Saunders, Wright, "User's guide for NPSOL 5.0", SOL 86-6R)
 * for the problem class the shipped examples and the headline metric use:
 *     minimise F(x)   subject to   A x = b   (linear equalities, lower == upper)
 * i.e.  - linear constraints are satisfied first and kept satisfied (NPSOL §"linear
 *         feasibility phase"); search directions lie in null(A);
 *       - quasi-Newton (BFGS) approximation of the (projected) Hessian, cold start
 *         from the identity; update skipped if y's is not sufficiently positive;
 *       - line search on F with sufficient decrease mu and line-search tolerance eta
 *         (NPSOL default eta = 0.9), safeguarded cubic interpolation, initial step 1
 *         limited by "step limit" (default 2.0): |alpha p| <= steplimit (1+|x|);
 *       - convergence test  alpha|p| <= sqrt(r)(1+|x|)  and
 *         |Z'g| <= sqrt(r)(1+max(1+|F|,|g|)),  r = optimality tolerance (eps^0.8);
 *       - inform: 0 optimal, 4 major-iteration limit, 6 line search failed,
 *         9 invalid/unsupported input (ntg.h / SURVEY.md §8 a14).
 * The inverse form W = H^-1 is kept instead of NPSOL's Cholesky factor R because
 * that is the form the MI355X kernels use (matvec + rank-2, no triangular solves);
 * the two are the same matrix in exact arithmetic.  Here W is a plain dense n x n
 * array updated in place -- the device keeps the same W as W0 + sum of rank-2 terms.
 *
 * Extension beyond NPSOL (opts.hessian = 1): W0 = Z (Z' H0 Z)^-1 Z' with
 * H0 = sum over cost active variables of trapezoid-weighted m m' (collocation
 * preconditioner).  hessian = 0 is the NPSOL-equivalent mode.
 *
 * PARITY STATUS: NPSOL iterates are unpinned (no source, no vectors).  Pinned at the
 * optimum by closed-form KKT solutions / independent solvers (tests/).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <stdio.h>
#include "oracle.h"

void orc_sqp_default_opts(orc_sqp_opts *o)
{
	o->itlim = 0; o->opttol = 0.0; o->steplimit = 2.0; o->ls_mu = 1e-4; o->ls_eta = 0.9;
	o->ls_maxfev = 20; o->hessian = 0; o->fixed_iters = 0; o->verbose = 0; o->qn_memory = 0;
}

/* ---------------- dense helpers (column-major, ld explicit) ---------------- */
#define M_(a, ld, r, c) ((a)[(size_t)(c) * (ld) + (r)])
static double dot_(const double *a, const double *b, int n)
{ int i; double s = 0.0; for (i = 0; i < n; i++) s += a[i] * b[i]; return s; }
static double nrm2_(const double *a, int n) { return sqrt(dot_(a, a, n)); }

/* lower Cholesky in place, returns 0 on success */
static int chol_(double *a, int n)
{
	int i, j, k;
	for (j = 0; j < n; j++) {
		double d = M_(a, n, j, j);
		for (k = 0; k < j; k++) d -= M_(a, n, j, k) * M_(a, n, j, k);
		if (!(d > 0.0)) return 1;
		d = sqrt(d); M_(a, n, j, j) = d;
		for (i = j + 1; i < n; i++) {
			double s = M_(a, n, i, j);
			for (k = 0; k < j; k++) s -= M_(a, n, i, k) * M_(a, n, j, k);
			M_(a, n, i, j) = s / d;
		}
	}
	return 0;
}
static void chol_solve_(const double *L, int n, double *b)
{
	int i, k;
	for (i = 0; i < n; i++) { double s = b[i]; for (k = 0; k < i; k++) s -= M_(L, n, i, k) * b[k]; b[i] = s / M_(L, n, i, i); }
	for (i = n - 1; i >= 0; i--) { double s = b[i]; for (k = i + 1; k < n; k++) s -= M_(L, n, k, i) * b[k]; b[i] = s / M_(L, n, i, i); }
}

/* ---------------- line search state machine ----------------
 * strong Wolfe (Nocedal & Wright alg. 3.5/3.6) with safeguarded cubic interpolation */
typedef struct {
	double phi0, dphi0, mu, eta, amax;
	int stage, nfev, maxfev;
	double a_prev, phi_prev, dphi_prev;
	double a_lo, phi_lo, dphi_lo, a_hi, phi_hi, dphi_hi;
	double a;
} ls_t;

static double cubic_min(double a0, double f0, double g0, double a1, double f1, double g1)
{
	double d1 = g0 + g1 - 3.0 * (f0 - f1) / (a0 - a1);
	double disc = d1 * d1 - g0 * g1, d2, den;
	if (!(disc >= 0.0)) return NAN;
	d2 = sqrt(disc); if (a1 < a0) d2 = -d2;
	den = g1 - g0 + 2.0 * d2;
	if (den == 0.0 || !isfinite(den)) return NAN;
	return a1 - (a1 - a0) * (g1 + d2 - d1) / den;
}
static void ls_init(ls_t *s, double phi0, double dphi0, double a1, double amax, double mu, double eta, int maxfev)
{
	s->phi0 = phi0; s->dphi0 = dphi0; s->mu = mu; s->eta = eta; s->amax = amax;
	s->stage = 0; s->nfev = 0; s->maxfev = maxfev;
	s->a_prev = 0.0; s->phi_prev = phi0; s->dphi_prev = dphi0;
	s->a = a1;
}
static double zoom_trial(const ls_t *s)
{
	double lo = s->a_lo < s->a_hi ? s->a_lo : s->a_hi, hi = s->a_lo < s->a_hi ? s->a_hi : s->a_lo;
	double w = hi - lo, c = cubic_min(s->a_lo, s->phi_lo, s->dphi_lo, s->a_hi, s->phi_hi, s->dphi_hi);
	if (!isfinite(c)) return 0.5 * (lo + hi);
	if (c < lo + 1e-5 * w) c = lo + 1e-5 * w;
	if (c > hi - 1e-5 * w) c = hi - 1e-5 * w;
	return c;
}
/* returns 0: evaluate at s->a; 1: accept the point just evaluated; 2: evaluate at s->a and
 * accept unconditionally; -1: failure */
static int ls_step(ls_t *s, double phi, double dphi)
{
	double a = s->a;
	int armijo = (phi <= s->phi0 + s->mu * a * s->dphi0);
	s->nfev++;
	if (s->stage == 0) {
		if (!armijo || (s->nfev > 1 && !(phi < s->phi_prev))) {
			s->a_lo = s->a_prev; s->phi_lo = s->phi_prev; s->dphi_lo = s->dphi_prev;
			s->a_hi = a; s->phi_hi = phi; s->dphi_hi = dphi;
			s->stage = 1;
		} else if (fabs(dphi) <= -s->eta * s->dphi0) {
			return 1;
		} else if (dphi >= 0.0) {
			s->a_lo = a; s->phi_lo = phi; s->dphi_lo = dphi;
			s->a_hi = s->a_prev; s->phi_hi = s->phi_prev; s->dphi_hi = s->dphi_prev;
			s->stage = 1;
		} else {
			double c, an;
			if (a >= s->amax || s->nfev >= s->maxfev) return 1; /* step limit / budget: Armijo holds */
			c = cubic_min(s->a_prev, s->phi_prev, s->dphi_prev, a, phi, dphi);
			if (!isfinite(c) || c < 1.1 * a) an = 4.0 * a;
			else an = c > 100.0 * a ? 100.0 * a : c;
			if (an > s->amax) an = s->amax;
			s->a_prev = a; s->phi_prev = phi; s->dphi_prev = dphi;
			s->a = an;
			return 0;
		}
	} else {
		if (!armijo || !(phi < s->phi_lo)) {
			s->a_hi = a; s->phi_hi = phi; s->dphi_hi = dphi;
		} else {
			if (fabs(dphi) <= -s->eta * s->dphi0) return 1;
			if (dphi * (s->a_hi - s->a_lo) >= 0.0) { s->a_hi = s->a_lo; s->phi_hi = s->phi_lo; s->dphi_hi = s->dphi_lo; }
			s->a_lo = a; s->phi_lo = phi; s->dphi_lo = dphi;
		}
	}
	/* in zoom: budget / interval collapse -> fall back on the best Armijo point */
	if (s->nfev >= s->maxfev ||
	    fabs(s->a_hi - s->a_lo) <= 1e-14 * fmax(fabs(s->a_hi), fabs(s->a_lo))) {
		if (!(s->a_lo > 0.0)) return -1;
		if (s->a_lo == a) return 1;
		s->a = s->a_lo;
		return 2;
	}
	s->a = zoom_trial(s);
	return 0;
}

/* ---------------- collocation preconditioner W0 = Z (Z'H0Z)^-1 Z' ---------------- */
static void add_av_terms(double *H0, int n, const orc_colloc *cc, const orc_AV *av, int nav, int bp, double w)
{
	int a, q1, q2;
	for (a = 0; a < nav; a++) {
		int o = av[a].output, r = av[a].deriv, k = cc->order[o], base = cc->iC[o] + cc->off[o][bp];
		const double *blk = cc->blk[o] + ((size_t)bp * k) * cc->maxderiv[o];
		for (q1 = 0; q1 < k; q1++)
			for (q2 = 0; q2 < k; q2++)
				M_(H0, n, base + q1, base + q2) += w * blk[q1 * cc->maxderiv[o] + r] * blk[q2 * cc->maxderiv[o] + r];
	}
}
/* Householder QR of A' (n x m): returns explicit Q (n x n, column-major) */
static void qr_full_q(const double *A, int m, int n, double *Q)
{
	double *R = malloc((size_t)n * m * sizeof(double)), *v = malloc(n * sizeof(double));
	int i, j, c;
	for (j = 0; j < m; j++) for (i = 0; i < n; i++) M_(R, n, i, j) = M_(A, m, j, i); /* R = A' */
	memset(Q, 0, (size_t)n * n * sizeof(double));
	for (i = 0; i < n; i++) M_(Q, n, i, i) = 1.0;
	for (j = 0; j < m && j < n; j++) {
		double nr = 0.0, alpha, vn = 0.0;
		for (i = j; i < n; i++) nr += M_(R, n, i, j) * M_(R, n, i, j);
		nr = sqrt(nr);
		if (nr == 0.0) continue;
		alpha = M_(R, n, j, j) > 0 ? -nr : nr;
		for (i = 0; i < n; i++) v[i] = 0.0;
		for (i = j; i < n; i++) v[i] = M_(R, n, i, j);
		v[j] -= alpha;
		for (i = j; i < n; i++) vn += v[i] * v[i];
		if (vn == 0.0) continue;
		for (c = j; c < m; c++) { /* R <- (I - 2vv'/v'v) R */
			double s = 0.0; for (i = j; i < n; i++) s += v[i] * M_(R, n, i, c);
			s = 2.0 * s / vn; for (i = j; i < n; i++) M_(R, n, i, c) -= s * v[i];
		}
		for (c = 0; c < n; c++) { /* Q <- Q (I - 2vv'/v'v) : rows of Q */
			double s = 0.0; for (i = j; i < n; i++) s += M_(Q, n, c, i) * v[i];
			s = 2.0 * s / vn; for (i = j; i < n; i++) M_(Q, n, c, i) -= s * v[i];
		}
	}
	free(R); free(v);
}
/* dense core: W0 = Z (Z'H0Z)^-1 Z' for one block (H0 n x n, AE m x n); returns 0 on success */
static int w0_dense(const double *H0, const double *AE, int m, int n, double *W0)
{
	int nr = n - m, i, j, k, rc;
	double *Q, *Hr, *T, *Zt, tr = 0.0;
	if (nr <= 0) return 1;
	Q = malloc((size_t)n * n * sizeof(double));
	if (m > 0) qr_full_q(AE, m, n, Q);
	else { memset(Q, 0, (size_t)n * n * sizeof(double)); for (i = 0; i < n; i++) M_(Q, n, i, i) = 1.0; }
	/* Z = Q[:, m:],  T = H0 Z (n x nr),  Hr = Z' T  (H0 is symmetric: its columns are read as rows) */
	T = malloc((size_t)n * nr * sizeof(double)); Hr = malloc((size_t)nr * nr * sizeof(double));
	for (j = 0; j < nr; j++) for (i = 0; i < n; i++) M_(T, n, i, j) = dot_(&M_(H0, n, 0, i), &M_(Q, n, 0, m + j), n);
	for (j = 0; j < nr; j++) for (i = 0; i < nr; i++) M_(Hr, nr, i, j) = dot_(&M_(Q, n, 0, m + i), &M_(T, n, 0, j), n);
	for (i = 0; i < nr; i++) tr += M_(Hr, nr, i, i);
	for (i = 0; i < nr; i++) M_(Hr, nr, i, i) += 1e-12 * tr / nr + 1e-300;
	rc = chol_(Hr, nr);
	if (!rc) {
		/* H0 singular on null(A) (e.g. no equality rows at all: constants and ramps cost nothing): the regularised
		 * inverse would scale those directions by 1e12.  No preconditioner then -- the caller starts from the identity. */
		double lo = 1e300, hi = 0.0;
		for (i = 0; i < nr; i++) { double dd = M_(Hr, nr, i, i) * M_(Hr, nr, i, i); if (dd < lo) lo = dd; if (dd > hi) hi = dd; }
		if (lo < 1e-9 * hi) { free(Q); free(T); free(Hr); return 2; }
	}
	if (rc) {
		/* not positive definite on null(A): regularise harder once */
		for (j = 0; j < nr; j++) for (i = 0; i < nr; i++) M_(Hr, nr, i, j) = dot_(&M_(Q, n, 0, m + i), &M_(T, n, 0, j), n);
		for (i = 0; i < nr; i++) M_(Hr, nr, i, i) += 1e-6 * tr / nr + 1e-300;
		rc = chol_(Hr, nr);
	}
	if (!rc) {
		/* W0 = Z Hr^-1 Z' : solve Hr X = Z' column by column (X is nr x n), W0 = Z X */
		double *X = malloc((size_t)nr * n * sizeof(double));
		Zt = malloc((size_t)nr * n * sizeof(double));   /* Zt[k + nr*i] = Z[i][k]: rows of Z contiguous */
		for (j = 0; j < n; j++) {
			for (i = 0; i < nr; i++) { M_(X, nr, i, j) = M_(Q, n, j, m + i); M_(Zt, nr, i, j) = M_(Q, n, j, m + i); }
			chol_solve_(Hr, nr, &M_(X, nr, 0, j));
		}
		for (j = 0; j < n; j++) for (i = 0; i < n; i++) M_(W0, n, i, j) = dot_(&M_(Zt, nr, 0, i), &M_(X, nr, 0, j), nr);
		for (j = 0; j < n; j++) for (i = 0; i < j; i++) {   /* symmetrise */
			double sym = 0.5 * (M_(W0, n, i, j) + M_(W0, n, j, i)); M_(W0, n, i, j) = sym; M_(W0, n, j, i) = sym;
		}
		free(X); free(Zt);
	}
	(void)k;
	free(Q); free(T); free(Hr);
	return rc;
}

/* returns 0 and fills W0 (n x n) on success.  H0 is block diagonal by output; outputs that no row of AE couples
 * give independent blocks of W0, which are built one at a time (a 12-output problem is twelve small
 * factorisations instead of one large one). */
static int build_colloc_W0(const orc_problem *p, const double *AE, int m, double *W0)
{
	const orc_colloc *cc = p->cc;
	int n = cc->nC, nout = cc->nout, i, j, o, r, P = cc->nbps, rc = 0;
	double *H0 = calloc((size_t)n * n, sizeof(double));
	int *comp = malloc(nout * sizeof(int)), *outof = malloc(n * sizeof(int));
	if (n - m <= 0) { free(H0); free(comp); free(outof); return 1; }
	for (i = 0; i < P; i++) {   /* trapezoid weight of breakpoint i */
		double w = 0.0;
		if (i > 0) w += (cc->bps[i] - cc->bps[i - 1]) / 2;
		if (i < P - 1) w += (cc->bps[i + 1] - cc->bps[i]) / 2;
		if (p->nucf) add_av_terms(H0, n, cc, p->tcostav, p->ntcostav, i, w);
	}
	if (p->nicf) add_av_terms(H0, n, cc, p->icostav, p->nicostav, 0, 1.0);
	if (p->nfcf) add_av_terms(H0, n, cc, p->fcostav, p->nfcostav, P - 1, 1.0);
	for (o = 0; o < nout; o++) { comp[o] = o; for (j = 0; j < cc->ncoef[o]; j++) outof[cc->iC[o] + j] = o; }
	for (r = 0; r < m; r++) {   /* merge the outputs a row touches (label = smallest output index) */
		int first = -1;
		for (j = 0; j < n; j++) if (M_(AE, m, r, j) != 0.0) {
			int c = comp[outof[j]];
			if (first < 0) first = c;
			else if (c != first) { int lo = c < first ? c : first, hi = c < first ? first : c; for (o = 0; o < nout; o++) if (comp[o] == hi) comp[o] = lo; first = lo; }
		}
	}
	memset(W0, 0, (size_t)n * n * sizeof(double));
	for (o = 0; o < nout && !rc; o++) {
		int nb = 0, mb = 0, *idx, *rows; double *Hb, *Ab, *Wb;
		if (comp[o] != o) continue;
		idx = malloc(n * sizeof(int)); rows = malloc((m + 1) * sizeof(int));
		for (j = 0; j < n; j++) if (comp[outof[j]] == o) idx[nb++] = j;
		for (r = 0; r < m; r++) { int hit = 0; for (j = 0; j < nb && !hit; j++) if (M_(AE, m, r, idx[j]) != 0.0) hit = 1; if (hit) rows[mb++] = r; }
		Hb = malloc((size_t)nb * nb * sizeof(double)); Ab = malloc((size_t)(mb + 1) * nb * sizeof(double)); Wb = malloc((size_t)nb * nb * sizeof(double));
		for (j = 0; j < nb; j++) for (i = 0; i < nb; i++) M_(Hb, nb, i, j) = M_(H0, n, idx[i], idx[j]);
		for (j = 0; j < nb; j++) for (i = 0; i < mb; i++) M_(Ab, mb, i, j) = M_(AE, m, rows[i], idx[j]);
		rc = w0_dense(Hb, Ab, mb, nb, Wb);
		if (!rc) for (j = 0; j < nb; j++) for (i = 0; i < nb; i++) M_(W0, n, idx[i], idx[j]) = M_(Wb, nb, i, j);
		free(idx); free(rows); free(Hb); free(Ab); free(Wb);
	}
	free(H0); free(comp); free(outof);
	return rc;
}

/* ---------------- the solver ---------------- */
typedef struct {
	const double *A; int n, m;   /* A: the EQUALITY rows, m x n column-major */
	double *S;   /* chol(A A') lower, m x m */
	double *tmpm;
} proj_t;
static void project(const proj_t *pj, const double *g, double *gp, double *lam_out)
{
	int n = pj->n, m = pj->m, i, j;
	const double *A = pj->A;
	memcpy(gp, g, n * sizeof(double));
	if (m == 0) return;
	for (i = 0; i < m; i++) { double s = 0.0; for (j = 0; j < n; j++) s += M_(A, m, i, j) * g[j]; pj->tmpm[i] = s; }
	chol_solve_(pj->S, m, pj->tmpm);
	for (j = 0; j < n; j++) { double s = 0.0; for (i = 0; i < m; i++) s += M_(A, m, i, j) * pj->tmpm[i]; gp[j] -= s; }
	if (lam_out) memcpy(lam_out, pj->tmpm, m * sizeof(double));
}

/* F_A and its gradient: the objective (orc_funobj) plus, when nonlinear constraints are present,
 * the Powell-Hestenes-Rockafellar augmented-Lagrangian terms of  bl <= c(x) <= bu :
 *   v = c + lam/mu,  p = clamp(v, bl, bu),  t = mu (v - p)   (= next multiplier estimate)
 *   F_A = F + sum_j (t_j^2 - lam_j^2) / (2 mu),   grad F_A = g + J' t
 * Also returns  rv = sqrt( sum_j ((c_j - p_j)/(1+|c_j|))^2 ): constraint violation and complementarity in one number. */
typedef struct {
	orc_problem *p; int n, nc; double mu; double *lam, *tnew, *c; int nfev;
	int nI; const int *irow;   /* linear INEQUALITY rows (indices into A): treated like constraints with a constant Jacobian */
} al_t;
static double al_eval(al_t *a, const double *x, double *g, double *rv_out, double *gnorm_f)
{
	orc_problem *p = a->p;
	int mode = 2, nstate = 0, n = a->n, nc = a->nc, nI = a->nI, i, j, m = p->nclin;
	double F, rv2 = 0.0, pen = 0.0;
	orc_funobj(p, &mode, x, &F, g, &nstate);
	a->nfev++;
	if (gnorm_f) *gnorm_f = nrm2_(g, n);
	if (nc > 0) {
		mode = 2;
		orc_funcon(p, &mode, x, a->c, NULL, &nstate);
	}
	for (j = 0; j < nc + nI; j++) {
		/* nonlinear rows first, then the linear inequality rows (c = A_r x) */
		double bl, bu, cj;
		if (j < nc) { bl = p->bl[n + m + j]; bu = p->bu[n + m + j]; cj = a->c[j]; }
		else {
			const int r = a->irow[j - nc];
			bl = p->bl[n + r]; bu = p->bu[n + r];
			cj = 0.0; for (i = 0; i < n; i++) cj += M_(p->A, m, r, i) * x[i];
			a->c[j] = cj;
		}
		{
			const double v = cj + a->lam[j] / a->mu;
			const double pj = v < bl ? bl : (v > bu ? bu : v);
			/* distance to the clamped shifted value: |c - b| for an active row, min(slack, lam/mu) for a feasible one --
			 * zero only when feasibility AND complementarity hold (the measure of LANCELOT / ALGENCAN) */
			const double tj = a->mu * (v - pj), rj = (cj - pj) / (1.0 + fabs(cj));
			a->tnew[j] = tj;
			pen += (tj - a->lam[j]) * (tj + a->lam[j]) / (2.0 * a->mu);   /* factored: no cancellation when c is tiny */
			rv2 += rj * rj;
		}
	}
	F += pen;
	if (nc > 0)
		for (i = 0; i < n; i++) { double sum = 0.0; for (j = 0; j < nc; j++) sum += M_(p->cJac, nc, j, i) * a->tnew[j]; g[i] += sum; }
	for (j = 0; j < nI; j++) { const int r = a->irow[j]; for (i = 0; i < n; i++) g[i] += M_(p->A, m, r, i) * a->tnew[nc + j]; }
	if (rv_out) *rv_out = sqrt(rv2);
	return F;
}

void orc_sqp_solve(orc_problem *p, double *x, const orc_sqp_opts *o, orc_sqp_result *res,
                   double *clambda, int *istate, double *R, double *trace, int trace_cap)
{
	const orc_colloc *cc = p->cc;
	int n = cc->nC, mall = p->nclin, m = 0, nI = 0, nc = p->ncnln, i, j, iter = 0, inform = 4, outer, nal;
	int itlim = o->itlim > 0 ? o->itlim : (50 > 3 * (n + mall) + 10 * nc ? 50 : 3 * (n + mall) + 10 * nc);
	int *erow = malloc((mall + 1) * sizeof(int)), *irow = malloc((mall + 1) * sizeof(int));
	double *AE = NULL, *bE = NULL;
	double r = o->opttol > 0 ? o->opttol : pow(DBL_EPSILON, 0.8), sr = sqrt(r), ftol = 1e-8;
	double *W, *W0 = NULL, *g, *gp, *gn, *gpn, *d, *pdir, *xt, *s, *y, *u, *t, *lam;
	double F = 0, Fn = 0, alpha = 0, pnorm = 0, gnf = 0, gnfn = 0, rv = 0, rvn = 0, rv_prev = HUGE_VAL;
	int weak = 0;
	proj_t pj;
	ls_t ls;
	al_t al;

	memset(res, 0, sizeof(*res));
	/* linear rows with lower == upper are kept satisfied by projection; the others join the nonlinear rows
	 * in the augmented Lagrangian */
	for (i = 0; i < mall; i++) { if (p->bl[n + i] == p->bu[n + i]) erow[m++] = i; else irow[nI++] = i; }
	nal = nc + nI;
	if (nal > 0 && o->fixed_iters) { res->inform = 9; free(erow); free(irow); return; }
	AE = malloc((size_t)(m + 1) * n * sizeof(double)); bE = malloc((m + 1) * sizeof(double));
	for (i = 0; i < m; i++) { bE[i] = p->bl[n + erow[i]]; for (j = 0; j < n; j++) M_(AE, m, i, j) = M_(p->A, mall, erow[i], j); }

	W = malloc((size_t)n * n * sizeof(double));
	g = malloc(n * sizeof(double)); gp = malloc(n * sizeof(double)); gn = malloc(n * sizeof(double));
	gpn = malloc(n * sizeof(double)); d = malloc(n * sizeof(double)); pdir = malloc(n * sizeof(double));
	xt = malloc(n * sizeof(double)); s = malloc(n * sizeof(double)); y = malloc(n * sizeof(double));
	u = malloc(n * sizeof(double)); t = malloc(n * sizeof(double)); lam = calloc(m + 1, sizeof(double));
	pj.A = AE; pj.n = n; pj.m = m; pj.S = NULL; pj.tmpm = malloc((m + 1) * sizeof(double));
	al.p = p; al.n = n; al.nc = nc; al.mu = 10.0; al.nfev = 0; al.nI = nI; al.irow = irow;
	al.lam = calloc(nal + 1, sizeof(double)); al.tnew = calloc(nal + 1, sizeof(double)); al.c = calloc(nal + 1, sizeof(double));
	if (m > 0) {
		pj.S = malloc((size_t)m * m * sizeof(double));
		for (i = 0; i < m; i++) for (j = 0; j < m; j++) {
			double sum = 0.0; int k; for (k = 0; k < n; k++) sum += M_(AE, m, i, k) * M_(AE, m, j, k);
			M_(pj.S, m, i, j) = sum;
		}
		if (chol_(pj.S, m)) { inform = 9; goto done; } /* rank-deficient A */
		/* feasibility: x += A' (AA')^-1 (b - A x) */
		for (i = 0; i < m; i++) { double sum = 0.0; for (j = 0; j < n; j++) sum += M_(AE, m, i, j) * x[j]; pj.tmpm[i] = bE[i] - sum; }
		chol_solve_(pj.S, m, pj.tmpm);
		for (j = 0; j < n; j++) { double sum = 0.0; for (i = 0; i < m; i++) sum += M_(AE, m, i, j) * pj.tmpm[i]; x[j] += sum; }
	}
	if (o->hessian == 1) {
		W0 = malloc((size_t)n * n * sizeof(double));
		if (build_colloc_W0(p, AE, m, W0)) { free(W0); W0 = NULL; }
	}

	/* outer loop: one pass when there are no nonlinear constraints; otherwise the multiplier /
	 * penalty iteration of the augmented Lagrangian (at most 30 passes) */
	for (outer = 0; outer < (nal > 0 ? 30 : 1); outer++) {
		/* inner tolerance: NPSOL's for the last passes, looser while the constraints are far off */
		const double sri = nal > 0 ? fmax(sr, fmin(1e-3, 0.1 * rv_prev)) : sr;
		int inner_inform = 4, stop = 0, at_x = 1, nupd = 0;
		weak = 0;
		if (outer > 0 && m > 0) {
			/* steps stay in null(A) only to rounding; hundreds of majors per pass can drift: re-apply
			 * x += A'(AA')^-1 (b - A x) before every further pass */
			for (i = 0; i < m; i++) { double sum = 0.0; for (j = 0; j < n; j++) sum += M_(AE, m, i, j) * x[j]; pj.tmpm[i] = bE[i] - sum; }
			chol_solve_(pj.S, m, pj.tmpm);
			for (j = 0; j < n; j++) { double sum = 0.0; for (i = 0; i < m; i++) sum += M_(AE, m, i, j) * pj.tmpm[i]; x[j] += sum; }
		}
		if (W0) memcpy(W, W0, (size_t)n * n * sizeof(double));
		else { memset(W, 0, (size_t)n * n * sizeof(double)); for (i = 0; i < n; i++) M_(W, n, i, i) = 1.0; }
		F = al_eval(&al, x, g, &rv, &gnf);
		project(&pj, g, gp, lam);
		for (i = 0; i < n; i++) { double sum = 0.0; for (j = 0; j < n; j++) sum += M_(W, n, i, j) * gp[j]; d[i] = sum; }

		for (;;) {
			double dphi0, xnorm, amax, a1, sy, yu, rho, cc2, tolg;
			int rc;
			if (iter >= itlim) { inner_inform = 4; stop = 1; break; }
			if (nal > 0 && m > 0) {
				/* Under a large penalty |g| >> |Z'g|: the rounding error of the projection, relative to |g|, is
				 * then a visible fraction of gp and of d = W gp, and x would creep off A x = b along the path.
				 * Projecting the direction itself leaves an error relative to |d| only. */
				project(&pj, d, xt, NULL);
				memcpy(d, xt, n * sizeof(double));
			}
			for (i = 0; i < n; i++) pdir[i] = -d[i];
			dphi0 = dot_(gp, pdir, n);
			pnorm = nrm2_(pdir, n); xnorm = nrm2_(x, n);
			tolg = sri * (1.0 + fmax(1.0 + fabs(F), gnf));
			if (pnorm == 0.0 || !(dphi0 < 0.0)) {
				/* already stationary in null(A) (or W lost definiteness numerically: restart once) */
				if (pnorm != 0.0) {
					nupd = 0;
					if (W0) memcpy(W, W0, (size_t)n * n * sizeof(double));
					else { memset(W, 0, (size_t)n * n * sizeof(double)); for (i = 0; i < n; i++) M_(W, n, i, i) = 1.0; }
					for (i = 0; i < n; i++) { double sum = 0.0; for (j = 0; j < n; j++) sum += M_(W, n, i, j) * gp[j]; d[i] = sum; pdir[i] = -sum; }
					dphi0 = dot_(gp, pdir, n); pnorm = nrm2_(pdir, n);
				}
				if (pnorm == 0.0 || !(dphi0 < 0.0)) { inner_inform = (nrm2_(gp, n) <= tolg) ? 0 : 6; break; }
			}
			/* already stationary to rounding level: a line search could only fail */
			if (!o->fixed_iters && nrm2_(gp, n) <= 1e-3 * tolg) { inner_inform = 0; break; }
			amax = o->steplimit * (1.0 + xnorm) / pnorm;
			a1 = amax < 1.0 ? amax : 1.0;
			ls_init(&ls, F, dphi0, a1, amax, o->ls_mu, o->ls_eta, o->ls_maxfev);
			for (;;) {
				double dphi;
				for (i = 0; i < n; i++) xt[i] = x[i] + ls.a * pdir[i];
				Fn = al_eval(&al, xt, gn, &rvn, &gnfn);
				project(&pj, gn, gpn, lam);
				dphi = dot_(gpn, pdir, n);
				rc = ls_step(&ls, Fn, dphi);
				if (rc == 1 || rc == -1) break;
				if (rc == 2) {
					for (i = 0; i < n; i++) xt[i] = x[i] + ls.a * pdir[i];
					Fn = al_eval(&al, xt, gn, &rvn, &gnfn);
					project(&pj, gn, gpn, lam);
					rc = 1; break;
				}
			}
			if (rc != 1) {
				/* line search failed: if the quasi-Newton matrix carries updates, drop them and retry from
				 * the same point with W0 (a poor W is the usual cause); fail only if W0 itself fails */
				if (nupd > 0 && nrm2_(gp, n) > tolg) {
					nupd = 0;
					if (W0) memcpy(W, W0, (size_t)n * n * sizeof(double));
					else { memset(W, 0, (size_t)n * n * sizeof(double)); for (i = 0; i < n; i++) M_(W, n, i, i) = 1.0; }
					for (i = 0; i < n; i++) { double sum = 0.0; for (j = 0; j < n; j++) sum += M_(W, n, i, j) * gp[j]; d[i] = sum; }
					continue;
				}
				/* no further decrease obtainable along a descent direction: converged if the projected gradient
				 * meets the tolerance, "optimal but not to the requested accuracy" (NPSOL inform 1) if it is
				 * within 10^3 of it (function values at rounding level), failure (6) otherwise */
				if (nrm2_(gp, n) <= tolg) inner_inform = 0;
				else if (nrm2_(gp, n) <= 1e3 * tolg) { inner_inform = 0; weak = 1; }
				else inner_inform = 6;
				at_x = 0; break;
			}
			alpha = ls.a;
			for (i = 0; i < n; i++) { s[i] = alpha * pdir[i]; y[i] = gpn[i] - gp[i]; }
			memcpy(x, xt, n * sizeof(double));
			if (nupd == (o->qn_memory > 0 ? o->qn_memory : 256)) {
				/* memory full (the device keeps the updates as pairs): restart the approximation from W0 */
				nupd = 0;
				if (W0) memcpy(W, W0, (size_t)n * n * sizeof(double));
				else { memset(W, 0, (size_t)n * n * sizeof(double)); for (i = 0; i < n; i++) M_(W, n, i, i) = 1.0; }
				for (i = 0; i < n; i++) { double sum = 0.0; for (j = 0; j < n; j++) sum += M_(W, n, i, j) * gp[j]; d[i] = sum; }
			}
			/* t = W gp+,  u = W y = t - d */
			for (i = 0; i < n; i++) { double sum = 0.0; for (j = 0; j < n; j++) sum += M_(W, n, i, j) * gpn[j]; t[i] = sum; }
			for (i = 0; i < n; i++) u[i] = t[i] - d[i];
			sy = dot_(s, y, n);
			if (sy > 1e-12 * nrm2_(s, n) * nrm2_(y, n)) {
				double a1s, a2u;
				yu = dot_(y, u, n); rho = 1.0 / sy; cc2 = rho * (1.0 + rho * yu);
				for (j = 0; j < n; j++) for (i = 0; i < n; i++)
					M_(W, n, i, j) += -rho * (s[i] * u[j] + u[i] * s[j]) + cc2 * s[i] * s[j];
				nupd++;
				a1s = dot_(s, gpn, n); a2u = dot_(u, gpn, n);
				for (i = 0; i < n; i++) d[i] = t[i] - rho * (s[i] * a2u + u[i] * a1s) + cc2 * s[i] * a1s;
			} else {
				memcpy(d, t, n * sizeof(double));
			}
			F = Fn; gnf = gnfn; rv = rvn; memcpy(g, gn, n * sizeof(double)); memcpy(gp, gpn, n * sizeof(double));
			if (trace && iter < trace_cap) {
				trace[4 * iter + 0] = F; trace[4 * iter + 1] = nrm2_(gp, n); trace[4 * iter + 2] = alpha; trace[4 * iter + 3] = (double)ls.nfev;
			}
			if (o->verbose) fprintf(stderr, "  maj %3d  F=%.15g |Zg|=%.3e alpha=%.3e nf=%d rv=%.2e mu=%g\n", iter, F, nrm2_(gp, n), alpha, ls.nfev, rv, al.mu);
			iter++;
			if (!o->fixed_iters &&
			    alpha * pnorm <= sri * (1.0 + nrm2_(x, n)) &&
			    nrm2_(gp, n) <= sri * (1.0 + fmax(1.0 + fabs(F), gnf))) { inner_inform = 0; break; }
		}
		if (nal == 0) { inform = (inner_inform == 0 && weak) ? 1 : inner_inform; break; }
		/* ---- multiplier / penalty update from the constraint values AT x: if the inner solve ended on a
		 *      rejected line-search trial, the last evaluation was elsewhere -> evaluate once more at x ---- */
		if (!at_x) { F = al_eval(&al, x, g, &rv, &gnf); at_x = 1; }
		if (inner_inform == 6) { inform = 6; break; }
		if (rv <= ftol && sri <= sr && inner_inform == 0) { memcpy(al.lam, al.tnew, nal * sizeof(double)); inform = weak ? 1 : 0; break; }
		if (stop) { inform = 4; break; }
		if (rv <= 0.25 * rv_prev) { memcpy(al.lam, al.tnew, nal * sizeof(double)); rv_prev = rv; }
		else al.mu *= 10.0;
		if (outer == 29) inform = 3;   /* nonlinear constraints not satisfied to tolerance */
	}
	if (nal > 0 && m > 0 && inform != 9) {
		/* many hundreds of majors under a large penalty let x drift off A x = b by rounding: restore it */
		for (i = 0; i < m; i++) { double sum = 0.0; for (j = 0; j < n; j++) sum += M_(AE, m, i, j) * x[j]; pj.tmpm[i] = bE[i] - sum; }
		chol_solve_(pj.S, m, pj.tmpm);
		for (j = 0; j < n; j++) { double sum = 0.0; for (i = 0; i < m; i++) sum += M_(AE, m, i, j) * pj.tmpm[i]; x[j] += sum; }
	}
done:
	res->inform = inform; res->iters = iter; res->nfev = al.nfev;
	/* objective WITHOUT the penalty terms (what NPSOL reports) */
	if (inform != 9) {
		if (nal > 0) { int mode = 0, nstate = 0; orc_funobj(p, &mode, x, &F, NULL, &nstate); }
		res->objective = F;
	}
	res->pg_norm = (inform == 9) ? 0.0 : nrm2_(gp, n);
	{ /* linear feasibility of the equality rows (and relative violation of everything in the AL) */
		double worst = 0.0;
		for (i = 0; i < m; i++) { double sum = 0.0; for (j = 0; j < n; j++) sum += M_(AE, m, i, j) * x[j]; worst = fmax(worst, fabs(sum - bE[i])); }
		res->feas = fmax(worst, nal > 0 && inform != 9 ? rv : 0.0);
	}
	if (clambda) {
		for (i = 0; i < n + mall + nc; i++) clambda[i] = 0.0;
		for (i = 0; i < m; i++) clambda[n + erow[i]] = lam[i];
		for (i = 0; i < nI; i++) clambda[n + irow[i]] = -al.lam[nc + i];
		for (i = 0; i < nc; i++) clambda[n + mall + i] = -al.lam[i]; /* NPSOL sign: grad F = A' lam_lin + J' lam_nl */
	}
	if (istate) {
		for (i = 0; i < n; i++) istate[i] = 0;
		for (i = 0; i < m; i++) istate[n + erow[i]] = 3;
		for (i = 0; i < nal; i++) {
			const int slot = i < nc ? n + mall + i : n + irow[i - nc];
			const double bl = p->bl[slot], bu = p->bu[slot], cv = al.c[i];
			istate[slot] = (bl == bu) ? 3 : (al.lam[i] != 0.0 ? (fabs(cv - bl) <= fabs(cv - bu) ? 1 : 2) : 0);
		}
	}
	if (R && inform != 9) {
		/* R'R = (W + A'(AA')^-1 A)^-1 when W0 is rank-deficient; W itself otherwise */
		double *Hf = malloc((size_t)n * n * sizeof(double)), *col = malloc(n * sizeof(double));
		memcpy(Hf, W, (size_t)n * n * sizeof(double));
		if (W0 && m > 0) {
			double *X = malloc((size_t)m * n * sizeof(double));
			for (j = 0; j < n; j++) { for (i = 0; i < m; i++) M_(X, m, i, j) = M_(AE, m, i, j); chol_solve_(pj.S, m, &M_(X, m, 0, j)); }
			for (j = 0; j < n; j++) for (i = 0; i < n; i++) { double sum = 0.0; int k; for (k = 0; k < m; k++) sum += M_(AE, m, k, i) * M_(X, m, k, j); M_(Hf, n, i, j) += sum; }
			free(X);
		}
		if (!chol_(Hf, n)) {
			double *Hinv = calloc((size_t)n * n, sizeof(double));
			for (j = 0; j < n; j++) { memset(col, 0, n * sizeof(double)); col[j] = 1.0; chol_solve_(Hf, n, col); for (i = 0; i < n; i++) M_(Hinv, n, i, j) = col[i]; }
			if (!chol_(Hinv, n)) { /* Hinv = L2 L2' ; R = L2' (upper) */
				memset(R, 0, (size_t)n * n * sizeof(double));
				for (j = 0; j < n; j++) for (i = j; i < n; i++) M_(R, n, j, i) = M_(Hinv, n, i, j);
			}
			free(Hinv);
		}
		free(Hf); free(col);
	}
	free(W); free(W0); free(g); free(gp); free(gn); free(gpn); free(d); free(pdir); free(xt);
	free(s); free(y); free(u); free(t); free(lam); free(pj.S); free(pj.tmpm);
	free(al.lam); free(al.tnew); free(al.c); free(erow); free(irow); free(AE); free(bE);
}
