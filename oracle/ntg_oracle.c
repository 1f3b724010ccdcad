/*
 * oracle/ntg_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, see oracle.h).
 *
 * Restatement of the reference's evaluation path, following the reference's loop
 * structure, row/column orders, dense temporaries and summation order so that it
 * doubles as the "reference-faithful" CPU timing baseline:
 *   colloc.c:15-117   collocation blocks + offsets        -> orc_colloc_make
 *   colloc.c:318-367  Zvalue / updateZ                    -> orc_updateZ
 *   colloc.c:243-316  CollocConcatMult{I,T,F}             -> mult_I / mult_T / mult_F
 *   colloc.c:369-423  dIdz2dIdZ{I,T,F}                    -> scat_I / scat_T / scat_F
 *   colloc.c:425-447  Z2zp{I,T,F}                         -> zp_I / zp_T / zp_F
 *   colloc.c:449-484  SplineInterp                        -> orc_spline_interp
 *   cost.c:4-174      Initial/Integrated/Final cost       -> cost_I / cost_T / cost_F
 *   integrator.c:16-62 trapezoid                          -> orc_integrate_*
 *   constraints.c:5-33 bounds                             -> orc_bounds
 *   constraints.c:36-195 nonlinear constraints            -> nl_I / nl_T / nl_F
 *   constraints.c:198-261 linear constraint matrix        -> lin_matrix
 *   ntg.c:274-371     NPfunobj / NPfuncon                 -> orc_funobj / orc_funcon
 *   ntg.c:374-389     linspace                            -> orc_linspace
 * Dense matrices are column-major with explicit leading dimension: M(r,c) = m[c*ld+r]
 * (the reference's FMatrix: elements[c][r], matrix.c:106-117).
 */
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <assert.h>
#include "oracle.h"

#define FM(m, ld, r, c) ((m)[(size_t)(c) * (ld) + (r)])
/* reference block[bp].matrix->elements[q][r] */
#define BLK(cc, o, bp, q, r) ((cc)->blk[o][((size_t)(bp) * (cc)->order[o] + (q)) * (cc)->maxderiv[o] + (r)])

void orc_linspace(double *v, double d0, double d1, int n)
{
	/* ntg.c:374-389: cumulative add, NOT d0+i*h */
	int i;
	double h;
	if (d0 == d1) { for (i = 0; i < n; i++) v[i] = d0; return; }
	h = (d1 - d0) / (n - 1);
	v[0] = d0;
	for (i = 1; i < n; i++) v[i] = v[i - 1] + h;
}

/* ---- colloc.c:57-117 for one output, colloc.c:15-55 for the concatenation ---- */
orc_colloc *orc_colloc_make(int nout, double **knots, const int *ninterv,
                            const double *bps, int nbps, const int *maxderiv,
                            const int *order, const int *mult)
{
	orc_colloc *cc = calloc(1, sizeof(*cc));
	int o, i, q, r;
	cc->nout = nout; cc->nbps = nbps;
	cc->order = malloc(nout * sizeof(int)); cc->mult = malloc(nout * sizeof(int));
	cc->ninterv = malloc(nout * sizeof(int)); cc->maxderiv = malloc(nout * sizeof(int));
	cc->ncoef = malloc(nout * sizeof(int));
	cc->iZ = malloc(nout * sizeof(int)); cc->iz = malloc(nout * sizeof(int)); cc->iC = malloc(nout * sizeof(int));
	cc->blk = malloc(nout * sizeof(double *)); cc->off = malloc(nout * sizeof(int *));
	cc->bps = malloc(nbps * sizeof(double));
	memcpy(cc->bps, bps, nbps * sizeof(double));
	cc->nz = 0; cc->nC = 0;
	for (o = 0; o < nout; o++) {
		int k = order[o], m = mult[o], l = ninterv[o], d = maxderiv[o];
		int n = l * (k - m) + m;              /* colloc.c:67 */
		int naug = n + k, nn, left, mflag;
		double *aug = malloc(naug * sizeof(double));
		double *a = malloc(k * k * sizeof(double));
		double *db = malloc(k * d * sizeof(double));
		cc->order[o] = k; cc->mult[o] = m; cc->ninterv[o] = l; cc->maxderiv[o] = d; cc->ncoef[o] = n;
		cc->blk[o] = calloc((size_t)nbps * k * d, sizeof(double));
		cc->off[o] = malloc(nbps * sizeof(int));
		orc_knots(knots[o], l, k, m, aug, &nn);            /* colloc.c:92-93 */
		assert(nn == n);
		for (i = 0; i < nbps; i++) {                       /* colloc.c:95-102 */
			double x = bps[i];
			orc_interv(aug, naug, x, &left, &mflag);
			orc_bsplvd(aug, k, x, left, a, db, d);
			/* dbiatx is (k x d) column-major; block = its transpose (FTranspose,
			 * matrix.c:75-82): elements[q][r] = dbiatx(q+1, r+1) */
			for (q = 0; q < k; q++)
				for (r = 0; r < d; r++)
					BLK(cc, o, i, q, r) = db[r * k + q];
		}
		for (i = 0; i < nbps; i++) {                       /* colloc.c:104-111 */
			orc_interv(knots[o], l + 1, bps[i], &left, &mflag);
			cc->off[o][i] = (left - 1) * (k - m);
		}
		free(aug); free(a); free(db);
		/* colloc.c:41-49 */
		if (o == 0) { cc->iZ[0] = 0; cc->iz[0] = 0; cc->iC[0] = 0; }
		else {
			cc->iZ[o] = cc->iZ[o - 1] + cc->maxderiv[o - 1] * nbps;
			cc->iz[o] = cc->iz[o - 1] + cc->maxderiv[o - 1];
			cc->iC[o] = cc->iC[o - 1] + cc->ncoef[o - 1];
		}
		cc->nz += d; cc->nC += n;
	}
	cc->nZ = cc->nz * nbps;                                /* colloc.c:50 */
	return cc;
}

void orc_colloc_free(orc_colloc *cc)
{
	int o;
	for (o = 0; o < cc->nout; o++) { free(cc->blk[o]); free(cc->off[o]); }
	free(cc->blk); free(cc->off); free(cc->order); free(cc->mult); free(cc->ninterv);
	free(cc->maxderiv); free(cc->ncoef); free(cc->iZ); free(cc->iz); free(cc->iC); free(cc->bps);
	free(cc);
}

/* colloc.c:318-331 */
static double zvalue(const orc_colloc *cc, const double *C, int o, int deriv, int bp)
{
	int j; double d = 0.0;
	for (j = 0; j < cc->order[o]; j++)
		d += BLK(cc, o, bp, j, deriv) * C[cc->iC[o] + cc->off[o][bp] + j];
	return d;
}
static int odb2lin(const orc_colloc *cc, int o, int deriv, int bp)
{
	return cc->iZ[o] + cc->maxderiv[o] * bp + deriv;
}

/* colloc.c:344-367 */
void orc_updateZ(double *Z, const orc_colloc *cc, const double *C, const orc_AV *av, int nav, int type)
{
	int i, j;
	switch (type) {
	case ORC_AVINITIAL:
		for (i = 0; i < nav; i++)
			Z[odb2lin(cc, av[i].output, av[i].deriv, 0)] = zvalue(cc, C, av[i].output, av[i].deriv, 0);
		break;
	case ORC_AVTRAJECTORY:
		for (i = 0; i < nav; i++)
			for (j = 0; j < cc->nbps; j++)
				Z[odb2lin(cc, av[i].output, av[i].deriv, j)] = zvalue(cc, C, av[i].output, av[i].deriv, j);
		break;
	case ORC_AVFINAL:
		for (i = 0; i < nav; i++)
			Z[odb2lin(cc, av[i].output, av[i].deriv, cc->nbps - 1)] =
				zvalue(cc, C, av[i].output, av[i].deriv, cc->nbps - 1);
		break;
	}
}

/* colloc.c:425-447 */
static void zp_I(double **zp, double *Z, const orc_colloc *cc)
{ int i; for (i = 0; i < cc->nout; i++) zp[i] = &Z[cc->iZ[i]]; }
static void zp_T(double **zp, double *Z, const orc_colloc *cc, int bp)
{ int i; for (i = 0; i < cc->nout; i++) zp[i] = &Z[cc->iZ[i] + bp * cc->maxderiv[i]]; }
static void zp_F(double **zp, double *Z, const orc_colloc *cc)
{
	int i;
	for (i = 0; i < cc->nout - 1; i++) zp[i] = &Z[cc->iZ[i + 1] - cc->maxderiv[i]];
	zp[cc->nout - 1] = &Z[cc->nZ - cc->maxderiv[cc->nout - 1]];
}

/* dIdz is (nz rows x ncon cols) col-major: dIdz(v, c) = user's dc[c][v]  (ntg.c:174-197,
 * constraints.c:104-105).  dIdZ is (rows x nZ) col-major.                              */
/* colloc.c:369-382 */
static void scat_I(double *dIdZ, int ldZ, int rows, const double *const *dIdz, const orc_colloc *cc)
{
	int i, j, k;
	for (i = 0; i < rows; i++)
		for (j = 0; j < cc->nout; j++)
			for (k = 0; k < cc->maxderiv[j]; k++)
				FM(dIdZ, ldZ, i, cc->iZ[j] + k) = dIdz[i][cc->iz[j] + k];
}
/* colloc.c:384-399 */
static void scat_T(double *dIdZ, int ldZ, int ncon, const double *const *dIdz, const orc_colloc *cc, int bp)
{
	int i, j, k;
	for (i = 0; i < ncon; i++)
		for (j = 0; j < cc->nout; j++)
			for (k = 0; k < cc->maxderiv[j]; k++)
				FM(dIdZ, ldZ, i * cc->nbps + bp, cc->iZ[j] + bp * cc->maxderiv[j] + k) = dIdz[i][cc->iz[j] + k];
}
/* colloc.c:401-423 */
static void scat_F(double *dIdZ, int ldZ, int rows, const double *const *dIdz, const orc_colloc *cc)
{
	int i, j, k;
	for (i = 0; i < rows; i++) {
		for (j = 0; j < cc->nout - 1; j++)
			for (k = 0; k < cc->maxderiv[j]; k++)
				FM(dIdZ, ldZ, i, cc->iZ[j + 1] - cc->maxderiv[j] + k) = dIdz[i][cc->iz[j] + k];
		for (k = 0; k < cc->maxderiv[cc->nout - 1]; k++)
			FM(dIdZ, ldZ, i, cc->nZ - cc->maxderiv[cc->nout - 1] + k) = dIdz[i][cc->iz[cc->nout - 1] + k];
	}
}

/* colloc.c:243-260: only the first `order` columns of each output are written */
static void mult_I(double *dIdC, int ldC, const double *dIdZ, int ldZ, int rows, const orc_colloc *cc)
{
	int I, J, j, k, l;
	for (I = 0; I < rows; I++)
		for (j = 0; j < cc->nout; j++)
			for (k = 0; k < cc->order[j]; k++) {
				J = cc->iC[j] + k;
				FM(dIdC, ldC, I, J) = 0;
				for (l = 0; l < cc->maxderiv[j]; l++)
					FM(dIdC, ldC, I, J) += FM(dIdZ, ldZ, I, cc->iZ[j] + l) * BLK(cc, j, 0, k, l);
			}
}
/* colloc.c:263-285: row I belongs to breakpoint I % nbps; band columns only */
static void mult_T(double *dIdC, int ldC, const double *dIdZ, int ldZ, int rows, const orc_colloc *cc)
{
	int I, i, j, k, l;
	for (I = 0; I < rows; I++) {
		i = I % cc->nbps;
		for (j = 0; j < cc->nout; j++)
			for (k = cc->off[j][i]; k < cc->off[j][i] + cc->order[j]; k++) {
				FM(dIdC, ldC, I, cc->iC[j] + k) = 0;
				for (l = 0; l < cc->maxderiv[j]; l++)
					FM(dIdC, ldC, I, cc->iC[j] + k) +=
						FM(dIdZ, ldZ, I, cc->iZ[j] + i * cc->maxderiv[j] + l) *
						BLK(cc, j, i, k - cc->off[j][i], l);
			}
	}
}
/* colloc.c:287-316 */
static void mult_F(double *dIdC, int ldC, const double *dIdZ, int ldZ, int rows, const orc_colloc *cc)
{
	int I, J, j, k, l, last = cc->nbps - 1;
	for (I = 0; I < rows; I++)
		for (j = 0; j < cc->nout; j++)
			for (k = 0; k < cc->order[j]; k++) {
				J = cc->iC[j] + cc->off[j][last] + k;
				FM(dIdC, ldC, I, J) = 0;
				for (l = 0; l < cc->maxderiv[j]; l++) {
					int zcol = (j == cc->nout - 1)
						? cc->iZ[j] + cc->maxderiv[j] * cc->nbps - cc->maxderiv[j] + l
						: cc->iZ[j + 1] - cc->maxderiv[j] + l;
					FM(dIdC, ldC, I, J) += FM(dIdZ, ldZ, I, zcol) * BLK(cc, j, last, k, l);
				}
			}
}

/* integrator.c:21-24 and :44-48 (TRAPEZOID is the only rule ever selected, cost.c:61,96,111,134) */
void orc_integrate_vector(double *I, const double *f, const double *t, int n)
{
	int i;
	for (i = 0, *I = 0.0; i < n - 1; i++) *I += (t[i + 1] - t[i]) * (f[i + 1] + f[i]) / 2;
}
void orc_integrate_cols(double *I, const double *f, int rows, int cols, const double *t)
{
	int i, j;
	for (i = 0; i < cols; i++)
		for (j = 0, I[i] = 0.0; j < rows - 1; j++)
			I[i] += (t[j + 1] - t[j]) * (FM(f, rows, j + 1, i) + FM(f, rows, j, i)) / 2;
}

/* cost.c:4-36 */
/* ---- scratch of one evaluation ----
 * The reference allocates its dense temporaries per call (calloc / free in cost.c:117-134, constraints.c:147 ...) and so does this
 * restatement.  For the TIMED multi-threaded CPU baseline (bench.py cpu_baseline) the allocations can be served from a per-thread
 * buffer that is reused from call to call (orc_set_scratch_reuse(1)): the loops and the zero-fills stay the reference's, but 128
 * threads no longer queue on the kernel's address-space lock for mmap / munmap / page faults of 0.3 - 100 MB per call -- that
 * lock is a property of the C library under threads, not of the reference (which is single-threaded).  Off by default: the
 * parity tests run the plain calloc / free path. */
static int orc_scratch_reuse = 0;
static __thread struct { char *base; size_t cap, top; } orc_arena;
void orc_set_scratch_reuse(int on) { orc_scratch_reuse = on; }
int orc_scratch_reuse_on(void) { return orc_scratch_reuse; }
static void orc_tmp_reset(void) { orc_arena.top = 0; }
static void *orc_tmp_malloc(size_t bytes)
{
	if (!orc_scratch_reuse) return malloc(bytes);
	bytes = (bytes + 63) & ~(size_t)63;
	if (!orc_arena.base) { orc_arena.cap = (size_t)768 << 20; orc_arena.base = malloc(orc_arena.cap); orc_arena.top = 0; }   /* virtual: pages are touched as used, once */
	if (!orc_arena.base || orc_arena.top + bytes > orc_arena.cap) return malloc(bytes);
	{ void *q = orc_arena.base + orc_arena.top; orc_arena.top += bytes; return q; }
}
static void *orc_tmp_calloc(size_t n, size_t sz)
{
	if (!orc_scratch_reuse) return calloc(n, sz);
	{ void *q = orc_tmp_malloc(n * sz); if (q) memset(q, 0, n * sz); return q; }
}
static void orc_tmp_free(void *q)
{
	if (orc_arena.base && (char *)q >= orc_arena.base && (char *)q < orc_arena.base + orc_arena.cap) return;   /* released by the next reset */
	free(q);
}

static void cost_I(int *mode, int *nstate, double *I, double *dI, orc_icf_t func, const orc_colloc *cc, double *Z)
{
	double **zp = orc_tmp_malloc(cc->nout * sizeof(double *));
	zp_I(zp, Z, cc);
	if (*mode == 0) {
		func(mode, nstate, I, NULL, zp);
	} else if (*mode == 1 || *mode == 2) {
		double *dIdz = orc_tmp_calloc(cc->nz, sizeof(double));
		double *dIdZ = orc_tmp_calloc(cc->nZ, sizeof(double));
		double *dIdC = orc_tmp_calloc(cc->nC, sizeof(double));
		const double *rowp[1];
		func(mode, nstate, I, dIdz, zp);
		rowp[0] = dIdz;
		scat_I(dIdZ, 1, 1, rowp, cc);
		mult_I(dIdC, 1, dIdZ, 1, 1, cc);
		memcpy(dI, dIdC, cc->nC * sizeof(double));
		orc_tmp_free(dIdz); orc_tmp_free(dIdZ); orc_tmp_free(dIdC);
	}
	orc_tmp_free(zp);
}
/* cost.c:141-174 */
static void cost_F(int *mode, int *nstate, double *I, double *dI, orc_icf_t func, const orc_colloc *cc, double *Z)
{
	double **zp = orc_tmp_malloc(cc->nout * sizeof(double *));
	zp_F(zp, Z, cc);
	if (*mode == 0) {
		func(mode, nstate, I, NULL, zp);
	} else if (*mode == 1 || *mode == 2) {
		double *dIdz = orc_tmp_calloc(cc->nz, sizeof(double));
		double *dIdZ = orc_tmp_calloc(cc->nZ, sizeof(double));
		double *dIdC = orc_tmp_calloc(cc->nC, sizeof(double));
		const double *rowp[1];
		func(mode, nstate, I, dIdz, zp);
		rowp[0] = dIdz;
		scat_F(dIdZ, 1, 1, rowp, cc);
		mult_F(dIdC, 1, dIdZ, 1, 1, cc);
		memcpy(dI, dIdC, cc->nC * sizeof(double));
		orc_tmp_free(dIdz); orc_tmp_free(dIdZ); orc_tmp_free(dIdC);
	}
	orc_tmp_free(zp);
}
/* cost.c:38-139 -- keeps the dense nbps x nC temporary and the full-column trapezoid */
static void cost_T(int *mode, int *nstate, double *I, double *dI, const double *bps,
                   orc_ucf_t func, const orc_colloc *cc, double *Z)
{
	double **zp = orc_tmp_malloc(cc->nout * sizeof(double *));
	int P = cc->nbps, i, j, k, l, offset;
	double *f = NULL, *d1 = NULL, *dIdz = NULL, *dIdC = NULL;
	if (*mode == 0 || *mode == 2) f = orc_tmp_malloc(P * sizeof(double));
	if (*mode == 1 || *mode == 2) {
		d1 = orc_tmp_malloc(cc->nz * sizeof(double));
		dIdz = orc_tmp_calloc((size_t)P * cc->nz, sizeof(double)); /* FMatrix(nbps rows, nz cols) */
	}
	for (i = 0; i < P; i++) {
		zp_T(zp, Z, cc, i);
		func(mode, nstate, &i, f ? f + i : NULL, d1, zp);
		if (d1) for (j = 0; j < cc->nz; j++) FM(dIdz, P, i, j) = d1[j];
	}
	if (f) { orc_integrate_vector(I, f, bps, P); orc_tmp_free(f); }
	if (d1) {
		orc_tmp_free(d1);
		dIdC = orc_tmp_calloc((size_t)P * cc->nC, sizeof(double));
		for (i = 0; i < cc->nout; i++)
			for (j = 0; j < P; j++) {
				offset = cc->off[i][j];
				for (k = 0; k < offset; k++) FM(dIdC, P, j, cc->iC[i] + k) = 0;
				for (; k < offset + cc->order[i]; k++) {
					FM(dIdC, P, j, cc->iC[i] + k) = 0;
					for (l = 0; l < cc->maxderiv[i]; l++)
						FM(dIdC, P, j, cc->iC[i] + k) += FM(dIdz, P, j, cc->iz[i] + l) * BLK(cc, i, j, k - offset, l);
				}
				for (; k < cc->ncoef[i]; k++) FM(dIdC, P, j, cc->iC[i] + k) = 0;
			}
		orc_tmp_free(dIdz);
		orc_integrate_cols(dI, dIdC, P, cc->nC, bps);
		orc_tmp_free(dIdC);
	}
	orc_tmp_free(zp);
}

/* constraints.c:5-33 */
void orc_bounds(double *bbar, const double *b, int nc, int nlic, int nltc, int nlfc,
                int nnlic, int nnltc, int nnlfc, int nbps, double bigbnd)
{
	int i, j, p = nc;
	const double *s = b;
	for (i = 0; i < nc; i++) bbar[i] = bigbnd;
	for (i = 0; i < nlic; i++) bbar[p++] = *s++;
	for (i = 0; i < nltc; i++, s++) for (j = 0; j < nbps; j++) bbar[p++] = *s;
	for (i = 0; i < nlfc; i++) bbar[p++] = *s++;
	for (i = 0; i < nnlic; i++) bbar[p++] = *s++;
	for (i = 0; i < nnltc; i++, s++) for (j = 0; j < nbps; j++) bbar[p++] = *s;
	for (i = 0; i < nnlfc; i++) bbar[p++] = *s++;
}

/* user Jacobian storage: DoubleFMatrix(nz rows, ncon cols) => dc[con][var], contiguous */
static double **make_dc(int nz, int ncon)
{
	double **d = orc_tmp_malloc((ncon > 0 ? ncon : 1) * sizeof(double *));
	int c;
	d[0] = orc_tmp_calloc((size_t)nz * (ncon > 0 ? ncon : 1), sizeof(double));
	for (c = 1; c < ncon; c++) d[c] = d[0] + (size_t)c * nz;
	return d;
}
static void free_dc(double **d) { orc_tmp_free(d[0]); orc_tmp_free(d); }

/* constraints.c:88-117 / :165-195.  dIdC points at row `row0` of the (ldJ x nC) Jacobian */
static void nl_IF(int final, int *mode, int *nstate, int ncon, double *c, double *dIdC, int ldJ,
                  orc_nlic_t func, const orc_colloc *cc, double *Z)
{
	double **zp = orc_tmp_malloc(cc->nout * sizeof(double *));
	if (final) zp_F(zp, Z, cc); else zp_I(zp, Z, cc);
	if (*mode == 0) {
		func(mode, nstate, c, NULL, zp);
	} else if (*mode == 1 || *mode == 2) {
		double **dIdz = make_dc(cc->nz, ncon);
		double *dIdZ = orc_tmp_calloc((size_t)ncon * cc->nZ, sizeof(double));
		func(mode, nstate, c, dIdz, zp);
		if (final) { scat_F(dIdZ, ncon, ncon, (const double *const *)dIdz, cc); mult_F(dIdC, ldJ, dIdZ, ncon, ncon, cc); }
		else       { scat_I(dIdZ, ncon, ncon, (const double *const *)dIdz, cc); mult_I(dIdC, ldJ, dIdZ, ncon, ncon, cc); }
		free_dc(dIdz); orc_tmp_free(dIdZ);
	}
	orc_tmp_free(zp);
}
/* constraints.c:120-162 -- keeps the dense (nbps*ncon x nZ) temporary, calloc'd per call */
static void nl_T(int *mode, int *nstate, int ncon, double *c, double *dIdC, int ldJ,
                 orc_nltc_t func, const orc_colloc *cc, double *Z)
{
	double **zp = orc_tmp_malloc(cc->nout * sizeof(double *));
	double *tmp = orc_tmp_malloc(ncon * sizeof(double));
	int P = cc->nbps, i, j;
	if (*mode == 0) {
		for (i = 0; i < P; i++) {
			zp_T(zp, Z, cc, i);
			func(mode, nstate, &i, tmp, NULL, zp);
			for (j = 0; j < ncon; j++) c[j * P + i] = tmp[j];
		}
	} else if (*mode == 1 || *mode == 2) {
		double **dIdz = make_dc(cc->nz, ncon);
		int rows = P * ncon;
		double *dIdZ = orc_tmp_calloc((size_t)rows * cc->nZ, sizeof(double));
		for (i = 0; i < P; i++) {
			zp_T(zp, Z, cc, i);
			func(mode, nstate, &i, tmp, dIdz, zp);
			for (j = 0; j < ncon; j++) c[j * P + i] = tmp[j];
			scat_T(dIdZ, rows, ncon, (const double *const *)dIdz, cc, i);
		}
		mult_T(dIdC, ldJ, dIdZ, rows, rows, cc);
		free_dc(dIdz); orc_tmp_free(dIdZ);
	}
	orc_tmp_free(tmp); orc_tmp_free(zp);
}

/* constraints.c:198-261: A = [lic; ltc (constraint-major x bp); lfc], col-major ld = nclin */
static void lin_matrix(double *A, int ldA, int nlic, double **lic, int nltc, double **ltc,
                       int nlfc, double **lfc, const orc_colloc *cc)
{
	int i1 = 0, i;
	if (nlic) {
		double *dIdZ = calloc((size_t)nlic * cc->nZ, sizeof(double));
		scat_I(dIdZ, nlic, nlic, (const double *const *)lic, cc);
		mult_I(A, ldA, dIdZ, nlic, nlic, cc);
		free(dIdZ); i1 = nlic;
	}
	if (nltc) {
		int rows = cc->nbps * nltc;
		double *dIdZ = calloc((size_t)rows * cc->nZ, sizeof(double));
		for (i = 0; i < cc->nbps; i++) scat_T(dIdZ, rows, nltc, (const double *const *)ltc, cc, i);
		mult_T(A + i1, ldA, dIdZ, rows, rows, cc);
		free(dIdZ); i1 += rows;
	}
	if (nlfc) {
		double *dIdZ = calloc((size_t)nlfc * cc->nZ, sizeof(double));
		scat_F(dIdZ, nlfc, nlfc, (const double *const *)lfc, cc);
		mult_F(A + i1, ldA, dIdZ, nlfc, nlfc, cc);
		free(dIdZ);
	}
}

orc_problem *orc_problem_make(
	int nout, double *bps, int nbps, int *kninterv, double **knots, int *order,
	int *mult, int *maxderiv,
	int nlic, double **lic, int nltc, double **ltc, int nlfc, double **lfc,
	int nnlic, orc_nlic_t nlicf, int nnltc, orc_nltc_t nltcf, int nnlfc, orc_nlic_t nlfcf,
	int nicav, orc_AV *icav, int ntcav, orc_AV *tcav, int nfcav, orc_AV *fcav,
	double *lowerb, double *upperb,
	int nicf, orc_icf_t icf, int nucf, orc_ucf_t ucf, int nfcf, orc_icf_t fcf,
	int nicostav, orc_AV *icostav, int ntcostav, orc_AV *tcostav, int nfcostav, orc_AV *fcostav)
{
	orc_problem *p = calloc(1, sizeof(*p));
	int n, ntot;
	p->cc = orc_colloc_make(nout, knots, kninterv, bps, nbps, maxderiv, order, mult);
	n = p->cc->nC;
	p->Z = calloc(p->cc->nZ, sizeof(double));                          /* ntg.c:119 */
	p->nlic = nlic; p->nltc = nltc; p->nlfc = nlfc;
	p->nnlic = nnlic; p->nnltc = nnltc; p->nnlfc = nnlfc;
	p->nlicf = nlicf; p->nltcf = nltcf; p->nlfcf = nlfcf;
	p->nicav = nicav; p->icav = icav; p->ntcav = ntcav; p->tcav = tcav; p->nfcav = nfcav; p->fcav = fcav;
	p->nicf = nicf; p->nucf = nucf; p->nfcf = nfcf; p->icf = icf; p->ucf = ucf; p->fcf = fcf;
	p->nicostav = nicostav; p->icostav = icostav; p->ntcostav = ntcostav; p->tcostav = tcostav;
	p->nfcostav = nfcostav; p->fcostav = fcostav;
	p->nclin = nlic + nltc * nbps + nlfc;                               /* ntg.c:156 */
	p->ncnln = nnlic + nnltc * nbps + nnlfc;                            /* ntg.c:157 */
	if (p->nclin == 0) p->A = calloc(1, sizeof(double));               /* ntg.c:162-166 */
	else {
		p->A = calloc((size_t)p->nclin * n, sizeof(double));
		lin_matrix(p->A, p->nclin, nlic, lic, nltc, ltc, nlfc, lfc, p->cc);
	}
	p->cJac = calloc((size_t)(p->ncnln ? p->ncnln : 1) * (p->ncnln ? n : 1), sizeof(double)); /* ntg.c:210-220 */
	ntot = n + p->nclin + p->ncnln;
	p->bu = calloc(ntot, sizeof(double)); p->bl = calloc(ntot, sizeof(double));
	orc_bounds(p->bu, upperb, n, nlic, nltc, nlfc, nnlic, nnltc, nnlfc, nbps, DBL_MAX);   /* ntg.c:224-229 */
	orc_bounds(p->bl, lowerb, n, nlic, nltc, nlfc, nnlic, nnltc, nnlfc, nbps, -DBL_MAX);
	return p;
}

void orc_problem_free(orc_problem *p)
{
	orc_colloc_free(p->cc);
	free(p->Z); free(p->A); free(p->cJac); free(p->bl); free(p->bu); free(p);
}

/* "cpu-opt" flavour of the running cost: same numbers as cost_T up to summation order, without its dense
 * nbps x nC temporary: Z of the cost variables at a breakpoint, the callback, the trapezoid as node weights, the banded
 * M' (w df) accumulated straight into the gradient.  No allocation per call beyond two small stack arrays. */
static void cost_T_banded(double *I, double *dI, const orc_problem *p, const double *x)
{
	const orc_colloc *cc = p->cc;
	int P = cc->nbps, i, a, q, md = 2, ns = 0, o;
	double z[64], df[64], *zp[16], f, acc = 0.0;
	for (i = 0; i < cc->nC; i++) dI[i] = 0.0;
	for (o = 0; o < cc->nout; o++) zp[o] = z + cc->iz[o];
	for (i = 0; i < cc->nz; i++) z[i] = 0.0;
	for (i = 0; i < P; i++) {
		double w = 0.0;
		if (i > 0) w += (cc->bps[i] - cc->bps[i - 1]) / 2;
		if (i < P - 1) w += (cc->bps[i + 1] - cc->bps[i]) / 2;
		for (a = 0; a < p->ntcostav; a++) {
			const int oo = p->tcostav[a].output, r = p->tcostav[a].deriv, k = cc->order[oo];
			const double *c = x + cc->iC[oo] + cc->off[oo][i];
			double s = 0.0;
			for (q = 0; q < k; q++) s += BLK(cc, oo, i, q, r) * c[q];
			z[cc->iz[oo] + r] = s;
		}
		p->ucf(&md, &ns, &i, &f, df, zp);
		acc += w * f;
		for (a = 0; a < p->ntcostav; a++) {
			const int oo = p->tcostav[a].output, r = p->tcostav[a].deriv, k = cc->order[oo];
			double *g = dI + cc->iC[oo] + cc->off[oo][i];
			const double wd = w * df[cc->iz[oo] + r];
			for (q = 0; q < k; q++) g[q] += wd * BLK(cc, oo, i, q, r);
		}
	}
	*I = acc;
}

/* ntg.c:274-335.  Quirk kept: mode 0 tests n?cf==1, modes 1/2 test !=0 (ntg.c:297-301 vs 309-313). */
void orc_funobj(orc_problem *p, int *mode, const double *x, double *y, double *yprime, int *nstate)
{
	const orc_colloc *cc = p->cc;
	double I = 0.0, In = 0.0, F = 0.0, *dI, *dIn, *dF;
	int i;
	orc_tmp_reset();
	if (p->banded && *mode == 2 && p->nicf == 0 && p->nfcf == 0 && p->nucf != 0 && cc->nz <= 64 && cc->nout <= 16) { cost_T_banded(y, yprime, p, x); return; }
	if (p->nicf != 0) orc_updateZ(p->Z, cc, x, p->icostav, p->nicostav, ORC_AVINITIAL);
	if (p->nucf != 0) orc_updateZ(p->Z, cc, x, p->tcostav, p->ntcostav, ORC_AVTRAJECTORY);
	if (p->nfcf != 0) orc_updateZ(p->Z, cc, x, p->fcostav, p->nfcostav, ORC_AVFINAL);
	switch (*mode) {
	case 0:
		if (p->nicf == 1) cost_I(mode, nstate, &I, NULL, p->icf, cc, p->Z);
		if (p->nucf == 1) cost_T(mode, nstate, &In, NULL, cc->bps, p->ucf, cc, p->Z);
		if (p->nfcf == 1) cost_F(mode, nstate, &F, NULL, p->fcf, cc, p->Z);
		*y = I + In + F;
		break;
	case 1:
	case 2:
		dI = orc_tmp_calloc(cc->nC, sizeof(double)); dIn = orc_tmp_calloc(cc->nC, sizeof(double)); dF = orc_tmp_calloc(cc->nC, sizeof(double));
		if (p->nicf != 0) cost_I(mode, nstate, &I, dI, p->icf, cc, p->Z);
		if (p->nucf != 0) cost_T(mode, nstate, &In, dIn, cc->bps, p->ucf, cc, p->Z);
		if (p->nfcf != 0) cost_F(mode, nstate, &F, dF, p->fcf, cc, p->Z);
		if (*mode == 2) *y = I + In + F;
		for (i = 0; i < cc->nC; i++) yprime[i] = dI[i] + dIn[i] + dF[i];   /* Vector3Add matrix.c:177 */
		orc_tmp_free(dI); orc_tmp_free(dIn); orc_tmp_free(dF);
		break;
	default:
		*nstate = -1;
	}
}

/* ntg.c:337-371 + constraints.c:36-85: rows = [initial; trajectory (con-major x bp); final] */
void orc_funcon(orc_problem *p, int *mode, const double *x, double *c, double *cJac, int *nstate)
{
	const orc_colloc *cc = p->cc;
	int i1 = 0, ldJ = p->ncnln ? p->ncnln : 1;
	double *J = cJac ? cJac : p->cJac;
	orc_tmp_reset();
	if (p->nnlic != 0) orc_updateZ(p->Z, cc, x, p->icav, p->nicav, ORC_AVINITIAL);
	if (p->nnltc != 0) orc_updateZ(p->Z, cc, x, p->tcav, p->ntcav, ORC_AVTRAJECTORY);
	if (p->nnlfc != 0) orc_updateZ(p->Z, cc, x, p->fcav, p->nfcav, ORC_AVFINAL);
	if (*mode < 0 || *mode > 2) { *mode = -1; return; }
	if (p->nnlic != 0) { nl_IF(0, mode, nstate, p->nnlic, c, J, ldJ, p->nlicf, cc, p->Z); i1 = p->nnlic; }
	if (p->nnltc != 0) { nl_T(mode, nstate, p->nnltc, c + i1, J + i1, ldJ, p->nltcf, cc, p->Z); i1 += p->nnltc * cc->nbps; }
	if (p->nnlfc != 0) nl_IF(1, mode, nstate, p->nnlfc, c + i1, J + i1, ldJ, p->nlfcf, cc, p->Z);
}

/* colloc.c:449-484 */
void orc_spline_interp(double *f, double x, double *knots, int ninterv, double *coefs,
                       int ncoefs, int order, int mult, int maxderiv)
{
	int n = ninterv * (order - mult) + mult, naug = n + order, nn, left1, left2, mflag, i, j, offset;
	double *aug = malloc(naug * sizeof(double));
	double *a = malloc(order * order * sizeof(double));
	double *db = malloc(order * maxderiv * sizeof(double));
	assert(n == ncoefs);
	orc_knots(knots, ninterv, order, mult, aug, &nn);
	orc_interv(aug, naug, x, &left1, &mflag);
	orc_bsplvd(aug, order, x, left1, a, db, maxderiv);
	orc_interv(knots, ninterv + 1, x, &left2, &mflag);
	offset = (left2 - 1) * (order - mult);
	for (i = 0; i < maxderiv; i++) {
		f[i] = 0.0;
		for (j = 0; j < order; j++) f[i] += db[i * order + j] * coefs[offset + j];
	}
	free(aug); free(a); free(db);
}
