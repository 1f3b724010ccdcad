/*
 * tests/drivers/dropin_drv.c -- exercises the drop-in boundary (include/ntg.h) the way the
 * reference's example programs do, without needing the reference tree at run time.
 * The problem DATA are those of examples/vanderpol.c:17-21,160-169 and
 * examples/kincar.c:133-137,319-339 (sizes, orders, boundary values); the callbacks are
 * written here from the problem statements (vanderpol.txt; kincar cost = xdd^2 + ydd^2).
 * "obstacle": the kincar lane change with order-6 splines on 10 intervals and the nonlinear trajectory
 * inequality (x-20)^2 + (y-0.5)^2 >= 9 (host callback) -- exercises the nonlinear-constraint path of ntg().
 * "ineq": the shipped kincar problem with two final-flag rows relaxed to ranges (linear inequality rows).
 * "sequence": vanderpol, kincar, vanderpol again in ONE process (ntg() keeps file-scope state like the reference,
 * ntg.c:17-41: repeated calls with different shapes must not leak into each other); options persist across the calls
 * as NPSOL's do.  "kincarR": kincar, then the returned factor R (row-major, n x n).
 * Usage: dropin_drv vanderpol|kincar|obstacle|ineq|sequence|kincarR   -> prints "RESULT inform objective c0 c1 ..."
 */
#include <math.h>
#include "ntg.h"

static void vdp_cost(int *mode, int *nstate, int *i, double *f, double *df, double **zp)
{
	double z = zp[0][0], zd = zp[0][1], zdd = zp[0][2], u = zdd + z - (1.0 - z * z) * zd;
	(void)nstate; (void)i;
	if (*mode == 0 || *mode == 2) *f = 0.5 * (z * z + zd * zd + u * u);
	if (*mode == 1 || *mode == 2) { df[0] = z + u * (1.0 + 2.0 * z * zd); df[1] = zd - u * (1.0 - z * z); df[2] = u; }
}
static void car_cost(int *mode, int *nstate, int *i, double *f, double *df, double **zp)
{
	(void)nstate; (void)i;
	if (*mode == 0 || *mode == 2) *f = zp[0][2] * zp[0][2] + zp[1][2] * zp[1][2];
	if (*mode == 1 || *mode == 2) { df[0] = df[1] = df[3] = df[4] = 0; df[2] = 2 * zp[0][2]; df[5] = 2 * zp[1][2]; }
}

static void obs_con(int *mode, int *nstate, int *i, double *c, double **dc, double **zp)
{
	double dx = zp[0][0] - 20.0, dy = zp[1][0] - 0.5; int v; (void)nstate; (void)i;
	if (*mode == 0 || *mode == 2) c[0] = dx * dx + dy * dy;
	if (*mode == 1 || *mode == 2) { for (v = 0; v < 6; v++) dc[0][v] = 0; dc[0][0] = 2 * dx; dc[0][3] = 2 * dy; }
}
static int g_obstacle = 0, g_printR = 0;

static int run(int nout, int order_, int mult_, int ninterv_, int nbps, int nlic, double **lic, int nlfc, double **lfc,
               double *lowerb, double *upperb, void (*ucf)(int *, int *, int *, double *, double *, double **),
               int ntav, AV *tav)
{
	int order[2] = {order_, order_}, mult[2] = {mult_, mult_}, nint[2] = {ninterv_, ninterv_}, md[2] = {3, 3};
	double *knots[2], *bps = calloc(nbps, sizeof(double));
	int ncoef = nout * (ninterv_ * (order_ - mult_) + mult_), nc = nlic + nlfc + (g_obstacle ? nbps : 0), i, inform;
	static AV ctav[2] = {{0, 0}, {1, 0}};
	double *coef = calloc(ncoef, sizeof(double)), objective;
	int *istate = calloc(ncoef + nc, sizeof(int));
	double *clambda = calloc(ncoef + nc, sizeof(double)), *R = calloc((ncoef + 1) * (ncoef + 1), sizeof(double));
	for (i = 0; i < nout; i++) { knots[i] = calloc(ninterv_ + 1, sizeof(double)); linspace(knots[i], 0, 5, ninterv_ + 1); }
	linspace(coef, 1, 1, ncoef);
	linspace(bps, 0, 5, nbps);
	npsoloption("nolist");
	npsoloption("summary file = 0");
	if (g_obstacle) npsoloption("print level 0");
	ntg(nout, bps, nbps, nint, knots, order, mult, md, coef,
	    nlic, lic, 0, NULL, nlfc, lfc, 0, NULL, g_obstacle ? 1 : 0, g_obstacle ? obs_con : NULL, 0, NULL,
	    0, NULL, g_obstacle ? 2 : 0, g_obstacle ? ctav : NULL, 0, NULL,
	    lowerb, upperb, 0, NULL, 1, ucf, 0, NULL, 0, NULL, ntav, tav, 0, NULL,
	    istate, clambda, R, &inform, &objective);
	printf("RESULT %d %.17g", inform, objective);
	for (i = 0; i < ncoef; i++) printf(" %.17g", coef[i]);
	printf("\n");
	{ /* SplineInterp at t = 0 and t = 5 for output 0 */
		double f0[3], f5[3];
		SplineInterp(f0, 0.0, knots[0], ninterv_, coef, ncoef / nout, order_, mult_, 3);
		SplineInterp(f5, 5.0, knots[0], ninterv_, coef, ncoef / nout, order_, mult_, 3);
		printf("INTERP %.17g %.17g %.17g %.17g %.17g %.17g\n", f0[0], f0[1], f0[2], f5[0], f5[1], f5[2]);
	}
	printf("ISTATE");
	for (i = 0; i < nc; i++) printf(" %d", istate[ncoef + i]);
	printf("\n");
	if (g_printR) {   /* R as NPSOL leaves it: column-major, leading dimension n (ntg.c:234-236: ldR = n) */
		int j;
		printf("RMAT %d", ncoef);
		for (i = 0; i < ncoef; i++) for (j = 0; j < ncoef; j++) printf(" %.17g", R[(size_t)j * ncoef + i]);
		printf("\n");
	}
	return inform;
}

static int one(int argc, char **argv);
int main(int argc, char **argv)
{
	if (argc > 1 && !strcmp(argv[1], "sequence")) {
		char *a1[2] = {argv[0], "vanderpol"}, *a2[2] = {argv[0], "kincar"};
		int rc = one(2, a1);
		rc |= one(2, a2);
		rc |= one(2, a1);
		return rc;
	}
	if (argc > 1 && !strcmp(argv[1], "kincarR")) { char *a2[2] = {argv[0], "kincar"}; g_printR = 1; return one(2, a2); }
	return one(argc, argv);
}
static int one(int argc, char **argv)
{
	if (argc > 1 && !strcmp(argv[1], "vanderpol")) {
		static AV tav[3] = {{0, 0}, {0, 1}, {0, 2}};
		double **lic = DoubleMatrix(2, 3), **lfc = DoubleMatrix(1, 3), lo[3], up[3];
		lic[0][0] = 1.0; lo[0] = up[0] = 1.0;
		lic[1][1] = 1.0; lo[1] = up[1] = 0.0;
		lfc[0][0] = -1.0; lfc[0][1] = 1.0; lo[2] = up[2] = 1.0;
		return run(1, 5, 3, 2, 20, 2, lic, 1, lfc, lo, up, vdp_cost, 3, tav);
	} else {
		static AV tav[2] = {{0, 2}, {1, 2}};
		Matrix *lic = MakeMatrix(6, 6), *lfc = MakeMatrix(6, 6);
		double lo[13], up[13], zi[6] = {0, 8, 0, -2, 0, 0}, zf[6] = {40, 8, 0, 2, 0, 0};
		int i;
		for (i = 0; i < 6; i++) { lic->elements[i][i] = 1.0; lfc->elements[i][i] = 1.0; lo[i] = up[i] = zi[i]; lo[6 + i] = up[6 + i] = zf[i]; }
		if (argc > 1 && !strcmp(argv[1], "ineq")) {   /* linear inequality rows: y(T) in [-1,1], y''(T) in [-0.5,0.5] */
			lo[6 + 3] = -1.0; up[6 + 3] = 1.0; lo[6 + 5] = -0.5; up[6 + 5] = 0.5;
			return run(2, 5, 3, 2, 20, 6, lic->elements, 6, lfc->elements, lo, up, car_cost, 2, tav);
		}
		if (argc > 1 && !strcmp(argv[1], "obstacle")) {
			g_obstacle = 1; lo[12] = 9.0; up[12] = 1e20;
			return run(2, 6, 3, 10, 51, 6, lic->elements, 6, lfc->elements, lo, up, car_cost, 2, tav);
		}
		return run(2, 5, 3, 2, 20, 6, lic->elements, 6, lfc->elements, lo, up, car_cost, 2, tav);
	}
}
