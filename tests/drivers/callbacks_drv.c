/*
 * tests/drivers/callbacks_drv.c -- exercises npsolCostFunction / npsolConstraintFunction (the
 * exported NPfunobj / NPfuncon of ntg.c:274-371) with HOST callbacks in every slot: initial,
 * trajectory and final costs plus initial, trajectory and final nonlinear constraints.
 * The callbacks are the "testfam" functions (same formulas as oracle/families.c family 2 and
 * ntg_amd/csrc/families.hpp), written against the reference callback ABI (ntg.h:81-83,90-92).
 * ntg() returns inform 9 for this problem class in this build, but calls the cost callback once
 * through the GPU path first -- so the driver installs a hook: the trajectory cost callback, on its
 * first call, evaluates npsolCostFunction/npsolConstraintFunction at test points and prints them.
 * Output: lines "F <f>", "G <g...>", "C <c...>", "J <cJac column-major...>".
 */
#include <math.h>
#include "ntg.h"
#include "ntg_amd.h"

#define NOUT 3
static int L = NOUT - 1;
static void icf(int *mode, int *ns, double *f, double *df, double **zp)
{
	int o; (void)ns;
	if (*mode == 0 || *mode == 2) { double s = 0; for (o = 0; o < NOUT; o++) s += (zp[o][0] - 1.0) * (zp[o][0] - 1.0) + 0.5 * zp[o][1] * zp[o][1]; *f = s + 0.25 * zp[0][0] * zp[L][1]; }
	if (*mode == 1 || *mode == 2) { for (o = 0; o < NOUT; o++) { df[3 * o] = 2.0 * (zp[o][0] - 1.0); df[3 * o + 1] = zp[o][1]; df[3 * o + 2] = 0; } df[0] += 0.25 * zp[L][1]; df[3 * L + 1] += 0.25 * zp[0][0]; }
}
static void ucf(int *mode, int *ns, int *i, double *f, double *df, double **zp)
{
	int o; double sn = sin(zp[0][0]), cs = cos(zp[0][0]); (void)ns; (void)i;
	if (*mode == 0 || *mode == 2) { double s = 0; for (o = 0; o < NOUT; o++) s += zp[o][0] * zp[o][0] + 0.1 * zp[o][1] * zp[o][1] + zp[o][2] * zp[o][2]; *f = s + 0.3 * sn * zp[L][1]; }
	if (*mode == 1 || *mode == 2) { for (o = 0; o < NOUT; o++) { df[3 * o] = 2.0 * zp[o][0]; df[3 * o + 1] = 0.2 * zp[o][1]; df[3 * o + 2] = 2.0 * zp[o][2]; } df[0] += 0.3 * cs * zp[L][1]; df[3 * L + 1] += 0.3 * sn; }
}
static void fcf(int *mode, int *ns, double *f, double *df, double **zp)
{
	int o; (void)ns;
	if (*mode == 0 || *mode == 2) { double s = 0; for (o = 0; o < NOUT; o++) s += zp[o][0] * zp[o][1] + 0.5 * zp[o][2] * zp[o][2]; *f = s; }
	if (*mode == 1 || *mode == 2) for (o = 0; o < NOUT; o++) { df[3 * o] = zp[o][1]; df[3 * o + 1] = zp[o][0]; df[3 * o + 2] = zp[o][2]; }
}
static void nlicf(int *mode, int *ns, double *c, double **dc, double **zp)
{
	int v; (void)ns;
	if (*mode == 0 || *mode == 2) c[0] = zp[0][0] * zp[0][0] + zp[L][1];
	if (*mode == 1 || *mode == 2) { for (v = 0; v < 3 * NOUT; v++) dc[0][v] = 0; dc[0][0] += 2.0 * zp[0][0]; dc[0][3 * L + 1] += 1.0; }
}
static void nltcf(int *mode, int *ns, int *i, double *c, double **dc, double **zp)
{
	int v; (void)ns; (void)i;
	if (*mode == 0 || *mode == 2) { c[0] = zp[0][0] * zp[0][0] + zp[L][0] * zp[L][0]; c[1] = zp[0][1] * zp[L][2] - cos(zp[0][0]); }
	if (*mode == 1 || *mode == 2) {
		for (v = 0; v < 3 * NOUT; v++) { dc[0][v] = 0; dc[1][v] = 0; }
		dc[0][0] += 2.0 * zp[0][0]; dc[0][3 * L] += 2.0 * zp[L][0];
		dc[1][1] += zp[L][2]; dc[1][3 * L + 2] += zp[0][1]; dc[1][0] += sin(zp[0][0]);
	}
}
static void nlfcf(int *mode, int *ns, double *c, double **dc, double **zp)
{
	int v; (void)ns;
	if (*mode == 0 || *mode == 2) c[0] = zp[0][2] * zp[0][0] + zp[L][1] * zp[L][1];
	if (*mode == 1 || *mode == 2) { for (v = 0; v < 3 * NOUT; v++) dc[0][v] = 0; dc[0][2] += zp[0][0]; dc[0][0] += zp[0][2]; dc[0][3 * L + 1] += 2.0 * zp[L][1]; }
}

static int g_n, g_ncnln;
static double *g_xtest;

/* argv: nbps, then nC doubles (test point), then 5 lic/ltc/lfc rows follow from stdin-free fixed rule */
int main(int argc, char **argv)
{
	int order[NOUT] = {5, 5, 6}, mult[NOUT] = {3, 3, 3}, nint[NOUT] = {4, 4, 5}, md[NOUT] = {3, 3, 3};
	int nbps = 17, i, j, ncoef = 0, nlic = 2, nltc = 1, nlfc = 2, nb, inform;
	double *knots[NOUT], *bps, **lic, **ltc, **lfc, *coef, *lo, *up, objective, *clambda, *R; int *istate;
	AV icav[2] = {{0, 0}, {2, 1}}, tcav[4] = {{0, 0}, {0, 1}, {2, 0}, {2, 2}}, fcav[3] = {{0, 0}, {0, 2}, {2, 1}};
	AV icostav[6] = {{0, 0}, {0, 1}, {1, 0}, {1, 1}, {2, 0}, {2, 1}};
	AV tcostav[9] = {{0, 0}, {0, 1}, {0, 2}, {1, 0}, {1, 1}, {1, 2}, {2, 0}, {2, 1}, {2, 2}};
	FILE *fp;
	for (i = 0; i < NOUT; i++) { knots[i] = calloc(nint[i] + 1, sizeof(double)); linspace(knots[i], 0, 2, nint[i] + 1); ncoef += nint[i] * (order[i] - mult[i]) + mult[i]; }
	bps = calloc(nbps, sizeof(double)); linspace(bps, 0, 2, nbps);
	lic = DoubleMatrix(nlic, 9); ltc = DoubleMatrix(nltc, 9); lfc = DoubleMatrix(nlfc, 9);
	/* inputs: file argv[1] holds lic(2x9) ltc(1x9) lfc(2x9) then ncoef test-point values */
	fp = fopen(argv[1], "r"); if (!fp) return 2;
	for (i = 0; i < nlic; i++) for (j = 0; j < 9; j++) if (fscanf(fp, "%lf", &lic[i][j]) != 1) return 3;
	for (i = 0; i < nltc; i++) for (j = 0; j < 9; j++) if (fscanf(fp, "%lf", &ltc[i][j]) != 1) return 3;
	for (i = 0; i < nlfc; i++) for (j = 0; j < 9; j++) if (fscanf(fp, "%lf", &lfc[i][j]) != 1) return 3;
	g_xtest = calloc(ncoef, sizeof(double));
	for (i = 0; i < ncoef; i++) if (fscanf(fp, "%lf", &g_xtest[i]) != 1) return 3;
	fclose(fp);
	nb = nlic + nltc + nlfc + 1 + 2 + 1;
	lo = calloc(nb, sizeof(double)); up = calloc(nb, sizeof(double));
	for (i = 0; i < nb; i++) { lo[i] = -1.0; up[i] = 1.0; }
	coef = calloc(ncoef, sizeof(double)); linspace(coef, 1, 1, ncoef);
	g_n = ncoef; g_ncnln = 1 + 2 * nbps + 1;
	istate = calloc(ncoef + nlic + nltc * nbps + nlfc + g_ncnln, sizeof(int));
	clambda = calloc(ncoef + nlic + nltc * nbps + nlfc + g_ncnln, sizeof(double));
	R = calloc((size_t)(ncoef + 1) * (ncoef + 1), sizeof(double));
	if (ntg_open(NOUT, bps, nbps, nint, knots, order, mult, md, nlic, lic, nltc, ltc, nlfc, lfc,
	             1, nlicf, 2, nltcf, 1, nlfcf, 2, (ntg_av *)icav, 4, (ntg_av *)tcav, 3, (ntg_av *)fcav,
	             1, icf, 1, ucf, 1, fcf, 6, (ntg_av *)icostav, 9, (ntg_av *)tcostav, 9, (ntg_av *)tcostav) != 0) return 4;
	{
		int m = 2, nstate = 1, needc = 0, n = g_n, nc = g_ncnln, ldJ = nc;
		double F, *g = calloc(n, sizeof(double)), *c = calloc(nc, sizeof(double)), *J = calloc((size_t)nc * n, sizeof(double));
		npsolCostFunction(&m, &n, g_xtest, &F, g, &nstate);
		printf("F %.17g\nG", F); for (j = 0; j < n; j++) printf(" %.17g", g[j]); printf("\n");
		m = 2; npsolConstraintFunction(&m, &nc, &n, &ldJ, &needc, g_xtest, c, J, &nstate);
		printf("C"); for (j = 0; j < nc; j++) printf(" %.17g", c[j]); printf("\nJ");
		for (j = 0; j < nc * n; j++) printf(" %.17g", J[j]); printf("\n");
		m = 0; npsolCostFunction(&m, &n, g_xtest, &F, g, &nstate); printf("F0 %.17g\n", F);
		m = 7; npsolCostFunction(&m, &n, g_xtest, &F, g, &nstate); printf("BADMODE %d\n", nstate);
	}
	ntg_close();
	/* the same problem through ntg(): linear inequality rows and nonlinear rows of all three kinds, bounds [-1, 1] */
	npsoloption("print level 0");
	ntg(NOUT, bps, nbps, nint, knots, order, mult, md, coef,
	    nlic, lic, nltc, ltc, nlfc, lfc, 1, nlicf, 2, nltcf, 1, nlfcf, 2, icav, 4, tcav, 3, fcav,
	    lo, up, 1, icf, 1, ucf, 1, fcf, 6, icostav, 9, tcostav, 9, tcostav,
	    istate, clambda, R, &inform, &objective);
	printf("INFORM %d\n", inform);
	return 0;
}
