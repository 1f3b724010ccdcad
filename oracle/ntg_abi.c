/*
 * oracle/ntg_abi.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, see oracle.h).
 *
 * Exposes the oracle behind the reference's public C ABI (ntg.h:72-104, matrix.h:37-45,
 * colloc.h:103-105) so that /root/reference/examples/{vanderpol,kincar}.c can be compiled
 * UNCHANGED against include/ntg.h and linked against liborc_ntg.so in the CPU test suite.
 * This pins the oracle on the only executable "tests" the reference has (its examples).
 * The product's own ntg() lives in ntg_amd/csrc and never calls into this file.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <ctype.h>
#include "oracle.h"

typedef struct { double **elements; int rows, cols; } Matrix; /* matrix.h:27-31 */

static orc_sqp_opts g_opts; static int g_opts_init = 0; static int g_print_level = 10;

void printNTGBanner(void)
{
	printf("\n  NTG-compatible CPU oracle (MI355X build test infrastructure)\n\n");
}

/* npsoloption(): ntg.c:269-272 forwards a free-form string to NPSOL's npoptn_.  The oracle
 * understands the strings used in-tree plus the solver knobs. */
void npsoloption(char *option)
{
	char buf[256]; char *eq; double v = 0; size_t i;
	if (!g_opts_init) { orc_sqp_default_opts(&g_opts); g_opts_init = 1; }
	strncpy(buf, option, sizeof(buf) - 1); buf[sizeof(buf) - 1] = 0;
	for (i = 0; buf[i]; i++) buf[i] = (char)tolower((unsigned char)buf[i]);
	eq = strchr(buf, '=');
	{ char *q = buf + strlen(buf); while (q > buf && (isdigit((unsigned char)q[-1]) || strchr(".e+-", q[-1]))) q--; v = atof(eq ? eq + 1 : q); }
	if (!strncmp(buf, "nolist", 6) || !strncmp(buf, "derivative level", 16) || !strncmp(buf, "summary file", 12)) return;
	if (!strncmp(buf, "print level", 11)) { g_print_level = (int)v; return; }
	if (!strncmp(buf, "major iteration limit", 21)) { g_opts.itlim = (int)v; return; }
	if (!strncmp(buf, "optimality tolerance", 20)) { g_opts.opttol = v; return; }
	if (!strncmp(buf, "line search tolerance", 21)) { g_opts.ls_eta = v; return; }
	if (!strncmp(buf, "step limit", 10)) { g_opts.steplimit = v; return; }
	if (!strncmp(buf, "hessian", 7)) { g_opts.hessian = strstr(buf, "colloc") ? 1 : 0; return; }
	fprintf(stderr, "ntg oracle: npsoloption '%s' ignored\n", option);
}

void linspace(double *v, double d0, double d1, int n) { orc_linspace(v, d0, d1, n); }

void ntg(int nout, double *bps, int nbps, int *kninterv, double **knots, int *order, int *mult,
         int *maxderiv, double *initialguess,
         int nlic, double **lic, int nltc, double **ltc, int nlfc, double **lfc,
         int nnlic, orc_nlic_t nlicf, int nnltc, orc_nltc_t nltcf, int nnlfc, orc_nlic_t nlfcf,
         int nicav, orc_AV *icav, int ntcav, orc_AV *tcav, int nfcav, orc_AV *fcav,
         double *lowerb, double *upperb,
         int nicf, orc_icf_t icf, int nucf, orc_ucf_t ucf, int nfcf, orc_icf_t fcf,
         int nicostav, orc_AV *icostav, int ntcostav, orc_AV *tcostav, int nfcostav, orc_AV *fcostav,
         int *istate, double *clambda, double *R, int *inform, double *objective)
{
	orc_problem *p; orc_sqp_result res;
	if (!g_opts_init) { orc_sqp_default_opts(&g_opts); g_opts_init = 1; }
	printNTGBanner();                                                   /* ntg.c:161 */
	p = orc_problem_make(nout, bps, nbps, kninterv, knots, order, mult, maxderiv,
		nlic, lic, nltc, ltc, nlfc, lfc, nnlic, nlicf, nnltc, nltcf, nnlfc, nlfcf,
		nicav, icav, ntcav, tcav, nfcav, fcav, lowerb, upperb,
		nicf, icf, nucf, ucf, nfcf, fcf, nicostav, icostav, ntcostav, tcostav, nfcostav, fcostav);
	orc_sqp_solve(p, initialguess, &g_opts, &res, clambda, istate, R, NULL, 0);
	*inform = res.inform; *objective = res.objective;
	if (g_print_level > 0)
		printf(" Exit oracle SQP - inform %d, majors %d, nfev %d, objective %.15g\n", res.inform, res.iters, res.nfev, res.objective);
	orc_problem_free(p);
}

void SplineInterp(double *f, double x, double *knots, int ninterv, double *coefs, int ncoefs,
                  int order, int mult, int maxderiv)
{ orc_spline_interp(f, x, knots, ninterv, coefs, ncoefs, order, mult, maxderiv); }

/* matrix.h:37-45 helpers the examples link (matrix.c:211-330) */
double **DoubleMatrix(int rows, int cols)
{
	double **t = malloc(rows * sizeof(double *)); int i;
	t[0] = calloc((size_t)rows * cols, sizeof(double));
	for (i = 1; i < rows; i++) t[i] = t[0] + (size_t)i * cols;
	return t;
}
void FreeDoubleMatrix(double **d) { free(d[0]); free(d); }
Matrix *MakeMatrix(int rows, int cols)
{ Matrix *m = malloc(sizeof(Matrix)); m->elements = DoubleMatrix(rows, cols); m->rows = rows; m->cols = cols; return m; }
void FreeMatrix(Matrix *m) { FreeDoubleMatrix(m->elements); free(m); }
static FILE *open_out(char *fn) { if (!strcmp(fn, "stdout")) return stdout; if (!strcmp(fn, "stderr")) return stderr; return fopen(fn, "w"); }
static void close_out(FILE *f) { if (f && f != stdout && f != stderr) fclose(f); }
void PrintMatrix(char *fn, Matrix *m)
{
	FILE *f = open_out(fn); int i, j; if (!f) return;
	for (i = 0; i < m->rows; i++) { for (j = 0; j < m->cols; j++) fprintf(f, "%f ", m->elements[i][j]); fprintf(f, "\n"); }
	fprintf(f, "\n\n\n"); close_out(f);
}
void PrintVector(char *fn, double *v, int n)
{ FILE *f = open_out(fn); int i; if (!f) return; for (i = 0; i < n; i++) fprintf(f, "%g ", v[i]); fprintf(f, "\n"); close_out(f); }
void PrintiVector(char *fn, int *v, int n)
{ FILE *f = open_out(fn); int i; if (!f) return; for (i = 0; i < n; i++) fprintf(f, "%d\n", v[i]); close_out(f); }
