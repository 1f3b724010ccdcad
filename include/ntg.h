/*
 * include/ntg.h -- drop-in public header of the MI355X-native NTG engine.
 *
 * Source-compatible with the reference's ntg.h (which transitively pulls av.h, colloc.h,
 * constraints.h, cost.h, matrix.h): a program written against murrayrm/ntg
 * (examples/vanderpol.c, examples/kincar.c) compiles unchanged with -I include and links
 * against libntg_amd.so.  Each declaration cites the reference interface it replaces.
 *
 * The examples include only "ntg.h" and rely on it for <stdio.h>, <float.h>, <assert.h>,
 * <stdlib.h>, <string.h> (reference ntg.h:25-31, colloc.h:25-29, matrix.h:22-25;
 * kincar.c:279 uses assert without including it).
 */
#ifndef NTG_AMD_NTG_H
#define NTG_AMD_NTG_H

#include <stdio.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAXNOUT 5 /* reference ntg.h:22 (unused by the reference, kept for source compat) */

/* active variable: reference av.h:18-26 */
#define AVINITIAL 0
#define AVTRAJECTORY 1
#define AVFINAL 2
typedef struct AVStruct {
	int output;
	int deriv;
} AV;

/* row-major matrix with a contiguous slab at elements[0]: reference matrix.h:27-31 */
typedef struct MatrixStruct {
	double **elements;
	int rows, cols;
} Matrix;

/* reference matrix.h:37-45 (matrix.c:211-330) */
Matrix *MakeMatrix(int rows, int cols);
void FreeMatrix(Matrix *matrix);
void PrintMatrix(char *filename, Matrix *matrix);
void PrintVector(char *filename, double *f, int nf);
void PrintiVector(char *filename, int *f, int nf);
double **DoubleMatrix(int rows, int cols);
void FreeDoubleMatrix(double **d);

/* reference colloc.h:103-105 (colloc.c:449-484): value and first maxderiv-1 derivatives of
 * one output at x from its B-spline coefficients. */
void SplineInterp(double *f, double x, double *knots, int ninterv, double *coeffs,
                  int ncoeffs, int order, int mult, int maxderiv);

/* reference ntg.h:72-99 (ntg.c:54-267).  Same argument meaning, ownership and outputs:
 * initialguess[nC] is in/out; istate/clambda have nC + nclin + ncnln entries; R >= nC*nC;
 * *inform uses NPSOL's codes (0 optimal, 1 optimal but not to the requested accuracy, 3 nonlinear
 * constraints not satisfied, 4 iteration limit, 6 no further progress, 9 invalid input / no GPU). */
void ntg(
	int nout, double *bps, int nbps, int *kninterv, double **knots,
	int *order, int *mult, int *max_deriv,
	double *initialguess,

	int nlic, double **lic,
	int nltc, double **ltc,
	int nlfc, double **lfc,

	int nnlic, void (*nlicf)(int *, int *, double *, double **, double **),
	int nnltc, void (*nltcf)(int *, int *, int *, double *, double **, double **),
	int nnlfc, void (*nlfcf)(int *, int *, double *, double **, double **),
	int ninitialconstrav, AV *initialconstrav,
	int ntrajectoryconstrav, AV *trajectoryconstrav,
	int nfinalconstrav, AV *finalconstrav,

	double *lowerb, double *upperb,

	int nicf, void (*icf)(int *, int *, double *, double *, double **),
	int nucf, void (*ucf)(int *, int *, int *, double *, double *, double **),
	int nfcf, void (*fcf)(int *, int *, double *, double *, double **),
	int ninitialcostav, AV *initialcostav,
	int ntrajectorycostav, AV *trajectorycostav,
	int nfinalcostav, AV *finalcostav,

	int *istate, double *clambda, double *R,
	int *inform, double *objective);

/* reference ntg.h:100 (ntg.c:269-272): NPSOL option string.  Understood keys: "nolist",
 * "print level N", "summary file = N", "derivative level = N" (accepted, no effect);
 * "major iteration limit = N", "optimality tolerance = X", "line search tolerance = X",
 * "step limit = X", "hessian = identity|colloc".  Unknown keys: warning on stderr. */
void npsoloption(char *type);

/* reference ntg.h:103 (ntg.c:374-389): cumulative-add linspace */
void linspace(double *v, double d0, double d1, int n);
/* reference ntg.h:104 (ntg.c:391-405) */
void printNTGBanner(void);

#ifdef __cplusplus
}
#endif
#endif
