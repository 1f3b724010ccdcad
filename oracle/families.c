#define _GNU_SOURCE
#include <sched.h>
/*
 * oracle/families.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, see oracle.h).
 *
 * Host callbacks (reference callback ABI, ntg.h:81-83,90-92) for the problem
 * families the MI355X kernels provide as device functors, plus the batched CPU
 * driver used by tests and by bench.py's cpu_baseline leg.
 *   family 0  kincar     running cost  sum_o (z_o'')^2            (examples/kincar.c:105-117,
 *                        generalised from 2 outputs to nout outputs = nout/2 cars stacked)
 *   family 1  vanderpol  running cost  (z^2 + z'^2 + u^2)/2,  u = z'' + z - (1-z^2) z'
 *                        (examples/vanderpol.c:206-241, examples/vanderpol.txt)
 *   family 2  testfam    synthetic: every callback slot (initial/trajectory/final cost and
 *                        nonlinear constraints) populated with smooth nonlinear functions,
 *                        used only to exercise constraints.c/cost.c row orders.
 *   family 3  obstacle   kincar running cost (2 outputs x, y) + one nonlinear trajectory constraint
 *                        c = (x - 20)^2 + (y - 0.5)^2  (>= r^2 through the bounds): a circular obstacle on the
 *                        lane-change path -- build-defined (BASELINE configs D/E are, too), used to exercise
 *                        the nonlinear-constraint path of NPfuncon / npsol_ end to end.
 *   family 4  quadrotor  4 flat outputs (x, y, z, yaw), maxderiv 5 (BASELINE config D).  Running cost
 *                        snap^2 of the position outputs + yaw acceleration^2; two nonlinear trajectory
 *                        constraints: c0 = x''^2 + y''^2 + (z'' + g)^2 (mass-normalised thrust squared,
 *                        bounded on both sides) and c1 = x'^2 + y'^2 + z'^2 (speed squared, bounded above).
 *   family 5  manipulator  nout = 3*narms joint angles, maxderiv 3 (BASELINE config E: 12 outputs = 4 planar
 *                        3-link arms).  Running cost sum of joint accelerations squared; one nonlinear
 *                        trajectory inequality per arm: tip height sin(qa) + sin(qa+qb) + sin(qa+qb+qc)
 *                        (bounded above by a ceiling through the bounds).
 * Families 0-3 and 5 use maxderiv == 3 for every output (flat index iz[o] = 3*o); family 4 uses 5.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "oracle.h"

static __thread int fam_nout = 2;
void orc_family_set_nout(int nout) { fam_nout = nout; }

/* ---- family 0 ---- */
static void kincar_ucf(int *mode, int *nstate, int *i, double *f, double *df, double **zp)
{
	int o; (void)nstate; (void)i;
	if (*mode == 0 || *mode == 2) { double s = 0.0; for (o = 0; o < fam_nout; o++) s += zp[o][2] * zp[o][2]; *f = s; }
	if (*mode == 1 || *mode == 2)
		for (o = 0; o < fam_nout; o++) { df[3 * o] = 0; df[3 * o + 1] = 0; df[3 * o + 2] = 2 * zp[o][2]; }
}
/* ---- family 1 ---- */
static void vdp_ucf(int *mode, int *nstate, int *i, double *f, double *df, double **zp)
{
	double z = zp[0][0], zd = zp[0][1], zdd = zp[0][2], t1 = z * z, u = zdd + z - (1.0 - t1) * zd;
	(void)nstate; (void)i;
	if (*mode == 0 || *mode == 2) *f = t1 / 2.0 + zd * zd / 2.0 + u * u / 2.0;
	if (*mode == 1 || *mode == 2) { df[0] = z + u * (1.0 + 2.0 * z * zd); df[1] = zd - u * (1.0 - t1); df[2] = u; }
}
/* ---- family 2 ---- */
static void tf_icf(int *mode, int *nstate, double *f, double *df, double **zp)
{
	int o, L = fam_nout - 1; (void)nstate;
	if (*mode == 0 || *mode == 2) { double s = 0; for (o = 0; o < fam_nout; o++) s += (zp[o][0] - 1.0) * (zp[o][0] - 1.0) + 0.5 * zp[o][1] * zp[o][1]; *f = s + 0.25 * zp[0][0] * zp[L][1]; }
	if (*mode == 1 || *mode == 2) {
		for (o = 0; o < fam_nout; o++) { df[3 * o] = 2.0 * (zp[o][0] - 1.0); df[3 * o + 1] = zp[o][1]; df[3 * o + 2] = 0; }
		df[0] += 0.25 * zp[L][1]; df[3 * L + 1] += 0.25 * zp[0][0];
	}
}
static void tf_ucf(int *mode, int *nstate, int *i, double *f, double *df, double **zp)
{
	int o, L = fam_nout - 1; double sn = sin(zp[0][0]), cs = cos(zp[0][0]); (void)nstate; (void)i;
	if (*mode == 0 || *mode == 2) { double s = 0; for (o = 0; o < fam_nout; o++) s += zp[o][0] * zp[o][0] + 0.1 * zp[o][1] * zp[o][1] + zp[o][2] * zp[o][2]; *f = s + 0.3 * sn * zp[L][1]; }
	if (*mode == 1 || *mode == 2) {
		for (o = 0; o < fam_nout; o++) { df[3 * o] = 2.0 * zp[o][0]; df[3 * o + 1] = 0.2 * zp[o][1]; df[3 * o + 2] = 2.0 * zp[o][2]; }
		df[0] += 0.3 * cs * zp[L][1]; df[3 * L + 1] += 0.3 * sn;
	}
}
static void tf_fcf(int *mode, int *nstate, double *f, double *df, double **zp)
{
	int o; (void)nstate;
	if (*mode == 0 || *mode == 2) { double s = 0; for (o = 0; o < fam_nout; o++) s += zp[o][0] * zp[o][1] + 0.5 * zp[o][2] * zp[o][2]; *f = s; }
	if (*mode == 1 || *mode == 2) for (o = 0; o < fam_nout; o++) { df[3 * o] = zp[o][1]; df[3 * o + 1] = zp[o][0]; df[3 * o + 2] = zp[o][2]; }
}
/* nnlic = 1 */
static void tf_nlicf(int *mode, int *nstate, double *c, double **dc, double **zp)
{
	int L = fam_nout - 1, v; (void)nstate;
	if (*mode == 0 || *mode == 2) c[0] = zp[0][0] * zp[0][0] + zp[L][1];
	if (*mode == 1 || *mode == 2) { for (v = 0; v < 3 * fam_nout; v++) dc[0][v] = 0; dc[0][0] += 2.0 * zp[0][0]; dc[0][3 * L + 1] += 1.0; }
}
/* nnltc = 2 */
static void tf_nltcf(int *mode, int *nstate, int *i, double *c, double **dc, double **zp)
{
	int L = fam_nout - 1, v; (void)nstate; (void)i;
	if (*mode == 0 || *mode == 2) {
		c[0] = zp[0][0] * zp[0][0] + zp[L][0] * zp[L][0];
		c[1] = zp[0][1] * zp[L][2] - cos(zp[0][0]);
	}
	if (*mode == 1 || *mode == 2) {
		for (v = 0; v < 3 * fam_nout; v++) { dc[0][v] = 0; dc[1][v] = 0; }
		dc[0][0] += 2.0 * zp[0][0]; dc[0][3 * L] += 2.0 * zp[L][0];
		dc[1][1] += zp[L][2]; dc[1][3 * L + 2] += zp[0][1]; dc[1][0] += sin(zp[0][0]);
	}
}
/* nnlfc = 1 */
static void tf_nlfcf(int *mode, int *nstate, double *c, double **dc, double **zp)
{
	int L = fam_nout - 1, v; (void)nstate;
	if (*mode == 0 || *mode == 2) c[0] = zp[0][2] * zp[0][0] + zp[L][1] * zp[L][1];
	if (*mode == 1 || *mode == 2) { for (v = 0; v < 3 * fam_nout; v++) dc[0][v] = 0; dc[0][2] += zp[0][0]; dc[0][0] += zp[0][2]; dc[0][3 * L + 1] += 2.0 * zp[L][1]; }
}

/* ---- family 3 ---- */
#define OBS_X 20.0
#define OBS_Y 0.5
static void obs_nltcf(int *mode, int *nstate, int *i, double *c, double **dc, double **zp)
{
	const double dx = zp[0][0] - OBS_X, dy = zp[1][0] - OBS_Y; int v; (void)nstate; (void)i;
	if (*mode == 0 || *mode == 2) c[0] = dx * dx + dy * dy;
	if (*mode == 1 || *mode == 2) { for (v = 0; v < 6; v++) dc[0][v] = 0; dc[0][0] = 2.0 * dx; dc[0][3] = 2.0 * dy; }
}

/* ---- family 4 ---- */
#define QUAD_G 9.81
static void quad_ucf(int *mode, int *nstate, int *i, double *f, double *df, double **zp)
{
	int v; (void)nstate; (void)i;
	if (*mode == 0 || *mode == 2) *f = zp[0][4] * zp[0][4] + zp[1][4] * zp[1][4] + zp[2][4] * zp[2][4] + zp[3][2] * zp[3][2];
	if (*mode == 1 || *mode == 2) {
		for (v = 0; v < 20; v++) df[v] = 0;
		df[4] = 2 * zp[0][4]; df[9] = 2 * zp[1][4]; df[14] = 2 * zp[2][4]; df[17] = 2 * zp[3][2];
	}
}
static void quad_nltcf(int *mode, int *nstate, int *i, double *c, double **dc, double **zp)
{
	const double ax = zp[0][2], ay = zp[1][2], az = zp[2][2] + QUAD_G; int v; (void)nstate; (void)i;
	if (*mode == 0 || *mode == 2) {
		c[0] = ax * ax + ay * ay + az * az;
		c[1] = zp[0][1] * zp[0][1] + zp[1][1] * zp[1][1] + zp[2][1] * zp[2][1];
	}
	if (*mode == 1 || *mode == 2) {
		for (v = 0; v < 20; v++) { dc[0][v] = 0; dc[1][v] = 0; }
		dc[0][2] = 2 * ax; dc[0][7] = 2 * ay; dc[0][12] = 2 * az;
		dc[1][1] = 2 * zp[0][1]; dc[1][6] = 2 * zp[1][1]; dc[1][11] = 2 * zp[2][1];
	}
}
/* ---- family 5 ---- */
static void manip_nltcf(int *mode, int *nstate, int *i, double *c, double **dc, double **zp)
{
	int j, v, narms = fam_nout / 3; (void)nstate; (void)i;
	for (j = 0; j < narms; j++) {
		const double qa = zp[3 * j][0], qb = zp[3 * j + 1][0], qc = zp[3 * j + 2][0];
		const double a1 = qa, a2 = qa + qb, a3 = qa + qb + qc;
		if (*mode == 0 || *mode == 2) c[j] = sin(a1) + sin(a2) + sin(a3);
		if (*mode == 1 || *mode == 2) {
			for (v = 0; v < 3 * fam_nout; v++) dc[j][v] = 0;
			dc[j][9 * j] = cos(a1) + cos(a2) + cos(a3);
			dc[j][9 * j + 3] = cos(a2) + cos(a3);
			dc[j][9 * j + 6] = cos(a3);
		}
	}
}

/* ---- second derivatives of the constraint callbacks above: Hz (nz x nz row-major) += sum_j t_j d2c_j ---- */
#define HZ(a, b) Hz[(a) * nz + (b)]
static void tf_nlic_hess(int *i, const double *t, double *Hz, double **zp)
{ int nz = 3 * fam_nout; (void)i; (void)zp; HZ(0, 0) += 2.0 * t[0]; }
static void tf_nltc_hess(int *i, const double *t, double *Hz, double **zp)
{
	int L = fam_nout - 1, nz = 3 * fam_nout; (void)i;
	HZ(0, 0) += 2.0 * t[0]; HZ(3 * L, 3 * L) += 2.0 * t[0];
	HZ(1, 3 * L + 2) += t[1]; HZ(3 * L + 2, 1) += t[1]; HZ(0, 0) += t[1] * cos(zp[0][0]);
}
static void tf_nlfc_hess(int *i, const double *t, double *Hz, double **zp)
{
	int L = fam_nout - 1, nz = 3 * fam_nout; (void)i; (void)zp;
	HZ(0, 2) += t[0]; HZ(2, 0) += t[0]; HZ(3 * L + 1, 3 * L + 1) += 2.0 * t[0];
}
static void obs_nltc_hess(int *i, const double *t, double *Hz, double **zp)
{ int nz = 6; (void)i; (void)zp; HZ(0, 0) += 2.0 * t[0]; HZ(3, 3) += 2.0 * t[0]; }
static void quad_nltc_hess(int *i, const double *t, double *Hz, double **zp)
{
	int nz = 20; (void)i; (void)zp;
	HZ(2, 2) += 2.0 * t[0]; HZ(7, 7) += 2.0 * t[0]; HZ(12, 12) += 2.0 * t[0];
	HZ(1, 1) += 2.0 * t[1]; HZ(6, 6) += 2.0 * t[1]; HZ(11, 11) += 2.0 * t[1];
}
static void manip_nltc_hess(int *i, const double *t, double *Hz, double **zp)
{
	int j, narms = fam_nout / 3, nz = 3 * fam_nout; (void)i;
	for (j = 0; j < narms; j++) {
		const double qa = zp[3 * j][0], qb = zp[3 * j + 1][0], qc = zp[3 * j + 2][0];
		const double s1 = sin(qa), s2 = sin(qa + qb), s3 = sin(qa + qb + qc);
		const int a = 9 * j, b = 9 * j + 3, c = 9 * j + 6;
		/* c_j = sin a1 + sin a2 + sin a3, a1 = qa, a2 = qa+qb, a3 = qa+qb+qc:  d2c = -(s1 e1e1' + s2 e2e2' + s3 e3e3') */
		HZ(a, a) -= t[j] * (s1 + s2 + s3); HZ(a, b) -= t[j] * (s2 + s3); HZ(a, c) -= t[j] * s3;
		HZ(b, a) -= t[j] * (s2 + s3); HZ(b, b) -= t[j] * (s2 + s3); HZ(b, c) -= t[j] * s3;
		HZ(c, a) -= t[j] * s3; HZ(c, b) -= t[j] * s3; HZ(c, c) -= t[j] * s3;
	}
}
#undef HZ
orc_nlhess_t orc_family_nlic_hess(int fam) { return fam == 2 ? tf_nlic_hess : NULL; }
orc_nlhess_t orc_family_nltc_hess(int fam) { return fam == 2 ? tf_nltc_hess : fam == 3 ? obs_nltc_hess : fam == 4 ? quad_nltc_hess : fam == 5 ? manip_nltc_hess : NULL; }
orc_nlhess_t orc_family_nlfc_hess(int fam) { return fam == 2 ? tf_nlfc_hess : NULL; }

orc_ucf_t orc_family_ucf(int fam) { return (fam == 0 || fam == 3 || fam == 5) ? kincar_ucf : fam == 1 ? vdp_ucf : fam == 2 ? tf_ucf : fam == 4 ? quad_ucf : NULL; }
orc_icf_t orc_family_icf(int fam) { return fam == 2 ? tf_icf : NULL; }
orc_icf_t orc_family_fcf(int fam) { return fam == 2 ? tf_fcf : NULL; }
orc_nlic_t orc_family_nlicf(int fam) { return fam == 2 ? tf_nlicf : NULL; }
orc_nltc_t orc_family_nltcf(int fam) { return fam == 2 ? tf_nltcf : fam == 3 ? obs_nltcf : fam == 4 ? quad_nltcf : fam == 5 ? manip_nltcf : NULL; }
orc_nlic_t orc_family_nlfcf(int fam) { return fam == 2 ? tf_nlfcf : NULL; }

/* test hook: the constraint Hessian callbacks against central differences of the Jacobian callbacks at one flag z
 * (zflat: nout x maxderiv, row-major), multipliers t; returns max |Hz - FD| / max(1, |Hz|) over the three kinds */
double orc_family_check_hess(int fam, int nout, int maxderiv, const double *zflat, const double *t)
{
	const int nz = nout * maxderiv;
	double *z = malloc(nz * sizeof(double)), *Hz = calloc((size_t)nz * nz, sizeof(double)), *dcp = malloc(16 * nz * sizeof(double)), *dcm = malloc(16 * nz * sizeof(double));
	double **zp = malloc(nout * sizeof(double *)), *rowp[16], *rowm[16], cdum[16], worst = 0.0, scale = 1.0;
	int kind, o, v, w, j, i0 = 0, md = 1, ns = 0;
	orc_family_set_nout(nout);
	for (o = 0; o < nout; o++) zp[o] = z + o * maxderiv;
	for (j = 0; j < 16; j++) { rowp[j] = dcp + (size_t)j * nz; rowm[j] = dcm + (size_t)j * nz; }
	for (kind = 0; kind < 3; kind++) {
		orc_nlhess_t hf = kind == 0 ? orc_family_nlic_hess(fam) : kind == 1 ? orc_family_nltc_hess(fam) : orc_family_nlfc_hess(fam);
		orc_nlic_t cf = kind == 0 ? orc_family_nlicf(fam) : kind == 2 ? orc_family_nlfcf(fam) : NULL;
		orc_nltc_t tf = kind == 1 ? orc_family_nltcf(fam) : NULL;
		const int ncon = kind == 1 ? (fam == 2 ? 2 : fam == 3 ? 1 : fam == 4 ? 2 : fam == 5 ? nout / 3 : 0) : (fam == 2 ? 1 : 0);
		if (!hf || ncon == 0) continue;
		memcpy(z, zflat, nz * sizeof(double));
		memset(Hz, 0, (size_t)nz * nz * sizeof(double));
		hf(&i0, t, Hz, zp);
		for (v = 0; v < nz * nz; v++) if (fabs(Hz[v]) > scale) scale = fabs(Hz[v]);
		for (w = 0; w < nz; w++) {
			const double h = 1e-5 * (1.0 + fabs(zflat[w]));
			memcpy(z, zflat, nz * sizeof(double)); z[w] = zflat[w] + h;
			md = 1; if (tf) tf(&md, &ns, &i0, cdum, rowp, zp); else cf(&md, &ns, cdum, rowp, zp);
			z[w] = zflat[w] - h;
			md = 1; if (tf) tf(&md, &ns, &i0, cdum, rowm, zp); else cf(&md, &ns, cdum, rowm, zp);
			for (v = 0; v < nz; v++) {
				double fd = 0.0;
				for (j = 0; j < ncon; j++) fd += t[j] * (rowp[j][v] - rowm[j][v]) / (2.0 * h);
				if (fabs(fd - Hz[v * nz + w]) > worst) worst = fabs(fd - Hz[v * nz + w]);
			}
		}
	}
	free(z); free(Hz); free(dcp); free(dcm); free(zp);
	return worst / scale;
}

/* ---------------- batched CPU driver ---------------- */
static double **rows_view(const double *flat, int nrows, int ncols)
{
	double **r; int i;
	if (nrows == 0) return NULL;
	r = malloc(nrows * sizeof(double *));
	for (i = 0; i < nrows; i++) r[i] = (double *)flat + (size_t)i * ncols;
	return r;
}
static orc_problem *make_from_spec(const orc_batch_spec *s, const double *lowerb, const double *upperb)
{
	int nz = 0, o; double **lic, **ltc, **lfc; orc_problem *p;
	for (o = 0; o < s->nout; o++) nz += s->maxderiv[o];
	lic = rows_view(s->lic, s->nlic, nz); ltc = rows_view(s->ltc, s->nltc, nz); lfc = rows_view(s->lfc, s->nlfc, nz);
	p = orc_problem_make(s->nout, (double *)s->bps, s->nbps, (int *)s->kninterv, (double **)s->knots,
		(int *)s->order, (int *)s->mult, (int *)s->maxderiv,
		s->nlic, lic, s->nltc, ltc, s->nlfc, lfc,
		s->nnlic, orc_family_nlicf(s->family), s->nnltc, orc_family_nltcf(s->family), s->nnlfc, orc_family_nlfcf(s->family),
		s->nicav, (orc_AV *)s->icav, s->ntcav, (orc_AV *)s->tcav, s->nfcav, (orc_AV *)s->fcav,
		(double *)lowerb, (double *)upperb,
		s->nicf, orc_family_icf(s->family), s->nucf, orc_family_ucf(s->family), s->nfcf, orc_family_fcf(s->family),
		s->nicostav, (orc_AV *)s->icostav, s->ntcostav, (orc_AV *)s->tcostav, s->nfcostav, (orc_AV *)s->fcostav);
	free(lic); free(ltc); free(lfc);
	p->nlic_hess = orc_family_nlic_hess(s->family); p->nltc_hess = orc_family_nltc_hess(s->family); p->nlfc_hess = orc_family_nlfc_hess(s->family);
	/* the same table as build_newton_tables() of the product (ntg_amd/csrc/plan.cpp) */
	if (s->family == ORC_FAM_OBSTACLE) { p->couple = 2; p->group_mask = (1ull << 0) | (1ull << 3); }
	else if (s->family == ORC_FAM_QUADROTOR) { p->couple = 4; p->group_mask = (1ull << 1) | (1ull << 2) | (1ull << 6) | (1ull << 7) | (1ull << 11) | (1ull << 12); }
	else if (s->family == ORC_FAM_MANIP) { p->couple = 3; p->group_mask = (1ull << 0) | (1ull << 3) | (1ull << 6); }
	return p;
}
static int spec_nb(const orc_batch_spec *s) { return s->nlic + s->nltc + s->nlfc + s->nnlic + s->nnltc + s->nnlfc; }

/* where the OpenMP threads of the batch drivers run: cpus[t] = the CPU thread t is on (sched_getcpu), for the bench line's affinity note */
int orc_thread_cpus(int nthreads, int *cpus)
{
	int t;
	if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static, 1) num_threads(nthreads)
	for (t = 0; t < nthreads; t++) cpus[t] = sched_getcpu();
	return 0;
}

int orc_solve_batch(const orc_batch_spec *s, int batch, const double *lowerb, const double *upperb,
                    double *x, const orc_sqp_opts *o, double *objective, int *inform, int *iters,
                    int *nfev, int nthreads)
{
	int b, nb = spec_nb(s);
	if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
	for (b = 0; b < batch; b++) {
		orc_problem *p; orc_sqp_result res;
		orc_family_set_nout(s->nout);
		p = make_from_spec(s, lowerb + (size_t)b * nb, upperb + (size_t)b * nb);
		orc_sqp_solve(p, x + (size_t)b * p->cc->nC, o, &res, NULL, NULL, NULL, NULL, 0);
		if (objective) objective[b] = res.objective;
		if (inform) inform[b] = res.inform;
		if (iters) iters[b] = res.iters;
		if (nfev) nfev[b] = res.nfev;
		orc_problem_free(p);
	}
	return 0;
}

/* f[batch], g[batch][nC], c[batch][ncnln], cJac[batch][ncnln*nC] (column-major per problem) */
int orc_eval_batch(const orc_batch_spec *s, int batch, const double *x, int mode,
                   double *f, double *g, double *c, double *cJac, int nthreads)
{
	int b, nb = spec_nb(s);
	double *zero = calloc(nb + 1, sizeof(double));
	if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
	for (b = 0; b < batch; b++) {
		orc_problem *p; int md = mode, ns = 1, n, nc;
		orc_family_set_nout(s->nout);
		p = make_from_spec(s, zero, zero);
		n = p->cc->nC; nc = p->ncnln;
		if (f || g) orc_funobj(p, &md, x + (size_t)b * n, f ? f + b : NULL, g ? g + (size_t)b * n : NULL, &ns);
		md = mode;
		if (nc > 0 && (c || cJac)) {
			orc_funcon(p, &md, x + (size_t)b * n, c ? c + (size_t)b * nc : NULL, NULL, &ns);
			if (cJac) memcpy(cJac + (size_t)b * nc * n, p->cJac, (size_t)nc * n * sizeof(double));
		}
		orc_problem_free(p);
	}
	free(zero);
	return 0;
}

/* export of the setup-time tables in the flat layout the tests compare against:
 * blk: outputs concatenated, per output [bp][q][r] (P*k_o*d_o doubles); off: [nout][P];
 * A column-major nclin x nC; bl/bu: nC+nclin+ncnln.  Any pointer may be NULL. */
int orc_spec_export(const orc_batch_spec *s, const double *lowerb, const double *upperb,
                    double *blk, int *off, double *A, double *bl, double *bu)
{
	orc_problem *p; int o, nb = spec_nb(s), ntot; size_t pos = 0;
	double *zero = calloc(nb + 1, sizeof(double));
	orc_family_set_nout(s->nout);
	p = make_from_spec(s, lowerb ? lowerb : zero, upperb ? upperb : zero);
	for (o = 0; o < s->nout; o++) {
		size_t cnt = (size_t)s->nbps * p->cc->order[o] * p->cc->maxderiv[o];
		if (blk) memcpy(blk + pos, p->cc->blk[o], cnt * sizeof(double));
		if (off) memcpy(off + (size_t)o * s->nbps, p->cc->off[o], s->nbps * sizeof(int));
		pos += cnt;
	}
	ntot = p->cc->nC + p->nclin + p->ncnln;
	if (A && p->nclin) memcpy(A, p->A, (size_t)p->nclin * p->cc->nC * sizeof(double));
	if (bl) memcpy(bl, p->bl, ntot * sizeof(double));
	if (bu) memcpy(bu, p->bu, ntot * sizeof(double));
	orc_problem_free(p); free(zero);
	return 0;
}

/* single-problem solve with trace (tests) */
int orc_solve_one(const orc_batch_spec *s, const double *lowerb, const double *upperb, double *x,
                  const orc_sqp_opts *o, orc_sqp_result *res, double *clambda, int *istate,
                  double *R, double *trace, int trace_cap)
{
	orc_problem *p;
	orc_family_set_nout(s->nout);
	p = make_from_spec(s, lowerb, upperb);
	orc_sqp_solve(p, x, o, res, clambda, istate, R, trace, trace_cap);
	orc_problem_free(p);
	return 0;
}
