// ntg_host.cpp -- the single-problem drop-in: ntg(), npsoloption(), linspace(), SplineInterp(),
// matrix helpers and the NPSOL-facing callbacks, with the reference's signatures (include/ntg.h).
//
// The user's cost/constraint functions are HOST function pointers (ntg.h:81-83,90-92), so this
// path alternates: HIP kernel updateZ -> host callbacks per breakpoint -> HIP kernel banded
// assembly + quadrature.  The SQP iteration around it (what npsol_ does at ntg.c:250) runs on
// the host with the same algorithm as the device sqp_kernel (linesearch.hpp is shared).
// Everything numerical between "x" and "f, g, c, cJac" is computed by kernels.hip; with no
// GPU, ntg() reports inform = 9 and prints why -- there is no CPU fallback.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cctype>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/ntg.h"
#include "ntg_dev.hpp"
#include "plan.hpp"
#include "linesearch.hpp"

typedef void (*icf_t)(int *, int *, double *, double *, double **);
typedef void (*ucf_t)(int *, int *, int *, double *, double *, double **);
typedef void (*nlic_t)(int *, int *, double *, double **, double **);
typedef void (*nltc_t)(int *, int *, int *, double *, double **, double **);

namespace {

struct HostProblem {
	ntg_plan *plan = nullptr;
	icf_t icf = nullptr, fcf = nullptr; ucf_t ucf = nullptr;
	nlic_t nlicf = nullptr, nlfcf = nullptr; nltc_t nltcf = nullptr;
	SmemLayout L;
	std::vector<double> Z;                // host mirror of GZ (ntg.c:119)
	double *d_x = nullptr, *d_Z = nullptr, *d_fT = nullptr, *d_dfT = nullptr, *d_fdI = nullptr, *d_fdF = nullptr,
	       *d_F = nullptr, *d_g = nullptr, *d_dc = nullptr, *d_cjac = nullptr;
	std::vector<double *> zp;
	int nfev = 0;
};
HostProblem *g_cur = nullptr;             // "current problem" like the reference's file-scope globals (ntg.c:17-41)

struct Options {
	ntg_solve_opts o; int print_level = 10; bool init = false;
} g_opt;
void ensure_opts() { if (!g_opt.init) { ntg_default_opts(&g_opt.o); g_opt.init = true; } }

bool hip_ok(hipError_t e, const char *what)
{
	if (e == hipSuccess) return true;
	fprintf(stderr, "ntg (MI355X): %s: %s\n", what, hipGetErrorString(e));
	return false;
}

// zp views into Z (colloc.c:425-447)
void zp_at(const NtgDims &D, std::vector<double> &Z, int bp, std::vector<double *> &zp)
{
	for (int o = 0; o < D.nout; o++) zp[o] = &Z[(size_t)D.iz[o] * D.P + (size_t)D.d[o] * bp];
}

// NPfunobj: ntg.c:274-335
bool host_funobj(HostProblem &hp, int mode, const double *x, double *f, double *g, int *nstate)
{
	const NtgDims &D = hp.plan->D;
	const int P = D.P, nz = D.nz, n = D.nC;
	if (!hip_ok(hipMemcpy(hp.d_x, x, (size_t)n * 8, hipMemcpyHostToDevice), "H2D x")) return false;
	if (!hip_ok(ntg_launch_hostz(D, hp.plan->T, hp.L, hp.d_x, D.nicf ? D.icost_mask : 0, D.nucf ? D.tcost_mask : 0,
	                             D.nfcf ? D.fcost_mask : 0, hp.d_Z, nullptr), "updateZ kernel")) return false;
	if (!hip_ok(hipMemcpy(hp.Z.data(), hp.d_Z, hp.Z.size() * 8, hipMemcpyDeviceToHost), "D2H Z")) return false;
	std::vector<double> fT(P, 0.0), dfT((size_t)P * nz, 0.0), fdI(nz + 1, 0.0), fdF(nz + 1, 0.0);
	int md = mode;
	// mode-0 quirk of the reference (costs evaluated only when n?cf == 1, ntg.c:297-301) kept
	const bool doI = mode == 0 ? D.nicf == 1 : D.nicf != 0, doT = mode == 0 ? D.nucf == 1 : D.nucf != 0,
	           doF = mode == 0 ? D.nfcf == 1 : D.nfcf != 0;
	if (doT)
		for (int i = 0; i < P; i++) {                                // cost.c:103-109
			zp_at(D, hp.Z, i, hp.zp);
			int ii = i;
			hp.ucf(&md, nstate, &ii, &fT[i], &dfT[(size_t)i * nz], hp.zp.data());
		}
	if (doI) { zp_at(D, hp.Z, 0, hp.zp); hp.icf(&md, nstate, &fdI[nz], fdI.data(), hp.zp.data()); }
	if (doF) { zp_at(D, hp.Z, P - 1, hp.zp); hp.fcf(&md, nstate, &fdF[nz], fdF.data(), hp.zp.data()); }
	if (!hip_ok(hipMemcpy(hp.d_fT, fT.data(), (size_t)P * 8, hipMemcpyHostToDevice), "H2D f") ||
	    !hip_ok(hipMemcpy(hp.d_dfT, dfT.data(), dfT.size() * 8, hipMemcpyHostToDevice), "H2D df") ||
	    !hip_ok(hipMemcpy(hp.d_fdI, fdI.data(), fdI.size() * 8, hipMemcpyHostToDevice), "H2D dfI") ||
	    !hip_ok(hipMemcpy(hp.d_fdF, fdF.data(), fdF.size() * 8, hipMemcpyHostToDevice), "H2D dfF")) return false;
	if (!hip_ok(ntg_launch_hostcost(D, hp.plan->T, hp.L, hp.d_fT, hp.d_dfT, hp.d_fdI, hp.d_fdF, hp.d_F, hp.d_g, nullptr),
	            "assembly kernel")) return false;
	double F;
	if (!hip_ok(hipMemcpy(&F, hp.d_F, 8, hipMemcpyDeviceToHost), "D2H F")) return false;
	if (f && mode != 1) *f = F;
	if (g && mode != 0 && !hip_ok(hipMemcpy(g, hp.d_g, (size_t)n * 8, hipMemcpyDeviceToHost), "D2H g")) return false;
	hp.nfev++;
	return true;
}

// NPfuncon: ntg.c:337-371, constraints.c:36-195.  cJac column-major ldJ x n.
bool host_funcon(HostProblem &hp, int mode, const double *x, double *c, double *cJac, int ldJ, int *nstate)
{
	const NtgDims &D = hp.plan->D;
	const int P = D.P, nz = D.nz, n = D.nC, nc = D.ncnln;
	if (nc == 0) return true;
	if (!hip_ok(hipMemcpy(hp.d_x, x, (size_t)n * 8, hipMemcpyHostToDevice), "H2D x")) return false;
	if (!hip_ok(ntg_launch_hostz(D, hp.plan->T, hp.L, hp.d_x, D.nnlic ? D.icon_mask : 0, D.nnltc ? D.tcon_mask : 0,
	                             D.nnlfc ? D.fcon_mask : 0, hp.d_Z, nullptr), "updateZ kernel")) return false;
	if (!hip_ok(hipMemcpy(hp.Z.data(), hp.d_Z, hp.Z.size() * 8, hipMemcpyDeviceToHost), "D2H Z")) return false;
	std::vector<double> dc((size_t)nc * nz, 0.0), cv(nc, 0.0);
	int md = mode;
	auto call = [&](int ncon, int row0, int stride, auto fn) {      // dc[con][var] (constraints.c:104,146)
		std::vector<double> slab((size_t)std::max(ncon, 1) * nz, 0.0), tmp(std::max(ncon, 1), 0.0);
		std::vector<double *> rows(std::max(ncon, 1));
		for (int j = 0; j < ncon; j++) rows[j] = &slab[(size_t)j * nz];
		fn(tmp.data(), mode == 0 ? nullptr : rows.data());
		for (int j = 0; j < ncon; j++) {
			cv[row0 + j * stride] = tmp[j];
			if (mode != 0) std::copy(rows[j], rows[j] + nz, &dc[(size_t)(row0 + j * stride) * nz]);
		}
	};
	if (D.nnlic) { zp_at(D, hp.Z, 0, hp.zp); call(D.nnlic, 0, 1, [&](double *cc, double **dd) { hp.nlicf(&md, nstate, cc, dd, hp.zp.data()); }); }
	if (D.nnltc)
		for (int i = 0; i < P; i++) {
			zp_at(D, hp.Z, i, hp.zp);
			int ii = i;
			call(D.nnltc, D.nnlic + i, P, [&](double *cc, double **dd) { hp.nltcf(&md, nstate, &ii, cc, dd, hp.zp.data()); });
		}
	if (D.nnlfc) { zp_at(D, hp.Z, P - 1, hp.zp); call(D.nnlfc, D.nnlic + D.nnltc * P, 1, [&](double *cc, double **dd) { hp.nlfcf(&md, nstate, cc, dd, hp.zp.data()); }); }
	if (c && mode != 1) std::copy(cv.begin(), cv.end(), c);
	if (cJac && mode != 0) {
		if (!hip_ok(hipMemcpy(hp.d_dc, dc.data(), dc.size() * 8, hipMemcpyHostToDevice), "H2D dc")) return false;
		if (!hip_ok(hipMemset(hp.d_cjac, 0, (size_t)nc * n * 8), "zero cJac")) return false;   // band positions only are written
		if (!hip_ok(ntg_launch_hostcon(D, hp.plan->T, hp.d_dc, nullptr, hp.d_cjac, nullptr), "Jacobian kernel")) return false;
		std::vector<double> J((size_t)nc * n);
		if (!hip_ok(hipMemcpy(J.data(), hp.d_cjac, J.size() * 8, hipMemcpyDeviceToHost), "D2H cJac")) return false;
		for (int col = 0; col < n; col++) for (int r = 0; r < nc; r++) cJac[(size_t)col * ldJ + r] = J[(size_t)col * nc + r];
	}
	return true;
}

bool alloc_host_problem(HostProblem &hp)
{
	const NtgDims &D = hp.plan->D;
	hp.L = ntg_make_layout(D, 128, 1, 1);
	hp.Z.assign((size_t)D.nz * D.P, 0.0);
	hp.zp.resize(D.nout);
	auto al = [&](double **p, size_t n) { return hip_ok(hipMalloc((void **)p, std::max<size_t>(n, 1) * 8), "hipMalloc"); };
	const size_t nc = D.ncnln;
	if (!(al(&hp.d_x, D.nC) && al(&hp.d_Z, hp.Z.size()) && al(&hp.d_fT, D.P) && al(&hp.d_dfT, (size_t)D.P * D.nz) &&
	      al(&hp.d_fdI, D.nz + 1) && al(&hp.d_fdF, D.nz + 1) && al(&hp.d_F, 1) && al(&hp.d_g, D.nC) &&
	      al(&hp.d_dc, nc * D.nz) && al(&hp.d_cjac, nc * D.nC))) return false;
	return hip_ok(hipMemset(hp.d_Z, 0, hp.Z.size() * 8), "zero Z");
}
void free_host_problem(HostProblem &hp)
{
	for (double *p : {hp.d_x, hp.d_Z, hp.d_fT, hp.d_dfT, hp.d_fdI, hp.d_fdF, hp.d_F, hp.d_g, hp.d_dc, hp.d_cjac})
		if (p) (void)hipFree(p);
	if (hp.plan) ntg_plan_destroy(hp.plan);
	hp.plan = nullptr;
}

// ---- tiny dense helpers (row-major) ----
bool chol_lower(std::vector<double> &a, int n)
{
	for (int j = 0; j < n; j++) {
		double d = a[(size_t)j * n + j];
		for (int k = 0; k < j; k++) d -= a[(size_t)j * n + k] * a[(size_t)j * n + k];
		if (!(d > 0.0)) return false;
		d = std::sqrt(d); a[(size_t)j * n + j] = d;
		for (int i = j + 1; i < n; i++) {
			double s = a[(size_t)i * n + j];
			for (int k = 0; k < j; k++) s -= a[(size_t)i * n + k] * a[(size_t)j * n + k];
			a[(size_t)i * n + j] = s / d;
		}
	}
	return true;
}
void chol_solve(const std::vector<double> &L, int n, double *b)
{
	for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[(size_t)i * n + k] * b[k]; b[i] = s / L[(size_t)i * n + i]; }
	for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= L[(size_t)k * n + i] * b[k]; b[i] = s / L[(size_t)i * n + i]; }
}
double dot(const std::vector<double> &a, const std::vector<double> &b) { double s = 0; for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i]; return s; }
double nrm2(const std::vector<double> &a) { return std::sqrt(dot(a, a)); }

} // namespace

// ---------------- exported NPSOL-facing callbacks (ntg.c:274-280, 337-346) ----------------
extern "C" void npsolCostFunction(int *mode, int *n, double *x, double *f, double *g, int *nstate)
{
	if (!g_cur || *n != g_cur->plan->D.nC || *mode < 0 || *mode > 2) { if (nstate) *nstate = -1; return; }  // ntg.c:332-333
	if (!host_funobj(*g_cur, *mode, x, f, g, nstate)) *mode = -1;
}
extern "C" void npsolConstraintFunction(int *mode, int *ncnln, int *n, int *ldJ, int *needc, double *x, double *c,
                                        double *cJac, int *nstate)
{
	(void)needc;                                                   // ignored by the reference too (ntg.c:337-371)
	if (!g_cur || *n != g_cur->plan->D.nC || *ncnln != g_cur->plan->D.ncnln || *mode < 0 || *mode > 2) { *mode = -1; return; } // ntg.c:368-369
	if (!host_funcon(*g_cur, *mode, x, c, cJac, *ldJ, nstate)) *mode = -1;
}

// ---------------- option strings (ntg.c:269-272 -> NPSOL npoptn_) ----------------
extern "C" void npsoloption(char *option)
{
	ensure_opts();
	std::string s(option);
	for (auto &ch : s) ch = (char)std::tolower((unsigned char)ch);
	double v = 0.0;
	size_t eq = s.find('=');
	if (eq != std::string::npos) v = atof(s.c_str() + eq + 1);
	else { size_t q = s.size(); while (q > 0 && (std::isdigit((unsigned char)s[q - 1]) || strchr(".e+-", s[q - 1]))) q--; v = atof(s.c_str() + q); }
	auto starts = [&](const char *k) { return s.compare(0, strlen(k), k) == 0; };
	if (starts("nolist") || starts("derivative level") || starts("summary file")) return;
	if (starts("print level")) { g_opt.print_level = (int)v; return; }
	if (starts("major iteration limit")) { g_opt.o.itlim = (int)v; return; }
	if (starts("optimality tolerance")) { g_opt.o.opttol = v; return; }
	if (starts("line search tolerance")) { g_opt.o.ls_eta = v; return; }
	if (starts("step limit")) { g_opt.o.steplimit = v; return; }
	if (starts("hessian")) { g_opt.o.hessian = s.find("colloc") != std::string::npos ? 1 : 0; return; }
	fprintf(stderr, "ntg (MI355X): npsoloption '%s' ignored\n", option);
}

extern "C" void linspace(double *v, double d0, double d1, int n)
{
	if (d0 == d1) { for (int i = 0; i < n; i++) v[i] = d0; return; }   // ntg.c:374-389
	const double h = (d1 - d0) / (n - 1);
	v[0] = d0;
	for (int i = 1; i < n; i++) v[i] = v[i - 1] + h;
}

extern "C" void printNTGBanner(void)
{
	printf("\n\n                     NTG-compatible trajectory generation, MI355X-native engine\n");
	printf("                     (C ABI of NTG 2.2; HIP kernels for gfx950)\n\n");
	printf("          *******************************************************\n\n\n");
}

// ---------------- the drop-in ----------------
#define NTG_PROBLEM_ARGS                                                                                     \
	int nout, double *bps, int nbps, int *kninterv, double **knots, int *order, int *mult, int *maxderiv,        \
	int nlic, double **lic, int nltc, double **ltc, int nlfc, double **lfc,                                      \
	int nnlic, nlic_t nlicf, int nnltc, nltc_t nltcf, int nnlfc, nlic_t nlfcf,                                   \
	int nicav, AV *icav, int ntcav, AV *tcav, int nfcav, AV *fcav,                                               \
	int nicf, icf_t icf, int nucf, ucf_t ucf, int nfcf, icf_t fcf,                                               \
	int nicostav, AV *icostav, int ntcostav, AV *tcostav, int nfcostav, AV *fcostav
#define NTG_PROBLEM_PASS                                                                                     \
	nout, bps, nbps, kninterv, knots, order, mult, maxderiv, nlic, lic, nltc, ltc, nlfc, lfc, nnlic, nlicf,      \
	nnltc, nltcf, nnlfc, nlfcf, nicav, icav, ntcav, tcav, nfcav, fcav, nicf, icf, nucf, ucf, nfcf, fcf,          \
	nicostav, icostav, ntcostav, tcostav, nfcostav, fcostav

// what ntg() does before calling npsol_ (ntg.c:114-229): collocation, A, globals
static HostProblem *open_host_problem(NTG_PROBLEM_ARGS)
{
	int nz = 0;
	for (int o = 0; o < nout; o++) nz += maxderiv[o];
	auto flat = [&](double **m, int rows) { std::vector<double> v((size_t)rows * nz); for (int i = 0; i < rows; i++) std::copy(m[i], m[i] + nz, &v[(size_t)i * nz]); return v; };
	std::vector<double> flic = flat(lic, nlic), fltc = flat(ltc, nltc), flfc = flat(lfc, nlfc);
	ntg_spec s;
	std::memset(&s, 0, sizeof(s));
	s.nout = nout; s.nbps = nbps; s.bps = bps; s.kninterv = kninterv; s.knots = (const double *const *)knots;
	s.order = order; s.mult = mult; s.maxderiv = maxderiv; s.family = NTG_FAM_HOST;
	s.nlic = nlic; s.nltc = nltc; s.nlfc = nlfc; s.lic = flic.data(); s.ltc = fltc.data(); s.lfc = flfc.data();
	s.nnlic = nnlic; s.nnltc = nnltc; s.nnlfc = nnlfc;
	s.nicav = nicav; s.ntcav = ntcav; s.nfcav = nfcav;
	s.icav = (const ntg_av *)icav; s.tcav = (const ntg_av *)tcav; s.fcav = (const ntg_av *)fcav;
	s.nicf = nicf; s.nucf = nucf; s.nfcf = nfcf;
	s.nicostav = nicostav; s.ntcostav = ntcostav; s.nfcostav = nfcostav;
	s.icostav = (const ntg_av *)icostav; s.tcostav = (const ntg_av *)tcostav; s.fcostav = (const ntg_av *)fcostav;

	HostProblem *hp = new HostProblem();
	int dev = 0;
	(void)hipGetDevice(&dev);
	if (ntg_plan_create(&s, dev, &hp->plan) != 0) {
		fprintf(stderr, "ntg (MI355X): cannot set the problem up: %s\n", ntg_last_error());
		delete hp;
		return nullptr;
	}
	hp->icf = icf; hp->ucf = ucf; hp->fcf = fcf; hp->nlicf = nlicf; hp->nltcf = nltcf; hp->nlfcf = nlfcf;
	if (!alloc_host_problem(*hp)) { free_host_problem(*hp); delete hp; return nullptr; }
	return hp;
}

// ntg_open()/ntg_close(): set up a problem exactly like ntg() does and leave it current, so that an
// external SQP/IPOPT driver can call npsolCostFunction / npsolConstraintFunction (include/ntg_amd.h)
static HostProblem *g_opened = nullptr;
extern "C" int ntg_open(
	int nout, double *bps, int nbps, int *kninterv, double **knots, int *order, int *mult, int *maxderiv,
	int nlic, double **lic, int nltc, double **ltc, int nlfc, double **lfc,
	int nnlic, nlic_t nlicf, int nnltc, nltc_t nltcf, int nnlfc, nlic_t nlfcf,
	int nicav, ntg_av *icav_, int ntcav, ntg_av *tcav_, int nfcav, ntg_av *fcav_,
	int nicf, icf_t icf, int nucf, ucf_t ucf, int nfcf, icf_t fcf,
	int nicostav, ntg_av *icostav_, int ntcostav, ntg_av *tcostav_, int nfcostav, ntg_av *fcostav_)
{
	if (g_opened) return NTG_E_BADARG;
	AV *icav = (AV *)icav_, *tcav = (AV *)tcav_, *fcav = (AV *)fcav_;          // same layout (av.h:22-26)
	AV *icostav = (AV *)icostav_, *tcostav = (AV *)tcostav_, *fcostav = (AV *)fcostav_;
	HostProblem *hp = open_host_problem(NTG_PROBLEM_PASS);
	if (!hp) return NTG_E_HIP;
	g_opened = hp; g_cur = hp;
	return 0;
}
extern "C" void ntg_close(void)
{
	if (!g_opened) return;
	if (g_cur == g_opened) g_cur = nullptr;
	free_host_problem(*g_opened);
	delete g_opened;
	g_opened = nullptr;
}

extern "C" void ntg(int nout, double *bps, int nbps, int *kninterv, double **knots, int *order, int *mult,
                    int *maxderiv, double *initialguess,
                    int nlic, double **lic, int nltc, double **ltc, int nlfc, double **lfc,
                    int nnlic, nlic_t nlicf, int nnltc, nltc_t nltcf, int nnlfc, nlic_t nlfcf,
                    int nicav, AV *icav, int ntcav, AV *tcav, int nfcav, AV *fcav,
                    double *lowerb, double *upperb,
                    int nicf, icf_t icf, int nucf, ucf_t ucf, int nfcf, icf_t fcf,
                    int nicostav, AV *icostav, int ntcostav, AV *tcostav, int nfcostav, AV *fcostav,
                    int *istate, double *clambda, double *R, int *inform, double *objective)
{
	ensure_opts();
	printNTGBanner();                                              // ntg.c:161
	*inform = 9; *objective = 0.0;
	HostProblem *hpp = open_host_problem(NTG_PROBLEM_PASS);
	if (!hpp) return;
	HostProblem &hp = *hpp;
	HostProblem *prev = g_cur;
	g_cur = &hp;

	const NtgDims &D = hp.plan->D;
	const int n = D.nC, m = D.nclin, ntot = n + m + D.ncnln;
	std::vector<double> x(initialguess, initialguess + n);
	int info = 4, iter = 0, nstate = 1;
	double F = 0.0;
	std::vector<double> W((size_t)n * n, 0.0);
	// expanded bounds (constraints.c:5-33) for the linear rows
	std::vector<double> bl(m), bu(m);
	for (int r = 0; r < m; r++) {
		int si;
		if (r < nlic) si = r; else if (r < nlic + nltc * nbps) si = nlic + (r - nlic) / nbps; else si = nlic + nltc + (r - nlic - nltc * nbps);
		bl[r] = lowerb[si]; bu[r] = upperb[si];
	}
	const int nc = D.ncnln;
	// linear rows with lower == upper are kept satisfied by projection (E); the others (I) join the nonlinear rows
	// in the augmented Lagrangian as constraints with a constant Jacobian row
	std::vector<int> erow, irow;
	for (int r = 0; r < m; r++) (bl[r] == bu[r] ? erow : irow).push_back(r);
	const int mE = (int)erow.size(), nI = (int)irow.size(), nal = nc + nI;
	std::vector<double> allam(nal, 0.0);         // augmented-Lagrangian multipliers: nonlinear rows, then rows I
	std::vector<double> cval(nal, 0.0);
	std::vector<double> lamE(mE, 0.0);
	{
		const std::vector<double> &A = hp.plan->h_Adense;            // row-major m x n, all linear rows
		std::vector<double> AE((size_t)mE * n), bE(mE);
		for (int i = 0; i < mE; i++) { bE[i] = bl[erow[i]]; std::copy(&A[(size_t)erow[i] * n], &A[(size_t)erow[i] * n] + n, &AE[(size_t)i * n]); }
		std::vector<double> S((size_t)mE * mE, 0.0);
		for (int i = 0; i < mE; i++) for (int j = 0; j < mE; j++) { double a = 0; for (int c = 0; c < n; c++) a += AE[(size_t)i * n + c] * AE[(size_t)j * n + c]; S[(size_t)i * mE + j] = a; }
		bool lin_ok = mE == 0 || chol_lower(S, mE);
		if (!lin_ok) fprintf(stderr, "ntg (MI355X): the linear equality rows are rank deficient (inform 9)\n");
		auto project = [&](const std::vector<double> &g, std::vector<double> &gp) {
			gp = g;
			if (!mE) return;
			for (int i = 0; i < mE; i++) { double a = 0; for (int c = 0; c < n; c++) a += AE[(size_t)i * n + c] * g[c]; lamE[i] = a; }
			chol_solve(S, mE, lamE.data());
			for (int c = 0; c < n; c++) { double a = 0; for (int i = 0; i < mE; i++) a += AE[(size_t)i * n + c] * lamE[i]; gp[c] -= a; }
		};
		auto make_feasible = [&]() { // linear feasibility phase
			if (!mE) return;
			std::vector<double> r(mE);
			for (int i = 0; i < mE; i++) { double a = 0; for (int c = 0; c < n; c++) a += AE[(size_t)i * n + c] * x[c]; r[i] = bE[i] - a; }
			chol_solve(S, mE, r.data());
			for (int c = 0; c < n; c++) { double a = 0; for (int i = 0; i < mE; i++) a += AE[(size_t)i * n + c] * r[i]; x[c] += a; }
		};
		if (lin_ok) make_feasible();
		const ntg_solve_opts &o = g_opt.o;
		const int itlim = o.itlim > 0 ? o.itlim : std::max(50, 3 * (n + m) + 10 * nc);
		const double sr = std::sqrt(o.opttol > 0 ? o.opttol : std::pow(DBL_EPSILON, 0.8));
		// expanded bounds of the nonlinear rows (constraints.c:24-30)
		std::vector<double> nbl(nc), nbu(nc);
		{
			const int b0 = nlic + nltc + nlfc;
			for (int r = 0; r < nc; r++) {
				int si;
				if (r < nnlic) si = b0 + r; else if (r < nnlic + nnltc * nbps) si = b0 + nnlic + (r - nnlic) / nbps; else si = b0 + nnlic + nnltc + (r - nnlic - nnltc * nbps);
				nbl[r] = lowerb[si]; nbu[r] = upperb[si];
			}
		}
		std::vector<double> tnew(nal, 0.0), J((size_t)std::max(nc, 1) * n, 0.0);
		for (int j = 0; j < nI; j++) { nbl.push_back(bl[irow[j]]); nbu.push_back(bu[irow[j]]); }
		double mu = 10.0, Fp = 0.0;
		bool okc = true;
		// F_A = F + sum (t^2 - lam^2)/(2 mu), grad = g + J't  (DESIGN.md section 4b); rv = relative violation
		auto al_eval = [&](const std::vector<double> &xx, std::vector<double> &gg, double &rv, double &gnf, double &Fpure) -> double {
			double Fv = 0.0;
			if (!host_funobj(hp, 2, xx.data(), &Fv, gg.data(), &nstate)) { okc = false; return 0.0; }
			nstate = 0;
			gnf = nrm2(gg); Fpure = Fv; rv = 0.0;
			if (nc > 0 && !host_funcon(hp, 2, xx.data(), cval.data(), J.data(), nc, &nstate)) { okc = false; return 0.0; }
			for (int j = 0; j < nI; j++) { double a = 0.0; const double *row = &A[(size_t)irow[j] * n]; for (int c = 0; c < n; c++) a += row[c] * xx[c]; cval[nc + j] = a; }
			if (nal > 0) {
				double pen = 0.0, rv2 = 0.0;
				for (int j = 0; j < nal; j++) {
					const double cj = cval[j], v = cj + allam[j] / mu;
					const double pj = v < nbl[j] ? nbl[j] : (v > nbu[j] ? nbu[j] : v);
					const double tj = mu * (v - pj), rj = (cj - pj) / (1.0 + std::fabs(cj));   // violation and complementarity (see sqp_kernel)
					tnew[j] = tj; pen += (tj - allam[j]) * (tj + allam[j]) / (2.0 * mu); rv2 += rj * rj;
				}
				Fv += pen;
				if (nc > 0) for (int c = 0; c < n; c++) { double a = 0.0; for (int j = 0; j < nc; j++) a += J[(size_t)c * nc + j] * tnew[j]; gg[c] += a; }
				for (int j = 0; j < nI; j++) { const double *row = &A[(size_t)irow[j] * n]; for (int c = 0; c < n; c++) gg[c] += row[c] * tnew[nc + j]; }
				rv = std::sqrt(rv2);
			}
			return Fv;
		};
		std::vector<double> g(n), gp(n), gn(n), gpn(n), d(n), p(n), xt(n), sv(n), y(n), u(n), t(n);
		double alpha = 0, pnorm = 0, gnf = 0, gnfn = 0, rv = 0, rvn = 0, rvprev = HUGE_VAL, Fpn = 0;
		LineSearch ls;
		if (!lin_ok) { okc = true; info = 9; }
		for (int outer = 0; lin_ok && okc && outer < (nal > 0 ? 30 : 1); outer++) {
			const double sri = nal > 0 ? std::max(sr, std::min(1e-3, 0.1 * rvprev)) : sr;
			int inner = 4, nupd = 0; bool stop = false, at_x = true, weak = false;
			if (outer > 0) make_feasible();
			std::fill(W.begin(), W.end(), 0.0); for (int i = 0; i < n; i++) W[(size_t)i * n + i] = 1.0;
			F = al_eval(x, g, rv, gnf, Fp);
			if (!okc) break;
			project(g, gp);
			d = gp;
			for (;;) {
				if (iter >= itlim) { inner = 4; stop = true; break; }
				if (nal > 0 && mE) {   // keep the direction in null(A) to rounding relative to |d|, not |g| (see sqp_kernel)
					const std::vector<double> keep = lamE;
					project(d, xt); d = xt; lamE = keep;
				}
				for (int i = 0; i < n; i++) p[i] = -d[i];
				double dphi0 = dot(gp, p);
				pnorm = nrm2(p);
				const double tolg = sri * (1.0 + std::max(1.0 + std::fabs(F), gnf));
				if (pnorm == 0.0 || !(dphi0 < 0.0)) {
					if (pnorm != 0.0) {
						nupd = 0;
						std::fill(W.begin(), W.end(), 0.0); for (int i = 0; i < n; i++) W[(size_t)i * n + i] = 1.0;
						d = gp; for (int i = 0; i < n; i++) p[i] = -d[i];
						dphi0 = dot(gp, p); pnorm = nrm2(p);
					}
					if (pnorm == 0.0 || !(dphi0 < 0.0)) { inner = nrm2(gp) <= tolg ? 0 : 6; break; }
				}
				if (nrm2(gp) <= 1e-3 * tolg) { inner = 0; break; }
				const double amax = (o.steplimit > 0 ? o.steplimit : 2.0) * (1.0 + nrm2(x)) / pnorm;
				ls.init(F, dphi0, amax < 1.0 ? amax : 1.0, amax, o.ls_mu, o.ls_eta, o.ls_maxfev);
				double Fn = 0; int rc;
				for (;;) {
					for (int i = 0; i < n; i++) xt[i] = x[i] + ls.a * p[i];
					Fn = al_eval(xt, gn, rvn, gnfn, Fpn);
					if (!okc) { rc = -1; break; }
					project(gn, gpn);
					rc = ls.step(Fn, dot(gpn, p));
					if (rc == 1 || rc == -1) break;
					if (rc == 2) {
						for (int i = 0; i < n; i++) xt[i] = x[i] + ls.a * p[i];
						Fn = al_eval(xt, gn, rvn, gnfn, Fpn);
						project(gn, gpn);
						rc = okc ? 1 : -1; break;
					}
				}
				if (!okc) break;
				if (rc != 1) {
					if (nupd > 0 && nrm2(gp) > tolg) {   // retry from the same point with W0
						nupd = 0;
						std::fill(W.begin(), W.end(), 0.0); for (int i = 0; i < n; i++) W[(size_t)i * n + i] = 1.0;
						d = gp;
						continue;
					}
					if (nrm2(gp) <= tolg) inner = 0; else if (nrm2(gp) <= 1e3 * tolg) { inner = 0; weak = true; } else inner = 6;
					at_x = false; break;
				}
				alpha = ls.a;
				for (int i = 0; i < n; i++) { sv[i] = alpha * p[i]; y[i] = gpn[i] - gp[i]; }
				x = xt;
				if (nupd == 256) {   // memory of the batched solver's pair history: restart from W0 (identity here) like it does
					nupd = 0;
					std::fill(W.begin(), W.end(), 0.0); for (int i = 0; i < n; i++) W[(size_t)i * n + i] = 1.0;
					d = gp;
				}
				for (int i = 0; i < n; i++) { double a = 0; for (int j = 0; j < n; j++) a += W[(size_t)i * n + j] * gpn[j]; t[i] = a; }
				for (int i = 0; i < n; i++) u[i] = t[i] - d[i];
				const double sy = dot(sv, y);
				if (sy > 1e-12 * nrm2(sv) * nrm2(y)) {
					const double rho = 1.0 / sy, c2 = rho * (1.0 + rho * dot(y, u));
					for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) W[(size_t)i * n + j] += -rho * (sv[i] * u[j] + u[i] * sv[j]) + c2 * sv[i] * sv[j];
					nupd++;
					const double a1 = dot(sv, gpn), a2 = dot(u, gpn);
					for (int i = 0; i < n; i++) d[i] = t[i] - rho * (sv[i] * a2 + u[i] * a1) + c2 * sv[i] * a1;
				} else d = t;
				F = Fn; Fp = Fpn; gnf = gnfn; rv = rvn; g = gn; gp = gpn;
				if (g_opt.print_level >= 5) printf("  maj %3d  F=%.15g |Zg|=%.3e alpha=%.3e nf=%d rv=%.2e mu=%g\n", iter, F, nrm2(gp), alpha, ls.nfev, rv, mu);
				iter++;
				if (alpha * pnorm <= sri * (1.0 + nrm2(x)) && nrm2(gp) <= sri * (1.0 + std::max(1.0 + std::fabs(F), gnf))) { inner = 0; break; }
			}
			if (!okc) break;
			if (nal == 0) { info = (inner == 0 && weak) ? 1 : inner; break; }
			if (!at_x) { F = al_eval(x, g, rv, gnf, Fp); if (!okc) break; }
			if (inner == 6) { info = 6; break; }
			if (rv <= 1e-8 && sri <= sr && inner == 0) { allam = tnew; info = weak ? 1 : 0; break; }
			if (stop) { info = 4; break; }
			if (rv <= 0.25 * rvprev) { allam = tnew; rvprev = rv; }
			else mu *= 10.0;
			if (outer == 29) info = 3;
		}
		if (!okc) info = 9;
		if (nal > 0 && okc && lin_ok) make_feasible();                // undo rounding drift off A x = b
		F = Fp;                                                       // objective without the penalty terms
		std::copy(x.begin(), x.end(), initialguess);                 // ntg.c:109: solution overwrites the guess
	}
	*inform = info; *objective = F;
	if (clambda) {   // NPSOL sign: grad F = A' lam_lin + J' lam_nl
		std::fill(clambda, clambda + ntot, 0.0);
		for (int i = 0; i < mE && info != 9; i++) clambda[n + erow[i]] = lamE[i];
		for (int j = 0; j < nI && info != 9; j++) clambda[n + irow[j]] = -allam[nc + j];
		for (int j = 0; j < nc && info != 9; j++) clambda[n + m + j] = -allam[j];
	}
	if (istate) {   // 3 equality, 1/2 at lower/upper bound, 0 inactive (NPSOL's istate codes)
		std::fill(istate, istate + ntot, 0);
		for (int i = 0; i < mE; i++) istate[n + erow[i]] = 3;
		for (int j = 0; j < nal && info != 9; j++) {
			const int slot = j < nc ? n + m + j : n + irow[j - nc];
			int si2;
			double lo2, up2;
			if (j < nc) {
				const int b0 = nlic + nltc + nlfc;
				if (j < nnlic) si2 = b0 + j; else if (j < nnlic + nnltc * nbps) si2 = b0 + nnlic + (j - nnlic) / nbps; else si2 = b0 + nnlic + nnltc + (j - nnlic - nnltc * nbps);
				lo2 = lowerb[si2]; up2 = upperb[si2];
			} else { lo2 = bl[irow[j - nc]]; up2 = bu[irow[j - nc]]; }
			istate[slot] = (lo2 == up2) ? 3 : (allam[j] != 0.0 ? (std::fabs(cval[j] - lo2) <= std::fabs(cval[j] - up2) ? 1 : 2) : 0);
		}
	}
	if (R && info != 9) { // R'R = W^-1 (upper triangular, ld = n, column-major like NPSOL's R)
		std::vector<double> Wc = W, H((size_t)n * n, 0.0), col(n);
		if (chol_lower(Wc, n)) {
			for (int j = 0; j < n; j++) { std::fill(col.begin(), col.end(), 0.0); col[j] = 1.0; chol_solve(Wc, n, col.data()); for (int i = 0; i < n; i++) H[(size_t)i * n + j] = col[i]; }
			if (chol_lower(H, n)) for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) R[(size_t)j * n + i] = i <= j ? H[(size_t)j * n + i] : 0.0;
		}
	}
	if (g_opt.print_level > 0)
		printf(" Exit NTG/MI355X SQP - inform %d, majors %d, nfev %d, objective %.15g\n", info, iter, hp.nfev, F);
	g_cur = prev;
	free_host_problem(hp);
	delete hpp;
}

// ---------------- SplineInterp (colloc.c:449-484): basis on the device, k-term dot on the host ----------------
extern "C" void SplineInterp(double *f, double x, double *knots, int ninterv, double *coefs, int ncoefs, int order,
                             int mult, int maxderiv)
{
	const int n = ninterv * (order - mult) + mult;
	assert(n == ncoefs);
	double *d_kn = nullptr, *d_x = nullptr, *d_blk = nullptr; int *d_off = nullptr;
	std::vector<double> blk((size_t)order * maxderiv); int off = 0;
	bool ok = hip_ok(hipMalloc((void **)&d_kn, (size_t)(ninterv + 1) * 8), "hipMalloc") && hip_ok(hipMalloc((void **)&d_x, 8), "hipMalloc") &&
	          hip_ok(hipMalloc((void **)&d_blk, blk.size() * 8), "hipMalloc") && hip_ok(hipMalloc((void **)&d_off, 4), "hipMalloc") &&
	          hip_ok(hipMemcpy(d_kn, knots, (size_t)(ninterv + 1) * 8, hipMemcpyHostToDevice), "H2D") &&
	          hip_ok(hipMemcpy(d_x, &x, 8, hipMemcpyHostToDevice), "H2D") &&
	          hip_ok(ntg_launch_basis(1, ninterv, order, mult, maxderiv, 1, d_kn, d_x, 0, 0, d_blk, d_off, nullptr), "basis kernel") &&
	          hip_ok(hipMemcpy(blk.data(), d_blk, blk.size() * 8, hipMemcpyDeviceToHost), "D2H") &&
	          hip_ok(hipMemcpy(&off, d_off, 4, hipMemcpyDeviceToHost), "D2H");
	for (void *p : {(void *)d_kn, (void *)d_x, (void *)d_blk, (void *)d_off}) if (p) (void)hipFree(p);
	if (!ok) { for (int i = 0; i < maxderiv; i++) f[i] = NAN; return; }
	for (int i = 0; i < maxderiv; i++) {
		f[i] = 0.0;
		for (int j = 0; j < order; j++) f[i] += blk[(size_t)j * maxderiv + i] * coefs[off + j];
	}
}

// ---------------- matrix helpers the examples link (matrix.h:37-45, matrix.c:211-330) ----------------
extern "C" double **DoubleMatrix(int rows, int cols)
{
	double **t = (double **)malloc(rows * sizeof(double *));
	t[0] = (double *)calloc((size_t)rows * cols, sizeof(double));
	for (int i = 1; i < rows; i++) t[i] = t[0] + (size_t)i * cols;
	return t;
}
extern "C" void FreeDoubleMatrix(double **d) { free(d[0]); free(d); }
extern "C" Matrix *MakeMatrix(int rows, int cols)
{
	Matrix *m = (Matrix *)malloc(sizeof(Matrix));
	m->elements = DoubleMatrix(rows, cols); m->rows = rows; m->cols = cols;
	return m;
}
extern "C" void FreeMatrix(Matrix *m) { FreeDoubleMatrix(m->elements); free(m); }
static FILE *open_out(const char *fn) { if (!strcmp(fn, "stdout")) return stdout; if (!strcmp(fn, "stderr")) return stderr; return fopen(fn, "w"); }
static void close_out(FILE *f) { if (f && f != stdout && f != stderr) fclose(f); }
extern "C" void PrintMatrix(char *fn, Matrix *m)
{
	FILE *f = open_out(fn); if (!f) return;
	for (int i = 0; i < m->rows; i++) { for (int j = 0; j < m->cols; j++) fprintf(f, "%f ", m->elements[i][j]); fprintf(f, "\n"); }
	fprintf(f, "\n\n\n"); close_out(f);
}
extern "C" void PrintVector(char *fn, double *v, int n)
{ FILE *f = open_out(fn); if (!f) return; for (int i = 0; i < n; i++) fprintf(f, "%g ", v[i]); fprintf(f, "\n"); close_out(f); }
extern "C" void PrintiVector(char *fn, int *v, int n)
{ FILE *f = open_out(fn); if (!f) return; for (int i = 0; i < n; i++) fprintf(f, "%d\n", v[i]); close_out(f); }
