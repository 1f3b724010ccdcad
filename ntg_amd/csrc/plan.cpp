// plan.cpp -- host side of the C ABI in include/ntg_amd.h: plan construction (the setup phase
// of ntg(), ntg.c:114-229), batched entry points, error handling.  No numerical fallback lives
// here: basis values, constraint rows, evaluation and the SQP all run in kernels.hip; the host
// only factors the tiny (nclin x nclin) A A' and, on request, builds the preconditioner.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <mutex>
#include <thread>
#include <atomic>
#include <vector>
#include <utility>
#include "ntg_dev.hpp"
#include "plan.hpp"
#include "qpdual.hpp"

static int build_newton_tables(ntg_plan *p);
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                                 \
	do {                                                                                          \
		hipError_t e_ = (x);                                                                      \
		if (e_ != hipSuccess) return fail(NTG_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
	} while (0)

extern "C" const char *ntg_last_error(void) { return g_err.c_str(); }
extern "C" int ntg_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}
extern "C" void ntg_default_opts(ntg_solve_opts *o)
{
	o->itlim = 0; o->opttol = 0.0; o->steplimit = 2.0; o->ls_mu = 1e-4; o->ls_eta = 0.9;
	o->ls_maxfev = 20; o->hessian = 0; o->fixed_iters = 0; o->block_threads = 0; o->qn_memory = 0; o->warm_start = 0;
}
extern "C" const char *ntg_solve_kernel_name(void) { return "sqp_kernel"; }
extern "C" const char *ntg_batch_solve_kernel(const ntg_plan *p, int batch, const ntg_solve_opts *o);
// diagnostic: LDS bytes and block size the solve / eval launches of this plan use
extern "C" int ntg_debug_layout(const ntg_plan *p, const ntg_solve_opts *o, int *lds_solve, int *lds_eval, int *nt_solve);

// ---------------- small dense helpers (row-major, host) ----------------
static bool chol_lower(std::vector<double> &a, int n)
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
static void chol_solve(const std::vector<double> &L, int n, double *b)
{
	for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[(size_t)i * n + k] * b[k]; b[i] = s / L[(size_t)i * n + i]; }
	for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= L[(size_t)k * n + i] * b[k]; b[i] = s / L[(size_t)i * n + i]; }
}

static u64 av_mask(const NtgDims &D, const ntg_av *av, int nav, bool *ok)
{
	u64 m = 0;
	for (int i = 0; i < nav; i++) {
		if (av[i].output < 0 || av[i].output >= D.nout || av[i].deriv < 0 || av[i].deriv >= D.d[av[i].output]) { *ok = false; continue; }
		m |= 1ull << (D.iz[av[i].output] + av[i].deriv);
	}
	return m;
}

template <class T> static int dev_upload(T **dst, const T *src, size_t n, std::vector<void *> &owned)
{
	*dst = nullptr;
	if (n == 0) return 0;
	HIPCHK(hipMalloc((void **)dst, n * sizeof(T)));
	owned.push_back(*dst);
	if (src) HIPCHK(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
	return 0;
}

extern "C" int ntg_plan_create(const ntg_spec *s, int device, ntg_plan **out)
{
	if (!s || !out) return fail(NTG_E_BADARG, "null spec");
	*out = nullptr;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
		return fail(NTG_E_NODEVICE, "no HIP device: libntg_amd has no CPU path");
	if (device < 0 || device >= ndev) return fail(NTG_E_BADARG, "bad device index");
	if (s->nout < 1 || s->nout > NTG_MAX_OUT) return fail(NTG_E_BADARG, "nout out of range (1..NTG_MAX_OUT)");
	if (s->nbps < 2) return fail(NTG_E_BADARG, "need at least 2 breakpoints");
	HIPCHK(hipSetDevice(device));

	ntg_plan *p = new ntg_plan();
	p->device = device;
	NtgDims &D = p->D;
	std::memset(&D, 0, sizeof(D));
	D.nout = s->nout; D.P = s->nbps; D.family = s->family;
	D.nlic = s->nlic; D.nltc = s->nltc; D.nlfc = s->nlfc;
	D.nnlic = s->nnlic; D.nnltc = s->nnltc; D.nnlfc = s->nnlfc;
	D.nicf = s->nicf; D.nucf = s->nucf; D.nfcf = s->nfcf;
	int nz = 0, nC = 0, sumk = 0;
	for (int o = 0; o < s->nout; o++) {
		const int k = s->order[o], m = s->mult[o], l = s->kninterv[o], d = s->maxderiv[o];
		if (k < 1 || k > NTG_MAX_ORDER || m < 0 || m >= k || l < 1 || d < 1 || d > k) {
			delete p; return fail(NTG_E_BADARG, "bad spline spec (order<=NTG_MAX_ORDER, 0<=mult<order, 1<=maxderiv<=order)");
		}
		D.order[o] = k; D.mult[o] = m; D.ninterv[o] = l; D.d[o] = d;
		D.ncoef[o] = l * (k - m) + m;                         // colloc.c:67
		D.iC[o] = nC; D.iz[o] = nz; D.koff[o] = sumk;         // colloc.c:41-49
		nC += D.ncoef[o]; nz += d; sumk += k;
	}
	if (nz > NTG_MAX_NZ) { delete p; return fail(NTG_E_BADARG, "sum(maxderiv) exceeds NTG_MAX_NZ"); }
	D.nC = nC; D.nz = nz; D.sumk = sumk;
	D.nclin = s->nlic + s->nltc * s->nbps + s->nlfc;           // ntg.c:156
	D.ncnln = s->nnlic + s->nnltc * s->nbps + s->nnlfc;        // ntg.c:157
	D.nbounds = s->nlic + s->nltc + s->nlfc + s->nnlic + s->nnltc + s->nnlfc;
	if (s->family != NTG_FAM_KINCAR && s->family != NTG_FAM_VANDERPOL && s->family != NTG_FAM_TESTFAM && s->family != NTG_FAM_OBSTACLE &&
	    s->family != NTG_FAM_QUADROTOR && s->family != NTG_FAM_MANIP && s->family != NTG_FAM_HOST) {
		delete p; return fail(NTG_E_BADARG, "unknown problem family");
	}
	if (s->family != NTG_FAM_HOST) {
		const int dm = s->family == NTG_FAM_QUADROTOR ? 5 : 3;   // Family<>::DM of families.hpp
		for (int o = 0; o < s->nout; o++)
			if (D.d[o] != dm) { delete p; return fail(NTG_E_UNSUPPORTED, "device family: wrong maxderiv (5 for the quadrotor family, 3 otherwise)"); }
	}
	if (s->family == NTG_FAM_QUADROTOR && (s->nout != 4 || s->nnlic || s->nnlfc || s->nnltc > 2)) { delete p; return fail(NTG_E_BADARG, "quadrotor family: 4 outputs, at most two trajectory constraints"); }
	if (s->family == NTG_FAM_MANIP && (s->nout % 3 || s->nnlic || s->nnlfc || s->nnltc > s->nout / 3)) { delete p; return fail(NTG_E_BADARG, "manipulator family: 3 outputs per arm, at most one trajectory constraint per arm"); }
	if (s->family == NTG_FAM_VANDERPOL && s->nout != 1) { delete p; return fail(NTG_E_BADARG, "vanderpol family has one output"); }
	if (s->family == NTG_FAM_TESTFAM && (s->nnlic > 1 || s->nnltc > 2 || s->nnlfc > 1)) { delete p; return fail(NTG_E_BADARG, "testfam has 1/2/1 nonlinear constraints"); }
	if ((s->family == NTG_FAM_KINCAR || s->family == NTG_FAM_VANDERPOL) && D.ncnln > 0) { delete p; return fail(NTG_E_BADARG, "family has no nonlinear constraints"); }
	if (s->family == NTG_FAM_OBSTACLE && (s->nout != 2 || s->nnlic || s->nnlfc || s->nnltc > 1)) { delete p; return fail(NTG_E_BADARG, "obstacle family: 2 outputs, at most one trajectory constraint"); }

	bool ok = true;
	D.icost_mask = av_mask(D, s->icostav, s->nicostav, &ok);
	D.tcost_mask = av_mask(D, s->tcostav, s->ntcostav, &ok);
	D.fcost_mask = av_mask(D, s->fcostav, s->nfcostav, &ok);
	D.icon_mask = av_mask(D, s->icav, s->nicav, &ok);
	D.tcon_mask = av_mask(D, s->tcav, s->ntcav, &ok);
	D.fcon_mask = av_mask(D, s->fcav, s->nfcav, &ok);
	if (!ok) { delete p; return fail(NTG_E_BADARG, "active variable out of range"); }
	// rows of the running-cost gradient kept on chip: the declared trajectory-cost active
	// variables (device functors return zero elsewhere); every flag entry for host callbacks,
	// whose df[] the reference uses in full (cost.c:107-108)
	D.ntav = 0;
	for (int v = 0; v < NTG_MAX_NZ; v++) D.tav_row[v] = -1;
	for (int v = 0; v < nz; v++)
		if (s->family == NTG_FAM_HOST || ((D.tcost_mask >> v) & 1ull)) D.tav_row[v] = (signed char)D.ntav++;
	D.ntav_cost = D.ntav;   // the evaluation's cost pass touches these rows only; the augmented Lagrangian of the solve the ones below too
	for (int v = 0; v < nz; v++)
		if (D.tav_row[v] < 0 && ((D.tcon_mask >> v) & 1ull)) D.tav_row[v] = (signed char)D.ntav++;

	// ---- basis classes: outputs with identical (knots, order, mult, maxderiv) share a table ----
	p->h_knots.resize(s->nout);
	for (int o = 0; o < s->nout; o++) p->h_knots[o].assign(s->knots[o], s->knots[o] + s->kninterv[o] + 1);
	D.nclass = 0;
	std::vector<int> rep;
	int blk_total = 0;
	for (int o = 0; o < s->nout; o++) {
		int c = -1;
		for (int j = 0; j < D.nclass; j++) {
			const int r = rep[j];
			if (D.order[r] == D.order[o] && D.mult[r] == D.mult[o] && D.d[r] == D.d[o] && D.ninterv[r] == D.ninterv[o] &&
			    p->h_knots[r] == p->h_knots[o]) { c = j; break; }
		}
		if (c < 0) {
			c = D.nclass++;
			rep.push_back(o);
			D.cls_blk[c] = blk_total;
			D.cls_k[c] = D.order[o]; D.cls_d[c] = D.d[o]; D.cls_l[c] = D.ninterv[o]; D.cls_m[c] = D.mult[o];
			blk_total += s->nbps * D.order[o] * D.d[o];
		}
		D.cls[o] = c;
	}
	D.blk_total = blk_total;
	D.tav_rmask = 0;
	for (int o = 0; o < s->nout; o++) for (int r = 0; r < D.d[o]; r++) if (D.tav_row[D.iz[o] + r] >= 0) D.tav_rmask |= 1 << r;
	D.uniform = D.nclass == 1;
	for (int o = 1; o < s->nout; o++) if (D.ncoef[o] != D.ncoef[0]) D.uniform = 0;
	p->class_rep = rep;

	// ---- device tables ----
	auto &own = p->owned;
	double *d_bps = nullptr, *d_blk = nullptr; int *d_off = nullptr;
	p->h_bps.assign(s->bps, s->bps + s->nbps);
	if (dev_upload(&d_bps, s->bps, (size_t)s->nbps, own)) { ntg_plan_destroy(p); return NTG_E_HIP; }
	if (dev_upload(&d_blk, (const double *)nullptr, (size_t)blk_total, own)) { ntg_plan_destroy(p); return NTG_E_HIP; }
	if (dev_upload(&d_off, (const int *)nullptr, (size_t)D.nclass * s->nbps, own)) { ntg_plan_destroy(p); return NTG_E_HIP; }
	for (int c = 0; c < D.nclass; c++) {
		const int o = rep[c];
		double *d_kn = nullptr;
		std::vector<void *> tmp_own;
		if (dev_upload(&d_kn, p->h_knots[o].data(), p->h_knots[o].size(), tmp_own)) { ntg_plan_destroy(p); return NTG_E_HIP; }
		hipError_t e = ntg_launch_basis(1, D.ninterv[o], D.order[o], D.mult[o], D.d[o], s->nbps, d_kn, d_bps, 0, 0,
		                                d_blk + D.cls_blk[c], d_off + (size_t)c * s->nbps, nullptr);
		hipError_t e2 = hipDeviceSynchronize();
		own.insert(own.end(), tmp_own.begin(), tmp_own.end());   // kept: ntg_batch_interp evaluates the basis at other times
		p->d_knots.push_back(d_kn);
		if (e != hipSuccess || e2 != hipSuccess) { ntg_plan_destroy(p); return fail(NTG_E_HIP, "basis kernel failed"); }
	}
	p->h_blk.resize(blk_total); p->h_off.resize((size_t)D.nclass * s->nbps);
	if (hipMemcpy(p->h_blk.data(), d_blk, (size_t)blk_total * 8, hipMemcpyDeviceToHost) != hipSuccess ||
	    hipMemcpy(p->h_off.data(), d_off, p->h_off.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { ntg_plan_destroy(p); return fail(NTG_E_HIP, "reading the basis tables back failed"); }
	NtgTables &T = p->T;
	std::memset(&T, 0, sizeof(T));
	T.bps = d_bps; T.blk = d_blk; T.off = d_off;
	// ---- active (class, derivative) channels: a derivative row is kept on chip only if some
	//      active variable (any of the six lists) uses it; host callbacks use all of them ----
	{
		const u64 all = D.icost_mask | D.tcost_mask | D.fcost_mask | D.icon_mask | D.tcon_mask | D.fcon_mask;
		std::vector<int> chrow((size_t)D.nclass * NTG_MAX_ORDER, -1), chcol((size_t)D.nclass * NTG_MAX_ORDER, -1);
		std::vector<double> rowv; std::vector<unsigned int> colp;
		if (s->nbps > 65535) { ntg_plan_destroy(p); return fail(NTG_E_UNSUPPORTED, "more than 65535 breakpoints"); }
		for (int c = 0; c < D.nclass; c++) {
			const int k = D.cls_k[c], dd = D.cls_d[c], P = s->nbps, nc = D.ncoef[rep[c]];
			const double *blk = p->h_blk.data() + D.cls_blk[c];
			const int *off = p->h_off.data() + (size_t)c * P;
			// support width of the column form (same for every derivative of the class)
			std::vector<int> cnt(nc, 0);
			for (int i = 0; i < P; i++) for (int q = 0; q < k; q++) cnt[off[i] + q]++;
			int W = 0; for (int v : cnt) W = std::max(W, v);
			const int W4 = (W + 3) & ~3;
			D.cls_W[c] = W4; D.cls_nc[c] = nc;
			for (int r = 0; r < dd; r++) {
				bool active = s->family == NTG_FAM_HOST;
				for (int o = 0; o < s->nout; o++) if (D.cls[o] == c && ((all >> (D.iz[o] + r)) & 1ull)) active = true;
				if (!active) continue;
				chrow[(size_t)c * NTG_MAX_ORDER + r] = (int)rowv.size();
				chcol[(size_t)c * NTG_MAX_ORDER + r] = (int)colp.size();
				for (int q = 0; q < k; q++) for (int i = 0; i < P; i++) rowv.push_back(blk[((size_t)i * k + q) * dd + r]);
				rowv.push_back(0.0);   // value index k*P: the zero the padding entries of a column point to
				// column cl: word 0 = its first breakpoint (the breakpoints of a column are consecutive because the block
				// offsets are non-decreasing), then W4 16-bit value indices q*P+i, two per word, padded with k*P
				if ((size_t)k * P >= 65535) { ntg_plan_destroy(p); return fail(NTG_E_UNSUPPORTED, "order*nbps exceeds the 16-bit column index"); }
				const size_t base = colp.size();
				const int WW = (W4 / 2 + 1 + 3) & ~3;   // == colp_words(W4) of solve_impl.hpp
				colp.resize(base + (size_t)WW * nc, 0u);
				std::vector<int> fill(nc, 0), first(nc, 0);
				std::vector<unsigned int> idx((size_t)nc * W4, (unsigned int)(k * P));
				for (int i = 0; i < P; i++) for (int q = 0; q < k; q++) {
					const int cl = off[i] + q, sidx = fill[cl]++;
					if (sidx == 0) first[cl] = i;
					else if (first[cl] + sidx != i) { ntg_plan_destroy(p); return fail(NTG_E_UNSUPPORTED, "breakpoints of a basis function are not consecutive"); }
					idx[(size_t)cl * W4 + sidx] = (unsigned int)(q * P + i);
				}
				for (int cl = 0; cl < nc; cl++) {
					unsigned int *w = &colp[base + (size_t)cl * WW];
					w[0] = (unsigned int)first[cl];
					for (int s2 = 0; s2 < W4; s2++) w[1 + s2 / 2] |= idx[(size_t)cl * W4 + s2] << (16 * (s2 & 1));
				}
			}
		}
		// the same columns by value (class 0; see NtgDims::colv_total)
		{
			std::vector<double> colv;
			const int c = 0, k = D.cls_k[c], P = s->nbps, nc = D.ncoef[rep[c]], W4 = D.cls_W[c], WP = W4 + 2, WW = (W4 / 2 + 1 + 3) & ~3;
			D.colv_stride = WP;
			for (int r = 0; r < NTG_MAX_ORDER; r++) D.ch_colv0[r] = -1;
			for (int r = 0; r < D.cls_d[c]; r++) {
				if (chcol[(size_t)c * NTG_MAX_ORDER + r] < 0) continue;
				D.ch_colv0[r] = (int)colv.size();
				const double *rv = rowv.data() + chrow[(size_t)c * NTG_MAX_ORDER + r];
				for (int cl = 0; cl < nc; cl++) {
					const unsigned int *w = &colp[(size_t)chcol[(size_t)c * NTG_MAX_ORDER + r] + (size_t)cl * WW];
					for (int s2 = 0; s2 < W4; s2++) colv.push_back(rv[(w[1 + s2 / 2] >> (16 * (s2 & 1))) & 0xffffu]);
					colv.push_back((double)w[0]);
					colv.push_back(0.0);
				}
			}
			(void)k; (void)P;
			D.colv_total = (int)colv.size();
			if (colv.empty()) colv.push_back(0.0);
			double *d_colv = nullptr;
			if (dev_upload(&d_colv, colv.data(), colv.size(), own)) { ntg_plan_destroy(p); return NTG_E_HIP; }
			T.colv = d_colv;
		}
		// breakpoint groups of class 0 (NtgDims::ig_n)
		{
			const int P = s->nbps; const int *off = p->h_off.data();
			int n = 0; bool ok = true;
			for (int i = 0; i < P && ok;) { int j = i; while (j < P && off[j] == off[i]) j++; if (n >= 64 || j - i > 6) ok = false; else D.igb[n++] = (unsigned short)i; i = j; }
			D.ig_n = ok ? n : 0;
			if (ok) D.igb[n] = (unsigned short)P;
		}
		D.row_total = (int)rowv.size(); D.col_total = (int)colp.size();
		if (colp.empty()) colp.push_back(0);
		double *d_rowv = nullptr; unsigned int *d_colp = nullptr; int *d_chrow = nullptr, *d_chcol = nullptr;
		if (dev_upload(&d_rowv, rowv.data(), rowv.size(), own) || dev_upload(&d_colp, colp.data(), colp.size(), own) ||
		    dev_upload(&d_chrow, chrow.data(), chrow.size(), own) ||
		    dev_upload(&d_chcol, chcol.data(), chcol.size(), own)) { ntg_plan_destroy(p); return NTG_E_HIP; }
		T.rowv = d_rowv; T.colp = d_colp; T.chrow = d_chrow; T.chcol = d_chcol;
		p->h_chrow = chrow;
		for (int r = 0; r < NTG_MAX_ORDER; r++) { D.ch_row0[r] = chrow[r]; D.ch_col0[r] = chcol[r]; }
	}

	// ---- linear constraint rows on the device, (A A')^-1 on the host ----
	if (D.nclin > 0) {
		double *d_lic = nullptr, *d_ltc = nullptr, *d_lfc = nullptr, *d_ab = nullptr, *d_sinv = nullptr; int *d_rbp = nullptr;
		std::vector<void *> tmp_own;
		if (dev_upload(&d_lic, s->lic, (size_t)s->nlic * nz, tmp_own) || dev_upload(&d_ltc, s->ltc, (size_t)s->nltc * nz, tmp_own) ||
		    dev_upload(&d_lfc, s->lfc, (size_t)s->nlfc * nz, tmp_own) ||
		    dev_upload(&d_ab, (const double *)nullptr, (size_t)D.nclin * sumk, own) ||
		    dev_upload(&d_rbp, (const int *)nullptr, (size_t)D.nclin, own)) { ntg_plan_destroy(p); return NTG_E_HIP; }
		hipError_t e = ntg_launch_linrows(D, T, d_lic, d_ltc, d_lfc, d_ab, d_rbp, nullptr);
		hipError_t e2 = hipDeviceSynchronize();
		for (void *q : tmp_own) hipFree(q);
		if (e != hipSuccess || e2 != hipSuccess) { ntg_plan_destroy(p); return fail(NTG_E_HIP, "linrows kernel failed"); }
		p->h_aband.resize((size_t)D.nclin * sumk); p->h_rbp.resize(D.nclin);
		if (hipMemcpy(p->h_aband.data(), d_ab, p->h_aband.size() * 8, hipMemcpyDeviceToHost) != hipSuccess ||
		    hipMemcpy(p->h_rbp.data(), d_rbp, p->h_rbp.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { ntg_plan_destroy(p); return fail(NTG_E_HIP, "reading the linear rows back failed"); }
		T.aband = d_ab; T.rbp = d_rbp;
		// dense rows; split into the equality rows (kept satisfied by projection) and the rows declared as
		// inequalities (spec->lin_ineq, handled by the augmented-Lagrangian loop like nonlinear rows)
		const int mall = D.nclin;
		std::vector<double> Aall((size_t)mall * nC, 0.0);
		ntg_plan_dense_A(p, Aall.data());
		std::vector<int> erow, irow, rowmap(mall), linflag((size_t)std::max(1, s->nlic + s->nltc + s->nlfc), 0);
		for (int r = 0; r < mall; r++) {
			int slot;
			if (r < s->nlic) slot = r;
			else if (r < s->nlic + s->nltc * s->nbps) slot = s->nlic + (r - s->nlic) / s->nbps;
			else slot = s->nlic + s->nltc + (r - s->nlic - s->nltc * s->nbps);
			const bool ineq = s->lin_ineq && s->lin_ineq[slot] != 0;
			linflag[slot] = ineq ? 1 : 0;
			if (ineq) { rowmap[r] = -(int)irow.size() - 1; irow.push_back(r); } else { rowmap[r] = (int)erow.size(); erow.push_back(r); }
		}
		const int m = (int)erow.size(), nI = (int)irow.size();
		D.mE = m; D.nI = nI;
		std::vector<double> Ad((size_t)std::max(m, 1) * nC, 0.0);
		for (int i = 0; i < m; i++) std::copy(&Aall[(size_t)erow[i] * nC], &Aall[(size_t)erow[i] * nC] + nC, &Ad[(size_t)i * nC]);
		{
			int *d_er = nullptr, *d_rm = nullptr, *d_lf = nullptr;
			if (erow.empty()) erow.push_back(0);
			if (dev_upload(&d_er, erow.data(), erow.size(), own) || dev_upload(&d_rm, rowmap.data(), rowmap.size(), own) ||
			    dev_upload(&d_lf, linflag.data(), linflag.size(), own)) { ntg_plan_destroy(p); return NTG_E_HIP; }
			T.erow = d_er; T.rowmap = d_rm; T.linflag = d_lf;
		}
		if (nI > 0) {   // inequality rows: CSR (c = A_r x) and CSC (g += A_r' t), exact zeros dropped
			std::vector<int> rptr(nI + 1, 0), rcol, cptr(nC + 1, 0), crow;
			std::vector<double> rval, cval;
			for (int j = 0; j < nI; j++) { const double *row = &Aall[(size_t)irow[j] * nC]; for (int c = 0; c < nC; c++) if (row[c] != 0.0) { rcol.push_back(c); rval.push_back(row[c]); } rptr[j + 1] = (int)rcol.size(); }
			for (int c = 0; c < nC; c++) { for (int j = 0; j < nI; j++) { const double v = Aall[(size_t)irow[j] * nC + c]; if (v != 0.0) { crow.push_back(j); cval.push_back(v); } } cptr[c + 1] = (int)crow.size(); }
			if (rcol.empty()) { rcol.push_back(0); rval.push_back(0.0); crow.push_back(0); cval.push_back(0.0); }
			int *d_ir = nullptr, *d_rp = nullptr, *d_rc = nullptr, *d_cp = nullptr, *d_cr = nullptr; double *d_rv = nullptr, *d_cv = nullptr;
			if (dev_upload(&d_ir, irow.data(), irow.size(), own) || dev_upload(&d_rp, rptr.data(), rptr.size(), own) ||
			    dev_upload(&d_rc, rcol.data(), rcol.size(), own) || dev_upload(&d_rv, rval.data(), rval.size(), own) ||
			    dev_upload(&d_cp, cptr.data(), cptr.size(), own) || dev_upload(&d_cr, crow.data(), crow.size(), own) ||
			    dev_upload(&d_cv, cval.data(), cval.size(), own)) { ntg_plan_destroy(p); return NTG_E_HIP; }
			T.irow = d_ir; T.icsr_ptr = d_rp; T.icsr_col = d_rc; T.icsr_val = d_rv; T.icsc_ptr = d_cp; T.icsc_row = d_cr; T.icsc_val = d_cv;
		}
		// S = A_E A_E' -> S^-1
		std::vector<double> S((size_t)m * m, 0.0);
		for (int i = 0; i < m; i++) for (int j = 0; j <= i; j++) {
			double a = 0.0;
			for (int c = 0; c < nC; c++) a += Ad[(size_t)i * nC + c] * Ad[(size_t)j * nC + c];
			S[(size_t)i * m + j] = a; S[(size_t)j * m + i] = a;
		}
		p->lin_ok = chol_lower(S, m);
		std::vector<double> Sinv((size_t)m * m, 0.0), col(m);
		if (p->lin_ok) {
			for (int j = 0; j < m; j++) {
				std::fill(col.begin(), col.end(), 0.0); col[j] = 1.0;
				chol_solve(S, m, col.data());
				for (int i = 0; i < m; i++) Sinv[(size_t)i * m + j] = col[i];
			}
			for (int i = 0; i < m; i++) for (int j = 0; j < i; j++) { // symmetrise
				const double a = 0.5 * (Sinv[(size_t)i * m + j] + Sinv[(size_t)j * m + i]);
				Sinv[(size_t)i * m + j] = a; Sinv[(size_t)j * m + i] = a;
			}
		}
		if (dev_upload(&d_sinv, Sinv.data(), Sinv.size(), own)) { ntg_plan_destroy(p); return NTG_E_HIP; }
		T.sinv = d_sinv;
		// sparse A, exact zeros dropped (a unit lic row touches one output, and a derivative at an
		// end point touches only the first/last r+1 coefficients)
		std::vector<int> rptr(m + 1, 0), rcol, cptr(nC + 1, 0), crow;
		std::vector<double> rval, cval;
		for (int i = 0; i < m; i++) { for (int c = 0; c < nC; c++) if (Ad[(size_t)i * nC + c] != 0.0) { rcol.push_back(c); rval.push_back(Ad[(size_t)i * nC + c]); } rptr[i + 1] = (int)rcol.size(); }
		for (int c = 0; c < nC; c++) { for (int i = 0; i < m; i++) if (Ad[(size_t)i * nC + c] != 0.0) { crow.push_back(i); cval.push_back(Ad[(size_t)i * nC + c]); } cptr[c + 1] = (int)crow.size(); }
		D.lin_nnz = (int)rcol.size();
		if (rcol.empty()) { rcol.push_back(0); rval.push_back(0.0); crow.push_back(0); cval.push_back(0.0); }
		int *d_rp = nullptr, *d_rc = nullptr, *d_cp = nullptr, *d_cr = nullptr; double *d_rv = nullptr, *d_cv = nullptr;
		if (dev_upload(&d_rp, rptr.data(), rptr.size(), own) || dev_upload(&d_rc, rcol.data(), rcol.size(), own) ||
		    dev_upload(&d_rv, rval.data(), rval.size(), own) || dev_upload(&d_cp, cptr.data(), cptr.size(), own) ||
		    dev_upload(&d_cr, crow.data(), crow.size(), own) || dev_upload(&d_cv, cval.data(), cval.size(), own)) { ntg_plan_destroy(p); return NTG_E_HIP; }
		T.csr_ptr = d_rp; T.csr_col = d_rc; T.csr_val = d_rv; T.csc_ptr = d_cp; T.csc_row = d_cr; T.csc_val = d_cv;
		p->h_csr_ptr = rptr; p->h_csr_col = rcol; p->h_csc_ptr = cptr; p->h_csc_row = crow; p->h_erow = erow;
		// (A A')^-1 as CSR, exact zeros dropped (block diagonal when constraint rows decouple)
		std::vector<int> sptr(m + 1, 0), scol; std::vector<double> sval;
		for (int i = 0; i < m; i++) { for (int j = 0; j < m; j++) if (Sinv[(size_t)i * m + j] != 0.0) { scol.push_back(j); sval.push_back(Sinv[(size_t)i * m + j]); } sptr[i + 1] = (int)scol.size(); }
		D.sinv_nnz = (int)scol.size();
		if (scol.empty()) { scol.push_back(0); sval.push_back(0.0); }
		int *d_sp = nullptr, *d_sc = nullptr; double *d_sv = nullptr;
		if (dev_upload(&d_sp, sptr.data(), sptr.size(), own) || dev_upload(&d_sc, scol.data(), scol.size(), own) ||
		    dev_upload(&d_sv, sval.data(), sval.size(), own)) { ntg_plan_destroy(p); return NTG_E_HIP; }
		T.sinv_ptr = d_sp; T.sinv_col = d_sc; T.sinv_val = d_sv;
		p->h_sinv_ptr = sptr; p->h_sinv_col = scol;
		// projector Q = A'(AA')^-1 A: only the coefficients some constraint touches have a non-zero
		// row; keep those rows as ELL (zero padded) when that is small
		{
			std::vector<double> SA((size_t)m * nC, 0.0), Q((size_t)nC * nC, 0.0);
			for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) { const double sij = Sinv[(size_t)i * m + j]; if (sij != 0.0) for (int c = 0; c < nC; c++) SA[(size_t)i * nC + c] += sij * Ad[(size_t)j * nC + c]; }
			for (int i = 0; i < m; i++) for (int a = 0; a < nC; a++) { const double aia = Ad[(size_t)i * nC + a]; if (aia != 0.0) for (int c = 0; c < nC; c++) Q[(size_t)a * nC + c] += aia * SA[(size_t)i * nC + c]; }
			std::vector<short> qidx(nC, -1); int nt = 0, w = 0;
			for (int a = 0; a < nC; a++) { int cnt = 0; for (int c = 0; c < nC; c++) if (Q[(size_t)a * nC + c] != 0.0) cnt++; if (cnt) { qidx[a] = (short)std::min(nt, 32000); nt++; w = std::max(w, cnt); } }
			if (nt > 0 && nt < 32000 && (size_t)nt * w * 12 + (size_t)nC * 2 <= 16 * 1024) {
				std::vector<int> qcol((size_t)nt * w, 0); std::vector<double> qval((size_t)nt * w, 0.0);
				for (int a = 0; a < nC; a++) if (qidx[a] >= 0) { int e = 0; for (int c = 0; c < nC; c++) if (Q[(size_t)a * nC + c] != 0.0) { qcol[(size_t)qidx[a] * w + e] = c; qval[(size_t)qidx[a] * w + e] = Q[(size_t)a * nC + c]; e++; } }
				short *d_qi = nullptr; int *d_qc = nullptr; double *d_qv = nullptr;
				if (dev_upload(&d_qi, qidx.data(), qidx.size(), own) || dev_upload(&d_qc, qcol.data(), qcol.size(), own) ||
				    dev_upload(&d_qv, qval.data(), qval.size(), own)) { ntg_plan_destroy(p); return NTG_E_HIP; }
				T.q_idx = d_qi; T.q_col = d_qc; T.q_val = d_qv;
				D.q_use = 1; D.q_nt = nt; D.q_w = w;
				// Rows that pin whole coefficients (the usual initial / final conditions: a derivative at an end point touches the first /
				// last r + 1 coefficients only): range(A') is spanned by m unit vectors and Q is the identity on those coefficients, zero
				// elsewhere -- g - Q g just zeroes the pinned entries (NtgDims::q_pin, used by the wave kernel).  Decided on the computed Q
				// itself, to 1e-9 (the accuracy (A A')^-1 gives it is ~1e-10): the reference's cumulative-add linspace (ntg.c:374-389) puts
				// the last breakpoint an ulp or two off the last knot, so rows of ~1e-15 for the neighbouring coefficients exist too.
				{
					int npin = 0; double dev = 0.0;
					std::vector<unsigned char> pinned(nC, 0);
					for (int a = 0; a < nC; a++) if (qidx[a] >= 0) {
						const bool pin = Q[(size_t)a * nC + a] > 0.5;
						npin += pin ? 1 : 0; pinned[a] = pin ? 1 : 0;
						for (int c = 0; c < nC; c++) dev = std::max(dev, std::fabs(Q[(size_t)a * nC + c] - ((pin && a == c) ? 1.0 : 0.0)));
					}
					D.q_pin = (npin == m && dev <= 1e-9) ? 1 : 0;
					unsigned char *d_pin = nullptr;
					if (D.q_pin && dev_upload(&d_pin, pinned.data(), pinned.size(), own)) { ntg_plan_destroy(p); return NTG_E_HIP; }
					T.q_pinned = d_pin;
					if (getenv("NTG_AMD_DEBUG_PLAN")) fprintf(stderr, "projector: %d non-zero rows, %d equality rows, width %d, %d pinned, max |Q - I_pinned| %.3e -> q_pin %d\n", nt, m, w, npin, dev, D.q_pin);
				}
				p->h_qidx = qidx; p->h_qcol = qcol; p->h_qval = qval;
			}
		}
		// the general three-step operator (A g, (AA')^-1, A' lam) is staged in LDS only when Q is not used
		D.lin_lds = (!D.q_use && ((size_t)D.lin_nnz * 24 + (size_t)D.sinv_nnz * 12 + (size_t)(2 * m + nC + 3) * 4) <= 24 * 1024) ? 1 : 0;
		p->h_Adense.swap(Aall);
		p->h_AE.swap(Ad);
	} else {
		p->lin_ok = true;
	}
	if (s->nlic > 0 && dev_upload(&p->d_lic, s->lic, (size_t)s->nlic * nz, own)) { ntg_plan_destroy(p); return NTG_E_HIP; }
	if (s->nlic > 0) p->h_lic.assign(s->lic, s->lic + (size_t)s->nlic * nz);
	if (s->nltc > 0) p->h_ltc.assign(s->ltc, s->ltc + (size_t)s->nltc * nz);
	if (s->nlfc > 0) p->h_lfc.assign(s->lfc, s->lfc + (size_t)s->nlfc * nz);
	// keep what the preconditioner build needs
	p->tcostav.assign(s->tcostav, s->tcostav + s->ntcostav);
	p->icostav.assign(s->icostav, s->icostav + s->nicostav);
	p->fcostav.assign(s->fcostav, s->fcostav + s->nfcostav);
	if (build_newton_tables(p)) { ntg_plan_destroy(p); return NTG_E_HIP; }
	*out = p;
	return 0;
}

// ---- structured Newton mode (newton.hpp): does the plan qualify, and its batch-shared tables ----
// Qualifies when: the family offers the per-group second-order blocks (Family::COUPLE / CG of families.hpp, mirrored here), one
// spline spec for every output, only trajectory nonlinear rows on exactly the flag entries the family's block covers, no
// linear inequality rows, and equality rows that pin a square invertible block of coefficients (the usual initial / final
// conditions): then null(A_E) = {pinned coefficients = 0} and the reduced Hessian is a principal submatrix of the band.
static int build_newton_tables(ntg_plan *p)
{
	NtgDims &D = p->D;
	D.nwt_on = 0; D.nwt_tab = 0; p->T.nwt_tu = p->T.nwt_g = nullptr;
	int go = 0, cg = 0;
	u64 gmask = 0;   // constraint flag entries of group 0, relative to the group's first flag entry
	const int dm = D.d[0];
	switch (D.family) {
	case NTG_FAM_OBSTACLE: go = 2; cg = 2; gmask = (1ull << 0) | (1ull << 3); break;
	// (x, y, z) couple through thrust and speed; the yaw output appears in no row: a FREE output -- its block of the model is the cost
	// model's, the same for every problem and every refresh, factored once here (nwt_lf) and solved by an otherwise idle wave
	case NTG_FAM_QUADROTOR: go = 3; cg = 6; gmask = (1ull << 1) | (1ull << 2) | (1ull << 6) | (1ull << 7) | (1ull << 11) | (1ull << 12); break;
	case NTG_FAM_MANIP: go = 3; cg = 3; gmask = (1ull << 0) | (1ull << 3) | (1ull << 6); break;
	default: return 0;
	}
	if (!D.uniform || D.nI > 0 || D.nnlic || D.nnlfc || D.nnltc <= 0 || !p->lin_ok) return 0;
	const int ngrp = D.nout / go, nfo = D.nout - ngrp * go;   // coupling groups; outputs left over are free (only the quadrotor family has one)
	if (ngrp < 1 || (nfo && D.family != NTG_FAM_QUADROTOR)) return 0;
	const int k = D.order[0], nco = D.ncoef[0], n = D.nC, P = D.P, hb = k * go - 1, m = D.mE, hbf = k - 1;
	if (hb > 32 || ngrp + nfo > 8) return 0;
	u64 want = 0;
	for (int g = 0; g < ngrp; g++) want |= gmask << (dm * go * g);
	if (D.tcon_mask != want) return 0;
	// pinned coefficients = the columns the equality rows touch; they must form a square system
	std::vector<char> pinned(n, 0);
	int npin = 0;
	// (entries at rounding level do not count: the last breakpoint of a cumulative-add linspace (ntg.c:385-388) may lie an ulp
	// past the last knot, where the spline is extrapolated and the other basis functions are ~1e-16 instead of 0; the solve
	// re-projects x and every direction onto A x = b anyway)
	for (int i = 0; i < m; i++) {
		double big = 0.0;
		for (int c = 0; c < n; c++) big = std::max(big, std::fabs(p->h_AE[(size_t)i * n + c]));
		for (int c = 0; c < n; c++) if (std::fabs(p->h_AE[(size_t)i * n + c]) > 1e-10 * big && !pinned[c]) { pinned[c] = 1; npin++; }
	}
	if (npin != m) return 0;
	std::vector<int> map, pos(n, -1);
	int ng = -1;
	for (int g = 0; g < ngrp; g++) {
		int cnt = 0;
		for (int cl = 0; cl < nco; cl++) for (int o = g * go; o < (g + 1) * go; o++) {
			const int c = D.iC[o] + cl;
			if (pinned[c]) continue;
			pos[c] = g * 1000000 + cnt; map.push_back(c); cnt++;
		}
		if (ng < 0) ng = cnt; else if (cnt != ng) return 0;
	}
	if (ng < 1) return 0;
	for (int c = 0; c < n; c++) if (pos[c] >= 0) pos[c] = (pos[c] / 1000000) * ng + pos[c] % 1000000;
	// free outputs: their free coefficients follow the groups' in the maps, output by output (ngf each)
	int ngf = 0;
	for (int f = 0; f < nfo; f++) {
		int cnt = 0;
		for (int cl = 0; cl < nco; cl++) {
			const int c = D.iC[ngrp * go + f] + cl;
			if (pinned[c]) continue;
			pos[c] = -2 - cnt; map.push_back(c); cnt++;   // provisional: -2 - index inside the output
		}
		if (f == 0) ngf = cnt; else if (cnt != ngf) return 0;
	}
	if (nfo && ngf < 1) return 0;
	for (int f = 0; f < nfo; f++) for (int cl = 0; cl < nco; cl++) { const int c = D.iC[ngrp * go + f] + cl; if (pos[c] <= -2) pos[c] = ngrp * ng + f * ngf + (-2 - pos[c]); }
	// breakpoint range of every local coefficient, from the block offsets (consecutive: checked when the column form was built)
	std::vector<short> lo(nco, (short)P), hi(nco, 0);
	const int *off = p->h_off.data();
	for (int i = 0; i < P; i++) for (int q = 0; q < k; q++) { const int cl = off[i] + q; lo[cl] = (short)std::min<int>(lo[cl], i); hi[cl] = (short)std::max<int>(hi[cl], i + 1); }
	// cost model: 2 w_i on the trajectory-cost variables, 2 on the initial / final ones (diagonal in the flag: same output only)
	const int ld = hb + 1;
	std::vector<double> k0((size_t)ngrp * ng * ld, 0.0);
	const int ldf = hbf + 1;
	std::vector<double> k0f((size_t)nfo * ngf * ldf, 0.0);   // cost model of the free outputs (band of half width k - 1 each)
	const double *blk = p->h_blk.data();
	auto add = [&](const std::vector<ntg_av> &av, int bp, double w) {
		for (const ntg_av &a : av) {
			const int o = a.output, r = a.deriv, g = o / go;
			for (int q1 = 0; q1 < k; q1++) for (int q2 = 0; q2 < k; q2++) {
				const int c1 = D.iC[o] + off[bp] + q1, c2 = D.iC[o] + off[bp] + q2;
				if (pos[c1] < 0 || pos[c2] < 0) continue;
				const double v = w * blk[((size_t)bp * k + q1) * dm + r] * blk[((size_t)bp * k + q2) * dm + r];
				if (o >= ngrp * go) {
					const int f = o - ngrp * go, p1 = pos[c1] - ngrp * ng - f * ngf, p2 = pos[c2] - ngrp * ng - f * ngf;
					if (p1 >= p2) k0f[((size_t)f * ngf + p1) * ldf + (p2 - p1 + hbf)] += v;
					continue;
				}
				const int p1 = pos[c1] - g * ng, p2 = pos[c2] - g * ng;
				if (p1 < p2) continue;
				k0[((size_t)g * ng + p1) * ld + (p2 - p1 + hb)] += v;
			}
		}
	};
	for (int i = 0; i < P; i++) {
		double w = 0.0;
		if (i > 0) w += (p->h_bps[i] - p->h_bps[i - 1]) / 2;
		if (i < P - 1) w += (p->h_bps[i + 1] - p->h_bps[i]) / 2;
		if (D.nucf) add(p->tcostav, i, 2.0 * w);
	}
	if (D.nicf) add(p->icostav, 0, 2.0);
	if (D.nfcf) add(p->fcostav, P - 1, 2.0);
	// breakpoint groups and colours; constraint flag entries of a group
	std::vector<int> ig;
	for (int i = 0; i < P;) { int j = i; while (j < P && off[j] == off[i]) j++; ig.push_back(i); ig.push_back(j - i); i = j; }
	const int nint = (int)ig.size() / 2;
	int cover = 1;
	for (int cl = 0; cl < nco; cl++) { int cnt = 0; for (int t = 0; t < nint; t++) { const int of = off[ig[2 * t]]; if (cl >= of && cl < of + k) cnt++; } cover = std::max(cover, cnt); }
	for (int t = 0; t + cover < nint; t++) if (off[ig[2 * (t + cover)]] < off[ig[2 * t]] + k) return 0;   // same-colour groups must not share coefficients
	u64 upack = 0;
	{
		int u = 0;
		for (int o = 0; o < go; o++) for (int r = 0; r < dm; r++) if ((gmask >> (dm * o + r)) & 1ull) { upack |= (u64)((o << 4) | r) << (8 * u); u++; }
		if (u != cg || cg > 8) return 0;
	}
	if (D.ig_n != nint) return 0;   // more than 64 groups or more than 6 breakpoints in one (see NtgDims::ig_n)
	D.nwt_nint = nint; D.nwt_cover = cover; D.nwt_upack = upack;
	// the free coefficients must be the same contiguous range [clo, chi) of every output (pinned ends)
	{
		int clo = -1, chi = -1;
		for (int cl = 0; cl < nco; cl++) if (!pinned[D.iC[0] + cl]) { if (clo < 0) clo = cl; chi = cl + 1; }
		if (clo < 0) return 0;
		for (int o = 0; o < D.nout; o++) for (int cl = 0; cl < nco; cl++) if ((bool)pinned[D.iC[o] + cl] != !(cl >= clo && cl < chi)) return 0;
		D.nwt_clo = clo; D.nwt_chi = chi;
	}
	// ---- tables of the QP-based SQP step's regime without constraint curvature (NtgTables::nwt_tu, nwt_g): K0 is then the model of every
	//      major iteration of every problem, so K0^-1 M_i' per breakpoint and M_k K0^-1 M_i' per pair of breakpoints are plan constants.
	//      Built when all coupling groups share one cost model and the tables stay under 48 MB. ----
	std::vector<double> tu, gt;
	{
		bool same = true;
		for (int g = 1; g < ngrp && same; g++) for (size_t e = 0; e < (size_t)ng * ld; e++) if (k0[(size_t)g * ng * ld + e] != k0[e]) { same = false; break; }
		const size_t ntu = (size_t)P * cg * ng, ngt2 = (size_t)P * P * cg * cg;
		if (same && (ntu + ngt2) * 8 <= (size_t)48 << 20 && !getenv("NTG_AMD_NO_QPTAB")) {
			// band Cholesky of K0 (group 0), compact lower band: L(i, j) at l[i * ld + (j - i + hb)]
			std::vector<double> l(k0.begin(), k0.begin() + (size_t)ng * ld);
			auto L = [&](int i, int j) -> double & { return l[(size_t)i * ld + (j - i + hb)]; };
			bool ok = true;
			for (int j = 0; j < ng && ok; j++) {
				double d = L(j, j);
				for (int t = std::max(0, j - hb); t < j; t++) d -= L(j, t) * L(j, t);
				if (!(d > 0.0)) { ok = false; break; }
				d = std::sqrt(d); L(j, j) = d;
				for (int i = j + 1; i <= std::min(ng - 1, j + hb); i++) {
					double sv = L(i, j);
					for (int t = std::max(0, i - hb); t < j; t++) sv -= L(i, t) * L(j, t);
					L(i, j) = sv / d;
				}
			}
			if (ok) {
				tu.assign(ntu, 0.0); gt.assign(ngt2, 0.0);
				std::vector<int> uo(cg), ur(cg);
				for (int u = 0; u < cg; u++) { uo[u] = (int)((upack >> (8 * u + 4)) & 15u); ur[u] = (int)((upack >> (8 * u)) & 15u); }
				const int clo = D.nwt_clo, chi = D.nwt_chi;
				for (int i = 0; i < P; i++) for (int u = 0; u < cg; u++) {
					double *t = tu.data() + ((size_t)i * cg + u) * ng;
					for (int q = 0; q < k; q++) { const int cl = off[i] + q; if (cl >= clo && cl < chi) t[(cl - clo) * go + uo[u]] = blk[((size_t)i * k + q) * dm + ur[u]]; }
					for (int r = 0; r < ng; r++) { double sv = t[r]; for (int c2 = std::max(0, r - hb); c2 < r; c2++) sv -= L(r, c2) * t[c2]; t[r] = sv / L(r, r); }
					for (int r = ng - 1; r >= 0; r--) { double sv = t[r]; for (int c2 = r + 1; c2 <= std::min(ng - 1, r + hb); c2++) sv -= L(c2, r) * t[c2]; t[r] = sv / L(r, r); }
				}
				for (int kb = 0; kb < P; kb++) for (int i = 0; i < P; i++) for (int v = 0; v < cg; v++) for (int u = 0; u < cg; u++) {
					const double *t = tu.data() + ((size_t)i * cg + u) * ng;
					double sv = 0.0;
					for (int q = 0; q < k; q++) { const int cl = off[kb] + q; if (cl >= clo && cl < chi) sv += blk[((size_t)kb * k + q) * dm + ur[v]] * t[(cl - clo) * go + uo[v]]; }
					gt[(((size_t)kb * P + i) * cg + v) * cg + u] = sv;
				}
			}
		}
	}
	// two-sided factorisation: two waves per group when the band is long enough and the largest workgroup has the waves (newton.hpp).  The
	// cost model is then split like the band: rows of the top part and the separator stay where they are, the entries of bottom rows move
	// to the reversed array (entry (i, j) -> row n - 1 - j, same band offset), which follows the groups' top arrays in the table.
	D.nwt_tw = 0; D.nwt_ja = D.nwt_jb = 0;
	// (only while every working wave still has a SIMD of its own: with four groups -- config E, eight waves -- the second wave of a group
	// shares its SIMD with another group's, both streams are issue bound, and the measured solve was 16 % SLOWER)
	if (ng >= 128 && 2 * ngrp + nfo <= 4 && !getenv("NTG_AMD_NO_TWOSIDED")) {
		const int jt = (ng - 32) / 16;
		D.nwt_tw = 1; D.nwt_ja = (jt + 1) / 2; D.nwt_jb = jt / 2;
		const int sepn = ng - 16 * (D.nwt_ja + D.nwt_jb), ngt = 16 * D.nwt_ja + sepn, brows = 16 * D.nwt_jb + 48;
		std::vector<double> kb((size_t)ngrp * brows * ld, 0.0);
		for (int g = 0; g < ngrp; g++)
			for (int i = ngt; i < ng; i++) for (int e = 0; e <= hb; e++) {
				const int j = i - hb + e;
				double &src = k0[((size_t)g * ng + i) * ld + e];
				if (j >= 0) kb[((size_t)g * brows + (ng - 1 - j)) * ld + e] = src;
				src = 0.0;
			}
		k0.insert(k0.end(), kb.begin(), kb.end());
	}
	// the free outputs' factor, in the layout nwt_solve_wave reads (row-major band, the diagonal inverted)
	std::vector<double> lf((size_t)nfo * ngf * ldf, 0.0);
	for (int f = 0; f < nfo; f++) {
		std::vector<double> a((size_t)ngf * ngf, 0.0);
		for (int i = 0; i < ngf; i++) for (int e = 0; e <= hbf; e++) { const int j = i - hbf + e; if (j >= 0) a[(size_t)i * ngf + j] = a[(size_t)j * ngf + i] = k0f[((size_t)f * ngf + i) * ldf + e]; }
		if (!chol_lower(a, ngf)) return 0;   // a cost that leaves a free output without curvature: no structured Newton mode
		for (int i = 0; i < ngf; i++) for (int e = 0; e <= hbf; e++) { const int j = i - hbf + e; if (j >= 0) lf[((size_t)f * ngf + i) * ldf + e] = (j == i) ? 1.0 / a[(size_t)i * ngf + i] : a[(size_t)i * ngf + j]; }
	}
	D.nwt_tab = 0; p->T.nwt_tu = p->T.nwt_g = nullptr;
	if (!tu.empty()) {
		double *d_tu = nullptr, *d_g = nullptr;
		if (dev_upload(&d_tu, tu.data(), tu.size(), p->owned) || dev_upload(&d_g, gt.data(), gt.size(), p->owned)) return NTG_E_HIP;
		p->T.nwt_tu = d_tu; p->T.nwt_g = d_g; D.nwt_tab = 1;
	}
	int *d_map = nullptr, *d_pos = nullptr; double *d_k0 = nullptr, *d_lf = nullptr; short *d_lo = nullptr, *d_hi = nullptr;
	if (dev_upload(&d_lf, lf.data(), lf.size(), p->owned)) return NTG_E_HIP;
	p->T.nwt_lf = d_lf; D.nwt_nfo = nfo; D.nwt_ngf = ngf; D.nwt_hbf = hbf;
	if (dev_upload(&d_map, map.data(), map.size(), p->owned) || dev_upload(&d_pos, pos.data(), pos.size(), p->owned) ||
	    dev_upload(&d_k0, k0.data(), k0.size(), p->owned) || dev_upload(&d_lo, lo.data(), lo.size(), p->owned) ||
	    dev_upload(&d_hi, hi.data(), hi.size(), p->owned)) return NTG_E_HIP;
	p->T.nwt_map = d_map; p->T.nwt_pos = d_pos; p->T.nwt_k0 = d_k0; p->T.nwt_lo = d_lo; p->T.nwt_hi = d_hi;
	D.nwt_on = 1; D.nwt_ngrp = ngrp; D.nwt_go = go; D.nwt_ng = ng; D.nwt_hb = hb; D.nwt_cg = cg;
	return 0;
}

// dense row-major [nclin][nC] A from the banded rows
void ntg_plan_dense_A(const ntg_plan *p, double *A)
{
	const NtgDims &D = p->D;
	std::fill(A, A + (size_t)D.nclin * D.nC, 0.0);
	for (int r = 0; r < D.nclin; r++)
		for (int o = 0; o < D.nout; o++) {
			const int col0 = D.iC[o] + p->h_off[(size_t)D.cls[o] * D.P + p->h_rbp[r]];
			for (int q = 0; q < D.order[o]; q++) A[(size_t)r * D.nC + col0 + q] = p->h_aband[(size_t)r * D.sumk + D.koff[o] + q];
		}
}

extern "C" void ntg_plan_destroy(ntg_plan *p)
{
	if (!p) return;
	hipSetDevice(p->device);
	for (void *q : p->owned) hipFree(q);
	for (void *q : p->grid_owned) hipFree(q);
	delete p;
}

extern "C" int ntg_plan_dims(const ntg_plan *p, int *nC, int *nz, int *nclin, int *ncnln, int *nbounds, int *sumk, int *nblk)
{
	if (!p) return fail(NTG_E_BADARG, "null plan");
	if (nC) *nC = p->D.nC; if (nz) *nz = p->D.nz; if (nclin) *nclin = p->D.nclin; if (ncnln) *ncnln = p->D.ncnln;
	if (nbounds) *nbounds = p->D.nbounds; if (sumk) *sumk = p->D.sumk;
	if (nblk) { int t = 0; for (int o = 0; o < p->D.nout; o++) t += p->D.P * p->D.order[o] * p->D.d[o]; *nblk = t; }
	return 0;
}

extern "C" int ntg_plan_tables(const ntg_plan *p, double *blk, int *off, double *A)
{
	if (!p) return fail(NTG_E_BADARG, "null plan");
	const NtgDims &D = p->D;
	size_t pos = 0;
	for (int o = 0; o < D.nout; o++) {
		const size_t cnt = (size_t)D.P * D.order[o] * D.d[o];
		if (blk) std::memcpy(blk + pos, p->h_blk.data() + D.cls_blk[D.cls[o]], cnt * 8);
		if (off) std::memcpy(off + (size_t)o * D.P, p->h_off.data() + (size_t)D.cls[o] * D.P, (size_t)D.P * 4);
		pos += cnt;
	}
	if (A && D.nclin) // column-major nclin x nC
		for (int r = 0; r < D.nclin; r++)
			for (int c = 0; c < D.nC; c++) A[(size_t)c * D.nclin + r] = p->h_Adense[(size_t)r * D.nC + c];
	return 0;
}

// Dense core of the preconditioner for one block: W0 = Z (Z' H0 Z)^-1 Z' with Z = null(A) from a Householder QR of A'.
// H0 (n x n), A (m x n) and W0 (n x n) are row-major.
static int precond_block(const std::vector<double> &H0, const std::vector<double> &A, int m, int n, std::vector<double> &W0)
{
	const int nr = n - m;
	if (nr <= 0) return fail(NTG_E_UNSUPPORTED, "no free directions");
	std::vector<double> Q((size_t)n * n, 0.0), R((size_t)n * std::max(m, 1), 0.0), v(n);
	for (int i = 0; i < n; i++) Q[(size_t)i * n + i] = 1.0;
	for (int j = 0; j < m; j++) for (int i = 0; i < n; i++) R[(size_t)i * m + j] = A[(size_t)j * n + i];
	for (int j = 0; j < m && j < n; j++) {
		double nrm = 0.0;
		for (int i = j; i < n; i++) nrm += R[(size_t)i * m + j] * R[(size_t)i * m + j];
		nrm = std::sqrt(nrm);
		if (nrm == 0.0) continue;
		const double alpha = R[(size_t)j * m + j] > 0 ? -nrm : nrm;
		std::fill(v.begin(), v.end(), 0.0);
		for (int i = j; i < n; i++) v[i] = R[(size_t)i * m + j];
		v[j] -= alpha;
		double vn = 0.0;
		for (int i = j; i < n; i++) vn += v[i] * v[i];
		if (vn == 0.0) continue;
		for (int c = j; c < m; c++) { double s = 0.0; for (int i = j; i < n; i++) s += v[i] * R[(size_t)i * m + c]; s = 2.0 * s / vn; for (int i = j; i < n; i++) R[(size_t)i * m + c] -= s * v[i]; }
		for (int c = 0; c < n; c++) { double s = 0.0; for (int i = j; i < n; i++) s += Q[(size_t)c * n + i] * v[i]; s = 2.0 * s / vn; for (int i = j; i < n; i++) Q[(size_t)c * n + i] -= s * v[i]; }
	}
	// Zt[j][:] = column m+j of Q ; T = H0 Z ; Hr = Z' T
	std::vector<double> Zt((size_t)nr * n), Tm((size_t)nr * n), Hr((size_t)nr * nr);
	for (int j = 0; j < nr; j++) for (int i = 0; i < n; i++) Zt[(size_t)j * n + i] = Q[(size_t)i * n + m + j];
	for (int j = 0; j < nr; j++) for (int i = 0; i < n; i++) { double s = 0.0; const double *h = &H0[(size_t)i * n], *z = &Zt[(size_t)j * n]; for (int k = 0; k < n; k++) s += h[k] * z[k]; Tm[(size_t)j * n + i] = s; }
	double tr = 0.0;
	auto form_hr = [&](double reg) {
		for (int i = 0; i < nr; i++) for (int j = 0; j <= i; j++) { double s = 0.0; const double *a = &Zt[(size_t)i * n], *b = &Tm[(size_t)j * n]; for (int k = 0; k < n; k++) s += a[k] * b[k]; Hr[(size_t)i * nr + j] = s; Hr[(size_t)j * nr + i] = s; }
		tr = 0.0; for (int i = 0; i < nr; i++) tr += Hr[(size_t)i * nr + i];
		for (int i = 0; i < nr; i++) Hr[(size_t)i * nr + i] += reg * tr / nr + 1e-300;
	};
	form_hr(1e-12);
	if (!chol_lower(Hr, nr)) { form_hr(1e-6); if (!chol_lower(Hr, nr)) return fail(NTG_E_UNSUPPORTED, "preconditioner not positive definite"); }
	{
		// H0 singular on null(A) (no equality rows: constants and ramps cost nothing): the regularised inverse would scale
		// those directions by 1e12 (1e6 after the harder regularisation) -- no preconditioner then, the solve starts from
		// the identity.  Checked after whichever factorisation succeeded.
		double lo = 1e300, hi = 0.0;
		for (int i = 0; i < nr; i++) { const double dd = Hr[(size_t)i * nr + i] * Hr[(size_t)i * nr + i]; lo = std::min(lo, dd); hi = std::max(hi, dd); }
		if (lo < 1e-9 * hi) return 1;
	}
	// X[:, c] = Hr^-1 Zt[:, c] ; W0 = Zt' X
	std::vector<double> X((size_t)n * nr), col(nr), Zc((size_t)n * nr);
	for (int c = 0; c < n; c++) { for (int i = 0; i < nr; i++) { col[i] = Zt[(size_t)i * n + c]; Zc[(size_t)c * nr + i] = col[i]; } chol_solve(Hr, nr, col.data()); for (int i = 0; i < nr; i++) X[(size_t)c * nr + i] = col[i]; }
	W0.assign((size_t)n * n, 0.0);
	for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) {
		double s = 0.0;
		const double *zi = &Zc[(size_t)i * nr], *xj = &X[(size_t)j * nr];
		for (int k = 0; k < nr; k++) s += zi[k] * xj[k];
		W0[(size_t)i * n + j] = s; W0[(size_t)j * n + i] = s;
	}
	// entries at rounding level relative to the diagonal are noise of the orthogonal factorisation: drop them
	for (int i = 0; i < n; i++) for (int j = 0; j < n; j++)
		if (i != j && std::fabs(W0[(size_t)i * n + j]) <= 1e-13 * std::sqrt(std::fabs(W0[(size_t)i * n + i] * W0[(size_t)j * n + j]))) W0[(size_t)i * n + j] = 0.0;
	return 0;
}

// W0 = Z (Z' H0 Z)^-1 Z', H0 = trapezoid-weighted sum of m m' over the cost active variables.
// Host, once per plan (shared by the whole batch).  H0 is block diagonal by output, so outputs that no row of A
// couples give independent blocks of W0: each block is factorised on its own (12 x 183^3 instead of 2196^3 for
// config E) and the result goes to HBM as ELL rows.
static int build_precond(ntg_plan *p)
{
	const NtgDims &D = p->D;
	const int n = D.nC, m = D.mE, P = D.P;
	if (n - m <= 0) return fail(NTG_E_UNSUPPORTED, "no free directions");
	if (n > 65535) return fail(NTG_E_UNSUPPORTED, "preconditioner: more than 65535 coefficients");
	// components of outputs under "some row of A touches both"
	std::vector<int> comp(D.nout), outof(n);
	for (int o = 0; o < D.nout; o++) { comp[o] = o; for (int j = 0; j < D.ncoef[o]; j++) outof[D.iC[o] + j] = o; }
	for (int r = 0; r < m; r++) {
		int first = -1;
		for (int j = 0; j < n; j++) if (p->h_AE[(size_t)r * n + j] != 0.0) {
			const int c = comp[outof[j]];
			if (first < 0) first = c;
			else if (c != first) { const int lo = std::min(c, first), hi = std::max(c, first); for (int o = 0; o < D.nout; o++) if (comp[o] == hi) comp[o] = lo; first = lo; }
		}
	}
	std::vector<std::vector<std::pair<int, double>>> rows(n);   // W0 row i: (column, value), zeros dropped
	std::vector<std::vector<double>> wblocks; int nb_first = -1; bool by_output = true;   // one dense block per output, few distinct?
	for (int o0 = 0; o0 < D.nout; o0++) {
		if (comp[o0] != o0) continue;
		std::vector<int> idx, rsel, loc(n, -1);
		for (int j = 0; j < n; j++) if (comp[outof[j]] == o0) { loc[j] = (int)idx.size(); idx.push_back(j); }
		for (int r = 0; r < m; r++) { bool hit = false; for (int j : idx) if (p->h_AE[(size_t)r * n + j] != 0.0) { hit = true; break; } if (hit) rsel.push_back(r); }
		const int nb = (int)idx.size(), mb = (int)rsel.size();
		std::vector<double> H0((size_t)nb * nb, 0.0), Ab((size_t)std::max(mb, 1) * nb, 0.0), Wb;
		auto add = [&](const std::vector<ntg_av> &av, int bp, double w) {
			for (const ntg_av &a : av) {
				const int o = a.output, r = a.deriv, k = D.order[o], d = D.d[o], c = D.cls[o];
				if (comp[o] != o0) continue;
				const int base = loc[D.iC[o]] + p->h_off[(size_t)c * P + bp];   // an output's coefficients are contiguous in idx
				const double *b = p->h_blk.data() + D.cls_blk[c] + (size_t)bp * k * d;
				for (int q1 = 0; q1 < k; q1++) for (int q2 = 0; q2 < k; q2++)
					H0[(size_t)(base + q1) * nb + base + q2] += w * b[q1 * d + r] * b[q2 * d + r];
			}
		};
		for (int i = 0; i < P; i++) {
			double w = 0.0;
			if (i > 0) w += (p->h_bps[i] - p->h_bps[i - 1]) / 2;
			if (i < P - 1) w += (p->h_bps[i + 1] - p->h_bps[i]) / 2;
			if (D.nucf) add(p->tcostav, i, w);
		}
		if (D.nicf) add(p->icostav, 0, 1.0);
		if (D.nfcf) add(p->fcostav, P - 1, 1.0);
		for (int i = 0; i < mb; i++) for (int j = 0; j < nb; j++) Ab[(size_t)i * nb + j] = p->h_AE[(size_t)rsel[i] * n + idx[j]];
		const int rc = precond_block(H0, Ab, mb, nb, Wb);
		if (rc > 0) { p->precond_ready = true; p->precond_singular = true; return 0; }   // singular model: identity start instead
		if (rc) return rc;
		{
			int nouts = 0; for (int o = 0; o < D.nout; o++) if (comp[o] == o0) nouts++;
			if (nouts != 1 || idx[0] != D.iC[o0] || (nb_first >= 0 && nb != nb_first)) by_output = false;
			else {
				nb_first = nb;
				int found = -1;
				for (size_t q = 0; q < wblocks.size(); q++) if (std::memcmp(Wb.data(), wblocks[q].data(), Wb.size() * sizeof(double)) == 0) found = (int)q;
				if (found < 0) { found = (int)wblocks.size(); wblocks.push_back(Wb); }
				p->D.n0_blk[o0] = found;
			}
		}
		for (int i = 0; i < nb; i++) for (int j = 0; j < nb; j++) if (Wb[(size_t)i * nb + j] != 0.0) rows[idx[i]].push_back({idx[j], Wb[(size_t)i * nb + j]});
	}
	if (by_output && nb_first > 0 && nb_first * D.nout == n && wblocks.size() <= 4) {
		// one dense block per output, kept once per distinct block: s-major (symmetric, so [s][row] == [row][s]),
		// rows padded with zeros to a multiple of 16
		const int spad = (nb_first + 15) & ~15;
		std::vector<double> all(wblocks.size() * (size_t)spad * nb_first + 16, 0.0);   // +16: the last row tile reads up to 15 words past a row
		for (size_t q = 0; q < wblocks.size(); q++) std::copy(wblocks[q].begin(), wblocks[q].end(), all.begin() + q * (size_t)spad * nb_first);
		double *d_wb = nullptr;
		if (dev_upload(&d_wb, all.data(), all.size(), p->owned)) return NTG_E_HIP;
		p->T.n0b = d_wb; p->T.n0b_n = nb_first; p->T.n0b_sp = spad; p->T.n0b_nblk = (int)wblocks.size();
		// the ELL form below stays as the fallback (block taller than the workgroup)
	}
	// ELL, s-major, zeros dropped
	int w = 0;
	for (int i = 0; i < n; i++) w = std::max(w, (int)rows[i].size());
	std::vector<double> ev((size_t)w * n, 0.0); std::vector<unsigned short> ec((size_t)w * n, 0);
	for (int i = 0; i < n; i++) { int e = 0; for (auto &cv : rows[i]) { ev[(size_t)e * n + i] = cv.second; ec[(size_t)e * n + i] = (unsigned short)cv.first; e++; } }
	double *d_n0 = nullptr; unsigned short *d_n0c = nullptr;
	if (dev_upload(&d_n0, ev.data(), ev.size(), p->owned) || dev_upload(&d_n0c, ec.data(), ec.size(), p->owned)) return NTG_E_HIP;
	p->T.n0 = d_n0; p->T.n0c = d_n0c; p->T.n0_w = w;
	p->precond_ready = true;
	return 0;
}

// workgroup size: breakpoints and coefficients are spread over the lanes
static int auto_threads(const NtgDims &D) { return (D.P <= 128 && D.nC <= 512) ? 128 : (D.nC <= 1024 ? 256 : 512); }

static int resolve_params(const ntg_plan *p, const ntg_solve_opts *o, SolveParams *sp, int *nt)
{
	const NtgDims &D = p->D;
	ntg_solve_opts def; ntg_default_opts(&def);
	if (!o) o = &def;
	sp->itlim = o->itlim > 0 ? o->itlim : std::max(50, 3 * (D.nC + D.nclin) + 10 * D.ncnln);
	sp->memcap = std::min(sp->itlim, o->qn_memory > 0 ? o->qn_memory : 256);
	sp->ls_maxfev = o->ls_maxfev > 0 ? o->ls_maxfev : 20;
	sp->hessian = o->hessian; sp->fixed_iters = o->fixed_iters; sp->warm = o->warm_start ? 1 : 0;
	// the structured Newton mode does not apply (or: per-problem grids -- its cost model and maps belong to the plan's grid): collocation
	// preconditioner.  Decided HERE, once: workspace size, layout and launch all see the same mode (a switch after the workspace was sized
	// for hessian = 2 -- no quasi-Newton history -- would let the quasi-Newton mode write its history past the end of the workspace).
	if (sp->hessian < 0 || sp->hessian > 3) sp->hessian = 0;
	// the QP-based SQP step (hessian = 3) rides on the structured Newton mode's band model: where that does not apply it does not either
	if (sp->hessian >= 2 && (!D.nwt_on || (p->grid_batch && !p->T.pp_k0))) sp->hessian = 1;
	sp->stamps = getenv("NTG_AMD_STAMPS") ? std::max(1, atoi(getenv("NTG_AMD_STAMPS"))) : 0;
	const double r = o->opttol > 0 ? o->opttol : std::pow(DBL_EPSILON, 0.8);
	sp->sr = std::sqrt(r);
	sp->steplimit = o->steplimit > 0 ? o->steplimit : 2.0;
	sp->ls_mu = o->ls_mu > 0 ? o->ls_mu : 1e-4; sp->ls_eta = o->ls_eta > 0 ? o->ls_eta : 0.9;
	int t = o->block_threads;
	if (t != 128 && t != 256 && t != 512) t = auto_threads(D);
	// the structured Newton mode runs one wavefront per coupling group: a workgroup too small for the plan's groups grows to hold them, and a
	// plan with more groups than the largest workgroup has waves takes the collocation preconditioner (decided here, once: workspace, layout
	// and launch all see the same mode)
	if (sp->hessian >= 2) {
		const int nwv = (D.nwt_tw ? 2 : 1) * D.nwt_ngrp + D.nwt_nfo;   // two waves per group with the two-sided factorisation
		while (t < 512 && nwv * 64 > t) t *= 2;
		if (nwv * 64 > t) sp->hessian = 1;
	}
	*nt = t;
	return 0;
}

// does the solve keep all its vectors in LDS, or only the two that are read across lanes (sqp_kernel, BIG)?
static int solve_layout(const NtgDims &D, int nt, SmemLayout *L, int *big, const SolveParams *sp = nullptr)
{
	// a short quasi-Newton memory keeps its pair scalars in LDS -- unless those bytes would cost a resident workgroup
	const int hrc = (sp && sp->hessian < 2 && sp->memcap < NTG_HRC_LDS) ? sp->memcap + 1 : 0;   // + 1: the one-vector-per-major form wants a slot more than pairs
	auto resident = [](int total) { return (160 * 1024) / std::max(total, 1); };
	*big = 0;
	const int qp = (sp && sp->hessian == 3) ? 1 : 0;   // the QP-based SQP step keeps its slots behind the Newton mode's solve vectors
	*L = ntg_make_layout(D, nt, 5, 1, hrc, qp);
	if (hrc && resident(L->total) < resident(L->total - 16 * hrc)) *L = ntg_make_layout(D, nt, 5, 1, 0, qp);
	if (L->total <= 160 * 1024) return 0;
	*big = 1;
	*L = ntg_make_layout(D, nt, 1, 0, hrc, qp);
	if (hrc && L->total > 160 * 1024) *L = ntg_make_layout(D, nt, 1, 0, 0, qp);
	return L->total <= 160 * 1024 ? 0 : -1;
}
static size_t hist_doubles(const NtgDims &D, int batch, const SolveParams &sp)
{
	if (sp.hessian >= 2) return 0;   // the structured Newton mode keeps no quasi-Newton pairs
	return (size_t)batch * sp.memcap * (2 * D.nC + 2);
}
static size_t al_doubles(const NtgDims &D, int batch) { return (size_t)batch * 2 * (D.ncnln + D.nI); }
// structured Newton mode: band matrix / factor of every group + the per-breakpoint blocks, per problem
static size_t nwt_doubles(const NtgDims &D, int batch, const SolveParams &sp, int nt)
{
	if (sp.hessian < 2 || !D.nwt_on) return 0;
	const size_t rev = D.nwt_tw ? (size_t)D.nwt_ngrp * (16 * D.nwt_jb + 48) * (D.nwt_hb + 1) : 0;   // the reversed arrays of the two-sided factorisation
	// QP-based SQP step: the slots' columns W J' and the QP's multipliers (sqp_kernel, qp_pp)
	const size_t qp = sp.hessian == 3 ? (size_t)ntg_qp_maxa(D.nwt_ngrp, nt) * ((D.nC + 1) & ~1) + (size_t)((D.ncnln + 1) & ~1) * (3 + D.nwt_cg + ntg_qp_maxa(D.nwt_ngrp, nt)) : 0;
	return (size_t)batch * ((size_t)D.nwt_ngrp * D.nwt_ng * (D.nwt_hb + 1) + rev + (size_t)D.nwt_ngrp * D.P * D.nwt_cg * D.nwt_cg + qp);
}

static int plan_ncu(const ntg_plan *p)
{
	if (p->ncu <= 0) {
		hipDeviceProp_t prop;
		p->ncu = (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
	}
	return p->ncu;
}

// which solve kernel takes the batch: the wave kernel (one wavefront per problem) for the plans it covers, else sqp_kernel.
// The answer depends on the options as ntg_batch_solve will see them (the preconditioner may turn out singular: hessian 1 -> 0).
static bool wave_takes(const ntg_plan *p, const SolveParams &sp, int batch, NtgWavePlan *w)
{
	if (sp.hessian == 1 && !p->precond_ready) return false;   // decided after the preconditioner exists (its block form is part of the test)
	return ntg_wave_plan(p->D, p->T, sp, batch, plan_ncu(p), w);
}

// name of the kernel ntg_batch_solve will launch for (plan, batch, options): "sqp_wave_kernel" (one wavefront per problem) or "sqp_kernel"
extern "C" const char *ntg_batch_solve_kernel(const ntg_plan *p, int batch, const ntg_solve_opts *o)
{
	if (!p) return "";
	SolveParams sp; int nt;
	resolve_params(p, o, &sp, &nt);
	if (sp.hessian == 1 && p->precond_ready && p->precond_singular) sp.hessian = 0;
	NtgWavePlan w;
	NtgTables T2 = p->T;
	if (sp.hessian == 1 && !p->precond_ready) { T2.n0b = (const double *)1; T2.n0b_n = p->D.ncoef[0]; }   // not built yet: assume the block form
	return ntg_wave_plan(p->D, T2, sp, batch, plan_ncu(p), &w) ? "sqp_wave_kernel" : "sqp_kernel";
}

extern "C" long long ntg_batch_workspace_bytes(const ntg_plan *p, int batch, const ntg_solve_opts *o)
{
	if (!p) return 0;
	SolveParams sp; int nt;
	resolve_params(p, o, &sp, &nt);
	SmemLayout L; int big;
	solve_layout(p->D, nt, &L, &big, &sp);
	const size_t npad = (size_t)((p->D.nC + 1) & ~1);
	size_t dbl = hist_doubles(p->D, batch, sp) + al_doubles(p->D, batch) + (big ? (size_t)batch * 5 * npad : 0) + nwt_doubles(p->D, batch, sp, nt);
	// the wave kernel's HBM tier of the direction chains (per resident wave, not per problem); sized for both of its instances
	for (int h = 0; h < 2; h++) {
		SolveParams s2 = sp; s2.hessian = h;
		NtgWavePlan w;
		if (ntg_wave_plan(p->D, p->T, s2, batch, plan_ncu(p), &w)) dbl = std::max(dbl, w.hist_doubles);
		else if (h == 1 && !p->grid_batch && !p->precond_ready) {
			// the preconditioner is not built yet: assume the wave kernel will take the solve (same shape test without the block form)
			NtgTables T2 = p->T; T2.n0b = (const double *)1; T2.n0b_n = p->D.ncoef[0];
			if (ntg_wave_plan(p->D, T2, s2, batch, plan_ncu(p), &w)) dbl = std::max(dbl, w.hist_doubles);
		}
	}
	return (long long)(dbl * 8 + 256);
}

extern "C" int ntg_batch_bounds(const ntg_plan *p, int batch, const double *d_lower, const double *d_upper,
                                double *d_bl, double *d_bu, void *stream)
{
	if (!p) return fail(NTG_E_BADARG, "null plan");
	HIPCHK(hipSetDevice(p->device));
	HIPCHK(ntg_launch_bounds(p->D, batch, d_lower, d_upper, d_bl, d_bu, (hipStream_t)stream));
	return 0;
}

extern "C" int ntg_batch_eval(const ntg_plan *p, int batch, const double *d_x, int mode, double *d_f, double *d_g,
                              double *d_c, double *d_jband, double *d_cjac, void *stream)
{
	if (!p) return fail(NTG_E_BADARG, "null plan");
	if (batch <= 0) return 0;
	if (!d_x) return fail(NTG_E_BADARG, "null x");
	if (mode < 0 || mode > 2) return fail(NTG_E_BADARG, "mode must be 0, 1 or 2");
	if (p->D.family == NTG_FAM_HOST) return fail(NTG_E_UNSUPPORTED, "host-callback plans evaluate through npsolCostFunction");
	if (p->grid_batch && batch != p->grid_batch) return fail(NTG_E_BADARG, "the plan carries per-problem grids for another batch size");
	if (batch <= 0) return 0;
	HIPCHK(hipSetDevice(p->device));
	const NtgDims &D = p->D;
	int nt = auto_threads(D);
	SmemLayout L = ntg_make_layout(D, nt, 0, 1);
	if (nt == 256 && L.total > 80 * 1024) { nt = 512; L = ntg_make_layout(D, nt, 0, 1); }   // one workgroup per CU anyway: give it more waves
	if (L.total > 160 * 1024) return fail(NTG_E_UNSUPPORTED, "problem tables exceed 160 KiB of LDS");
	hipStream_t st = (hipStream_t)stream;
	if (d_cjac && D.ncnln) HIPCHK(hipMemsetAsync(d_cjac, 0, (size_t)batch * D.ncnln * D.nC * 8, st)); // GcJac starts zeroed (ntg.c:217)
	// persistent grid = what is resident at once: LDS-limited workgroups per CU, capped by the waves a
	// CU holds (32) and by the register budget the kernel was compiled for (NTG_EVAL_WAVES per SIMD is a
	// lower bound; 8 workgroups of 128 threads = 4 waves per SIMD is the most that can ever be resident)
	int ncu = 256;
	{ hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount; }
	const int wg_per_cu = std::max(1, std::min(std::min(8, 32 / (nt / 64)), (160 * 1024) / std::max(L.total, 1)));
	int grid = std::min(batch, ncu * wg_per_cu);
	if (const char *eg = getenv("NTG_AMD_EVAL_GRID")) grid = std::max(1, std::min(grid, atoi(eg)));   // experiments only
	EvalArgs ea{nt, grid, ncu, batch, mode, d_x, d_f, d_g, d_c, d_jband, d_cjac, st};
	HIPCHK(ntg_launch_eval(D, p->T, L, ea));
	return 0;
}

extern "C" int ntg_batch_solve(const ntg_plan *pc, int batch, const double *d_lower, const double *d_upper, double *d_x,
                               const ntg_solve_opts *o, double *d_objective, int *d_inform, int *d_iters, int *d_nfev,
                               double *d_clambda, void *d_work, long long work_bytes, void *stream)
{
	ntg_plan *p = const_cast<ntg_plan *>(pc);
	if (!p) return fail(NTG_E_BADARG, "null plan");
	if (batch <= 0) return 0;
	if (!d_x || !d_lower || !d_upper) return fail(NTG_E_BADARG, "null argument");
	if (p->D.family == NTG_FAM_HOST) return fail(NTG_E_UNSUPPORTED, "host-callback plans are solved by ntg()");
	if (p->grid_batch && batch != p->grid_batch) return fail(NTG_E_BADARG, "the plan carries per-problem grids for another batch size");
	if (!p->lin_ok) return fail(NTG_E_BADARG, "linear constraint rows are rank deficient");
	if (batch <= 0) return 0;
	HIPCHK(hipSetDevice(p->device));
	SolveParams sp; int nt;
	resolve_params(p, o, &sp, &nt);
	if (work_bytes < ntg_batch_workspace_bytes(p, batch, o) || !d_work) return fail(NTG_E_BADARG, "workspace too small");
	if (p->grid_batch && sp.hessian == 1 && !p->T.pp_n0b) return fail(NTG_E_BADARG, "per-problem grids were set without the preconditioner (with_precond = 0): solve with hessian = 0");
	if (sp.hessian == 1) {   // built on first use, once: two threads or streams may first-solve the same plan
		std::lock_guard<std::mutex> lk(p->precond_mutex);
		if (!p->precond_ready) { int rc = build_precond(p); if (rc) return rc; }
	}
	if (sp.hessian == 1 && p->precond_singular) sp.hessian = 0;
	SmemLayout L; int big;
	if (solve_layout(p->D, nt, &L, &big, &sp)) return fail(NTG_E_UNSUPPORTED, "problem state exceeds 160 KiB of LDS");
	double *alw = (double *)d_work + hist_doubles(p->D, batch, sp);   // [batch][2][ncnln] multipliers, estimates
	double *vecw = alw + al_doubles(p->D, batch);                     // [batch][5][npad] x, gp, gp+, d, g (BIG only)
	double *nwtw = vecw + (big ? (size_t)batch * 5 * ((p->D.nC + 1) & ~1) : 0);   // structured Newton mode: bands and blocks
	SqpArgs sa{nt, big, batch, d_lower, d_upper, d_x, d_objective, d_inform, d_iters, d_nfev, d_clambda, (double *)d_work, alw,
	           big ? vecw : nullptr, sp.hessian >= 2 ? nwtw : nullptr, (hipStream_t)stream,
	           (unsigned int *)((char *)d_work + ((ntg_batch_workspace_bytes(p, batch, o) - 256) & ~(long long)7))};
	NtgWavePlan wp;
	if (wave_takes(p, sp, batch, &wp)) {   // one wavefront per problem
		HIPCHK(ntg_launch_sqp_wave(p->D, p->T, sp, sa, wp));
		return 0;
	}
	HIPCHK(ntg_launch_sqp(p->D, p->T, L, sp, sa));
	return 0;
}

// A receding-horizon run: nsteps times (solve, shift).  The first step runs directly (it may build the preconditioner);
// the (solve, count, shift) sequence of the remaining steps is captured once into a hipGraph and replayed, so that a
// step costs one graph launch instead of three kernel launches from the host loop.
// The multipliers' share of the receding-horizon step (see ntg_amd.h): the estimates of the trajectory rows in the workspace move shift_bp
// breakpoints towards the start of the horizon.
extern "C" int ntg_batch_mpc_shift_multipliers(const ntg_plan *p, int batch, int shift_bp, const ntg_solve_opts *o, void *d_work, long long work_bytes,
                                               void *stream)
{
	if (!p) return fail(NTG_E_BADARG, "null plan");
	if (batch <= 0) return 0;
	const NtgDims &D = p->D;
	if (D.ncnln + D.nI == 0 || D.nnltc == 0 || shift_bp == 0) return 0;
	if (shift_bp < 0 || shift_bp >= D.P) return fail(NTG_E_BADARG, "shift out of range");
	if (!d_work || work_bytes < ntg_batch_workspace_bytes(p, batch, o)) return fail(NTG_E_BADARG, "workspace too small");
	SolveParams sp; int nt;
	resolve_params(p, o, &sp, &nt);
	HIPCHK(hipSetDevice(p->device));
	double *alw = (double *)d_work + hist_doubles(D, batch, sp);   // [batch][2][ncnln + nI]: multipliers, estimates (as ntg_batch_solve lays it out)
	HIPCHK(ntg_launch_mpc_shift_lambda(D, batch, shift_bp, alw, (hipStream_t)stream));
	return 0;
}

extern "C" int ntg_batch_mpc_run(const ntg_plan *p, int batch, int nsteps, int shift_bp, int shift_knots, double *d_x,
                                 double *d_lower, double *d_upper, const ntg_solve_opts *o, int *d_inform, int *d_notconv,
                                 void *d_work, long long work_bytes, void *stream)
{
	if (!p) return fail(NTG_E_BADARG, "null plan");
	if (batch <= 0 || nsteps <= 0) return 0;
	if (!d_x || !d_lower || !d_upper || !d_inform) return fail(NTG_E_BADARG, "null argument");
	if (shift_bp < 0 || shift_bp >= p->D.P || shift_knots < 0) return fail(NTG_E_BADARG, "shift out of range");
	if (p->grid_batch && batch != p->grid_batch) return fail(NTG_E_BADARG, "the plan carries per-problem grids for another batch size");
	HIPCHK(hipSetDevice(p->device));
	hipStream_t st = (hipStream_t)stream, own = nullptr;
	if (!st) { HIPCHK(hipStreamCreateWithFlags(&own, hipStreamNonBlocking)); st = own; }   // the legacy default stream cannot be captured
	// The FIRST solve is cold whatever the caller's options say: a warm start reads the multiplier estimates of the previous solve from
	// d_work (solve_impl.hpp: `if (sp.warm) al_lam[j] = al_t[j]`), and before the first solve the workspace holds nothing of the kind.
	// The estimates it leaves are shifted with the horizon, so the captured / replayed steps start warm as asked.
	ntg_solve_opts cold;
	if (o) { cold = *o; cold.warm_start = 0; }
	auto step = [&](const ntg_solve_opts *so) -> int {
		int rc = ntg_batch_solve(p, batch, d_lower, d_upper, d_x, so, nullptr, d_inform, nullptr, nullptr, nullptr, d_work, work_bytes, st);
		if (rc) return rc;
		if (d_notconv) { hipError_t e = ntg_launch_count_notconv(batch, d_inform, d_notconv, st); if (e != hipSuccess) return fail(NTG_E_HIP, hipGetErrorString(e)); }
		rc = ntg_batch_mpc_shift(p, batch, shift_bp, shift_knots, d_x, d_lower, d_upper, st);
		if (rc) return rc;
		if (o && o->warm_start) rc = ntg_batch_mpc_shift_multipliers(p, batch, shift_bp, o, d_work, work_bytes, st);   // the multipliers travel with the horizon
		return rc;
	};
	int rc = 0;
	if (own) rc = hipDeviceSynchronize() == hipSuccess ? 0 : NTG_E_HIP;   // order after work queued on the default stream
	if (!rc) rc = step(o ? &cold : nullptr);
	if (!rc && nsteps > 1) {
		hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
		hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
		if (e == hipSuccess) {
			rc = step(o);
			hipError_t e2 = hipStreamEndCapture(st, &graph);
			if (!rc && e2 != hipSuccess) rc = fail(NTG_E_HIP, hipGetErrorString(e2));
			if (!rc && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) rc = fail(NTG_E_HIP, "hipGraphInstantiate failed");
			for (int s = 1; !rc && s < nsteps; s++) if (hipGraphLaunch(exec, st) != hipSuccess) rc = fail(NTG_E_HIP, "hipGraphLaunch failed");
			if (exec) (void)hipGraphExecDestroy(exec);
			if (graph) (void)hipGraphDestroy(graph);
		} else {
			for (int s = 1; !rc && s < nsteps; s++) rc = step(o);   // capture unavailable: plain launches
		}
	}
	if (own) { if (hipStreamSynchronize(own) != hipSuccess && !rc) rc = NTG_E_HIP; (void)hipStreamDestroy(own); }
	return rc;
}

extern "C" int ntg_batch_interp(const ntg_plan *p, int batch, const double *d_x, int ntimes, const double *d_times, double *d_z,
                                void *stream)
{
	// documented layout of d_times: [ntimes] on the plan's grid, [batch][ntimes] with per-problem grids
	return ntg_batch_interp_strided(p, batch, d_x, ntimes, d_times, (p && p->grid_batch) ? (long long)ntimes : 0, d_z, stream);
}

extern "C" int ntg_batch_interp_strided(const ntg_plan *p, int batch, const double *d_x, int ntimes, const double *d_times,
                                        long long times_stride, double *d_z, void *stream)
{
	if (!p) return fail(NTG_E_BADARG, "null plan");
	if (batch <= 0 || ntimes <= 0) return 0;
	if (!d_x || !d_times || !d_z) return fail(NTG_E_BADARG, "null argument");
	if (p->grid_batch && batch != p->grid_batch) return fail(NTG_E_BADARG, "the plan carries per-problem grids for another batch size");
	if (times_stride != 0 && times_stride < ntimes) return fail(NTG_E_BADARG, "times_stride must be 0 (one time vector for the batch) or >= ntimes");
	if (times_stride != 0 && !p->grid_batch) return fail(NTG_E_BADARG, "per-problem times need per-problem grids (ntg_plan_set_grids); pass times_stride = 0");
	HIPCHK(hipSetDevice(p->device));
	const NtgDims &D = p->D;
	hipStream_t st = (hipStream_t)stream;
	if (p->grid_batch) {
		// per-problem grids: every problem on its own knots (one basis class), at its own times (row b of d_times at b * times_stride) or,
		// times_stride == 0, all at the same times
		double *d_tb = nullptr; int *d_to = nullptr;
		const size_t per = (size_t)ntimes * D.cls_k[0] * D.cls_d[0];
		hipError_t e2 = hipMallocAsync((void **)&d_tb, (size_t)batch * per * 8, st);
		if (e2 == hipSuccess) e2 = hipMallocAsync((void **)&d_to, (size_t)batch * ntimes * 4, st);
		if (e2 == hipSuccess) e2 = ntg_launch_basis(batch, D.cls_l[0], D.cls_k[0], D.cls_m[0], D.cls_d[0], ntimes, p->d_grid_knots, d_times, D.cls_l[0] + 1, times_stride, d_tb, d_to, st);
		if (e2 == hipSuccess) e2 = ntg_launch_interp(D, batch, ntimes, d_x, d_tb, d_to, nullptr, d_z, st, 1);
		if (d_tb) (void)hipFreeAsync(d_tb, st);
		if (d_to) (void)hipFreeAsync(d_to, st);
		if (e2 != hipSuccess) return fail(NTG_E_HIP, hipGetErrorString(e2));
		return 0;
	}
	// basis of every class at the requested times (the same kernel that builds the collocation tables), stream ordered
	double *d_tblk = nullptr; int *d_toff = nullptr, *d_base = nullptr;
	std::vector<int> base(D.nclass);
	size_t tot = 0;
	for (int c = 0; c < D.nclass; c++) { base[c] = (int)tot; tot += (size_t)ntimes * D.cls_k[c] * D.cls_d[c]; }
	// (the three buffers are released on every path: failures fall through to the frees below)
	hipError_t e = hipMallocAsync((void **)&d_tblk, tot * 8, st);
	if (e == hipSuccess) e = hipMallocAsync((void **)&d_toff, (size_t)D.nclass * ntimes * 4, st);
	if (e == hipSuccess) e = hipMallocAsync((void **)&d_base, (size_t)D.nclass * 4, st);
	if (e == hipSuccess) e = hipMemcpyAsync(d_base, base.data(), (size_t)D.nclass * 4, hipMemcpyHostToDevice, st);
	for (int c = 0; c < D.nclass && e == hipSuccess; c++)
		e = ntg_launch_basis(1, D.cls_l[c], D.cls_k[c], D.cls_m[c], D.cls_d[c], ntimes, p->d_knots[c], d_times, 0, 0,
		                     d_tblk + base[c], d_toff + (size_t)c * ntimes, st);
	if (e == hipSuccess) e = ntg_launch_interp(D, batch, ntimes, d_x, d_tblk, d_toff, d_base, d_z, st);
	const hipError_t es = hipStreamSynchronize(st);   // `base` is read by the async copy above
	if (d_tblk) (void)hipFreeAsync(d_tblk, st);
	if (d_toff) (void)hipFreeAsync(d_toff, st);
	if (d_base) (void)hipFreeAsync(d_base, st);
	HIPCHK(e);
	HIPCHK(es);
	return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-problem grids: the setup phase of ntg() (ntg.c:114-229: CollocMatrix per output, LinearConstraintsMatrix) run for every
// problem of a batch on its own break sequence and breakpoints.  The basis blocks come from basis_kernel (one launch for the whole
// batch); what the host derives from them per problem is small dense algebra over the equality rows ((A A')^-1 is nclin x nclin), done
// here on a few host threads like the one-grid setup of ntg_plan_create.
// ---------------------------------------------------------------------------------------------------------------------
static void dense_AE_pp(const ntg_plan *p, const double *blk, std::vector<double> &AE)   // AE: [mE][nC] row-major
{
	const NtgDims &D = p->D;
	const int n = D.nC, m = D.mE, P = D.P, nz = D.nz;
	AE.assign((size_t)std::max(m, 1) * n, 0.0);
	for (int e = 0; e < m; e++) {
		const int r = p->h_erow[e];
		const double *row; int bp;
		if (r < D.nlic) { row = p->h_lic.data() + (size_t)r * nz; bp = 0; }
		else if (r < D.nlic + D.nltc * P) { const int rr = r - D.nlic; row = p->h_ltc.data() + (size_t)(rr / P) * nz; bp = rr % P; }
		else { row = p->h_lfc.data() + (size_t)(r - D.nlic - D.nltc * P) * nz; bp = P - 1; }
		for (int o = 0; o < D.nout; o++) {
			const int k = D.order[o], d = D.d[o], col0 = D.iC[o] + p->h_off[bp];   // one basis class: offsets of class 0
			for (int q = 0; q < k; q++) {
				double acc = 0.0;
				for (int l = 0; l < d; l++) acc += row[D.iz[o] + l] * blk[((size_t)bp * k + q) * d + l];
				AE[(size_t)e * n + col0 + q] = acc;
			}
		}
	}
}

// preconditioner blocks of one grid (the distinct blocks of build_precond, same order): all[q][spad][nb]
static int precond_blocks_pp(const ntg_plan *p, const double *blk, const double *bps, const std::vector<double> &AE, double *all)
{
	const NtgDims &D = p->D;
	const int n = D.nC, m = D.mE, P = D.P, nb = p->T.n0b_n, spad = p->T.n0b_sp;
	for (int q = 0; q < p->T.n0b_nblk; q++) {
		int o0 = -1;
		for (int o = 0; o < D.nout; o++) if (D.n0_blk[o] == q) { o0 = o; break; }
		if (o0 < 0) return NTG_E_UNSUPPORTED;
		const int k = D.order[o0], d = D.d[o0], c0 = D.iC[o0];
		std::vector<double> H0((size_t)nb * nb, 0.0), Wb;
		auto add = [&](const std::vector<ntg_av> &av, int bp, double w) {
			for (const ntg_av &a : av) {
				if (a.output != o0) continue;
				const int base = p->h_off[bp], r = a.deriv;
				const double *b = blk + (size_t)bp * k * d;
				for (int q1 = 0; q1 < k; q1++) for (int q2 = 0; q2 < k; q2++) H0[(size_t)(base + q1) * nb + base + q2] += w * b[q1 * d + r] * b[q2 * d + r];
			}
		};
		for (int i = 0; i < P; i++) {
			double w = 0.0;
			if (i > 0) w += (bps[i] - bps[i - 1]) / 2;
			if (i < P - 1) w += (bps[i + 1] - bps[i]) / 2;
			if (D.nucf) add(p->tcostav, i, w);
		}
		if (D.nicf) add(p->icostav, 0, 1.0);
		if (D.nfcf) add(p->fcostav, P - 1, 1.0);
		std::vector<int> rsel;
		for (int r = 0; r < m; r++) { bool hit = false; for (int j = 0; j < nb && !hit; j++) if (p->h_AE[(size_t)r * n + c0 + j] != 0.0) hit = true; if (hit) rsel.push_back(r); }
		const int mb = (int)rsel.size();
		std::vector<double> Ab((size_t)std::max(mb, 1) * nb, 0.0);
		for (int i = 0; i < mb; i++) for (int j = 0; j < nb; j++) Ab[(size_t)i * nb + j] = AE[(size_t)rsel[i] * n + c0 + j];
		if (precond_block(H0, Ab, mb, nb, Wb)) return NTG_E_UNSUPPORTED;
		std::copy(Wb.begin(), Wb.end(), all + (size_t)q * spad * nb);
	}
	return 0;
}

extern "C" void ntg_plan_clear_grids(ntg_plan *p)
{
	if (!p || !p->grid_batch) return;
	hipSetDevice(p->device);
	hipDeviceSynchronize();
	for (void *q : p->grid_owned) hipFree(q);
	p->grid_owned.clear();
	p->T = p->T_shared;
	p->grid_batch = 0; p->d_grid_knots = nullptr;
}

extern "C" int ntg_plan_set_grids(ntg_plan *p, int batch, const double *d_knots, const double *d_bps, int with_precond, void *stream)
{
	if (!p) return fail(NTG_E_BADARG, "null plan");
	if (batch <= 0 || !d_knots || !d_bps) return fail(NTG_E_BADARG, "bad argument");
	const NtgDims &D = p->D;
	if (D.family == NTG_FAM_HOST) return fail(NTG_E_UNSUPPORTED, "host-callback plans have one grid");
	if (D.nclass != 1) return fail(NTG_E_UNSUPPORTED, "per-problem grids need one basis class (every output on the same knots / order / multiplicity)");
	// Nonlinear rows are fine: their evaluation reads the same per-problem tables, and the structured Newton mode / the QP-based SQP step get
	// the cost model and the free-output factors of every grid (step 4b, grids.hip grid_nwt_kernel).
	if (D.nI > 0) return fail(NTG_E_UNSUPPORTED, "per-problem grids: plans without linear inequality rows so far");
	HIPCHK(hipSetDevice(p->device));
	ntg_plan_clear_grids(p);
	if (with_precond) {
		std::lock_guard<std::mutex> lk(p->precond_mutex);
		if (!p->precond_ready) { int rc = build_precond(p); if (rc) return rc; }
		if (p->precond_singular || !p->T.n0b) return fail(NTG_E_UNSUPPORTED, "per-problem grids with the preconditioner need its block form (one dense block per output)");
	}
	hipStream_t st = (hipStream_t)stream;
	const int P = D.P, k = D.cls_k[0], d = D.cls_d[0], l = D.cls_l[0], n = D.nC, m = D.mE, nz = D.nz;
	const size_t nblk = (size_t)P * k * d;
	const int row_total = D.row_total, lin_nnz = std::max(D.lin_nnz, 1), sinv_nnz = std::max(D.sinv_nnz, 1), qn = D.q_use ? D.q_nt * D.q_w : 0;
	// batch-shared inputs of the device algebra, uploaded once per plan
	if (!p->d_planoff) {
		if (dev_upload(&p->d_planoff, p->h_off.data(), (size_t)P, p->owned)) return NTG_E_HIP;
		std::vector<double> rows((size_t)std::max(D.nclin, 1) * nz, 0.0);
		if (D.nlic) std::copy(p->h_lic.begin(), p->h_lic.end(), rows.begin());
		if (D.nltc) std::copy(p->h_ltc.begin(), p->h_ltc.end(), rows.begin() + (size_t)D.nlic * nz);
		if (D.nlfc) std::copy(p->h_lfc.begin(), p->h_lfc.end(), rows.begin() + (size_t)(D.nlic + D.nltc) * nz);
		if (dev_upload(&p->d_linrows, rows.data(), rows.size(), p->owned)) return NTG_E_HIP;
		std::vector<int> er(std::max(m, 1), 0);
		for (int e2 = 0; e2 < m; e2++) er[e2] = p->h_erow[e2];
		if (dev_upload(&p->d_erow, er.data(), er.size(), p->owned)) return NTG_E_HIP;
		std::vector<int> r2c(std::max(D.q_nt, 1), 0);
		std::vector<unsigned char> pad((size_t)std::max(qn, 1), 0);
		if (D.q_use) {
			for (int a = 0; a < n; a++) if (p->h_qidx[a] >= 0) r2c[p->h_qidx[a]] = a;
			// ELL padding: entries whose plan value is exactly 0 and that repeat column 0 behind the row's real entries
			for (int t = 0; t < D.q_nt; t++) for (int w2 = 1; w2 < D.q_w; w2++)
				if (p->h_qval[(size_t)t * D.q_w + w2] == 0.0 && p->h_qcol[(size_t)t * D.q_w + w2] == 0) pad[(size_t)t * D.q_w + w2] = 1;
		}
		if (dev_upload(&p->d_qrow2coef, r2c.data(), r2c.size(), p->owned) || dev_upload(&p->d_qpad, pad.data(), pad.size(), p->owned)) return NTG_E_HIP;
	}
	// 1. basis blocks and offsets of every problem: one launch of basis_kernel (bsplvd at every collocation point, colloc.c:95-111)
	double *d_blk = nullptr; int *d_off = nullptr, *d_err = nullptr;
	double *d_rowv = nullptr, *d_bpsc = nullptr, *d_csr = nullptr, *d_csc = nullptr, *d_sinv = nullptr, *d_q = nullptr, *d_n0b = nullptr, *d_knc = nullptr;
	auto fail_free = [&](int code, const std::string &msg) {
		for (void *q : {(void *)d_blk, (void *)d_off, (void *)d_err, (void *)d_rowv, (void *)d_bpsc, (void *)d_csr, (void *)d_csc, (void *)d_sinv, (void *)d_q, (void *)d_n0b, (void *)d_knc}) if (q) hipFree(q);
		return fail(code, msg);
	};
	const size_t n0b_sz = with_precond ? (size_t)p->T.n0b_nblk * p->T.n0b_sp * p->T.n0b_n + 16 : 0;
	if (hipMalloc((void **)&d_blk, (size_t)batch * nblk * 8) != hipSuccess || hipMalloc((void **)&d_off, (size_t)batch * P * 4) != hipSuccess ||
	    hipMalloc((void **)&d_err, 16) != hipSuccess || hipMalloc((void **)&d_rowv, (size_t)batch * row_total * 8) != hipSuccess ||
	    hipMalloc((void **)&d_bpsc, (size_t)batch * P * 8) != hipSuccess || hipMalloc((void **)&d_csr, (size_t)batch * lin_nnz * 8) != hipSuccess ||
	    hipMalloc((void **)&d_csc, (size_t)batch * lin_nnz * 8) != hipSuccess || hipMalloc((void **)&d_sinv, (size_t)batch * sinv_nnz * 8) != hipSuccess ||
	    hipMalloc((void **)&d_q, (size_t)batch * std::max(qn, 1) * 8) != hipSuccess || hipMalloc((void **)&d_knc, (size_t)batch * (l + 1) * 8) != hipSuccess ||
	    (with_precond && hipMalloc((void **)&d_n0b, (size_t)batch * n0b_sz * 8) != hipSuccess)) return fail_free(NTG_E_HIP, "hipMalloc (per-problem grids)");
	hipError_t e = hipMemsetAsync(d_err, 0, 16, st);
	if (e == hipSuccess) e = hipMemsetAsync(d_rowv, 0, (size_t)batch * row_total * 8, st);
	if (e == hipSuccess) e = hipMemcpyAsync(d_bpsc, d_bps, (size_t)batch * P * 8, hipMemcpyDeviceToDevice, st);
	if (e == hipSuccess) e = hipMemcpyAsync(d_knc, d_knots, (size_t)batch * (l + 1) * 8, hipMemcpyDeviceToDevice, st);   // kept: ntg_batch_interp evaluates the basis at other times
	if (e == hipSuccess) e = ntg_launch_basis(batch, l, k, D.cls_m[0], d, P, d_knots, d_bps, l + 1, P, d_blk, d_off, st);
	// 2. channel rows in the kernels' layout + the structure check (every breakpoint in the plan's knot interval); 3. the algebra of the
	//    linear rows -- A_E on the plan's patterns, (A A')^-1, Q -- one wavefront per problem, all on the device (grids.hip)
	if (e == hipSuccess) e = ntg_launch_grid_rows(D, batch, d_blk, d_off, p->d_planoff, d_rowv, d_err, st);
	if (e == hipSuccess && m > 0) {
		NtgGridLin g{d_blk, p->d_linrows, p->d_planoff, p->d_erow, p->T.csr_ptr, p->T.csr_col, p->T.csc_ptr, p->T.csc_row, p->T.sinv_ptr, p->T.sinv_col,
		             p->T.q_col, p->d_qrow2coef, p->d_qpad, d_csr, d_csc, d_sinv, d_q, d_err};
		e = ntg_launch_grid_lin(D, batch, g, st);
	}
	int herr[4] = {0, 0, 0, 0};
	if (e == hipSuccess) e = hipMemcpyAsync(herr, d_err, 12, hipMemcpyDeviceToHost, st);
	if (e == hipSuccess) e = hipStreamSynchronize(st);
	if (e != hipSuccess) return fail_free(NTG_E_HIP, hipGetErrorString(e));
	if (herr[0] == 1) return fail_free(NTG_E_BADARG, "per-problem grid: a breakpoint lies in another knot interval than in the plan's grid (problem " + std::to_string(herr[1]) + ", breakpoint " + std::to_string(herr[2]) + ")");
	if (herr[0] == 2) return fail_free(NTG_E_UNSUPPORTED, "per-problem grid: a linear-constraint entry outside the plan's sparsity pattern (problem " + std::to_string(herr[1]) + ", row " + std::to_string(herr[2]) + ")");
	if (herr[0] == 3) return fail_free(NTG_E_BADARG, "per-problem grid: linear constraint rows are rank deficient (problem " + std::to_string(herr[1]) + ")");
	// 4. the preconditioner blocks of every grid (hessian = 1).  On the device (grids.hip, grid_prec_kernel) when the equality rows that touch
	//    a block pin whole coefficients -- null(A) is then spanned by unit vectors and W0 is the inverse of a principal submatrix of H0;
	//    decided once per plan from the shared grid's A.  Otherwise: dense n_o x n_o algebra (Householder null space) on host threads.
	if (with_precond && p->prec_dev < 0) {
		const int nb = p->T.n0b_n, nq = p->T.n0b_nblk;
		std::vector<int> fidx((size_t)nq * nb, -1), binfo((size_t)nq * 4, 0);
		bool okdev = !getenv("NTG_AMD_HOST_PRECOND");
		int nrmax = 0;
		for (int q = 0; q < nq && okdev; q++) {
			int o0 = -1;
			for (int o = 0; o < D.nout; o++) if (D.n0_blk[o] == q) { o0 = o; break; }
			if (o0 < 0 || D.ncoef[o0] != nb) { okdev = false; break; }
			const int c0 = D.iC[o0];
			std::vector<char> pin(nb, 0);
			int mb = 0, npin = 0;
			for (int r = 0; r < m; r++) {
				double big = 0.0; bool hit = false;
				for (int c = 0; c < n; c++) big = std::max(big, std::fabs(p->h_AE[(size_t)r * n + c]));
				for (int j = 0; j < nb; j++) if (p->h_AE[(size_t)r * n + c0 + j] != 0.0) hit = true;
				if (!hit) continue;
				mb++;
				for (int j = 0; j < nb; j++) if (std::fabs(p->h_AE[(size_t)r * n + c0 + j]) > 1e-10 * big && !pin[j]) { pin[j] = 1; npin++; }
				// a row that also touches another block couples the blocks: not this structure
				for (int c = 0; c < n; c++) if ((c < c0 || c >= c0 + nb) && std::fabs(p->h_AE[(size_t)r * n + c]) > 1e-10 * big) okdev = false;
			}
			if (npin != mb || nb - npin < 1) { okdev = false; break; }
			int cnt = 0;
			for (int j = 0; j < nb; j++) fidx[(size_t)q * nb + j] = pin[j] ? -1 : cnt++;
			binfo[4 * q] = cnt; nrmax = std::max(nrmax, cnt);
			auto mask_of = [&](const std::vector<ntg_av> &av) { int mk = 0; for (const ntg_av &a : av) if (a.output == o0) mk |= 1 << a.deriv; return mk; };
			binfo[4 * q + 1] = D.nucf ? mask_of(p->tcostav) : 0; binfo[4 * q + 2] = D.nicf ? mask_of(p->icostav) : 0; binfo[4 * q + 3] = D.nfcf ? mask_of(p->fcostav) : 0;
		}
		if (okdev && 2 * (size_t)nrmax * (nrmax + 1) * 8 > 160 * 1024) okdev = false;
		if (okdev && (dev_upload(&p->d_pfidx, fidx.data(), fidx.size(), p->owned) || dev_upload(&p->d_pbinfo, binfo.data(), binfo.size(), p->owned))) return fail_free(NTG_E_HIP, "upload (preconditioner tables)");
		p->prec_dev = okdev ? 1 : 0; p->prec_nrmax = nrmax;
	}
	if (with_precond && p->prec_dev == 1) {
		NtgGridPrec g{d_blk, d_bpsc, p->d_planoff, p->d_pfidx, p->d_pbinfo, d_n0b, d_err, p->T.n0b_nblk, p->T.n0b_n, p->T.n0b_sp, (int)n0b_sz, p->prec_nrmax};
		hipError_t e3 = hipMemsetAsync(d_n0b, 0, (size_t)batch * n0b_sz * 8, st);
		if (e3 == hipSuccess) e3 = ntg_launch_grid_prec(D, batch, g, st);
		if (e3 == hipSuccess) e3 = hipMemcpyAsync(herr, d_err, 12, hipMemcpyDeviceToHost, st);
		if (e3 == hipSuccess) e3 = hipStreamSynchronize(st);
		if (e3 != hipSuccess) return fail_free(NTG_E_HIP, hipGetErrorString(e3));
		if (herr[0] == 3) return fail_free(NTG_E_UNSUPPORTED, "per-problem grid: preconditioner block not positive definite (problem " + std::to_string(herr[1]) + ")");
	} else if (with_precond) {
		std::vector<double> hblk((size_t)batch * nblk), hbps((size_t)batch * P), n0bv((size_t)batch * n0b_sz, 0.0);
		if (hipMemcpy(hblk.data(), d_blk, hblk.size() * 8, hipMemcpyDeviceToHost) != hipSuccess ||
		    hipMemcpy(hbps.data(), d_bps, hbps.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail_free(NTG_E_HIP, "reading the basis blocks back failed");
		std::atomic<int> next(0), err(0);
		auto worker = [&]() {
			std::vector<double> AE;
			for (;;) {
				const int b = next.fetch_add(1);
				if (b >= batch || err.load()) break;
				const double *blk = hblk.data() + (size_t)b * nblk;
				dense_AE_pp(p, blk, AE);
				if (precond_blocks_pp(p, blk, hbps.data() + (size_t)b * P, AE, n0bv.data() + (size_t)b * n0b_sz)) { err.store(3); break; }
			}
		};
		const unsigned nthr = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
		std::vector<std::thread> pool;
		for (unsigned t = 0; t + 1 < nthr; t++) pool.emplace_back(worker);
		worker();
		for (auto &th : pool) th.join();
		if (err.load()) return fail_free(NTG_E_UNSUPPORTED, "per-problem grid: preconditioner block not positive definite");
		if (hipMemcpy(d_n0b, n0bv.data(), n0bv.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return fail_free(NTG_E_HIP, "uploading the preconditioner blocks failed");
	}
	// 4b. the structured Newton mode's cost model and free-output factors of every grid (hessian = 2 / 3 then run on per-problem grids like on
	//     the plan's own; the QP-based SQP step's plan tables belong to the plan's grid and stay unused there)
	double *d_k0pp = nullptr, *d_lfpp = nullptr;
	const size_t k0_sz = D.nwt_on ? (size_t)D.nwt_ngrp * D.nwt_ng * (D.nwt_hb + 1) + (D.nwt_tw ? (size_t)D.nwt_ngrp * (16 * D.nwt_jb + 48) * (D.nwt_hb + 1) : 0) : 0;
	const size_t lf_sz = D.nwt_on ? std::max<size_t>((size_t)D.nwt_nfo * D.nwt_ngf * (D.nwt_hbf + 1), 1) : 0;
	if (D.nwt_on) {
		if (hipMalloc((void **)&d_k0pp, (size_t)batch * k0_sz * 8) != hipSuccess || hipMalloc((void **)&d_lfpp, (size_t)batch * lf_sz * 8) != hipSuccess) {
			if (d_k0pp) hipFree(d_k0pp);
			return fail_free(NTG_E_HIP, "hipMalloc (per-problem cost models)");
		}
		NtgGridNwt gn{d_rowv, d_bpsc, p->d_planoff, p->T.nwt_lo, p->T.nwt_hi, d_k0pp, d_lfpp, d_err, (int)k0_sz, (int)lf_sz};
		hipError_t e4 = hipMemsetAsync(d_err, 0, 16, st);
		if (e4 == hipSuccess) e4 = ntg_launch_grid_nwt(D, batch, gn, st);
		if (e4 == hipSuccess) e4 = hipMemcpyAsync(herr, d_err, 12, hipMemcpyDeviceToHost, st);
		if (e4 == hipSuccess) e4 = hipStreamSynchronize(st);
		if (e4 != hipSuccess || herr[0] == 3) {
			hipFree(d_k0pp); hipFree(d_lfpp);
			return e4 != hipSuccess ? fail_free(NTG_E_HIP, hipGetErrorString(e4)) : fail_free(NTG_E_UNSUPPORTED, "per-problem grid: the cost model of a free output is not positive definite (problem " + std::to_string(herr[1]) + ")");
		}
	}
	hipFree(d_off); hipFree(d_err);   // (d_blk is kept: the receding-horizon shift evaluates the whole flag at a breakpoint)
	// 5. the kernels add b * stride to the value pointers (NtgTables::pp_*)
	p->T_shared = p->T;
	for (void *q : {(void *)d_rowv, (void *)d_bpsc, (void *)d_csr, (void *)d_csc, (void *)d_sinv, (void *)d_q, (void *)d_n0b, (void *)d_knc, (void *)d_blk, (void *)d_k0pp, (void *)d_lfpp}) if (q) p->grid_owned.push_back(q);
	p->d_grid_knots = d_knc;
	NtgTables &T = p->T;
	T.rowv = d_rowv; T.pp_rowv = row_total;
	T.blk = d_blk; T.pp_blk = (long long)nblk;   // one basis class: cls_blk[0] == 0
	T.bps = d_bpsc; T.pp_bps = P;
	if (m > 0) { T.csr_val = d_csr; T.csc_val = d_csc; T.pp_lin = lin_nnz; T.sinv_val = d_sinv; T.pp_sinv = sinv_nnz; }
	if (D.q_use) { T.q_val = d_q; T.pp_q = qn; }
	if (with_precond) { T.n0b = d_n0b; T.pp_n0b = (long long)n0b_sz; T.n0 = nullptr; T.n0c = nullptr; }
	if (D.nwt_on) { T.nwt_k0 = d_k0pp; T.pp_k0 = (long long)k0_sz; T.nwt_lf = d_lfpp; T.pp_lf = (long long)lf_sz; }
	p->grid_batch = batch;
	return 0;
}

extern "C" int ntg_batch_kincar_reverse(const ntg_plan *p, int batch, int ntimes, const double *d_z, double wheelbase, int reverse_gear,
                                        double *d_state, void *stream)
{
	if (!p) return fail(NTG_E_BADARG, "null plan");
	if (batch <= 0 || ntimes <= 0) return 0;
	if (!d_z || !d_state) return fail(NTG_E_BADARG, "null argument");
	const NtgDims &D = p->D;
	if ((D.family != NTG_FAM_KINCAR && D.family != NTG_FAM_OBSTACLE) || D.nout % 2 || D.nz != 3 * D.nout)
		return fail(NTG_E_UNSUPPORTED, "kincar_flat_reverse needs a kincar-family plan (two outputs per car, three flag entries per output)");
	if (!(wheelbase > 0.0)) return fail(NTG_E_BADARG, "wheelbase must be positive");
	HIPCHK(hipSetDevice(p->device));
	HIPCHK(ntg_launch_kincar_reverse((long long)batch * ntimes, D.nz, D.nout / 2, wheelbase, reverse_gear, d_z, d_state, (hipStream_t)stream));
	return 0;
}

extern "C" int ntg_basis_batch(int ngrids, int ninterv, int order, int mult, int maxderiv, int nbps, const double *d_knots,
                               const double *d_bps, double *d_blk, int *d_off, void *stream)
{
	if (order < 1 || order > NTG_MAX_ORDER || mult < 0 || mult >= order || maxderiv < 1 || maxderiv > order || ninterv < 1 || nbps < 1)
		return fail(NTG_E_BADARG, "bad spline spec");
	if (ngrids <= 0) return 0;
	if (ngrids > 65535) return fail(NTG_E_BADARG, "at most 65535 grids per call");
	HIPCHK(ntg_launch_basis(ngrids, ninterv, order, mult, maxderiv, nbps, d_knots, d_bps, ninterv + 1, nbps, d_blk, d_off,
	                        (hipStream_t)stream));
	return 0;
}

extern "C" int ntg_debug_layout(const ntg_plan *p, const ntg_solve_opts *o, int *lds_solve, int *lds_eval, int *nt_solve)
{
	if (!p) return NTG_E_BADARG;
	SolveParams sp; int nt;
	resolve_params(p, o, &sp, &nt);
	SmemLayout L; int big;
	solve_layout(p->D, nt, &L, &big, &sp);
	if (getenv("NTG_AMD_DEBUG"))
		fprintf(stderr, "solve layout (big %d): rowv %d colp %d chrow %d off %d bps %d wts %d x %d dfz %d fvals %d red %d dfi %d vecs %d lam %d rho %d oinfo %d ls %d tI %d q_idx %d q_col %d q_val %d csr_ptr %d sinv_val %d total %d | row_total %d col_total %d ntav %d q_nt %d q_w %d lin_nnz %d sinv_nnz %d lin_lds %d\n",
		        big, L.rowv, L.colp, L.chrow, L.off, L.bps, L.wts, L.x, L.dfz, L.fvals, L.red, L.dfi, L.vecs, L.lam, L.rho, L.oinfo, L.ls, L.tI, L.q_idx, L.q_col, L.q_val, L.csr_ptr, L.sinv_val, L.total,
		        p->D.row_total, p->D.col_total, p->D.ntav, p->D.q_nt, p->D.q_w, p->D.lin_nnz, p->D.sinv_nnz, p->D.lin_lds);
	if (lds_solve) *lds_solve = big ? -L.total : L.total;   // negative: only the cross-lane vectors are in LDS
	if (lds_eval) *lds_eval = ntg_make_layout(p->D, auto_threads(p->D), 0, 1).total;
	if (nt_solve) *nt_solve = nt;
	return 0;
}

extern "C" int ntg_batch_mpc_shift(const ntg_plan *p, int batch, int shift_bp, int shift_knots, double *d_x,
                                   double *d_lower, double *d_upper, void *stream)
{
	if (!p) return fail(NTG_E_BADARG, "null plan");
	if (batch <= 0) return 0;
	if (!d_x || !d_lower || !d_upper) return fail(NTG_E_BADARG, "null argument");
	if (shift_bp < 0 || shift_bp >= p->D.P || shift_knots < 0) return fail(NTG_E_BADARG, "shift out of range");
	// the shift re-pins the initial bounds with the basis blocks of the grid in force (T.blk: the plan's, or every problem's own after
	// ntg_plan_set_grids, which keeps them for this)
	if (p->grid_batch && batch != p->grid_batch) return fail(NTG_E_BADARG, "the plan carries per-problem grids for another batch size");
	HIPCHK(hipSetDevice(p->device));
	HIPCHK(ntg_launch_mpc_shift(p->D, p->T, batch, shift_bp, shift_knots, p->d_lic, d_x, d_lower, d_upper, (hipStream_t)stream));
	return 0;
}
