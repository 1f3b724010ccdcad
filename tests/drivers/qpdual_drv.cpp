// Host unit check of csrc/qpdual.hpp (the scalar code the device runs on one lane per coupling group): random working sets
//   S = J J' (m <= NTG_QP_MAXA rows of n columns, some nearly dependent), random J W g and r, bound sides +-1;
// the kernel's outer loop restated here (most violated row enters, qp_passive_solve re-solves) must end at the KKT point of
//   min 1/2 nu'H nu + q'nu, nu >= 0:   nu >= 0,  H nu + q >= -tol,  nu_a (H nu + q)_a = 0.
// usage: qpdual_drv [seeds]
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include "../../ntg_amd/csrc/qpdual.hpp"

int main(int argc, char **argv)
{
	const int seeds = argc > 1 ? atoi(argv[1]) : 200;
	int bad = 0, maxit = 0;
	for (int seed = 0; seed < seeds; seed++) {
		std::mt19937_64 rng(seed);
		std::normal_distribution<double> N(0.0, 1.0);
		const int m = 2 + seed % (NTG_QP_MAXA - 1), n = 24;
		std::vector<double> J((size_t)m * n), base(ntg_qp_doubles(NTG_QP_MAXA), 0.0);
		for (auto &v : J) v = N(rng);
		if (seed % 3 == 0) for (int a = 1; a < m; a += 2) for (int c = 0; c < n; c++) J[(size_t)a * n + c] = J[(size_t)(a - 1) * n + c] + 1e-6 * N(rng);   // adjacent breakpoints
		QpSlotsT<double *, int *> s(base.data(), NTG_QP_MAXA);
		*s.ns = m;
		for (int a = 0; a < m; a++) {
			for (int b = 0; b <= a; b++) { double v = 0.0; for (int c = 0; c < n; c++) v += J[(size_t)a * n + c] * J[(size_t)b * n + c]; s.S[NTG_QP_TR(a, b)] = v; }
			s.jwg[a] = N(rng); s.rr[a] = N(rng); s.sgn[a] = (seed + a) % 4 == 0 ? -1 : 1; s.inP[a] = 0; s.nu[a] = 0.0; s.nu0[a] = 0.0; s.row[a] = a;
		}
		auto H = [&](int a, int b) { const double sa = s.sgn[a] < 0 ? -1.0 : 1.0, sb = s.sgn[b] < 0 ? -1.0 : 1.0; return sa * sb * s.S[a >= b ? NTG_QP_TR(a, b) : NTG_QP_TR(b, a)] + (a == b ? 1e-10 * s.S[NTG_QP_TR(a, a)] : 0.0); };
		auto q = [&](int a) { return (s.sgn[a] < 0 ? -1.0 : 1.0) * (s.jwg[a] + s.rr[a]); };
		int it = 0;
		for (; it < 10 * m + 10; it++) {
			int best = -1; double wb = 1e-9;
			for (int a = 0; a < m; a++) if (!s.inP[a]) { double w = -q(a); for (int b = 0; b < m; b++) w -= H(a, b) * s.nu[b]; if (w > wb) { wb = w; best = a; } }
			if (best < 0) break;
			s.inP[best] = 1;
			qp_passive_solve(s);
		}
		if (it > maxit) maxit = it;
		double worst = 0.0;
		for (int a = 0; a < m; a++) {
			// (residuals relative to the size of the terms they are the sum of: nearly dependent rows carry multipliers of 1e6)
			double grad = q(a), mag = 1.0 + fabs(q(a)); for (int b = 0; b < m; b++) { grad += H(a, b) * s.nu[b]; mag += fabs(H(a, b) * s.nu[b]); }
			if (s.nu[a] < 0.0) worst = fmax(worst, 1.0);
			if (grad < -1e-8 * mag) worst = fmax(worst, -grad / mag);
			if (s.nu[a] > 0.0) worst = fmax(worst, fabs(grad) / mag);
		}
		if (!(worst <= 1e-9) || it >= 10 * m + 10) { printf("seed %d m %d: KKT residual %.3e after %d iterations\n", seed, m, worst, it); bad++; }
	}
	printf("%d seeds, %d failures, at most %d outer iterations\n", seeds, bad, maxit);
	return bad ? 1 : 0;
}
