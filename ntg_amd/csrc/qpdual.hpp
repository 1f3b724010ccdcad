// qpdual.hpp -- the dual active-set solve of the QP-based SQP step (ntg_solve_opts.hessian = 3), scalar code shared
// by the device kernel (one lane per coupling group runs it on that group's slots in LDS) and by the host unit test
// (tests/drivers/qpdual_drv.cpp).
//
// Role in the reference: the QP subproblem NPSOL solves with the Jacobian ntg() hands it (ntg.c:217-220,250-253; constraints.c:120-162):
//     min 1/2 p'K p + g'p   s.t.  bl - c <= J p <= bu - c      (A_E p = 0 is built into W = Z (Z'KZ)^-1 Z')
// through its dual in the multipliers of the rows in the working set ("slots"):
//     min 1/2 nu'H nu + q'nu,  nu >= 0,   H = D S D + diag(del),  S = J W J' (the dense J K^-1 J' block of the slots),  q = D (J W g + r) - del nu0,
// D = +-1 (upper / lower bound; +1 for an equality row, whose variable is free), r = bound - c, del_a = 1e-10 S_aa (a proximal term
// around the multipliers nu0 of the previous major iteration: nearly dependent rows of adjacent breakpoints stay factorable, a fixed
// point is not moved).  Lawson & Hanson's finite active-set method carried over to a positive definite H: the kernel adds the most
// violated row (a slot enters the passive set), this routine re-solves on the passive set by Cholesky and steps back to the first
// sign change (slots leave) until the multipliers of all passive slots are positive.
#pragma once
#include <math.h>
#ifndef NTG_HD
#ifdef __HIPCC__
#define NTG_HD __host__ __device__ __forceinline__
#else
#define NTG_HD inline
#endif
#endif

#define NTG_QP_MAXA 32                 // largest number of slots (rows in the working set of one major iteration) of a coupling group
#define NTG_QP_MAXCG 6                 // constraint flag entries per group (Family::CG)
#define NTG_QP_RED 128                 // doubles of the entering-row search's scratch: [waves <= 8][row functions <= 8][2]
// slots per coupling group of a plan: 32 where one group has the LDS of a large workgroup to itself (quadrotor: rows stay active along arcs of the
// trajectory), 16 for several groups (manipulator: a couple of rows per arm) and for the small workgroups that share a CU (obstacle class)
NTG_HD int ntg_qp_maxa(int ngrp, int nthreads) { return (ngrp == 1 && nthreads >= 256) ? 32 : 16; }
// doubles of LDS per coupling group: S and its factor (packed lower triangles), six vectors, the slots' derivative rows, the integer state
NTG_HD int ntg_qp_doubles(int maxa) { return 2 * (maxa * (maxa + 1) / 2) + 6 * maxa + maxa * NTG_QP_MAXCG + 2 + (5 * maxa + 1) / 2 + 2; }

// view of one group's slots (DP / IP: address-space qualified pointers on the device, plain on the host)
template <class DP, class IP>
struct QpSlotsT {
	DP S, H, nu, nu0, z, jwg, rr, sd, ar;
	IP ns, row, sgn, inP, flag, tab, pl;   // tab[a] = 1: the slot's column comes from the plan's tables (not stored); pl: the passive slots in order (device solve)
	NTG_HD QpSlotsT(DP base, int maxa)
	{
		const int tri = maxa * (maxa + 1) / 2;
		S = base; H = S + tri; nu = H + tri; nu0 = nu + maxa; z = nu0 + maxa; jwg = z + maxa;
		rr = jwg + maxa; sd = rr + maxa; ar = sd + maxa;
		IP ib = (IP)(ar + maxa * NTG_QP_MAXCG);
		ns = ib; flag = ib + 1; row = ib + 4; sgn = row + maxa; inP = sgn + maxa; tab = inP + maxa; pl = tab + maxa;   // 4 + 5 maxa ints
	}
};
#define NTG_QP_TR(a, b) ((a) * ((a) + 1) / 2 + (b))   // a >= b

// Solve on the passive set with the ratio test (the inner loop of Lawson & Hanson).  On return nu >= 0 on the passive slots, 0 elsewhere;
// slots whose multiplier reached zero have left (inP = 0).  Returns the number of passive-set solves.
template <class Q>
NTG_HD int qp_passive_solve(Q &s)
{
	const int ns = *s.ns;
	int P[NTG_QP_MAXA], np = 0, solves = 0;
	for (int a = 0; a < ns; a++) if (s.inP[a]) P[np++] = a;
	for (int round = 0; np > 0 && round < 3 * NTG_QP_MAXA + 8; round++) {   // (bounded: see the device routine)
		// H_PP = D S D + diag(del), right-hand side -q_P
		int ok = 0;
		for (int attempt = 0; attempt < 2 && !ok; attempt++) {
			const double shift = attempt ? 1e4 : 1.0;
			for (int k = 0; k < np; k++) {
				const int a = P[k]; const double sa = s.sgn[a] < 0 ? -1.0 : 1.0;
				for (int l = 0; l <= k; l++) {
					const int b = P[l]; const double sb = s.sgn[b] < 0 ? -1.0 : 1.0;
					double h = sa * sb * s.S[a >= b ? NTG_QP_TR(a, b) : NTG_QP_TR(b, a)];
					if (k == l) h += shift * 1e-10 * s.S[NTG_QP_TR(a, a)];
					s.H[NTG_QP_TR(k, l)] = h;
				}
			}
			ok = 1;   // packed Cholesky, in place
			for (int k = 0; k < np && ok; k++) {
				for (int l = 0; l <= k; l++) {
					double v = s.H[NTG_QP_TR(k, l)];
					for (int t = 0; t < l; t++) v -= s.H[NTG_QP_TR(k, t)] * s.H[NTG_QP_TR(l, t)];
					if (l == k) { if (!(v > 0.0)) { ok = 0; break; } s.H[NTG_QP_TR(k, k)] = sqrt(v); }
					else s.H[NTG_QP_TR(k, l)] = v / s.H[NTG_QP_TR(l, l)];
				}
			}
		}
		if (!ok) {   // not factorable even with the larger shift: give the working set up (the step is then the unconstrained one)
			for (int k = 0; k < np; k++) { s.nu[P[k]] = 0.0; s.inP[P[k]] = 0; }
			np = 0;
			break;
		}
		for (int k = 0; k < np; k++) {
			const int a = P[k]; const double sa = s.sgn[a] < 0 ? -1.0 : 1.0;
			double v = -(sa * (s.jwg[a] + s.rr[a]) - 1e-10 * s.S[NTG_QP_TR(a, a)] * s.nu0[a]);
			for (int t = 0; t < k; t++) v -= s.H[NTG_QP_TR(k, t)] * s.z[t];
			s.z[k] = v / s.H[NTG_QP_TR(k, k)];
		}
		for (int k = np - 1; k >= 0; k--) {
			double v = s.z[k];
			for (int t = k + 1; t < np; t++) v -= s.H[NTG_QP_TR(t, k)] * s.z[t];
			s.z[k] = v / s.H[NTG_QP_TR(k, k)];
		}
		solves++;
		double amin = 1.0; int neg = 0;
		for (int k = 0; k < np; k++) {
			const int a = P[k];
			if (s.sgn[a] != 0 && !(s.z[k] > 0.0)) { const double al = s.nu[a] / (s.nu[a] - s.z[k]); neg = 1; if (al < amin) amin = al; }
		}
		if (!neg) { for (int k = 0; k < np; k++) s.nu[P[k]] = s.z[k]; break; }
		if (!(amin >= 0.0)) amin = 0.0;
		int l = 0;
		for (int k = 0; k < np; k++) {
			const int a = P[k];
			s.nu[a] += amin * (s.z[k] - s.nu[a]);
			if (s.sgn[a] != 0 && !(s.nu[a] > 1e-14 * (1.0 + fabs(s.z[k])))) { s.nu[a] = 0.0; s.inP[a] = 0; }
			else P[l++] = a;
		}
		np = l;
	}
	return solves;
}
