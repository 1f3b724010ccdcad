// linesearch.hpp -- strong-Wolfe line search as a reverse-communication state machine, usable
// from the device SQP kernel (every lane runs the same scalar code on broadcast values, so
// control flow stays wave-uniform) and from the host driver of the ntg() drop-in.
//
// Role in the reference: NPSOL's line search inside npsol_ (ntg.c:250); NPSOL is absent from
// the reference tree, so this follows its published contract (sufficient decrease mu, "line
// search tolerance" eta, step limit) with bracketing/zoom and safeguarded cubic interpolation
// (Nocedal & Wright alg. 3.5/3.6).  DESIGN.md §4 is the normative statement.
#pragma once
#include <math.h>
#ifndef NTG_HD
#ifdef __HIPCC__
#define NTG_HD __host__ __device__ __forceinline__
#else
#define NTG_HD inline
#endif
#endif

struct LineSearch {
	double phi0, dphi0, mu, eta, amax;
	double a_prev, phi_prev, dphi_prev;
	double a_lo, phi_lo, dphi_lo, a_hi, phi_hi, dphi_hi;
	double a;
	int stage, nfev, maxfev;

	static NTG_HD double cubic_min(double a0, double f0, double g0, double a1, double f1, double g1)
	{
		double d1 = g0 + g1 - 3.0 * (f0 - f1) / (a0 - a1);
		double disc = d1 * d1 - g0 * g1;
		if (!(disc >= 0.0)) return NAN;
		double d2 = sqrt(disc);
		if (a1 < a0) d2 = -d2;
		double den = g1 - g0 + 2.0 * d2;
		if (den == 0.0 || !isfinite(den)) return NAN;
		return a1 - (a1 - a0) * (g1 + d2 - d1) / den;
	}
	NTG_HD void init(double phi0_, double dphi0_, double a1, double amax_, double mu_, double eta_, int maxfev_)
	{
		phi0 = phi0_; dphi0 = dphi0_; mu = mu_; eta = eta_; amax = amax_;
		stage = 0; nfev = 0; maxfev = maxfev_;
		a_prev = 0.0; phi_prev = phi0_; dphi_prev = dphi0_;
		a_lo = a_hi = phi_lo = phi_hi = dphi_lo = dphi_hi = 0.0;
		a = a1;
	}
#ifdef __HIPCC__
	// The wave kernel keeps the state in registers: every lane runs the same scalar code on the same values.  Passing the fields that
	// step() rewrites through v_readfirstlane tells the compiler so -- they then live in scalar registers (or their spill lanes), and
	// every branch of step() is a scalar branch instead of an exec-mask region.
	__device__ __forceinline__ static double uni_(double v)
	{
		return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
	}
	__device__ __forceinline__ void make_uniform()
	{
		a_prev = uni_(a_prev); phi_prev = uni_(phi_prev); dphi_prev = uni_(dphi_prev);
		a_lo = uni_(a_lo); phi_lo = uni_(phi_lo); dphi_lo = uni_(dphi_lo);
		a_hi = uni_(a_hi); phi_hi = uni_(phi_hi); dphi_hi = uni_(dphi_hi);
		a = uni_(a);
		stage = __builtin_amdgcn_readfirstlane(stage); nfev = __builtin_amdgcn_readfirstlane(nfev);
	}
#endif
	NTG_HD double zoom_trial() const
	{
		double lo = a_lo < a_hi ? a_lo : a_hi, hi = a_lo < a_hi ? a_hi : a_lo;
		double w = hi - lo, c = cubic_min(a_lo, phi_lo, dphi_lo, a_hi, phi_hi, dphi_hi);
		if (!isfinite(c)) return 0.5 * (lo + hi);
		if (c < lo + 1e-5 * w) c = lo + 1e-5 * w;
		if (c > hi - 1e-5 * w) c = hi - 1e-5 * w;
		return c;
	}
	// feed phi(a), phi'(a) of the trial just evaluated.
	// 0: evaluate at `a` next; 1: accept the point just evaluated; 2: evaluate at `a` and
	// accept unconditionally; -1: failure.
	NTG_HD int step(double phi, double dphi)
	{
		const double at = a;
		const bool armijo = (phi <= phi0 + mu * at * dphi0);
		nfev++;
		if (stage == 0) {
			if (!armijo || (nfev > 1 && !(phi < phi_prev))) {
				a_lo = a_prev; phi_lo = phi_prev; dphi_lo = dphi_prev;
				a_hi = at; phi_hi = phi; dphi_hi = dphi;
				stage = 1;
			} else if (fabs(dphi) <= -eta * dphi0) {
				return 1;
			} else if (dphi >= 0.0) {
				a_lo = at; phi_lo = phi; dphi_lo = dphi;
				a_hi = a_prev; phi_hi = phi_prev; dphi_hi = dphi_prev;
				stage = 1;
			} else {
				if (at >= amax || nfev >= maxfev) return 1;
				double c = cubic_min(a_prev, phi_prev, dphi_prev, at, phi, dphi), an;
				if (!isfinite(c) || c < 1.1 * at) an = 4.0 * at;
				else an = c > 100.0 * at ? 100.0 * at : c;
				if (an > amax) an = amax;
				a_prev = at; phi_prev = phi; dphi_prev = dphi;
				a = an;
				return 0;
			}
		} else {
			if (!armijo || !(phi < phi_lo)) {
				a_hi = at; phi_hi = phi; dphi_hi = dphi;
			} else {
				if (fabs(dphi) <= -eta * dphi0) return 1;
				if (dphi * (a_hi - a_lo) >= 0.0) { a_hi = a_lo; phi_hi = phi_lo; dphi_hi = dphi_lo; }
				a_lo = at; phi_lo = phi; dphi_lo = dphi;
			}
		}
		if (nfev >= maxfev || fabs(a_hi - a_lo) <= 1e-14 * fmax(fabs(a_hi), fabs(a_lo))) {
			if (!(a_lo > 0.0)) return -1;
			if (a_lo == at) return 1;
			a = a_lo;
			return 2;
		}
		a = zoom_trial();
		return 0;
	}
};
