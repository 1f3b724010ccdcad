/*
 * oracle/pgs.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).  Not part of the product path.
 *
 * Restatement of the four PGS (C. de Boor, "A Practical Guide to Splines",
 * public-domain Fortran, fetched unpinned by /root/reference/pgs/Makefile:6,73-77)
 * routines the reference calls: knots, interv, bsplvb, bsplvd.  The Fortran
 * sources are NOT present under /root/reference (pgs/README:1-19), so these are
 * written from the published algorithms (PGS 2nd ed., ch. X "The stable evaluation
 * of B-splines and splines"), with the reference's call sites as the contract:
 *   colloc.c:92-93   side_.m = mult; knots_(knots,&ninterv,&order,augknots,&n)
 *   colloc.c:98      interv_(augknots,&naugknots,&x,&left,&mflag)
 *   colloc.c:99      bsplvd_(augknots,&order,&x,&left,a,dbiatx,&maxderiv)
 *   colloc.c:107     interv_(knots,&nknots,&x,&left,&mflag)
 * Everything is fp64 (reference promotes REAL with -fdefault-real-8, Makefile:19).
 *
 * PARITY STATUS: unpinned by the reference (it ships no tests or vectors for PGS);
 * pinned independently by closed-form end-point values, partition of unity and
 * scipy.interpolate.BSpline (tests/test_oracle_pgs.py).
 *
 * Indices in the public functions below are 1-based exactly like the Fortran
 * (left, mflag semantics), arrays are passed as C pointers to element 1.
 */
#include <stdlib.h>
#include "oracle.h"

/* knots(break,l,kpm,t,n) with COMMON /side/ m: t = break(1) x kpm, every interior
 * break x (kpm-m), break(l+1) x kpm; n = l*(kpm-m)+m.  (call site colloc.c:92-93;
 * cross-check: examples/vanderpol.m:18 augknt(knots,order,order-mult)) */
void orc_knots(const double *brk, int l, int kpm, int m, double *t, int *n)
{
	int k = kpm - m, i, j, pos = 0;
	*n = l * k + m;
	for (j = 0; j < kpm; j++) t[pos++] = brk[0];
	for (i = 1; i < l; i++)
		for (j = 0; j < k; j++) t[pos++] = brk[i];
	for (j = 0; j < kpm; j++) t[pos++] = brk[l];
}

/* interv(xt,lxt,x,left,mflag), 2nd-edition semantics:
 *   x <  xt(1)            : left = 1,  mflag = -1
 *   xt(i) <= x < xt(i+1)  : left = i (the largest such i), mflag = 0
 *   x >= xt(lxt)          : left = max{ i < lxt : xt(i) < xt(lxt) }, mflag = (x==xt(lxt)) ? 0 : 1
 * (the Fortran keeps a SAVEd search start `ilo`; the result does not depend on it) */
void orc_interv(const double *xt, int lxt, double x, int *left, int *mflag)
{
	int i;
	if (x < xt[0]) { *left = 1; *mflag = -1; return; }
	if (x >= xt[lxt - 1]) {
		*mflag = (x == xt[lxt - 1]) ? 0 : 1;
		for (i = lxt - 1; i >= 1; i--)
			if (xt[i - 1] < xt[lxt - 1]) { *left = i; return; }
		*left = 1; *mflag = 1;
		return;
	}
	/* binary search for largest i (1-based) with xt(i) <= x */
	{
		int lo = 1, hi = lxt;              /* xt(lo) <= x < xt(hi) */
		while (hi - lo > 1) {
			int mid = (lo + hi) / 2;
			if (x >= xt[mid - 1]) lo = mid; else hi = mid;
		}
		*left = lo; *mflag = 0;
	}
}

/* bsplvb: values of the jhigh B-splines of order jhigh not vanishing at x,
 * Cox-de Boor triangle.  index==1 starts from order 1, index==2 continues from the
 * order reached in the previous call (state kept in *st, the Fortran's SAVE). */
void orc_bsplvb(const double *t, int jhigh, int index, double x, int left,
                double *biatx, orc_bsplvb_state *st)
{
	int i, jp1;
	double saved, term;
	if (index == 1) {
		st->j = 1;
		biatx[0] = 1.0;
		if (st->j >= jhigh) return;
	}
	do {
		jp1 = st->j + 1;
		st->deltar[st->j - 1] = t[left + st->j - 1] - x;      /* t(left+j) - x   */
		st->deltal[st->j - 1] = x - t[left - st->j];          /* x - t(left+1-j) */
		saved = 0.0;
		for (i = 1; i <= st->j; i++) {
			term = biatx[i - 1] / (st->deltar[i - 1] + st->deltal[jp1 - i - 1]);
			biatx[i - 1] = saved + st->deltar[i - 1] * term;
			saved = st->deltal[jp1 - i - 1] * term;
		}
		biatx[jp1 - 1] = saved;
		st->j = jp1;
	} while (st->j < jhigh);
}

/* bsplvd: dbiatx(i,m) = D^{m-1} B_{left-k+i,k}(x), i=1..k, m=1..nderiv,
 * Fortran column-major dbiatx(k,nderiv): element (i,m) at dbiatx[(m-1)*k+(i-1)].
 * a(k,k) is scratch (column-major). */
void orc_bsplvd(const double *t, int k, double x, int left, double *a,
                double *dbiatx, int nderiv)
{
	orc_bsplvb_state st;
	int mhigh, kp1, ideriv, m, j, jp1mid, i, jlow, il, kp1mm, ld;
	double fkp1mm, factor, sum;
#define A_(r, c) a[((c) - 1) * k + ((r) - 1)]
#define DB_(r, c) dbiatx[((c) - 1) * k + ((r) - 1)]
	mhigh = nderiv < k ? nderiv : k;
	if (mhigh < 1) mhigh = 1;
	kp1 = k + 1;
	orc_bsplvb(t, kp1 - mhigh, 1, x, left, dbiatx, &st);
	if (mhigh == 1) return;
	/* fill column ideriv with the order-(k+1-ideriv) values, raising the order by one
	 * each pass, so that column m ends up holding the order k+1-m values */
	ideriv = mhigh;
	for (m = 2; m <= mhigh; m++) {
		jp1mid = 1;
		for (j = ideriv; j <= k; j++) {
			DB_(j, ideriv) = DB_(jp1mid, 1);
			jp1mid++;
		}
		ideriv--;
		orc_bsplvb(t, kp1 - ideriv, 2, x, left, dbiatx, &st);
	}
	/* a = identity (lower triangle zeroed as needed) */
	jlow = 1;
	for (i = 1; i <= k; i++) {
		for (j = jlow; j <= k; j++) A_(j, i) = 0.0;
		jlow = i;
		A_(i, i) = 1.0;
	}
	/* difference the coefficient table, combine with lower-order values */
	for (m = 2; m <= mhigh; m++) {
		kp1mm = kp1 - m;
		fkp1mm = (double)kp1mm;
		il = left;
		i = k;
		for (ld = 1; ld <= kp1mm; ld++) {
			factor = fkp1mm / (t[il + kp1mm - 1] - t[il - 1]);
			for (j = 1; j <= i; j++)
				A_(i, j) = (A_(i, j) - A_(i - 1, j)) * factor;
			il--;
			i--;
		}
		for (i = 1; i <= k; i++) {
			sum = 0.0;
			jlow = i > m ? i : m;
			for (j = jlow; j <= k; j++)
				sum = A_(j, i) * DB_(j, m) + sum;
			DB_(i, m) = sum;
		}
	}
#undef A_
#undef DB_
}
