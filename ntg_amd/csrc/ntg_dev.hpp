// ntg_dev.hpp -- shared host/device declarations of the MI355X NTG engine (gfx950 only).
//
// Data layout in HBM (all fp64, indices int32):
//   blk   per basis CLASS (outputs with identical knots/order/mult/maxderiv share one table):
//         [bp][q][r] = D^r B_{off+q}(bps[bp])      == reference block[bp].matrix->elements[q][r]
//         (colloc.c:100-101); classes concatenated, class c starts at cls_blk[c]
//   off   [class][bp]  coefficient offset of the block (colloc.c:104-111)
//   aband [nclin][sumk] banded rows of the linear-constraint matrix A (constraints.c:198-261):
//         row r lives at breakpoint rbp[r]; for output o the k_o entries koff[o].. are the
//         columns iC[o]+off[cls[o]][rbp[r]]+q of the dense A the reference builds
//   x, g  [batch][nC]   coefficient vectors, problem-major (coalesced per workgroup)
//   hist  [batch][memcap][2][nC]  quasi-Newton pairs (s_i, u_i = W_i y_i)
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ntg_amd.h"

typedef unsigned long long u64;

struct NtgDims {
	int nout, P, nz, nC, nclin, ncnln, nbounds, sumk;
	int nlic, nltc, nlfc, nnlic, nnltc, nnlfc;
	int nicf, nucf, nfcf;
	int family;
	int order[NTG_MAX_OUT], mult[NTG_MAX_OUT], ninterv[NTG_MAX_OUT], d[NTG_MAX_OUT];
	int ncoef[NTG_MAX_OUT], iC[NTG_MAX_OUT], iz[NTG_MAX_OUT], koff[NTG_MAX_OUT], cls[NTG_MAX_OUT];
	int nclass, blk_total;
	int cls_blk[NTG_MAX_OUT], cls_k[NTG_MAX_OUT], cls_d[NTG_MAX_OUT], cls_l[NTG_MAX_OUT], cls_m[NTG_MAX_OUT];
	u64 icost_mask, tcost_mask, fcost_mask, icon_mask, tcon_mask, fcon_mask;
	// running-cost gradient rows kept in LDS: only the flag entries that can be non-zero
	// (the declared trajectory-cost active variables; all of them for host callbacks)
	int ntav, ntav_cost;   // rows of the weighted-gradient area: all trajectory active variables; those of the cost come first
	signed char tav_row[NTG_MAX_NZ];   // flat flag index -> compact row, or -1
	int uniform;                        // 1: one basis class and equal ncoef for every output
	int mE, nI;                         // linear rows kept by projection (equalities) / handled by the AL loop (declared inequalities)
	int lin_nnz, lin_lds;               // sparse A_E (exact zeros dropped); 1: staged in LDS
	int sinv_nnz;                       // sparse (A A')^-1 (block diagonal when the rows decouple)
	int q_use, q_nt, q_w;               // projector Q = A'(AA')^-1 A kept as ELL over its non-zero rows
	int q_pin;                          // 1: the equality rows pin whole coefficients (as many touched columns as rows: the usual initial / final
	                                    // conditions) -- Q is then the identity on its non-zero rows, and g - Q g just zeroes those entries of g
	// collocation matrix of every ACTIVE (class, derivative) channel, in two sparse forms
	int row_total, col_total;           // doubles in rowv / entries in colv+coli
	int cls_W[NTG_MAX_OUT];             // padded (multiple of 4) support width of the column form, per class
	int cls_nc[NTG_MAX_OUT];            // coefficients per output of the class
	int n0_blk[NTG_MAX_OUT];            // which dense preconditioner block an output uses (NtgTables::n0b)
	int ch_row0[NTG_MAX_ORDER], ch_col0[NTG_MAX_ORDER];   // host copy of class 0's channel offsets (NtgTables::chrow/chcol)
	int tav_rmask;                      // union over outputs of the derivative indices with a cost AV
	// column form by VALUE for the lean evaluation kernel (class 0 only): per active derivative channel [nco][colv_stride]
	// doubles = the W basis values of the column at its consecutive breakpoints, then its first breakpoint (as a double);
	// colv_stride = W + 2: lanes that are consecutive columns read 16-byte words from 16 distinct bank groups
	int colv_total, colv_stride, ch_colv0[NTG_MAX_ORDER];
	// breakpoint groups of class 0 (consecutive breakpoints with the same block offset = one knot interval): ig_n groups,
	// group t = breakpoints [igb[t], igb[t+1]); ig_n = 0 when there are more than 64 groups or a group has more than 6 breakpoints
	int ig_n;
	unsigned short igb[66];
	// structured Newton mode (newton.hpp): coupling groups of nwt_go outputs, nwt_ng free coefficients each (interleaved by
	// output), half bandwidth nwt_hb, nwt_cg constraint flag entries per group; nwt_on = 0: the plan does not qualify
	int nwt_on, nwt_ngrp, nwt_go, nwt_ng, nwt_hb, nwt_cg;
	int nwt_tw, nwt_ja, nwt_jb;      // two-sided factorisation (newton.hpp, nwt_factor_pairs): on, block columns eliminated from the top / from the bottom
	int nwt_nfo, nwt_ngf, nwt_hbf;   // free outputs (in no nonlinear row): count, free coefficients and band half width of each; their factor is NtgTables::nwt_lf
	// breakpoint groups (consecutive breakpoints with the same block offset = one knot interval), how many consecutive
	// groups overlap a coefficient (colours of the assembly), and the constraint flag entries of a group packed one byte
	// each: (output within the group) << 4 | derivative
	int nwt_nint, nwt_cover;
	int nwt_tab;   // NtgTables::nwt_tu / nwt_g are there
	u64 nwt_upack;
	int nwt_clo, nwt_chi;               // the free local coefficients of every output: [clo, chi); free index p = (cl - clo) go + o
};

struct NtgTables {
	const double *bps;     // [P]
	const double *blk;     // [blk_total]
	const int *off;        // [nclass][P]
	const double *aband;   // [nclin][sumk]
	const int *rbp;        // [nclin]
	const double *sinv;    // [nclin][nclin]  (A A')^-1
	// preconditioner W0 = Z(Z'H0Z)^-1 Z' as ELL, s-major ([s][nC]) so that lanes with consecutive rows
	// read consecutive words; exact zeros dropped (block diagonal when outputs decouple); nullptr: none
	const double *n0; const unsigned short *n0c; int n0_w;
	// ... or, when W0 is block diagonal by output with equal block sizes: the distinct blocks, each s-major [s][row]
	// (see apply_n0_block); NtgDims::n0_blk maps outputs to blocks
	const double *n0b; int n0b_n, n0b_sp, n0b_nblk;   // rows per block, rows padded to 16 (zeros), distinct blocks
	// sparse A: CSR (rows) and CSC (columns)
	const int *csr_ptr, *csr_col; const double *csr_val;
	const int *csc_ptr, *csc_row; const double *csc_val;
	const int *sinv_ptr, *sinv_col; const double *sinv_val;   // CSR of (A A')^-1
	// rowv[chrow + q*P + bp]  = D^r B_{off(bp)+q}(bps[bp])      (by breakpoint: Z = M C, Jacobian rows)
	// colp[chcol + cl*WW ..] = column cl of the same matrix (its non-zeros sit at consecutive breakpoints): word 0 = the
	// first breakpoint i0, then W 16-bit value indices (q*P+i relative to chrow, two per word; padding = k*P, a stored
	// zero); the s-th entry multiplies the weighted gradient at breakpoint i0+s.  WW = colp_words(W) words per column.
	// chrow/chcol[class*NTG_MAX_ORDER + r] = channel offsets, -1 when no active variable uses D^r
	const double *rowv; const unsigned int *colp; const int *chrow, *chcol;
	const double *colv;    // see NtgDims::colv_total
	// Per-problem grids (ntg_plan_set_grids): the index tables above stay shared, the VALUES become per problem.  Problem b reads
	// rowv + b pp_rowv, bps + b pp_bps, csr_val + b pp_lin, csc_val + b pp_lin, sinv_val + b pp_sinv, q_val + b pp_q, n0b + b pp_n0b
	// (strides in elements; all 0 = one shared grid).
	long long pp_rowv, pp_bps, pp_lin, pp_sinv, pp_q, pp_n0b;
	long long pp_blk;   // ... and blk + b pp_blk (the full basis blocks: the receding-horizon shift evaluates the whole flag)
	long long pp_k0, pp_lf;   // ... nwt_k0 + b pp_k0, nwt_lf + b pp_lf: the structured Newton mode's cost model and free-output factors of every grid (grids.hip, grid_nwt_kernel)
	// linear rows: erow[mE] = original row of equality e; rowmap[nclin] = e, or -(j+1) for inequality j; linflag[slot]
	// = 1 for slots declared as inequalities; inequality rows as CSR (by row) and CSC (by coefficient)
	const int *erow, *rowmap, *linflag, *irow;
	const int *icsr_ptr, *icsr_col, *icsc_ptr, *icsc_row; const double *icsr_val, *icsc_val;
	// structured Newton mode: nwt_map[g][p] = coefficient of free entry p of group g; nwt_pos[c] = g * nwt_ng + p, or -1 for a
	// pinned coefficient; nwt_k0 = cost-model band [g][p][hb+1]; nwt_lo/hi[cl] = breakpoints [lo, hi) in whose block cl lies
	const int *nwt_map, *nwt_pos; const double *nwt_k0; const short *nwt_lo, *nwt_hi;
	// QP-based SQP step in the regime without constraint curvature (the model is the cost model K0, the same for every problem and major):
	// nwt_tu[i][u][p] = (K0^-1 M_i' e_u)[p], the column of W for flag entry u of a group at breakpoint i; nwt_g[k][i][v][u] = e_v' M_k K0^-1 M_i' e_u.
	// A slot's column W J' and its J U over all rows are then short combinations of table rows instead of a band solve and a breakpoint pass
	// (NtgDims::nwt_tab = 1: every coupling group has the same cost model, tables built with the plan)
	const double *nwt_tu, *nwt_g;
	const double *nwt_lf;   // [nwt_nfo][nwt_ngf][nwt_hbf + 1]: band Cholesky factor (diagonal inverted) of the free outputs' cost model, built with the plan
	const short *q_idx;    // [nC] row of coefficient c in the compact Q, or -1
	const unsigned char *q_pinned;   // [nC] 1: coefficient c is pinned by the equality rows (NtgDims::q_pin plans only)
	const int *q_col;      // [q_nt][q_w]
	const double *q_val;   // [q_nt][q_w], zero padded
};

// words per column of the column form (see NtgTables::colp)
__host__ __device__ constexpr int colp_words(int W) { return (W / 2 + 1 + 3) & ~3; }
// zero doubles kept after the last weighted-gradient row: a column's W reads start at its first breakpoint and may run past P
__host__ __device__ inline int ntg_dfz_tail(const NtgDims &D) { int w = 16; for (int c = 0; c < D.nclass; c++) w = D.cls_W[c] > w ? D.cls_W[c] : w; return w; }

// byte offsets into dynamic LDS, computed on the host (kernels.hip: make_layout)
#define NTG_HRC_LDS 64   // quasi-Newton memories up to this many pairs keep the pair scalars (rho, c2) in LDS, when that costs no residency

struct SmemLayout {
	int rowv, colp, chrow, chcol, off, bps, wts, x, dfz, fvals, red, dfi, dff, vecs, lam, rho, c2;
	int csr_ptr, csr_col, csr_val, csc_ptr, csc_row, csc_val, sinv_ptr, sinv_col, sinv_val, oinfo, tavrow, tcomp, q_idx, q_col, q_val, ls, tI, total;
	int with_lin;   // the linear-constraint operator (projector / CSR tables) is staged: solve layouts only, the evaluation never applies it
	int tav_rows;   // active-variable rows the cost pass may touch: all (solve) or the cost's only (evaluation: the others are not even allocated)
	int dfz_rows;   // rows of the weighted-gradient area: every active variable (solve) or the cost's / one constraint chunk's (evaluation)
	int hrc_n;   // pairs whose scalars (rho, c2) live in LDS at L.rho (0: they travel with the pair in HBM)
	int nwt_y;   // structured Newton mode: byte offset (inside the dfz area, which is idle between evaluations) of the solve vectors; panels follow
	int emit;    // evaluation layouts with trajectory constraint rows: byte offset of the row emission's per-lane decode (eval_constraints), -1: none
};

struct SolveParams {
	int itlim, memcap, ls_maxfev, hessian, fixed_iters;
	int stamps;   // diagnostic: clambda[b][0..7] receives per-phase cycle counts instead of multipliers
	int warm;     // augmented-Lagrangian rows: start from the multiplier estimates the previous solve left in the workspace
	double sr, steplimit, ls_mu, ls_eta;
};

// launcher arguments (host side)
struct EvalArgs { int nt, grid, ncu, batch, mode; const double *x; double *f, *g, *c, *jb, *cj; hipStream_t st; };
struct SqpArgs {
	int nt, big, batch; const double *lo, *up; double *x, *obj; int *inf, *it, *nf; double *cl, *hist, *alw, *vecw, *nwtw; hipStream_t st;
	unsigned int *counter;   // problem queue of the wave kernel (4 bytes inside the workspace)
};

