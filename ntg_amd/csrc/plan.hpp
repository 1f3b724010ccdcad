// plan.hpp -- the plan object behind include/ntg_amd.h and the launcher prototypes.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <mutex>
#include "ntg_dev.hpp"

struct ntg_plan {
	int device = 0;
	int *d_pfidx = nullptr, *d_pbinfo = nullptr; int prec_dev = -1, prec_nrmax = 0;   // device path of the per-problem preconditioner blocks: tables, applicability (-1: not decided)
	double *d_grid_knots = nullptr;   // per-problem grids: device copy of the knots [grid_batch][ninterv + 1] (ntg_batch_interp)
	mutable int ncu = 0;                        // compute units of the device (queried on first use)
	NtgDims D;
	NtgTables T;
	bool lin_ok = true;
	std::mutex precond_mutex;                   // guards the lazy build of the preconditioner
	bool precond_ready = false;                 // W0 tables (T.n0 or T.n0b) built
	bool precond_singular = false;              // the cost Hessian model is singular on null(A_E): hessian = 1 falls back to the identity
	std::vector<void *> owned;                  // device allocations
	std::vector<std::vector<double>> h_knots;   // host mirrors of the setup tables
	std::vector<double> h_bps, h_blk, h_aband, h_Adense, h_AE;   // h_Adense: all linear rows; h_AE: the equality rows
	std::vector<int> h_off, h_rbp, class_rep;
	std::vector<ntg_av> icostav, tcostav, fcostav;
	// host copies of the shared index tables (per-problem grids recompute the values behind them)
	std::vector<int> h_chrow, h_csr_ptr, h_csr_col, h_csc_ptr, h_csc_row, h_sinv_ptr, h_sinv_col, h_erow, h_qcol;
	std::vector<short> h_qidx;
	std::vector<double> h_lic, h_ltc, h_lfc;    // the user's linear rows [n][nz]
	// per-problem grids (ntg_plan_set_grids): number of problems they were set for (0: shared grid), their device arrays
	int grid_batch = 0;
	std::vector<void *> grid_owned;
	NtgTables T_shared;                         // the tables of the shared grid, restored by ntg_plan_clear_grids
	double *d_lic = nullptr;                    // [nlic][nz] kept for the receding-horizon shift
	// per-problem grids on the device (grids.hip): the user's linear rows stacked [nclin][nz], row -> coefficient map and padding mask of
	// the projector's ELL pattern, the plan's block offsets; the plan's own projector values (host) to tell padding from pattern
	double *d_linrows = nullptr; int *d_qrow2coef = nullptr, *d_planoff = nullptr, *d_erow = nullptr; unsigned char *d_qpad = nullptr;
	std::vector<double> h_qval;
	std::vector<double *> d_knots;              // break sequence of every basis class (ntg_batch_interp)
};

void ntg_plan_dense_A(const ntg_plan *p, double *A);

// LDS carve-up: tables + (with_x) the coefficient vector + nvec further vectors of nC doubles.
//   eval_kernel   nvec 0 (the gradient is assembled into the x buffer once x is no longer needed)
//   host path     nvec 1
//   sqp_kernel    nvec 5 (xt, gp, gp+, d, g) with x -- or, BIG, nvec 1 (xt) without x: the rest lives in HBM/L2
SmemLayout ntg_make_layout(const NtgDims &D, int nthreads, int nvec, int with_x, int hrc_pairs = 0, int qp = 0);
hipError_t ntg_launch_eval(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const EvalArgs &a);
hipError_t ntg_launch_sqp(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const SolveParams &sp, const SqpArgs &a);
// one wavefront per problem (solve_wave.hpp, fam_kincar_wave.hip): does it take this solve, and its launch shape / HBM workspace
struct NtgWavePlan { int fat, nwv, grid, cap, hbm_slots, nlds, ppg, noagpr; size_t lds, hist_doubles; };   // ppg: per-problem grids (wave-private value tables)
bool ntg_wave_plan(const NtgDims &D, const NtgTables &T, const SolveParams &sp, int batch, int ncu, NtgWavePlan *w);
hipError_t ntg_launch_sqp_wave(const NtgDims &D, const NtgTables &T, const SolveParams &sp, const SqpArgs &a, const NtgWavePlan &w);
hipError_t ntg_launch_basis(int ngrids, int l, int k, int m, int d, int P, const double *knots, const double *bps,
                            long long knots_stride, long long bps_stride, double *blk, int *off, hipStream_t st);
hipError_t ntg_launch_interp(const NtgDims &D, int batch, int ntimes, const double *x, const double *tblk, const int *toff,
                             const int *tblk_base, double *z, hipStream_t st, int pp = 0);
hipError_t ntg_launch_kincar_reverse(long long nsamp, int nz, int ncars, double wheelbase, int reverse_gear, const double *z, double *out, hipStream_t st);
hipError_t ntg_launch_count_notconv(int batch, const int *inform, int *count, hipStream_t st);
hipError_t ntg_launch_linrows(const NtgDims &D, const NtgTables &T, const double *lic, const double *ltc,
                              const double *lfc, double *aband, int *rbp, hipStream_t st);
hipError_t ntg_launch_bounds(const NtgDims &D, int batch, const double *lo, const double *up, double *bl, double *bu,
                             hipStream_t st);
hipError_t ntg_launch_hostz(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const double *x, u64 mI,
                            u64 mT, u64 mF, double *Z, hipStream_t st);
hipError_t ntg_launch_hostcost(const NtgDims &D, const NtgTables &T, const SmemLayout &L, const double *fT,
                               const double *dfT, const double *fdI, const double *fdF, double *F, double *g,
                               hipStream_t st);
hipError_t ntg_launch_hostcon(const NtgDims &D, const NtgTables &T, const double *dc, double *jband, double *cjac,
                              hipStream_t st);
hipError_t ntg_launch_mpc_shift_lambda(const NtgDims &D, int batch, int sbp, double *alw, hipStream_t st);
// per-problem grids: device-side setup algebra (grids.hip)
struct NtgGridLin {
	const double *blk, *linrows; const int *plan_off, *erow, *csr_ptr, *csr_col, *csc_ptr, *csc_row, *sinv_ptr, *sinv_col, *q_col, *q_row2coef;
	const unsigned char *q_pad; double *csr_val, *csc_val, *sinv_val, *q_val; int *err;
};
hipError_t ntg_launch_grid_rows(const NtgDims &D, int batch, const double *blk, const int *off, const int *plan_off, double *rowv, int *err, hipStream_t st);
hipError_t ntg_launch_grid_lin(const NtgDims &D, int batch, const NtgGridLin &g, hipStream_t st);
// preconditioner blocks of every grid on the device (grids.hip, grid_prec_kernel)
struct NtgGridPrec { const double *blk, *bps; const int *plan_off, *fidx, *binfo; double *n0b; int *err; int nblk, nb, spad, n0b_sz, nrmax; };
hipError_t ntg_launch_grid_prec(const NtgDims &D, int batch, const NtgGridPrec &g, hipStream_t st);
// structured Newton mode on per-problem grids: cost model (band, every coupling group; two-sided layout where the plan uses it) and the free
// outputs' band factors of every grid, from the per-problem channel rows and breakpoints
struct NtgGridNwt { const double *rowv, *bps; const int *plan_off; const short *lo, *hi; double *k0, *lf; int *err; int k0_sz, lf_sz; };
hipError_t ntg_launch_grid_nwt(const NtgDims &D, int batch, const NtgGridNwt &g, hipStream_t st);
hipError_t ntg_launch_mpc_shift(const NtgDims &D, const NtgTables &T, int batch, int sbp, int sknot, const double *lic,
                                double *x, double *lower, double *upper, hipStream_t st);
