"""Pins for the oracle's SQP (NPSOL replacement) at the optimum: closed-form KKT solutions for
the kincar QPs and an independent scipy solve for vanderpol (BASELINE.md §2 tolerances:
|dF| <= 1e-9, |dC|inf <= 1e-6, constraint violation <= 1e-8)."""
import numpy as np
import pytest
import scipy.optimize as so

import orc
from ntg_amd import configs as cf


def kkt_kincar(spec, b):
    """min C'HC/2 s.t. AC=b with H = 2 D2' diag(w) D2 built from the oracle's tables."""
    tab = orc.export_tables(spec, b, b)
    n, P = spec.nC, spec.nbps
    w = np.zeros(P); dt = np.diff(spec.bps); w[:-1] += dt / 2; w[1:] += dt / 2
    H = np.zeros((n, n)); pos = 0; base = 0
    for o in range(spec.nout):
        k, d, no = spec.order[o], spec.maxderiv[o], spec.ncoef[o]
        blk = tab["blk"][pos:pos + P * k * d].reshape(P, k, d); pos += P * k * d
        D2 = np.zeros((P, no))
        for i in range(P):
            D2[i, tab["off"][o, i]:tab["off"][o, i] + k] = blk[i, :, 2]
        H[base:base + no, base:base + no] = 2 * D2.T @ (w[:, None] * D2); base += no
    A = tab["A"]; m = A.shape[0]
    K = np.block([[H, A.T], [A, np.zeros((m, m))]])
    sol = np.linalg.solve(K, np.concatenate([np.zeros(n), b]))
    x = sol[:n]
    return x, 0.5 * x @ H @ x


def test_kincar_shipped_known_answer():
    spec = cf.config_K0(); lo, up = cf.bounds_K0_shipped()
    xs, fs = kkt_kincar(spec, lo)
    assert abs(fs - 2.457581141950512) < 1e-10
    np.testing.assert_allclose(xs, [0, 5, 10, 20, 30, 35, 40, -2, -2, -2, 0, 2, 2, 2], atol=1e-9)
    for h in (0, 1):
        r = orc.solve_one(spec, lo, up, np.ones(spec.nC), orc.default_opts(hessian=h))
        assert r["inform"] == 0
        assert abs(r["objective"] - fs) <= 1e-9
        assert np.abs(r["x"] - xs).max() <= 1e-6
        assert r["feas"] <= 1e-8
        assert list(r["istate"][spec.nC:]) == [3] * spec.nclin


def vdp_objective(spec):
    tab = orc.export_tables(spec)
    P, k, d = spec.nbps, spec.order[0], 3
    blk = tab["blk"].reshape(P, k, d)
    M = np.zeros((d, P, spec.nC))
    for i in range(P):
        M[:, i, tab["off"][0, i]:tab["off"][0, i] + k] = blk[i].T

    def F(c):
        z, zd, zdd = M[0] @ c, M[1] @ c, M[2] @ c
        u = zdd + z - (1 - z * z) * zd
        f = 0.5 * (z * z + zd * zd + u * u)
        return float(np.sum(np.diff(spec.bps) * (f[1:] + f[:-1]) / 2))
    return F, tab["A"]


def test_vanderpol_known_answer():
    spec = cf.config_A(); lo, up = cf.bounds_A()
    F, A = vdp_objective(spec)
    res = so.minimize(F, np.ones(spec.nC), method="SLSQP", constraints=[{"type": "eq", "fun": lambda c: A @ c - lo}],
                      options=dict(ftol=1e-15, maxiter=500))
    assert abs(res.fun - 1.7022142628309958) < 1e-9
    cstar = [1, 1, 0.3937399093, -0.0369580060, -0.4395320819, -0.7168653229, -0.2449741945]
    for h in (0, 1):
        r = orc.solve_one(spec, lo, up, np.ones(spec.nC), orc.default_opts(hessian=h))
        assert r["inform"] == 0
        assert abs(r["objective"] - 1.7022142628309958) <= 1e-9
        assert np.abs(r["x"] - cstar).max() <= 1e-6
        assert r["feas"] <= 1e-8


@pytest.mark.parametrize("cfg,ncars", [(cf.config_B, 1), (cf.config_M, 3)])
def test_config_B_M_optimum_matches_kkt(cfg, ncars):
    spec = cfg(); lo, up = cf.kincar_random_bounds(ncars, 2)
    for p in range(2):
        xs, fs = kkt_kincar(spec, lo[p])
        for h in (0, 1):
            r = orc.solve_one(spec, lo[p], up[p], np.ones(spec.nC), orc.default_opts(hessian=h))
            assert r["inform"] == 0
            assert abs(r["objective"] - fs) <= 1e-9 * max(1.0, abs(fs))
            assert np.abs(r["x"] - xs).max() <= 1e-6 * max(1.0, np.abs(xs).max())
            assert r["feas"] <= 1e-8
        assert r["iters"] <= 5            # preconditioned: the QP is solved in a few majors


def test_fixed_iteration_mode_and_limits():
    spec = cf.config_B(); lo, up = cf.kincar_random_bounds(1, 1)
    r = orc.solve_one(spec, lo[0], up[0], np.ones(spec.nC), orc.default_opts(itlim=50, fixed_iters=1), trace_cap=64)
    assert r["inform"] == 4 and r["iters"] == 50
    assert np.all(np.diff(r["trace"][:, 0]) < 0)          # monotone decrease of F
    r2 = orc.solve_one(spec, lo[0], up[0], np.ones(spec.nC), orc.default_opts(itlim=5))
    assert r2["inform"] == 4 and r2["iters"] == 5


def test_unsupported_inputs_are_loud():
    O = cf.config_O(10); lo, up = cf.obstacle_bounds(1)
    r = orc.solve_one(O, lo[0], up[0], np.ones(O.nC), orc.default_opts(itlim=50, fixed_iters=1))
    assert r["inform"] == 9                                # fixed-work mode is defined without AL rows only


def test_linear_inequality_rows_vs_scipy():
    """Linear rows with lower < upper (here: final lateral position anywhere in [-1, 1], final heading rate in
    [-0.5, 0.5]) join the augmented Lagrangian; the problem stays a convex QP, so scipy's SLSQP is an independent
    reference for the optimum."""
    import scipy.optimize as so
    spec = cf.config_K0(); lo, up = cf.bounds_K0_shipped()
    lo = lo.copy(); up = up.copy()
    lo[6 + 3], up[6 + 3] = -1.0, 1.0          # y(T)   in [-1, 1]   (row of output 1, derivative 0)
    lo[6 + 5], up[6 + 5] = -0.5, 0.5          # y''(T) in [-0.5, 0.5]
    xs, fs = kkt_kincar(spec, cf.bounds_K0_shipped()[0])   # reuse H from the helper via a second call below
    tab = orc.export_tables(spec)
    A = tab["A"]
    P = spec.nbps; w = np.zeros(P); dt = np.diff(spec.bps); w[:-1] += dt / 2; w[1:] += dt / 2
    H = np.zeros((spec.nC, spec.nC)); pos = 0; base = 0
    for o in range(2):
        k, d, no = spec.order[o], 3, spec.ncoef[o]
        blk = tab["blk"][pos:pos + P * k * d].reshape(P, k, d); pos += P * k * d
        D2 = np.zeros((P, no))
        for i in range(P):
            D2[i, tab["off"][o, i]:tab["off"][o, i] + k] = blk[i, :, 2]
        H[base:base + no, base:base + no] = 2 * D2.T @ (w[:, None] * D2); base += no
    eq = [i for i in range(12) if lo[i] == up[i]]; iq = [i for i in range(12) if lo[i] != up[i]]
    cons = [{"type": "eq", "fun": lambda x: A[eq] @ x - lo[eq], "jac": lambda x: A[eq]},
            {"type": "ineq", "fun": lambda x: A[iq] @ x - lo[iq], "jac": lambda x: A[iq]},
            {"type": "ineq", "fun": lambda x: up[iq] - A[iq] @ x, "jac": lambda x: -A[iq]}]
    ref = so.minimize(lambda x: 0.5 * x @ H @ x, xs, jac=lambda x: H @ x, constraints=cons, method="SLSQP",
                      options=dict(ftol=1e-15, maxiter=500))
    for h in (0, 1):
        r = orc.solve_one(spec, lo, up, np.ones(spec.nC), orc.default_opts(hessian=h))
        assert r["inform"] in (0, 1)
        assert abs(r["objective"] - ref.fun) <= 1e-7 * max(1.0, ref.fun)
        assert np.abs(r["x"] - ref.x).max() <= 1e-5 * np.abs(ref.x).max()
        Ax = A @ r["x"]
        assert np.abs(Ax[eq] - lo[eq]).max() <= 1e-8 and (Ax[iq] >= lo[iq] - 1e-7).all() and (Ax[iq] <= up[iq] + 1e-7).all()
        assert list(r["istate"][spec.nC + np.array(eq)]) == [3] * len(eq)
    assert ref.fun < 2.457581141950512                     # relaxing the final flag can only lower the cost


@pytest.mark.parametrize("hessian", [0, 1])
def test_obstacle_nonlinear_inequality_kkt(hessian):
    """Nonlinear trajectory inequality (family 3) through the augmented-Lagrangian outer loop: KKT conditions
    at the returned point, checked independently of the solver (NPSOL sign convention of clambda)."""
    spec = cf.config_O(10)
    lo, up = cf.obstacle_bounds(6)
    nact = 0
    for p in range(6):
        r = orc.solve_one(spec, lo[p], up[p], np.ones(spec.nC), orc.default_opts(hessian=hessian))
        assert r["inform"] == 0
        ev = orc.eval_batch(spec, r["x"][None], 2)
        g, J, c = ev["g"][0], ev["cJac"][0], ev["c"][0]
        A = orc.export_tables(spec, lo[p], up[p])["A"]
        lam = r["clambda"]; ll, ln = lam[spec.nC:spec.nC + spec.nclin], lam[spec.nC + spec.nclin:]
        assert np.abs(g - A.T @ ll - J.T @ ln).max() <= 2e-6 * np.abs(g).max()      # stationarity
        assert np.abs(A @ r["x"] - lo[p][:spec.nclin]).max() <= 1e-8                # linear feasibility
        assert (c - 9.0).min() >= -1e-7 * 9.0                                        # c >= r^2
        assert ln.min() >= -1e-9                                                      # dual feasibility
        assert np.abs(ln * (c - 9.0)).max() <= 1e-5 * max(1.0, np.abs(ln).max())     # complementarity
        nact += int((ln > 1e-9).sum())
        assert list(r["istate"][spec.nC:spec.nC + spec.nclin]) == [3] * spec.nclin
    assert nact >= 3                                       # the obstacle really is active in this sample


def test_R_factor_is_cholesky_of_hessian_estimate():
    spec = cf.config_K0(); lo, up = cf.bounds_K0_shipped()
    r = orc.solve_one(spec, lo, up, np.ones(spec.nC), want_R=True)
    R = r["R"]
    assert np.allclose(R, np.triu(R))
    H = R.T @ R
    assert np.all(np.linalg.eigvalsh(H) > 0)


def _de_case(name):
    if name == "D":
        return cf.config_D(ninterv=4), cf.quadrotor_bounds(3)
    return cf.config_E(ninterv=4, narms=1), cf.manipulator_bounds(3, narms=1)


@pytest.mark.parametrize("name", ["D", "E"])
def test_families_4_5_derivatives_by_finite_differences(name):
    """The quadrotor / manipulator callbacks are build-defined: their hand-written gradients and Jacobians are
    pinned by central differences of their own values."""
    spec, _ = _de_case(name)
    rng = np.random.default_rng(3)
    x = rng.normal(size=spec.nC) * 0.4 + 0.8
    ev = orc.eval_batch(spec, x[None], 2)
    h = 1e-6
    dirs = rng.normal(size=(6, spec.nC))
    xp = np.concatenate([x[None] + h * dirs, x[None] - h * dirs])
    e2 = orc.eval_batch(spec, xp, 0)
    df = (e2["f"][:6] - e2["f"][6:]) / (2 * h)
    dc = (e2["c"][:6] - e2["c"][6:]) / (2 * h)
    assert np.abs(df - dirs @ ev["g"][0]).max() <= 1e-6 * max(1.0, np.abs(df).max())
    assert np.abs(dc - dirs @ ev["cJac"][0].T).max() <= 1e-6 * max(1.0, np.abs(dc).max())


@pytest.mark.parametrize("name", ["D", "E"])
def test_families_4_5_reduced_grid_vs_slsqp(name):
    """Reduced configs D / E against scipy's SLSQP on the same functions: the augmented-Lagrangian solve must
    reach the same constrained optimum (the problems are smooth and, at this size, have one minimum nearby)."""
    spec, (lo, up) = _de_case(name)
    A = orc.export_tables(spec)["A"]
    P, nl = spec.nbps, spec.nclin
    nact = 0
    for b in range(3):
        r = orc.solve_one(spec, lo[b], up[b], np.ones(spec.nC), orc.default_opts(hessian=1))
        assert r["inform"] in (0, 1)
        lcb = np.repeat(lo[b][nl:], P); ucb = np.repeat(up[b][nl:], P)
        fin_l = lcb > -1e19; fin_u = ucb < 1e19

        def cons(x):
            c = orc.eval_batch(spec, x[None], 0)["c"][0]
            return np.concatenate([(c - lcb)[fin_l], (ucb - c)[fin_u]])

        def cons_jac(x):
            J = orc.eval_batch(spec, x[None], 2)["cJac"][0]
            return np.concatenate([J[fin_l], -J[fin_u]])
        def slsqp(start):
            return so.minimize(lambda x: orc.eval_batch(spec, x[None], 0)["f"][0], start, jac=lambda x: orc.eval_batch(spec, x[None], 2)["g"][0],
                               constraints=[{"type": "eq", "fun": lambda x: A @ x - lo[b][:nl], "jac": lambda x: A},
                                            {"type": "ineq", "fun": cons, "jac": cons_jac}],
                               method="SLSQP", options=dict(ftol=1e-14, maxiter=400))
        # SLSQP at ftol 1e-14 usually stops with status 8 ("positive directional derivative") at the optimum
        ref = slsqp(np.ones(spec.nC))
        assert ref.status in (0, 8) and cons(ref.x).min() >= -1e-7 and np.abs(A @ ref.x - lo[b][:nl]).max() <= 1e-7
        assert cons(r["x"]).min() >= -1e-6 and np.abs(A @ r["x"] - lo[b][:nl]).max() <= 1e-8
        # the arm problem is not convex: from the same start SLSQP may stop in a worse local minimum, never a better one
        assert r["objective"] <= ref.fun + 2e-6 * max(1.0, abs(ref.fun))
        if name == "D":
            assert abs(r["objective"] - ref.fun) <= 2e-6 * max(1.0, abs(ref.fun))
        # and SLSQP started at the returned point stays there: it is a constrained local minimum
        pol = slsqp(r["x"])
        assert abs(r["objective"] - pol.fun) <= 2e-6 * max(1.0, abs(pol.fun))
        nact += int((np.abs(r["clambda"][spec.nC + spec.nclin:]) > 1e-8).any())
    assert nact >= 1                                       # an inequality binds for at least one of the draws
