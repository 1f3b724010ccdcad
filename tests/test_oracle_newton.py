"""CPU tests of the oracle's structured Newton mode (oracle/sqp.c, opts.hessian = 2): the constraint Hessian callbacks,
the mode against the quasi-Newton mode on the same problems, and the committed golden solutions of configs D / E."""
import ctypes as C
import os
import numpy as np
import pytest

import orc
from ntg_amd import configs as cf

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("fam,nout,dm", [(2, 3, 3), (3, 2, 3), (4, 4, 5), (5, 6, 3)])
def test_constraint_hessians_match_central_differences(fam, nout, dm):
    L = orc.lib()
    L.orc_family_check_hess.restype = C.c_double
    rng = np.random.default_rng(fam)
    z = rng.normal(size=nout * dm); t = rng.normal(size=16)
    err = L.orc_family_check_hess(fam, nout, dm, z.ctypes.data_as(orc.dp), t.ctypes.data_as(orc.dp))
    assert err <= 1e-8, err


def _kkt_ok(spec, lo, up, r, stat_tol):
    """stationarity / feasibility / multiplier signs of one oracle result (NPSOL's sign: g = A' lam_lin + J' lam_nl)"""
    tb = orc.export_tables(spec, lo, up)
    ev = orc.eval_batch(spec, r["x"][None], 2)
    g, c, J = ev["g"][0], ev["c"][0], ev["cJac"][0]
    n, m = spec.nC, spec.nclin
    lam = r["clambda"]
    res = g - tb["A"].T @ lam[n:n + m] - J.T @ lam[n + m:]
    assert np.abs(res).max() <= stat_tol * max(1.0, np.abs(g).max())
    bl, bu = tb["bl"][n + m:], tb["bu"][n + m:]
    assert ((c - bl) / (1 + np.abs(bl)) >= -1e-7).all() and ((bu - c) / (1 + np.abs(bu)) >= -1e-7).all()


@pytest.mark.parametrize("name", ["O", "D8", "E8"])
def test_newton_mode_reaches_the_quasi_newton_optimum(name):
    if name == "O":
        spec = cf.config_O(); lo, up = cf.obstacle_bounds(3)
    elif name == "D8":
        spec = cf.config_D(ninterv=8); lo, up = cf.quadrotor_bounds(3)
    else:
        spec = cf.config_E(ninterv=8, narms=2); lo, up = cf.manipulator_bounds(3, narms=2)
    for b in range(3):
        x0 = np.ones(spec.nC)
        r2 = orc.solve_one(spec, lo[b], up[b], x0, orc.default_opts(hessian=2))
        r1 = orc.solve_one(spec, lo[b], up[b], x0, orc.default_opts(hessian=1))
        assert r2["inform"] == 0 and r1["inform"] in (0, 1)
        _kkt_ok(spec, lo[b], up[b], r2, 1e-6)
        # same local optimum unless the problem is non-convex (manipulator): then both are KKT points, the Newton one not worse by much
        if name != "E8":
            assert abs(r2["objective"] - r1["objective"]) <= 2e-5 * abs(r1["objective"])
        assert r2["iters"] <= max(60, r1["iters"])


def test_newton_mode_falls_back_where_it_does_not_apply():
    """testfam (two spline specs, initial / final nonlinear rows): hessian = 2 behaves like hessian = 1"""
    spec = cf.config_T()
    rng = np.random.default_rng(3)
    nb = spec.nlic + spec.nltc + spec.nlfc + spec.nnlic + spec.nnltc + spec.nnlfc
    lo = np.full(nb, -1.0); up = np.full(nb, 1.0)
    lo[:spec.nlic + spec.nltc + spec.nlfc] = up[:spec.nlic + spec.nltc + spec.nlfc] = 0.1   # linear rows: equalities
    up[:spec.nlic + spec.nltc + spec.nlfc] = lo[:spec.nlic + spec.nltc + spec.nlfc]
    lo[-(spec.nnlic + spec.nnltc + spec.nnlfc):] = -5.0; up[-(spec.nnlic + spec.nnltc + spec.nnlfc):] = 5.0
    x0 = 0.3 * np.ones(spec.nC)
    r2 = orc.solve_one(spec, lo, up, x0, orc.default_opts(hessian=2, itlim=40))
    r1 = orc.solve_one(spec, lo, up, x0, orc.default_opts(hessian=1, itlim=40))
    assert r2["inform"] == r1["inform"] and r2["iters"] == r1["iters"] and r2["objective"] == r1["objective"]


@pytest.mark.parametrize("name", ["D", "E"])
def test_golden_solutions_are_kkt_points(name):
    """tests/golden/sol_{D,E}.npz (made by tests/golden/make_solutions.py): inform 0 within 50 / 70 majors, KKT conditions hold"""
    gold = np.load(os.path.join(GOLD, f"sol_{name}.npz"))
    spec = cf.config_D() if name == "D" else cf.config_E()
    assert (gold["inform"] == 0).all()
    assert gold["iters"].max() <= (50 if name == "D" else 70)
    for b in (0, 5):
        _kkt_ok(spec, gold["lower"][b], gold["upper"][b], dict(x=gold["x"][b], clambda=gold["clambda"][b]), 6e-7)


def test_golden_solution_is_reproduced():
    """the generating script is deterministic: one problem of config D re-solved here equals the fixture"""
    gold = np.load(os.path.join(GOLD, "sol_D.npz"))
    spec = cf.config_D()
    r = orc.solve_one(spec, gold["lower"][1], gold["upper"][1], np.ones(spec.nC), orc.default_opts(hessian=2))
    assert r["iters"] == gold["iters"][1]
    assert abs(r["objective"] - gold["objective"][1]) <= 1e-12 * abs(gold["objective"][1])
    assert np.abs(r["x"] - gold["x"][1]).max() <= 1e-10
