"""CPU tests of the QP-based SQP step (opts.hessian = 3): the oracle's statement (oracle/sqp.c sqpqp_run: what NPSOL does with the Jacobian
the reference hands it, ntg.c:217-220,250-253) reaches KKT points in a fraction of the Newton mode's major iterations, its committed
solutions of configs D / E are KKT points and reproducible, and the scalar active-set routine the device runs (csrc/qpdual.hpp) ends at the
KKT point of its dual QP on random and nearly dependent working sets."""
import os
import subprocess
import numpy as np
import pytest

import orc
from ntg_amd import configs as cf
from test_oracle_newton import _kkt_ok

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


@pytest.mark.parametrize("name", ["O", "D8", "E8"])
def test_qp_sqp_reaches_kkt_points_in_fewer_majors(name):
    if name == "O":
        spec = cf.config_O(); lo, up = cf.obstacle_bounds(6)
    elif name == "D8":
        spec = cf.config_D(ninterv=8); lo, up = cf.quadrotor_bounds(6)
    else:
        spec = cf.config_E(ninterv=8, narms=2); lo, up = cf.manipulator_bounds(6, narms=2)
    it3 = it2 = 0
    for b in range(6):
        r3 = orc.solve_one(spec, lo[b], up[b], np.ones(spec.nC), orc.default_opts(hessian=3))
        r2 = orc.solve_one(spec, lo[b], up[b], np.ones(spec.nC), orc.default_opts(hessian=2))
        assert r3["inform"] == 0 and r2["inform"] == 0
        _kkt_ok(spec, lo[b], up[b], r3, 1e-6)
        if name != "E8":   # the same optimum (the arm problem is not convex: both are KKT points)
            assert abs(r3["objective"] - r2["objective"]) <= 1e-6 * abs(r2["objective"])
        it3 += r3["iters"]; it2 += r2["iters"]
    assert it3 <= 0.6 * it2, (it3, it2)


@pytest.mark.parametrize("name", ["D", "E"])
def test_golden_qp_solutions_are_kkt_points(name):
    """tests/golden/sol_qp_{D,E}.npz (tests/golden/make_solutions.py qp): inform 0 within 10 / 25 majors, KKT conditions with the reported multipliers"""
    gold = np.load(os.path.join(GOLD, f"sol_qp_{name}.npz"))
    spec = cf.config_D() if name == "D" else cf.config_E()
    assert (gold["inform"] == 0).all()
    assert gold["iters"].max() <= (10 if name == "D" else 25)
    for b in (0, 2):
        _kkt_ok(spec, gold["lower"][b], gold["upper"][b], dict(x=gold["x"][b], clambda=gold["clambda"][b]), 6e-7)


def test_golden_qp_solution_is_reproduced():
    gold = np.load(os.path.join(GOLD, "sol_qp_D.npz"))
    spec = cf.config_D()
    r = orc.solve_one(spec, gold["lower"][0], gold["upper"][0], np.ones(spec.nC), orc.default_opts(hessian=3))
    assert r["iters"] == gold["iters"][0]
    assert abs(r["objective"] - gold["objective"][0]) <= 1e-12 * abs(gold["objective"][0])
    assert np.abs(r["x"] - gold["x"][0]).max() <= 1e-10


def test_dual_active_set_routine_on_the_host(tmp_path):
    """csrc/qpdual.hpp compiled for the host: 600 working sets (a third with nearly dependent adjacent rows), KKT residual <= 1e-9 relative"""
    exe = str(tmp_path / "qpdual_drv")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "drivers", "qpdual_drv.cpp")])
    r = subprocess.run([exe, "600"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
