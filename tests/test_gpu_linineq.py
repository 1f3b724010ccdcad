"""-m gpu: linear rows declared as inequalities (ntg_spec.lin_ineq) in the batched solver: they join the
augmented-Lagrangian loop of sqp_kernel like nonlinear rows with a constant Jacobian.  Reference: the oracle
(same algorithm; it reads lower < upper from the bounds like NPSOL)."""
import numpy as np
import pytest
import torch

import orc
from ntg_amd import api, configs as cf
from gpu_common import dev

pytestmark = pytest.mark.gpu


def _spec(name, flags):
    s = cf.config_K0() if name == "K0" else cf.config_B()
    s.lin_ineq = flags
    return s


def _bounds(name, nb):
    if name == "K0":
        lo, up = cf.bounds_K0_shipped()
        lo = np.tile(lo, (nb, 1)); up = np.tile(up, (nb, 1))
        lo[:, 0] += np.linspace(0.0, 1.0, nb); up[:, 0] = lo[:, 0]     # a small family of problems
    else:
        lo, up = cf.kincar_random_bounds(1, nb)
    lo = lo.copy(); up = up.copy()
    yT = lo[:, 6 + 3].copy()
    lo[:, 6 + 3], up[:, 6 + 3] = yT - 1.0, yT + 0.5      # y(T) in a window around the pinned value
    lo[:, 6 + 5], up[:, 6 + 5] = -0.05, 0.05              # y''(T) nearly free
    return lo, up


@pytest.mark.parametrize("name", ["K0", "B"])
@pytest.mark.parametrize("hessian", [0, 1])
def test_linear_inequality_rows_match_oracle(name, hessian):
    flags = [0] * 12; flags[6 + 3] = 1; flags[6 + 5] = 1
    spec = _spec(name, flags)
    p = api.Plan(spec, 0)
    nb = 8
    lo, up = _bounds(name, nb)
    x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    # NPSOL's default major-iteration limit 3(n + nclin) is short for several multiplier passes from a cold start
    out = p.solve(dev(lo), dev(up), x, api.default_opts(hessian=hessian, itlim=3000), want_lambda=True)
    torch.cuda.synchronize()
    inf = out["inform"].cpu().numpy(); obj = out["objective"].cpu().numpy(); lam = out["clambda"].cpu().numpy()
    xg = x.cpu().numpy()
    A = p.tables()["A"]
    nact = 0
    for i in range(nb):
        ref = orc.solve_one(spec, lo[i], up[i], np.ones(spec.nC), orc.default_opts(hessian=hessian, itlim=3000))
        assert ref["inform"] in (0, 1) and inf[i] in (0, 1)
        assert abs(obj[i] - ref["objective"]) <= 1e-7 * max(1.0, abs(ref["objective"]))
        assert np.abs(xg[i] - ref["x"]).max() <= 1e-5 * np.abs(ref["x"]).max()
        Ax = A @ xg[i]
        eq = [r for r in range(12) if not flags[r]]; iq = [r for r in range(12) if flags[r]]
        assert np.abs(Ax[eq] - lo[i][eq]).max() <= 1e-8
        assert (Ax[iq] >= lo[i][iq] - 1e-7).all() and (Ax[iq] <= up[i][iq] + 1e-7).all()
        # multipliers in NPSOL's layout [coefficients; all linear rows; nonlinear rows]: stationarity g = A' lam with
        # the kernel's own numbers (equality AND inequality rows), sign and complementarity on the inequality rows;
        # the oracle's estimates agree to the accuracy of a first-order estimate
        ll, lr = lam[i, spec.nC:spec.nC + 12], ref["clambda"][spec.nC:spec.nC + 12]
        g = orc.eval_batch(spec, xg[i][None], 2)["g"][0]
        # multiplier estimates are first order: accurate to (penalty x residual x |A|), a few 1e-4 of |g| here
        assert np.abs(g - A.T @ ll).max() <= 2e-3 * max(1.0, np.abs(g).max())
        for r in iq:
            at_lo, at_up = Ax[r] <= lo[i][r] + 1e-6, Ax[r] >= up[i][r] - 1e-6
            assert (ll[r] >= -1e-8 if at_lo else True) and (ll[r] <= 1e-8 if at_up else True)
            assert abs(ll[r]) <= 1e-6 * max(1.0, np.abs(ll).max()) or at_lo or at_up
        assert np.abs(ll - lr).max() <= 2e-2 * max(1.0, np.abs(lr).max())
        nact += int((np.abs(ll[iq]) > 1e-8).any())
    assert nact >= 2                                   # the windows bind for some of the problems


def test_equality_declared_row_with_a_range_is_refused():
    flags = [0] * 12; flags[6 + 3] = 1; flags[6 + 5] = 1
    spec = _spec("K0", flags)
    p = api.Plan(spec, 0)
    lo, up = _bounds("K0", 3)
    up[1, 2] = lo[1, 2] + 0.1                            # row 2 was NOT declared an inequality: problem 1 is invalid
    lo[2, 6 + 3], up[2, 6 + 3] = 1.0, -1.0               # empty range on a declared row: problem 2 is invalid
    x = torch.ones((3, spec.nC), dtype=torch.float64, device="cuda:0")
    out = p.solve(dev(lo), dev(up), x, api.default_opts())
    inf = out["inform"].cpu().numpy()
    assert inf[0] in (0, 1) and inf[1] == 9 and inf[2] == 9
    assert torch.equal(x[1], torch.ones_like(x[1]))      # x untouched for refused problems


def test_all_rows_equalities_is_the_old_path():
    """lin_ineq all zero must give exactly the plan without flags (same tables, tuned instance)."""
    s0 = cf.config_B(); s1 = cf.config_B(); s1.lin_ineq = [0] * 12
    lo, up = cf.kincar_random_bounds(1, 4)
    xs = []
    for s in (s0, s1):
        x = torch.ones((4, s.nC), dtype=torch.float64, device="cuda:0")
        api.Plan(s, 0).solve(dev(lo), dev(up), x, api.default_opts(hessian=1))
        xs.append(x.clone())
    assert torch.equal(xs[0], xs[1])


@pytest.mark.parametrize("hessian", [0, 1])
def test_every_linear_row_a_range(hessian):
    """No equality row left: nothing to project, every linear row goes through the augmented Lagrangian."""
    flags = [1] * 12
    spec = _spec("K0", flags)
    p = api.Plan(spec, 0)
    lo0, up0 = cf.bounds_K0_shipped()
    nb = 4
    lo = np.tile(lo0, (nb, 1)) - 0.05 * (1 + np.arange(nb))[:, None]
    up = np.tile(up0, (nb, 1)) + 0.05 * (1 + np.arange(nb))[:, None]
    x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    out = p.solve(dev(lo), dev(up), x, api.default_opts(hessian=hessian, itlim=3000), want_lambda=True)
    inf = out["inform"].cpu().numpy(); obj = out["objective"].cpu().numpy(); xg = x.cpu().numpy()
    A = p.tables()["A"]
    for i in range(nb):
        ref = orc.solve_one(spec, lo[i], up[i], np.ones(spec.nC), orc.default_opts(hessian=hessian, itlim=3000))
        assert inf[i] in (0, 1) and ref["inform"] in (0, 1)
        assert abs(obj[i] - ref["objective"]) <= 1e-6 * max(1.0, abs(ref["objective"]))
        Ax = A @ xg[i]
        assert (Ax >= lo[i] - 1e-6).all() and (Ax <= up[i] + 1e-6).all()
    assert (np.diff(obj) < 0).all()          # wider ranges, lower optimal cost


def test_linear_trajectory_inequality_rows():
    """A linear TRAJECTORY row declared as a range (lateral position y(t) <= a ceiling at every breakpoint): 101 rows of
    the augmented Lagrangian whose Jacobian rows are rows of A."""
    spec = cf.config_B()
    ltc = np.zeros((1, spec.nz)); ltc[0, 3] = 1.0            # y
    spec.ltc = ltc
    spec.lin_ineq = [0] * 6 + [1] + [0] * 6                   # lic (6), ltc (1), lfc (6)
    p = api.Plan(spec, 0)
    nb = 6
    lo0, up0 = cf.kincar_random_bounds(1, nb)
    ymax = np.maximum(lo0[:, 3], lo0[:, 9]) + 0.05           # a ceiling just above both end points
    lo = np.concatenate([lo0[:, :6], np.full((nb, 1), -cf.INF_BOUND), lo0[:, 6:]], axis=1)
    up = np.concatenate([up0[:, :6], ymax[:, None], up0[:, 6:]], axis=1)
    x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    out = p.solve(dev(lo), dev(up), x, api.default_opts(hessian=1), want_lambda=True)
    inf = out["inform"].cpu().numpy(); obj = out["objective"].cpu().numpy(); xg = x.cpu().numpy()
    A = p.tables()["A"]; P = spec.nbps
    nact = 0
    for i in range(nb):
        ref = orc.solve_one(spec, lo[i], up[i], np.ones(spec.nC), orc.default_opts(hessian=1))
        assert inf[i] in (0, 1) and ref["inform"] in (0, 1)
        assert abs(obj[i] - ref["objective"]) <= 1e-6 * max(1.0, abs(ref["objective"]))
        y = (A @ xg[i])[6:6 + P]
        assert y.max() <= ymax[i] + 1e-6
        nact += int(y.max() >= ymax[i] - 1e-6)
    assert nact >= 2
