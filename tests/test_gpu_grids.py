"""-m gpu: per-problem grids (ntg_plan_set_grids): every problem of a batch on its own break sequence and breakpoints
-- the setup phase of ntg() (ntg.c:114-229) per problem -- against the oracle built once per problem on that grid.

Tolerances: evaluation 1e-12 relative (same arithmetic, other summation order); optimum as tests/test_gpu_solve.py
(|dF| <= 1e-9 max(1,|F|), |dx| <= 1e-6 max(1,|x|inf), linear feasibility 1e-8 relative to the row's largest entry:
short horizons scale the derivative rows by 1/h^r)."""
import dataclasses
import numpy as np
import pytest
import torch

import orc
from ntg_amd import api, configs as cf
from gpu_common import dev, rel

pytestmark = pytest.mark.gpu


def grids_for(spec, nb, seed=5, warp=0.0):
    """nb horizons in [0.6, 1.6] x the plan's, optionally with non-uniform breaks (a smooth monotone warp of the knots);
    breakpoints stay inside the plan's knot intervals: interpolated between the problem's own knots"""
    rng = np.random.default_rng(seed)
    l, P = spec.kninterv[0], spec.nbps
    k0 = np.asarray(spec.knots[0]); T0 = k0[-1] - k0[0]
    # position of every plan breakpoint inside its knot interval (interval index, fraction)
    j = np.minimum(np.searchsorted(k0, spec.bps, side="right") - 1, l - 1)
    fr = (spec.bps - k0[j]) / (k0[j + 1] - k0[j])
    knots = np.zeros((nb, l + 1)); bps = np.zeros((nb, P))
    for b in range(nb):
        s = (k0 - k0[0]) / T0
        a = warp * rng.uniform(-1, 1)
        kn = k0[0] + T0 * rng.uniform(0.6, 1.6) * (s + a * s * (1 - s))
        knots[b] = kn
        bp = kn[j] + fr * (kn[j + 1] - kn[j])
        # keep every breakpoint in the plan's interval in floating point too (linspace puts some an ulp beside a knot)
        bp = np.maximum(bp, kn[j]); inner = j < l - 1
        bp[inner] = np.minimum(bp[inner], np.nextafter(kn[j + 1][inner], -np.inf))
        if spec.bps[-1] >= k0[-1]: bp[-1] = max(bp[-1], kn[-1])
        bps[b] = bp
    return knots, bps


def spec_on(spec, knots, bps):
    return dataclasses.replace(spec, bps=bps.copy(), knots=[knots.copy() for _ in range(spec.nout)])


@pytest.mark.parametrize("name,warp", [("B", 0.0), ("M", 0.3), ("K0", 0.3)])
def test_eval_on_per_problem_grids(name, warp):
    spec = {"B": cf.config_B, "M": cf.config_M, "K0": cf.config_K0}[name]()
    nb = 16
    knots, bps = grids_for(spec, nb, warp=warp)
    p = api.Plan(spec, 0)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((nb, spec.nC))
    shared = p.eval(dev(x))
    f_shared = shared["f"].cpu().numpy().copy()
    p.set_grids(dev(knots), dev(bps), with_precond=False)
    out = p.eval(dev(x))
    torch.cuda.synchronize()
    f = out["f"].cpu().numpy(); g = out["g"].cpu().numpy()
    for b in range(nb):
        ref = orc.eval_batch(spec_on(spec, knots[b], bps[b]), x[b:b + 1])
        assert abs(f[b] - ref["f"][0]) <= 1e-12 * max(1.0, abs(ref["f"][0]))
        assert rel(g[b], ref["g"][0]) <= 1e-12
    assert np.abs(f - f_shared).max() > 1e-3          # the grids do differ
    with pytest.raises(api.NtgError):                  # the grids are for exactly this batch
        p.eval(dev(x[:3]))
    p.clear_grids()
    again = p.eval(dev(x))
    np.testing.assert_array_equal(again["f"].cpu().numpy(), f_shared)


@pytest.mark.parametrize("name,ncars", [("B", 1), ("M", 3)])
@pytest.mark.parametrize("hessian", [0, 1])
def test_solve_on_per_problem_horizons(name, ncars, hessian):
    """free-final-time style batch: the same boundary conditions reached over 16 different horizons"""
    spec = {"B": cf.config_B, "M": cf.config_M}[name]()
    nb = 16
    knots, bps = grids_for(spec, nb, warp=0.2)
    lo, up = cf.kincar_random_bounds(ncars, nb)
    p = api.Plan(spec, 0)
    p.set_grids(dev(knots), dev(bps), with_precond=bool(hessian))
    x = dev(np.ones((nb, spec.nC)))
    # which kernel: the wave kernel's per-problem-grid instances (wave-private value tables; with the preconditioner: its per-problem
    # blocks read from HBM)
    assert p.solve_kernel(nb, api.default_opts(hessian=hessian)) == "sqp_wave_kernel"
    out = p.solve(dev(lo), dev(up), x, api.default_opts(hessian=hessian))
    torch.cuda.synchronize()
    xs = x.cpu().numpy(); obj = out["objective"].cpu().numpy(); inform = out["inform"].cpu().numpy(); iters = out["iters"].cpu().numpy()
    assert (inform == 0).all()
    objs = []
    for b in range(nb):
        sb = spec_on(spec, knots[b], bps[b])
        ref = orc.solve_one(sb, lo[b], up[b], np.ones(spec.nC), orc.default_opts(hessian=hessian))
        assert ref["inform"] == 0
        assert abs(obj[b] - ref["objective"]) <= 1e-9 * max(1.0, abs(ref["objective"]))
        assert np.abs(xs[b] - ref["x"]).max() <= 1e-6 * max(1.0, np.abs(ref["x"]).max())
        A = orc.export_tables(sb)["A"]
        assert (np.abs(A @ xs[b] - lo[b]) <= 1e-8 * np.maximum(1.0, np.abs(A).max(axis=1))).all()
        objs.append(ref["objective"])
        if hessian == 1:
            assert abs(int(iters[b]) - ref["iters"]) <= 1
    if hessian == 1:
        assert iters.max() <= 5          # the per-problem preconditioner is the exact reduced Hessian of each grid
    assert np.ptp(objs) > 1e-3


def test_device_and_host_preconditioner_blocks_agree(monkeypatch):
    """the per-problem preconditioner blocks built on the device (grid_prec_kernel: inverse of the free coefficients' principal submatrix)
    and on host threads (Householder null space, NTG_AMD_HOST_PRECOND=1) are the same operator: identical major-iteration counts,
    objectives to 1e-9 and optima to 1e-6 (the parity tolerances; measured 1e-11 relative) on 16 horizons of config M"""
    spec = cf.config_M(); nb = 16
    knots, bps = grids_for(spec, nb, warp=0.2, seed=11)
    lo, up = cf.kincar_random_bounds(3, nb)
    res = []
    for host in (False, True):
        if host: monkeypatch.setenv("NTG_AMD_HOST_PRECOND", "1")
        p = api.Plan(spec, 0)
        p.set_grids(dev(knots), dev(bps), with_precond=True)
        x = dev(np.ones((nb, spec.nC)))
        out = p.solve(dev(lo), dev(up), x, api.default_opts(hessian=1))
        torch.cuda.synchronize()
        res.append((out["iters"].cpu().numpy().copy(), out["objective"].cpu().numpy().copy(), x.cpu().numpy().copy(), out["inform"].cpu().numpy().copy()))
    assert (res[0][3] == 0).all() and (res[1][3] == 0).all()
    assert np.array_equal(res[0][0], res[1][0])
    assert (np.abs(res[0][1] - res[1][1]) <= 1e-9 * np.maximum(1.0, np.abs(res[1][1]))).all()          # both converged inside the same tolerance ball
    assert np.abs(res[0][2] - res[1][2]).max() <= 1e-6 * max(1.0, np.abs(res[1][2]).max())


def test_interp_on_per_problem_grids():
    """ntg_batch_interp after ntg_plan_set_grids: every problem at its own times on its own knots == SplineInterp (colloc.c:449-484,
    restated in the oracle) with that problem's knots; a shared time vector is refused (shape), a batch of another size too"""
    import ctypes as C
    spec = cf.config_M(); nb = 6
    knots, bps = grids_for(spec, nb, warp=0.3, seed=3)
    p = api.Plan(spec, 0)
    p.set_grids(dev(knots), dev(bps), with_precond=False)
    rng = np.random.default_rng(8)
    x = rng.normal(size=(nb, spec.nC))
    nt = 17
    times = np.stack([np.concatenate([[knots[b, 0], knots[b, -1]], knots[b, 1:4], rng.uniform(knots[b, 0], knots[b, -1], nt - 5)]) for b in range(nb)])
    z = p.interp(dev(x), dev(times)).cpu().numpy()
    assert z.shape == (nb, nt, spec.nz)
    dp = C.POINTER(C.c_double)
    iz = np.concatenate([[0], np.cumsum(spec.maxderiv)]); iC = np.concatenate([[0], np.cumsum(spec.ncoef)])
    ref = np.zeros_like(z)
    for b in range(nb):
        kn = np.ascontiguousarray(knots[b])
        for o in range(spec.nout):
            co = np.ascontiguousarray(x[b, iC[o]:iC[o + 1]])
            for ti in range(nt):
                f = np.zeros(spec.maxderiv[o])
                orc.lib().orc_spline_interp(f.ctypes.data_as(dp), C.c_double(float(times[b, ti])), kn.ctypes.data_as(dp), int(spec.kninterv[o]),
                                            co.ctypes.data_as(dp), int(spec.ncoef[o]), int(spec.order[o]), int(spec.mult[o]), int(spec.maxderiv[o]))
                ref[b, ti, iz[o]:iz[o + 1]] = f
    assert rel(z, ref) <= 1e-13
    with pytest.raises(api.NtgError):
        p.interp(dev(x[:3]), dev(times[:3]))
    # a 1-D time vector with grids in force is an explicit broadcast (times_stride = 0 of ntg_batch_interp_strided): every problem at the
    # SAME times on its OWN knots -- not an ntimes-long buffer read as [batch][ntimes]
    tshared = np.linspace(max(knots[:, 0]), min(knots[:, -1]), nt)
    zs = p.interp(dev(x), dev(tshared)).cpu().numpy()
    zr = p.interp(dev(x), dev(np.tile(tshared, (nb, 1)))).cpu().numpy()
    assert np.array_equal(zs, zr)
    p.clear_grids()
    z0 = p.interp(dev(x), dev(times[0])).cpu().numpy()          # back on the plan's knots: one shared time vector
    assert z0.shape == (nb, nt, spec.nz)
    with pytest.raises(api.NtgError):                            # per-problem times without per-problem grids: refused, not "row 0 for everyone"
        p.interp(dev(x), dev(times))
    bad = api.lib().ntg_batch_interp_strided(p.h, nb, api._ptr(dev(x)), nt, api._ptr(dev(times)), nt, api._ptr(torch.empty((nb, nt, spec.nz), dtype=torch.float64, device="cuda:0")), None)
    assert bad != 0                                              # ... and by the C entry itself


def test_grid_structure_mismatch_is_refused():
    spec = cf.config_B()
    nb = 4
    knots, bps = grids_for(spec, nb)
    p = api.Plan(spec, 0)
    bad = bps.copy()
    bad[2, 7] = knots[2, 2] + 1e-3           # breakpoint 7 belongs to knot interval 1, moved into interval 2
    bad[2] = np.sort(bad[2])
    with pytest.raises(api.NtgError, match="knot interval"):
        p.set_grids(dev(knots), dev(bad), with_precond=False)
    with pytest.raises(api.NtgError):
        p.set_grids(dev(knots[:, :-1]), dev(bps))


@pytest.mark.parametrize("name", ["O", "D8", "E8", "D"])   # D: BASELINE's size -- the two-sided band layout, the yaw output's factor per grid
def test_nonlinear_rows_on_per_problem_horizons(name):
    """free final time with nonlinear trajectory rows (obstacle avoidance; the quadrotor's thrust and speed bounds; the arms' ceiling): evaluation
    (values, gradient, residuals, dense Jacobian) on 8 horizons against the oracle built on each grid, then the structured Newton mode
    (hessian = 2) and the QP-based SQP step (hessian = 3) ON those grids -- the cost model and the free outputs' factors of every grid come
    from the device (grids.hip, grid_nwt_kernel; VERDICT r3 item 5: the reference builds its collocation per ntg() call, colloc.c:57-117) --
    each problem against the oracle's solve in the same mode on that grid: objective 1e-9, x 1e-6."""
    if name == "O":
        spec = cf.config_O(); lo, up = cf.obstacle_bounds(8)
    elif name == "E8":
        spec = cf.config_E(ninterv=8, narms=2); lo, up = cf.manipulator_bounds(8, narms=2)
    else:
        spec = cf.config_D(ninterv=8) if name == "D8" else cf.config_D(); lo, up = cf.quadrotor_bounds(8)
    nb = 8
    knots, bps = grids_for(spec, nb, warp=0.2, seed=9)
    if name in ("D8", "D"):   # horizons in [0.9, 1.3] x the plan's: shorter flights violate the thrust bound outright
        knots, bps = grids_for(spec, nb, warp=0.1 if name == "D8" else 0.0, seed=9)   # (40 intervals: a warped break sequence moves breakpoints across knots)
        s0 = knots[:, :1]; sc = (0.9 + 0.4 * (knots[:, -1:] - s0 - 0.6 * 5.0) / 5.0)
        knots = s0 + (knots - s0) / (knots[:, -1:] - s0) * 5.0 * sc; bps = s0 + (bps - s0) / (bps[:, -1:] - s0) * 5.0 * sc
        # the affine map rounds: keep every breakpoint in the plan's knot interval in floating point (as grids_for does)
        k0 = np.asarray(spec.knots[0]); l = spec.kninterv[0]
        j = np.minimum(np.searchsorted(k0, spec.bps, side="right") - 1, l - 1); inner = j < l - 1
        for b in range(nb):
            bps[b] = np.maximum(bps[b], knots[b][j])
            bps[b][inner] = np.minimum(bps[b][inner], np.nextafter(knots[b][j + 1][inner], -np.inf))
            if spec.bps[-1] >= k0[-1]: bps[b][-1] = max(bps[b][-1], knots[b][-1])
    p = api.Plan(spec, 0)
    p.set_grids(dev(knots), dev(bps), with_precond=True)
    x = np.random.default_rng(4).normal(size=(nb, spec.nC)) * 0.3 + 1.0
    ev = p.eval(dev(x), 2, want_dense_jac=True)
    for b in range(nb):
        ref = orc.eval_batch(spec_on(spec, knots[b], bps[b]), x[b:b + 1], 2)
        assert rel(ev["f"][b:b + 1].cpu().numpy(), ref["f"]) <= 1e-12 and rel(ev["g"][b].cpu().numpy(), ref["g"][0]) <= 1e-12
        assert rel(ev["c"][b].cpu().numpy(), ref["c"][0]) <= 1e-12 and rel(ev["cJac"][b].cpu().numpy(), ref["cJac"][0]) <= 1e-12
    objs = []
    for hess in (2, 3):
        assert p.solve_kernel(nb, api.default_opts(hessian=hess)) == "sqp_kernel"
        xs = dev(np.ones((nb, spec.nC)))
        out = p.solve(dev(lo), dev(up), xs, api.default_opts(hessian=hess))
        torch.cuda.synchronize()
        inform = out["inform"].cpu().numpy(); obj = out["objective"].cpu().numpy(); it = out["iters"].cpu().numpy()
        assert (inform == 0).all(), (hess, inform)
        for b in range(nb):
            ref = orc.solve_one(spec_on(spec, knots[b], bps[b]), lo[b], up[b], np.ones(spec.nC), orc.default_opts(hessian=hess))
            assert ref["inform"] == 0
            assert abs(int(it[b]) - ref["iters"]) <= 3, (hess, b, it[b], ref["iters"])   # (not a run of the quasi-Newton mode: that takes hundreds)
            assert abs(obj[b] - ref["objective"]) <= 1e-9 * max(1.0, abs(ref["objective"])), (hess, b, obj[b], ref["objective"])
            assert np.abs(xs[b].cpu().numpy() - ref["x"]).max() <= 1e-6 * max(1.0, np.abs(ref["x"]).max()), (hess, b)
            objs.append(ref["objective"])
    assert np.ptp(objs) > 1e-3


def test_receding_horizon_on_per_problem_grids():
    """solve, advance one knot interval, re-pin, shift, re-solve -- every problem on its own horizon: each re-solve against the oracle built
    on that problem's grid, the shift against a numpy statement with that problem's basis blocks; then the same loop through
    ntg_batch_mpc_run (hipGraph) gives the same iterates"""
    from test_gpu_mpc import shift_numpy
    spec = cf.config_B(); nb, nsteps = 5, 3
    knots, bps = grids_for(spec, nb, warp=0.0, seed=13)      # uniform knots per problem (the coefficient shift assumes them), horizons differ
    lo, up = cf.kincar_random_bounds(1, nb)
    p = api.Plan(spec, 0)
    p.set_grids(dev(knots), dev(bps), with_precond=True)
    specs = [spec_on(spec, knots[b], bps[b]) for b in range(nb)]
    tabs = [orc.export_tables(sb) for sb in specs]
    x = dev(np.ones((nb, spec.nC))); lo_d, up_d = dev(lo), dev(up)
    x2 = x.clone(); lo2, up2 = lo_d.clone(), up_d.clone()
    opts = api.default_opts(hessian=1)
    for step in range(nsteps):
        lo_h, up_h, x_h = lo_d.cpu().numpy(), up_d.cpu().numpy(), x.cpu().numpy()
        out = p.solve(lo_d, up_d, x, opts)
        torch.cuda.synchronize()
        assert (out["inform"].cpu().numpy() == 0).all()
        xg = x.cpu().numpy()
        for b in range(nb):
            ref = orc.solve_one(specs[b], lo_h[b], up_h[b], x_h[b], orc.default_opts(hessian=1))
            assert ref["inform"] == 0
            assert np.abs(xg[b] - ref["x"]).max() <= 1e-6 * max(1.0, np.abs(ref["x"]).max()), (step, b)
        exp = [shift_numpy(specs[b], tabs[b], xg[b], lo_h[b], up_h[b], 5, 1) for b in range(nb)]
        p.mpc_shift(x, lo_d, up_d, 5, 1)
        np.testing.assert_allclose(x.cpu().numpy(), np.stack([e[0] for e in exp]), rtol=0, atol=0)
        np.testing.assert_allclose(lo_d.cpu().numpy(), np.stack([e[1] for e in exp]), rtol=1e-13, atol=1e-12)
    # the captured loop
    res = p.mpc_run(x2, lo2, up2, nsteps, 5, 1, opts)
    torch.cuda.synchronize()
    np.testing.assert_allclose(x2.cpu().numpy(), x.cpu().numpy(), rtol=0, atol=1e-9 * max(1.0, float(x.abs().max())))
    with pytest.raises(api.NtgError):
        p.mpc_shift(x[:2].contiguous(), lo_d[:2].contiguous(), up_d[:2].contiguous(), 5, 1)
