"""-m gpu: receding-horizon MPC (BASELINE config C): solve, advance one knot
interval along the solution, re-pin the initial flag, shift-warm-start, re-solve -- against the same
loop run with the CPU oracle and a numpy restatement of the shift."""
import numpy as np
import pytest
import torch

import orc
from ntg_amd import api, configs as cf
from gpu_common import plan_for, dev

pytestmark = pytest.mark.gpu


def shift_numpy(spec, tab, x, lo, up, sbp, sknot):
    """numpy statement of ntg_batch_mpc_shift for one problem."""
    P = spec.nbps
    z = np.zeros(spec.nz); pos = 0; iC = 0; iz = 0
    xn = x.copy()
    for o in range(spec.nout):
        k, d, n = spec.order[o], spec.maxderiv[o], spec.ncoef[o]
        blk = tab["blk"][pos:pos + P * k * d].reshape(P, k, d); pos += P * k * d
        off = tab["off"][o, sbp]
        for r in range(d):
            z[iz + r] = sum(blk[sbp, q, r] * x[iC + off + q] for q in range(k))
        sh = sknot * (k - spec.mult[o])
        for cl in range(n):
            xn[iC + cl] = x[iC + min(cl + sh, n - 1)]
        iC += n; iz += d
    lo2, up2 = lo.copy(), up.copy()
    lo2[:spec.nlic] = spec.lic @ z; up2[:spec.nlic] = spec.lic @ z
    return xn, lo2, up2


def test_mpc_loop_matches_oracle_loop():
    spec = cf.config_B(); p = plan_for("B")
    nb, nsteps, sknot = 6, 4, 1
    sbp = 5 * sknot                                   # P = 5 l + 1: knot j is breakpoint 5 j
    lo, up = cf.kincar_random_bounds(1, nb)
    tab = orc.export_tables(spec)
    x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    lo_d, up_d = dev(lo), dev(up)
    opts = api.default_opts(hessian=1)
    for step in range(nsteps):
        # the oracle solves exactly the problem the GPU is about to solve (same re-pinned bounds, same warm start);
        # two free-running loops would drift apart by the 1e-7 accuracy of each re-pinned initial flag
        lo_h, up_h, x_h = lo_d.cpu().numpy(), up_d.cpu().numpy(), x.cpu().numpy()
        out = p.solve(lo_d, up_d, x, opts)
        ref = orc.solve_batch(spec, lo_h, up_h, x_h, orc.default_opts(hessian=1), nthreads=4)
        assert (out["inform"].cpu().numpy() == 0).all() and (ref["inform"] == 0).all()
        xg = x.cpu().numpy()
        assert np.abs(xg - ref["x"]).max() <= 1e-6 * np.abs(ref["x"]).max(), step
        assert np.abs(out["objective"].cpu().numpy() - ref["objective"]).max() <= 1e-9 * np.abs(ref["objective"]).max()
        # shift: GPU kernel vs numpy statement on the same input
        exp = [shift_numpy(spec, tab, xg[i], lo_h[i], up_h[i], sbp, sknot) for i in range(nb)]
        p.mpc_shift(x, lo_d, up_d, sbp, sknot)
        np.testing.assert_allclose(x.cpu().numpy(), np.stack([e[0] for e in exp]), rtol=0, atol=0)
        np.testing.assert_allclose(lo_d.cpu().numpy(), np.stack([e[1] for e in exp]), rtol=1e-13, atol=1e-12)
        assert torch.equal(lo_d, up_d)
    # warm starts pay off: later re-solves need no more majors than the cold first one
    assert int(out["iters"].max()) <= 5


def test_mpc_full_config_C_properties():
    """config C size (100 re-solves x 1024): every re-solve converges and stays feasible."""
    spec = cf.config_B(); p = plan_for("B")
    nb = 1024
    lo, up = cf.kincar_random_bounds(1, nb)
    lo_d, up_d = dev(lo), dev(up)
    x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    A = dev(p.tables()["A"])
    opts = api.default_opts(hessian=1)
    work = torch.empty(p.workspace_bytes(nb, opts), dtype=torch.uint8, device="cuda:0")
    worst_feas = 0.0; ninf = 0
    for step in range(100):
        out = p.solve(lo_d, up_d, x, opts, work=work)
        ninf += int((out["inform"] != 0).sum().item())
        if step % 25 == 0:
            worst_feas = max(worst_feas, (x @ A.T - lo_d).abs().max().item())
        p.mpc_shift(x, lo_d, up_d, 5, 1)
    assert ninf == 0
    assert worst_feas <= 1e-8


def test_mpc_run_graph_replay_is_the_host_loop():
    """ntg_batch_mpc_run (first step direct, the rest replayed as a hipGraph) == the same steps issued one by one."""
    spec = cf.config_B(); p = plan_for("B")
    nb = 256
    lo, up = cf.kincar_random_bounds(1, nb)
    opts = api.default_opts(hessian=1)
    work = torch.empty(p.workspace_bytes(nb, opts), dtype=torch.uint8, device="cuda:0")
    lo1, up1 = dev(lo), dev(up)
    x1 = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    for _ in range(12):
        p.solve(lo1, up1, x1, opts, work=work)
        p.mpc_shift(x1, lo1, up1, 5, 1)
    lo2, up2 = dev(lo), dev(up)
    x2 = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    inform, bad = p.mpc_run(x2, lo2, up2, 12, 5, 1, opts, work=work)
    torch.cuda.synchronize()
    assert int(bad.item()) == 0 and (inform == 0).all()
    assert torch.equal(x1, x2) and torch.equal(lo1, lo2) and torch.equal(up1, up2)


@pytest.mark.parametrize("hess", [1, 3])
def test_obstacle_mpc_with_multiplier_carry_over(hess):
    """Receding horizon with a nonlinear inequality row (kincar + circular obstacle): the multiplier estimates travel with the horizon
    (ntg_solve_opts.warm_start + ntg_batch_mpc_shift_multipliers -- the use NPSOL's clambda was meant for, ntg.h:64-68).  Every re-solve
    is compared with the oracle started from the SAME multipliers, and the carry-over has to pay: fewer majors than re-solving cold.
    hess = 1: augmented-Lagrangian passes from the carried-over estimates; hess = 3: the QP-based SQP step, whose first working set is the
    rows the carried-over multipliers name (no pass on the objective alone)."""
    spec = cf.config_O(20); p = api.Plan(spec, 0)
    nb, nsteps, sknot = 6, 4, 1
    sbp = 5 * sknot
    P, n0 = spec.nbps, spec.nC + spec.nclin
    lo, up = cf.obstacle_bounds(nb)
    cold, warm = api.default_opts(hessian=hess), api.default_opts(hessian=hess, warm_start=1)
    work = torch.empty(p.workspace_bytes(nb, warm), dtype=torch.uint8, device="cuda:0")

    def loop(carry):
        x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
        lo_d, up_d = dev(lo), dev(up)
        majors, lam_prev = 0, None
        for step in range(nsteps):
            o = warm if (carry and step > 0) else cold
            lo_h, up_h, x_h = lo_d.cpu().numpy(), up_d.cpu().numpy(), x.cpu().numpy()
            out = p.solve(lo_d, up_d, x, o, work=work, want_lambda=True)
            torch.cuda.synchronize()
            assert (out["inform"].cpu().numpy() == 0).all(), (step, out["inform"])
            majors += int(out["iters"].sum().item())
            if carry:
                for i in range(nb):   # the oracle on exactly this re-solve: same bounds, same start, same starting multipliers
                    ref = orc.solve_one(spec, lo_h[i], up_h[i], x_h[i], orc.default_opts(hessian=hess), warm_lam=None if lam_prev is None else lam_prev[i])
                    assert ref["inform"] == 0
                    assert abs(out["objective"][i].item() - ref["objective"]) <= 1e-9 * max(1.0, abs(ref["objective"])), (step, i)
                    assert np.abs(x[i].cpu().numpy() - ref["x"]).max() <= 1e-6 * max(1.0, np.abs(ref["x"]).max()), (step, i)
                    assert abs(int(out["iters"][i]) - ref["iters"]) <= 2, (step, i, int(out["iters"][i]), ref["iters"])
            lam = -out["clambda"][:, n0:].cpu().numpy()                 # internal sign of the nonlinear rows' estimates
            p.mpc_shift(x, lo_d, up_d, sbp, sknot)
            if carry:
                p.mpc_shift_multipliers(nb, sbp, warm, work)
                lam_prev = np.concatenate([lam[:, sbp:], np.zeros((nb, sbp))], axis=1)   # numpy statement of the multiplier shift
        return majors

    m_warm = loop(True)
    m_cold = loop(False)
    assert m_warm < m_cold, (m_warm, m_cold)


@pytest.mark.parametrize("hess", [1, 3])
def test_mpc_run_with_warm_start_is_the_host_loop_with_a_cold_first_step(hess):
    """ntg_batch_mpc_run with warm_start = 1 on a plan with nonlinear rows: the library solves the FIRST step cold (the workspace holds no
    multiplier estimates yet -- here it is filled with NaN bytes on purpose) and every later step warm, exactly like the hand-written host
    loop of three launches per step (solve, shift, multiplier shift)."""
    spec = cf.config_O(20); p = api.Plan(spec, 0)
    nb, nsteps, sknot = 6, 4, 1
    sbp = 5 * sknot
    lo, up = cf.obstacle_bounds(nb)
    cold, warm = api.default_opts(hessian=hess), api.default_opts(hessian=hess, warm_start=1)
    nbytes = p.workspace_bytes(nb, warm)
    work1 = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    x1 = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    lo1, up1 = dev(lo), dev(up)
    for step in range(nsteps):
        out = p.solve(lo1, up1, x1, warm if step > 0 else cold, work=work1)
        assert (out["inform"].cpu().numpy() == 0).all(), (step, out["inform"])
        p.mpc_shift(x1, lo1, up1, sbp, sknot)
        p.mpc_shift_multipliers(nb, sbp, warm, work1)
    work2 = torch.full((nbytes,), 0xFF, dtype=torch.uint8, device="cuda:0")   # every double a NaN: a warm first step would read these
    x2 = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    lo2, up2 = dev(lo), dev(up)
    inform, bad = p.mpc_run(x2, lo2, up2, nsteps, sbp, sknot, warm, work=work2)
    torch.cuda.synchronize()
    assert int(bad.item()) == 0 and (inform == 0).all()
    assert torch.isfinite(x2).all()
    assert torch.equal(x1, x2) and torch.equal(lo1, lo2) and torch.equal(up1, up2)
