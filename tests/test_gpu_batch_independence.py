"""-m gpu: a problem's result does not depend on what else is in the launch -- bitwise, through the C ABI -- for the paths added in round 3:
the structured Newton mode with the two-sided factorisation and a free output (config D), the full-request instance of the evaluation
kernel, and the wave kernel's per-problem-grid instances (with and without the preconditioner).  Persistent kernels, problem queues and
workgroup-private tables make this a property worth asserting: a stale table, a missed reset of per-problem state or a read past a batch's
end would show up here first."""
import numpy as np
import pytest
import torch

from ntg_amd import api, configs as cf
from gpu_common import dev
from test_gpu_grids import grids_for

pytestmark = pytest.mark.gpu


def _solve(plan, lo, up, nC, opts):
    x = torch.ones((lo.shape[0], nC), dtype=torch.float64, device="cuda:0")
    out = plan.solve(dev(lo), dev(up), x, opts)
    torch.cuda.synchronize()
    return x.cpu().numpy(), out["objective"].cpu().numpy(), out["inform"].cpu().numpy(), out["iters"].cpu().numpy()


def test_newton_mode_config_D():
    spec = cf.config_D(); p = api.Plan(spec, 0)
    lo, up = cf.quadrotor_bounds(300)
    o = api.default_opts(hessian=2)
    xa, oa, ia, ta = _solve(p, lo, up, spec.nC, o)
    xb, ob, ib, tb = _solve(p, lo[:7], up[:7], spec.nC, o)
    x1, o1, i1, t1 = _solve(p, lo[6:7], up[6:7], spec.nC, o)
    assert np.array_equal(xa[:7], xb) and np.array_equal(oa[:7], ob) and np.array_equal(ta[:7], tb) and np.array_equal(ia[:7], ib)
    assert np.array_equal(xa[6:7], x1) and np.array_equal(ta[6:7], t1)


def test_full_request_evaluation_config_D():
    spec = cf.config_D(); p = api.Plan(spec, 0)
    x = np.random.default_rng(1).normal(size=(4096, spec.nC))
    ea = p.eval(dev(x), 2); eb = p.eval(dev(x[:5].copy()), 2)
    for k in ("f", "g", "c", "jband"):
        assert torch.equal(ea[k][:5], eb[k]), k


@pytest.mark.parametrize("hessian", [0, 1])
def test_wave_kernel_on_per_problem_grids(hessian):
    spec = cf.config_M(); p = api.Plan(spec, 0)
    kn, bp = grids_for(spec, 64, warp=0.2, seed=2)
    lo, up = cf.kincar_random_bounds(3, 64)
    o = api.default_opts(hessian=hessian)
    p.set_grids(dev(kn), dev(bp), with_precond=True)
    assert p.solve_kernel(64, o) == "sqp_wave_kernel"
    xa, oa, ia, ta = _solve(p, lo, up, spec.nC, o)
    p.set_grids(dev(kn[:3].copy()), dev(bp[:3].copy()), with_precond=True)
    xb, ob, ib, tb = _solve(p, lo[:3], up[:3], spec.nC, o)
    assert (ia == 0).all()
    assert np.array_equal(xa[:3], xb) and np.array_equal(oa[:3], ob) and np.array_equal(ta[:3], tb)
