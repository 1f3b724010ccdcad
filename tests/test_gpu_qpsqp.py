"""-m gpu: the QP-based SQP step on the band model (ntg_solve_opts.hessian = 3; csrc/qpdual.hpp, sqp_kernel<..., QPM>) through
the C ABI -- what NPSOL does with the Jacobian the reference hands it (ntg.c:217-220,250-253; constraints.c:120-162): against the oracle's
statement of the same algorithm (oracle/sqp.c sqpqp_run) on reduced grids, against committed oracle solutions at BASELINE's sizes, through
the KKT conditions at the bench batches, and the rule that a problem whose working set does not fit continues in the Newton mode."""
import os
import numpy as np
import pytest
import torch

import orc
from ntg_amd import api, configs as cf
from gpu_common import dev
from test_gpu_newton import _case, _kkt

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _solve(spec, lo, up, hessian=3, want_lambda=False):
    p = api.Plan(spec, 0)
    nb = lo.shape[0]
    x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    out = p.solve(dev(lo), dev(up), x, api.default_opts(hessian=hessian), want_lambda=want_lambda)
    torch.cuda.synchronize()
    return p, x, out


@pytest.mark.parametrize("name,nb", [("O", 24), ("D2", 16), ("E2", 12)])
def test_qp_sqp_matches_oracle(name, nb):
    """same algorithm on both sides: every problem inform 0, the SAME number of major iterations (the active-set QPs have unique solutions, so
    the two paths agree to rounding), objective to 1e-9, x to 1e-6 -- and far fewer majors than the augmented-Lagrangian Newton mode"""
    spec, bounds = _case(name)
    lo, up = bounds(nb)
    p, x, out = _solve(spec, lo, up)
    ref = orc.solve_batch(spec, lo, up, np.ones((nb, spec.nC)), orc.default_opts(hessian=3), nthreads=8)
    inf = out["inform"].cpu().numpy(); it = out["iters"].cpu().numpy(); obj = out["objective"].cpu().numpy()
    assert (inf == 0).all() and (ref["inform"] == 0).all(), (inf, ref["inform"])
    assert np.abs(it - ref["iters"]).max() <= 1, (it, ref["iters"])
    assert (np.abs(obj - ref["objective"]) <= 1e-9 * np.maximum(1.0, np.abs(ref["objective"]))).all(), np.abs(obj - ref["objective"]).max()
    assert np.abs(x.cpu().numpy() - ref["x"]).max() <= 1e-6
    _, _, out2 = _solve(spec, lo, up, hessian=2)
    assert it.mean() <= 0.5 * out2["iters"].float().mean().item()
    assert out["nfev"].float().mean().item() <= 0.5 * out2["nfev"].float().mean().item()


@pytest.mark.parametrize("name", ["D", "E"])
def test_qp_sqp_full_size_against_golden_solutions(name):
    """BASELINE.json sizes against tests/golden/sol_qp_{D,E}.npz (oracle, hessian = 3; tests/golden/make_solutions.py qp): objective 1e-9, x 1e-6,
    multipliers 1e-4 of their scale, KKT conditions with the reported multipliers.  Config E follows the oracle's path major by major
    (measured: objective 2e-13, x 5e-12); config D reaches the unconstrained optimum (phase 0) in one major fewer than the oracle (the yaw
    output is solved apart), the QP majors then agree one to one (measured: objective 2e-10)."""
    gold = np.load(os.path.join(GOLD, f"sol_qp_{name}.npz"))
    spec, _ = _case(name)
    lo, up = gold["lower"], gold["upper"]
    p, x, out = _solve(spec, lo, up, want_lambda=True)
    inf = out["inform"].cpu().numpy(); obj = out["objective"].cpu().numpy(); lam = out["clambda"].cpu().numpy()
    assert (inf == 0).all() and (gold["inform"] == 0).all()
    assert np.abs(out["iters"].cpu().numpy() - gold["iters"]).max() <= 1
    tol = 1e-9
    assert (np.abs(obj - gold["objective"]) <= tol * np.abs(gold["objective"])).all(), (np.abs(obj - gold["objective"]) / np.abs(gold["objective"])).max()
    assert np.abs(x.cpu().numpy() - gold["x"]).max() <= 1e-6 * max(1.0, np.abs(gold["x"]).max())
    nl = slice(spec.nC + spec.nclin, None)
    assert np.abs(lam[:, nl] - gold["clambda"][:, nl]).max() <= 1e-4 * max(1.0, np.abs(gold["clambda"][:, nl]).max())
    stat = _kkt(spec, p, x, lo, up, lam, 6e-7)
    assert np.median(stat) <= 5e-8


@pytest.mark.parametrize("name,batch,maj_mean,maj_max,fell_max", [("E", 1024, 20, 100, 0.01), ("O", 2048, 8, 40, 0.01), ("D", 512, 8, 50, 0.05)])
def test_qp_sqp_bench_batches(name, batch, maj_mean, maj_max, fell_max):
    """the batches bench.py runs: inform 0 for >= 99 % (config E: every problem), nothing but 0 / 1; KKT conditions of a sample with the reported
    multipliers; the share of problems that left the mode for the augmented-Lagrangian passes (working set full / no acceptable step) is
    small -- config D has 32 slots per group for its rows, which stay active along arcs of the trajectory"""
    spec, bounds = _case(name)
    lo, up = bounds(batch)
    p, x, out = _solve(spec, lo, up, want_lambda=True)
    inf = out["inform"].cpu().numpy(); it = out["iters"].cpu().numpy()
    assert np.isin(inf, (0, 1)).all(), np.bincount(inf)
    # (config D: the problems that continue in the Newton mode end like that mode ends them -- a few at inform 1, "optimal, not to the requested accuracy")
    assert (inf == 0).mean() >= {"E": 1.0, "O": 0.99, "D": 0.98}[name], np.bincount(inf)
    assert it.mean() <= maj_mean and it.max() <= maj_max, (it.mean(), it.max())
    sel = np.arange(0, batch, batch // 8)[:8]
    lam = out["clambda"].cpu().numpy()
    _kkt(spec, p, x[sel].contiguous(), lo[sel], up[sel], lam[sel], 6e-7)
    os.environ["NTG_AMD_STAMPS"] = "3"   # the kernel's work counters in place of the multipliers
    try:
        x2 = torch.ones((batch, spec.nC), dtype=torch.float64, device="cuda:0")
        o2 = p.solve(dev(lo), dev(up), x2, api.default_opts(hessian=3), want_lambda=True)
        torch.cuda.synchronize()
    finally:
        del os.environ["NTG_AMD_STAMPS"]
    cnt = o2["clambda"][:, :10].cpu().numpy()
    assert (cnt[:, 9] > 0).mean() <= fell_max, (cnt[:, 9] > 0).mean()
    assert torch.equal(x2, x)   # (and the solve is deterministic)


def test_full_working_set_continues_in_the_newton_mode():
    """a problem the QP step cannot finish (a working set larger than the group's slots, or -- config D's case now that it has 32 slots -- a step the
    l1 merit function does not accept in 25 halvings) must not stall or end at inform 6: it continues with the augmented-Lagrangian passes from
    where it is and ends at the same optimum as the Newton mode alone"""
    spec, bounds = _case("D")
    lo, up = bounds(512)
    p, x, out = _solve(spec, lo, up)
    os.environ["NTG_AMD_STAMPS"] = "3"
    try:
        x2 = torch.ones((512, spec.nC), dtype=torch.float64, device="cuda:0")
        o2 = p.solve(dev(lo), dev(up), x2, api.default_opts(hessian=3), want_lambda=True)
        torch.cuda.synchronize()
    finally:
        del os.environ["NTG_AMD_STAMPS"]
    fell = (o2["clambda"][:, 9] > 0).cpu().numpy()
    assert fell.sum() >= 1   # the batch holds such problems (tests the rule, not the luck of the draw)
    _, xn, outn = _solve(spec, lo, up, hessian=2)
    inf = out["inform"].cpu().numpy(); infn = outn["inform"].cpu().numpy()
    both = fell & (inf == 0) & (infn == 0)
    assert both.sum() >= 1
    obj = out["objective"].cpu().numpy(); objn = outn["objective"].cpu().numpy()
    assert (np.abs(obj[both] - objn[both]) <= 1e-6 * np.abs(objn[both])).all()
    assert np.isin(inf[fell], (0, 1)).all()


def test_qp_mode_falls_back_where_the_band_model_does_not_apply():
    """hessian = 3 on a plan without the structured Newton mode's band model (kincar: no nonlinear rows) is hessian = 1, bit for bit"""
    spec = cf.config_B()
    p = api.Plan(spec, 0)
    lo, up = cf.kincar_random_bounds(1, 8)
    xa = torch.ones((8, spec.nC), dtype=torch.float64, device="cuda:0"); xb = xa.clone()
    oa = p.solve(dev(lo), dev(up), xa, api.default_opts(hessian=3)); ob = p.solve(dev(lo), dev(up), xb, api.default_opts(hessian=1))
    assert torch.equal(xa, xb) and torch.equal(oa["iters"], ob["iters"])


@pytest.mark.parametrize("name,nb", [("O", 40), ("E2", 12)])
def test_qp_sqp_result_does_not_depend_on_the_batch(name, nb):
    """one workgroup per problem, nothing shared but the plan's tables: a problem's result is bit-identical whether it is solved alone, in
    a batch, or in a permuted batch (slots, row caches and columns are per problem; the working set is rebuilt in row order)"""
    spec, bounds = _case(name)
    lo, up = bounds(nb)
    p, x, out = _solve(spec, lo, up)
    perm = np.random.default_rng(11).permutation(nb)
    _, xp, outp = _solve(spec, lo[perm], up[perm])
    assert torch.equal(xp, x[torch.as_tensor(perm, device=x.device)])
    assert torch.equal(outp["iters"], out["iters"][torch.as_tensor(perm, device=x.device)])
    _, x1, out1 = _solve(spec, lo[3:4], up[3:4])
    assert torch.equal(x1[0], x[3]) and int(out1["iters"][0]) == int(out["iters"][3])
