"""-m gpu: the structured Newton mode of sqp_kernel (ntg_solve_opts.hessian = 2, ntg_amd/csrc/newton.hpp) through the C ABI:
the band Cholesky / triangular solves on their own, the solve against the oracle's statement of the same algorithm, and
BASELINE configs D and E at full size against the committed golden solutions, their KKT conditions and the inform /
major-iteration histograms at the bench batch."""
import os
import subprocess
import numpy as np
import pytest
import torch

import orc
from ntg_amd import api, configs as cf
from gpu_common import dev

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def unit_exe():
    """tests/drivers/nwt_unit.hip: one wavefront factors and solves a random SPD band against a host Cholesky"""
    src = os.path.join(HERE, "drivers", "nwt_unit.hip"); exe = os.path.join(HERE, "drivers", "nwt_unit")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(HERE, "..", "include"),
                               "-Wno-unused-value", "-Wno-pass-failed", "-o", exe, src], cwd=os.path.join(HERE, "drivers"))
    return exe


@pytest.mark.parametrize("ng,hb,spread", [(40, 11, 0), (114, 11, 0), (531, 17, 0), (531, 17, 6), (616, 31, 0), (616, 31, 8), (50, 32, 0), (7, 5, 0)])
def test_band_cholesky_and_solves(unit_exe, ng, hb, spread):
    """factor (inverse diagonal) and L^-T L^-1 y to rounding level, including shapes that end inside a tile and badly scaled matrices"""
    r = subprocess.run([unit_exe, str(ng), str(hb), str(spread)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("ng,hb,spread", [(462, 23, 0), (462, 23, 6), (531, 17, 6), (616, 31, 3), (128, 11, 0), (200, 29, 0), (143, 17, 0)])
def test_two_sided_band_cholesky(unit_exe, ng, hb, spread):
    """two waves per group (nwt_factor_pairs / nwt_solve_pairs): top sweep, reversed bottom sweep, merged separator; separator widths 32 .. 47"""
    r = subprocess.run([unit_exe, str(ng), str(hb), str(spread), "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("row", [5, 200, 215, 240, 300, 461])   # top part, last top block, separator (208 .. 253), bottom part, last row
def test_indefinite_matrix_is_reported_by_both_factorisations(unit_exe, row):
    """a strict factorisation (the curvature attempt of a refresh) must report a non-positive pivot wherever it sits: in the top sweep,
    the reversed bottom sweep or the merged separator of the two-sided form, and in the one-sided routine"""
    r = subprocess.run([unit_exe, "462", "23", "0", "3", str(row)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


def _case(name):
    if name == "O":
        return cf.config_O(), cf.obstacle_bounds
    if name == "D2":
        return cf.config_D(ninterv=10), cf.quadrotor_bounds
    if name == "E2":
        return cf.config_E(ninterv=20, narms=2), lambda n: cf.manipulator_bounds(n, narms=2)
    if name == "D":
        return cf.config_D(), cf.quadrotor_bounds
    return cf.config_E(), cf.manipulator_bounds


@pytest.mark.parametrize("name,nb", [("O", 24), ("D2", 16), ("E2", 12)])
def test_newton_solve_matches_oracle(name, nb):
    """same algorithm on both sides: inform, major-iteration counts within 3, objective to 1e-9 (one problem: 1e-8), x to 1e-6"""
    spec, bounds = _case(name)
    p = api.Plan(spec, 0)
    lo, up = bounds(nb)
    x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    out = p.solve(dev(lo), dev(up), x, api.default_opts(hessian=2))
    torch.cuda.synchronize()
    ref = orc.solve_batch(spec, lo, up, np.ones((nb, spec.nC)), orc.default_opts(hessian=2), nthreads=8)
    inf = out["inform"].cpu().numpy(); it = out["iters"].cpu().numpy(); obj = out["objective"].cpu().numpy()
    # The device calls every problem of these batches optimal (inform 0).  The oracle ends one E2 problem with inform 1 ("optimal, not
    # to the requested accuracy": its last line search ran out of representable decrease at a different rounding) -- the counts are
    # explicit, and EVERY problem both sides call optimal is compared (round 2 allowed 10 % to drop out).
    ref_inform1 = {"O": 0, "D2": 0, "E2": 1}[name]
    assert (inf == 0).all(), inf
    assert np.isin(ref["inform"], (0, 1)).all() and int((ref["inform"] == 1).sum()) == ref_inform1, ref["inform"]
    ok = (inf == 0) & (ref["inform"] == 0)
    assert int(ok.sum()) == nb - ref_inform1
    assert np.abs(it - ref["iters"])[ok].max() <= 3, (it, ref["iters"])
    # objective: 1e-9 for all but one problem -- since the device solves the quadrotor's yaw output apart from (x, y, z) (a free output:
    # constant factor from the plan) the two sides no longer round alike, and one D2 problem whose last pass stops at a
    # violation of a few 1e-9 differs by 4.8e-9 in the objective (3.6e-8 in x; first order in the position along the active rows'
    # normals).  The full-size golden comparisons below keep 1e-9.
    dobj = np.abs(obj - ref["objective"]) / np.abs(ref["objective"])
    assert (dobj[ok] <= 1e-8).all() and int((dobj[ok] > 1e-9).sum()) <= 1, dobj
    assert np.abs(x.cpu().numpy() - ref["x"])[ok].max() <= 1e-6 * max(1.0, np.abs(ref["x"]).max())
    # and the mode is not the quasi-Newton mode in disguise: far fewer majors on the constrained problems
    x1 = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    out1 = p.solve(dev(lo), dev(up), x1, api.default_opts(hessian=1))
    assert it.mean() < out1["iters"].float().mean().item()


def _kkt(spec, p, x, lo, up, lam, stat_tol, feas_tol=1e-8):
    ev = p.eval(x, 2, want_dense_jac=True)
    g = ev["g"].cpu().numpy(); J = ev["cJac"].cpu().numpy(); c = ev["c"].cpu().numpy()
    A = p.tables()["A"]; xg = x.cpu().numpy(); P = spec.nbps
    nl0 = spec.lic.shape[0] + spec.ltc.shape[0] + spec.lfc.shape[0]
    stat = []
    for i in range(xg.shape[0]):
        ll, ln = lam[i, spec.nC:spec.nC + spec.nclin], lam[i, spec.nC + spec.nclin:]
        stat.append(np.abs(g[i] - A.T @ ll - J[i].T @ ln).max() / max(1.0, np.abs(g[i]).max()))
        rowscale = np.abs(A).max(axis=1) * max(1.0, np.abs(xg[i]).max())
        assert (np.abs(A @ xg[i] - lo[i][:spec.nclin]) <= 1e-9 * rowscale + 1e-9).all()
        for j in range(spec.nnltc):
            cj = c[i, j * P:(j + 1) * P]; lj = ln[j * P:(j + 1) * P]
            l, u = lo[i, nl0 + j], up[i, nl0 + j]
            assert cj.min() >= l - feas_tol * (1 + abs(l)) and cj.max() <= u + feas_tol * (1 + abs(u))
            inactive = (cj > l + 1e-5 * (1 + abs(l))) & (cj < u - 1e-5 * (1 + abs(u)))
            assert np.abs(lj[inactive]).max(initial=0.0) <= 1e-6 * max(1.0, np.abs(lj).max())
            assert lj[cj >= u - 1e-5 * (1 + abs(u))].max(initial=0.0) <= 1e-9 and lj[cj <= l + 1e-5 * (1 + abs(l))].min(initial=0.0) >= -1e-9
    stat = np.array(stat)
    # the stopping rule is NPSOL's |Z'g| <= sqrt(eps^0.8) (1 + max(1 + |F|, |g|)) = 5.5e-7 scaled; Newton steps end far below it
    assert stat.max() <= stat_tol, stat.max()
    return stat


@pytest.mark.parametrize("name", ["D", "E"])
def test_full_size_against_golden_solutions(name):
    """BASELINE.json sizes (nC 656 / 2196, 402 / 1204 nonlinear rows): x*, objective and multipliers of 8 problems against
    tests/golden/sol_{D,E}.npz (oracle, same algorithm): |dF| <= 1e-9 |F|, |dx| <= 1e-6, multipliers to 1e-4 of their scale"""
    gold = np.load(os.path.join(GOLD, f"sol_{name}.npz"))
    spec, _ = _case(name)
    p = api.Plan(spec, 0)
    lo, up = gold["lower"], gold["upper"]
    nb = lo.shape[0]
    x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    out = p.solve(dev(lo), dev(up), x, api.default_opts(hessian=2), want_lambda=True)
    torch.cuda.synchronize()
    inf = out["inform"].cpu().numpy(); obj = out["objective"].cpu().numpy(); lam = out["clambda"].cpu().numpy()
    assert (inf == 0).all()
    assert np.abs(out["iters"].cpu().numpy() - gold["iters"]).max() <= 3
    assert (np.abs(obj - gold["objective"]) <= 1e-9 * np.abs(gold["objective"])).all()
    assert np.abs(x.cpu().numpy() - gold["x"]).max() <= 1e-6 * max(1.0, np.abs(gold["x"]).max())
    nl = slice(spec.nC + spec.nclin, None)
    assert np.abs(lam[:, nl] - gold["clambda"][:, nl]).max() <= 1e-4 * max(1.0, np.abs(gold["clambda"][:, nl]).max())
    stat = _kkt(spec, p, x, lo, up, lam, 6e-7)
    assert np.median(stat) <= 5e-8


@pytest.mark.parametrize("name,batch,maj_mean,maj_max", [("D", 512, 20, 50), ("E", 1024, 70, 120)])
def test_bench_batch_inform_histogram(name, batch, maj_mean, maj_max):
    """the batch bench.py runs (4096 / 8 and 8192 / 8 problems): inform 0 for >= 99 %, nothing but 0 / 1, majors bounded; KKT for a sample"""
    spec, bounds = _case(name)
    p = api.Plan(spec, 0)
    lo, up = bounds(batch)
    x = torch.ones((batch, spec.nC), dtype=torch.float64, device="cuda:0")
    out = p.solve(dev(lo), dev(up), x, api.default_opts(hessian=2), want_lambda=True)
    torch.cuda.synchronize()
    inf = out["inform"].cpu().numpy(); it = out["iters"].cpu().numpy()
    assert np.isin(inf, (0, 1)).all(), np.bincount(inf)
    assert (inf == 0).mean() >= 0.99, np.bincount(inf)
    assert it.mean() <= maj_mean and it.max() <= maj_max, (it.mean(), it.max())
    sel = np.arange(0, batch, batch // 8)[:8]
    lam = out["clambda"].cpu().numpy()
    _kkt(spec, p, x[sel].contiguous(), lo[sel], up[sel], lam[sel], 6e-7)
    nlam = lam[:, spec.nC + spec.nclin:]
    assert (np.abs(nlam).max(axis=1) > 1e-8).mean() > 0.3   # the constraints matter for a good share of the batch


def test_newton_mode_falls_back_like_the_oracle():
    """hessian = 2 on a plan that does not qualify (kincar: no nonlinear rows) is hessian = 1, bit for bit"""
    spec = cf.config_B()
    p = api.Plan(spec, 0)
    lo, up = cf.kincar_random_bounds(1, 8)
    xa = torch.ones((8, spec.nC), dtype=torch.float64, device="cuda:0"); xb = xa.clone()
    oa = p.solve(dev(lo), dev(up), xa, api.default_opts(hessian=2)); ob = p.solve(dev(lo), dev(up), xb, api.default_opts(hessian=1))
    assert torch.equal(xa, xb) and torch.equal(oa["iters"], ob["iters"])
