"""-m gpu: the batched SQP kernel (replacement of the npsol_ call, ntg.c:250) against the oracle
and against closed-form optima, through the C ABI.

Tolerances: at the optimum BASELINE.md §2 (|dF| <= 1e-9 relative to max(1,|F|), |dC|inf <= 1e-6
scaled by max(1,|C|inf), linear feasibility <= 1e-8).  For a fixed number of majors the GPU and
the oracle run the same algorithm with different summation orders; iterates agree to 1e-7 relative
in F after 50 majors (quasi-Newton recurrences amplify rounding), counts of majors/evaluations
are exact."""
import numpy as np
import pytest
import torch

import orc
from ntg_amd import api, configs as cf
from gpu_common import plan_for, dev, rel, same_work
from test_oracle_known_answers import kkt_kincar

pytestmark = pytest.mark.gpu


def solve(name, lo, up, x0, **kw):
    p = plan_for(name)
    x = dev(x0)
    out = p.solve(dev(lo), dev(up), x, api.default_opts(**kw), want_lambda=True)
    torch.cuda.synchronize()
    return x.cpu().numpy(), {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("hessian", [0, 1])
def test_kincar_shipped_known_answer(hessian):
    spec = cf.config_K0(); lo, up = cf.bounds_K0_shipped()
    x, out = solve("K0", lo[None], up[None], np.ones((1, spec.nC)), hessian=hessian)
    assert out["inform"][0] == 0
    assert abs(out["objective"][0] - 2.457581141950512) <= 1e-9
    np.testing.assert_allclose(x[0], [0, 5, 10, 20, 30, 35, 40, -2, -2, -2, 0, 2, 2, 2], atol=1e-6)


@pytest.mark.parametrize("hessian", [0, 1])
def test_vanderpol_known_answer(hessian):
    spec = cf.config_A(); lo, up = cf.bounds_A()
    x, out = solve("A", lo[None], up[None], np.ones((1, spec.nC)), hessian=hessian)
    assert out["inform"][0] == 0
    assert abs(out["objective"][0] - 1.7022142628309958) <= 1e-9
    np.testing.assert_allclose(x[0], [1, 1, 0.3937399093, -0.0369580060, -0.4395320819, -0.7168653229, -0.2449741945], atol=1e-6)
    o = orc.solve_one(spec, lo, up, np.ones(spec.nC), orc.default_opts(hessian=hessian))
    assert out["iters"][0] == o["iters"] and out["nfev"][0] == o["nfev"]


@pytest.mark.parametrize("name,ncars", [("B", 1), ("M", 3), ("M4", 2), ("M4b", 2)])
@pytest.mark.parametrize("hessian", [0, 1])
def test_optimum_matches_kkt_and_oracle(name, ncars, hessian):
    spec = plan_for(name).spec
    nb = 24
    lo, up = cf.kincar_random_bounds(ncars, nb)
    x, out = solve(name, lo, up, np.ones((nb, spec.nC)), hessian=hessian)
    ref = orc.solve_batch(spec, lo, up, np.ones((nb, spec.nC)), orc.default_opts(hessian=hessian), nthreads=8)
    assert (out["inform"] == 0).all()
    t = plan_for(name).tables()
    for p in range(nb):
        if p < 4:
            xs, fs = kkt_kincar(spec, lo[p])
            assert abs(out["objective"][p] - fs) <= 1e-9 * max(1.0, abs(fs))
            assert np.abs(x[p] - xs).max() <= 1e-6 * max(1.0, np.abs(xs).max())
        assert abs(out["objective"][p] - ref["objective"][p]) <= 1e-9 * max(1.0, abs(ref["objective"][p]))
        assert np.abs(x[p] - ref["x"][p]).max() <= 1e-6 * max(1.0, np.abs(ref["x"][p]).max())
        assert np.abs(t["A"] @ x[p] - lo[p]).max() <= 1e-8
    if hessian == 1:
        assert out["iters"].max() <= 5
        # majors agree (+-1 at the rounding-level exit test); the evaluation count of the LAST line search (decrease at rounding
        # level next to the optimum) legitimately depends on summation order
        assert np.abs(out["iters"] - ref["iters"]).max() <= 1


@pytest.mark.parametrize("name,ncars", [("B", 1), ("M", 3), ("M4", 2), ("M4b", 2)])
def test_fixed_50_majors_parity_with_oracle(name, ncars):
    """The benchmark mode: exactly 50 majors, identity cold start."""
    spec = plan_for(name).spec
    nb = 16
    lo, up = cf.kincar_random_bounds(ncars, nb)
    x, out = solve(name, lo, up, np.ones((nb, spec.nC)), itlim=50, fixed_iters=1)
    ref = orc.solve_batch(spec, lo, up, np.ones((nb, spec.nC)), orc.default_opts(itlim=50, fixed_iters=1), nthreads=8)
    assert (out["iters"] == 50).all() and (out["inform"] == 4).all()
    assert same_work(out["nfev"], ref["nfev"])
    assert rel(out["objective"], ref["objective"]) <= 1e-7
    assert np.abs(x - ref["x"]).max() <= 1e-5 * np.abs(ref["x"]).max()


@pytest.mark.parametrize("itlim,memory", [(30, 0), (70, 0), (50, 20)])
def test_fixed_majors_parity_in_every_history_form(itlim, memory):
    """The device keeps the quasi-Newton operator in three forms, all the matrix of DESIGN 4a.4: one stored direction per major with the
    link scalars in LDS (short runs: itlim <= memory < 64; the headline mode), pairs (s, u) with (rho, c2) in LDS (memory < 64 < itlim:
    restarts when full), pairs with their scalars in HBM (longer memories).  Same iterates as the oracle's dense W in each."""
    spec = plan_for("M").spec
    nb = 8
    lo, up = cf.kincar_random_bounds(3, nb)
    kw = dict(itlim=itlim, fixed_iters=1, qn_memory=memory)
    x, out = solve("M", lo, up, np.ones((nb, spec.nC)), **kw)
    ref = orc.solve_batch(spec, lo, up, np.ones((nb, spec.nC)), orc.default_opts(**kw), nthreads=8)
    assert (out["iters"] == itlim).all()
    assert same_work(out["nfev"], ref["nfev"])
    assert rel(out["objective"], ref["objective"]) <= 1e-7
    assert np.abs(x - ref["x"]).max() <= 1e-5 * np.abs(ref["x"]).max()


def test_multipliers_and_feasibility_outputs():
    spec = cf.config_K0(); lo, up = cf.bounds_K0_shipped()
    x, out = solve("K0", lo[None], up[None], np.ones((1, spec.nC)))
    o = orc.solve_one(spec, lo, up, np.ones(spec.nC))
    lam = out["clambda"][0]
    assert np.all(lam[:spec.nC] == 0)
    np.testing.assert_allclose(lam[spec.nC:], o["clambda"][spec.nC:], rtol=1e-6, atol=1e-8)


def test_unsupported_inputs_are_loud_not_wrong():
    spec = cf.config_K0(); lo, up = cf.bounds_K0_shipped()
    up2 = up.copy(); up2[3] += 0.5
    x, out = solve("K0", np.stack([lo, lo]), np.stack([up, up2]), np.ones((2, spec.nC)))
    assert out["inform"][0] == 0 and out["inform"][1] == 9       # inequality: flagged, per problem
    assert np.array_equal(x[1], np.ones(spec.nC))                 # and left untouched


def test_iteration_limit_and_ragged_batch():
    spec = cf.config_B()
    for nb in (1, 3, 65):
        lo, up = cf.kincar_random_bounds(1, nb)
        x, out = solve("B", lo, up, np.ones((nb, spec.nC)), itlim=7)
        assert (out["iters"] == 7).all() and (out["inform"] == 4).all()


def test_full_batch_properties_config_M():
    """BASELINE size (4096 x config M): size-independent properties instead of an oracle run:
    linear feasibility of every solution, KKT stationarity of the projected gradient,
    idempotence (re-solving from the solution does not move it) and batch-order invariance."""
    p = plan_for("M"); spec = p.spec
    nb = 4096
    lo, up = cf.kincar_random_bounds(3, nb)
    lo_d, up_d = dev(lo), dev(up)
    x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    out = p.solve(lo_d, up_d, x, api.default_opts(hessian=1))
    assert int((out["inform"] != 0).sum()) == 0
    A = dev(p.tables()["A"])
    feas = (x @ A.T - lo_d).abs().max().item()
    assert feas <= 1e-8
    ev = p.eval(x, 2)
    g = ev["g"]
    lam = torch.linalg.solve(A @ A.T, A @ g.T)                   # projected gradient must vanish
    pg = g - (A.T @ lam).T
    assert (pg.norm(dim=1) / (1 + g.norm(dim=1))).max().item() <= 1e-6
    # idempotence
    x2 = x.clone()
    out2 = p.solve(lo_d, up_d, x2, api.default_opts(hessian=1))
    assert (out2["inform"] == 0).all() and (x2 - x).abs().max().item() <= 1e-7 * x.abs().max().item()
    # order invariance: a permuted batch gives the permuted, bit-identical results
    perm = torch.randperm(nb, device="cuda:0")
    xp = torch.ones_like(x)
    p.solve(lo_d[perm].contiguous(), up_d[perm].contiguous(), xp, api.default_opts(hessian=1))
    assert torch.equal(xp, x[perm])


def test_c_abi_error_returns():
    """Bad arguments come back as negative NTG_E_* codes with a message -- never a crash, never a silent result."""
    import ctypes as C
    import copy
    L = api.lib()
    spec = cf.config_K0()
    # plan creation: unknown family, maxderiv that the family does not have, bad spline spec
    for mutate, what in ((lambda s: setattr(s, "family", 77), "family"),
                         (lambda s: setattr(s, "maxderiv", [2, 2]), "maxderiv"),
                         (lambda s: setattr(s, "order", [11, 11]), "order")):
        s2 = copy.deepcopy(spec); mutate(s2)
        if what == "maxderiv":
            s2.lic = np.eye(4); s2.lfc = np.eye(4); s2.tcostav = [(0, 1), (1, 1)]
        with pytest.raises(api.NtgError):
            api.Plan(s2, 0)
    with pytest.raises(api.NtgError):
        api.Plan(spec, 99)                                         # no such device
    p = plan_for("K0")
    lo, up = cf.bounds_K0_shipped()
    lo_d, up_d = dev(lo[None]), dev(up[None])
    x = torch.ones((1, spec.nC), dtype=torch.float64, device="cuda:0")
    o = api.default_opts()
    small = torch.empty(16, dtype=torch.uint8, device="cuda:0")
    rc = L.ntg_batch_solve(p.h, 1, lo_d.data_ptr(), up_d.data_ptr(), x.data_ptr(), C.byref(o), None, None, None, None, None,
                           small.data_ptr(), 16, None)
    assert rc == -2 and b"workspace" in L.ntg_last_error()         # NTG_E_BADARG
    rc = L.ntg_batch_solve(p.h, 1, None, up_d.data_ptr(), x.data_ptr(), C.byref(o), None, None, None, None, None, None, 0, None)
    assert rc == -2
    assert L.ntg_batch_solve(p.h, 0, None, None, None, None, None, None, None, None, None, None, 0, None) == 0   # empty batch: nothing to do
    rc = L.ntg_batch_eval(p.h, 1, x.data_ptr(), 5, None, None, None, None, None, None)
    assert rc == -2 and b"mode" in L.ntg_last_error()
    assert torch.equal(x, torch.ones_like(x))                      # nothing was touched


@pytest.mark.parametrize("name,ncars", [("B", 1), ("M4", 2), ("M4b", 2), ("M", 3)])
def test_wave_kernel_is_what_runs_and_agrees_with_the_workgroup_kernel(name, ncars):
    """The kincar class is solved by sqp_wave_kernel (one wavefront per problem, solve_wave.hpp); NTG_AMD_NOWAVE=1 forces sqp_kernel.
    Same algorithm, different summation orders: identical evaluation counts in the fixed-work mode, objectives to 1e-7 (both are
    within 1e-7 of the oracle, test_fixed_50_majors_parity_with_oracle), optima to 1e-9 / 1e-6; every other plan reports sqp_kernel."""
    import os
    p = plan_for(name); spec = p.spec
    nb = 37                                                      # not a multiple of the waves per workgroup
    lo, up = cf.kincar_random_bounds(ncars, nb)
    fixed, conv = dict(itlim=50, fixed_iters=1), dict(hessian=1)
    assert p.solve_kernel(nb, api.default_opts(**fixed)) == "sqp_wave_kernel" and p.solve_kernel(nb, api.default_opts(**conv)) == "sqp_wave_kernel"
    assert plan_for("K0").solve_kernel(nb, api.default_opts()) == "sqp_kernel"      # order 5, 2 intervals: outside the class
    res = {}
    for mode in ("wave", "wg"):
        if mode == "wg":
            os.environ["NTG_AMD_NOWAVE"] = "1"
        try:
            res[mode] = (solve(name, lo, up, np.ones((nb, spec.nC)), **fixed), solve(name, lo, up, np.ones((nb, spec.nC)), **conv))
        finally:
            os.environ.pop("NTG_AMD_NOWAVE", None)
    (xf_w, of_w), (xc_w, oc_w) = res["wave"]
    (xf_g, of_g), (xc_g, oc_g) = res["wg"]
    assert same_work(of_w["nfev"], of_g["nfev"]) and (of_w["iters"] == 50).all()
    assert rel(of_w["objective"], of_g["objective"]) <= 2e-7
    assert (oc_w["inform"] == 0).all() and (oc_g["inform"] == 0).all()
    assert rel(oc_w["objective"], oc_g["objective"]) <= 1e-9
    assert np.abs(xc_w - xc_g).max() <= 1e-6 * np.abs(xc_g).max()
    # multipliers of the equality rows (final pass at x) agree as well
    np.testing.assert_allclose(oc_w["clambda"][:, spec.nC:], oc_g["clambda"][:, spec.nC:], rtol=1e-6, atol=1e-7 * np.abs(oc_g["clambda"]).max())


@pytest.mark.parametrize("itlim,memory", [(70, 8), (40, 3)])
def test_wave_kernel_restarts_with_a_short_memory(itlim, memory):
    """Several restarts of the quasi-Newton memory inside one solve (each stores its first pair as two chain slots):
    same iterates as the oracle's dense W restarted at the same counts."""
    spec = plan_for("M").spec
    nb = 6
    lo, up = cf.kincar_random_bounds(3, nb)
    kw = dict(itlim=itlim, fixed_iters=1, qn_memory=memory)
    assert plan_for("M").solve_kernel(nb, api.default_opts(**kw)) == "sqp_wave_kernel"
    x, out = solve("M", lo, up, np.ones((nb, spec.nC)), **kw)
    ref = orc.solve_batch(spec, lo, up, np.ones((nb, spec.nC)), orc.default_opts(**kw), nthreads=8)
    assert (out["iters"] == itlim).all()
    assert same_work(out["nfev"], ref["nfev"])
    assert rel(out["objective"], ref["objective"]) <= 1e-7


def test_wave_kernel_cold_start_to_convergence_and_outputs():
    """NPSOL-equivalent cold start run to convergence on the wave kernel (the long-memory instance): optimum, multipliers of the
    equality rows, feasibility against the oracle; a problem whose bounds are not equalities is refused per problem (inform 9)."""
    spec = plan_for("B").spec
    nb = 5
    lo, up = cf.kincar_random_bounds(1, nb)
    up2 = up.copy(); up2[3, 2] += 0.25                                   # problem 3: a range where the plan has an equality row
    x, out = solve("B", lo, up2, np.ones((nb, spec.nC)), hessian=0)
    ref = orc.solve_batch(spec, lo, up, np.ones((nb, spec.nC)), orc.default_opts(hessian=0), nthreads=8)
    good = np.array([0, 1, 2, 4])
    assert (out["inform"][good] == 0).all() and out["inform"][3] == 9
    assert np.array_equal(x[3], np.ones(spec.nC)) and (out["clambda"][3] == 0).all()
    assert rel(out["objective"][good], ref["objective"][good]) <= 1e-9
    assert np.abs(x[good] - ref["x"][good]).max() <= 1e-6 * np.abs(ref["x"]).max()
    A = plan_for("B").tables()["A"]
    assert np.abs(x[good] @ A.T - lo[good]).max() <= 1e-8
    for i in good[:2]:
        o = orc.solve_one(spec, lo[i], up[i], np.ones(spec.nC), orc.default_opts(hessian=0))
        lam = out["clambda"][i]
        assert np.all(lam[:spec.nC] == 0)
        # (first-order quantities of two solutions that agree to 1e-6 in x: the cost's curvature amplifies the difference)
        np.testing.assert_allclose(lam[spec.nC:], o["clambda"][spec.nC:], rtol=1e-3, atol=1e-4 * np.abs(o["clambda"]).max())


def test_fallback_instance_without_register_slots():
    """NTG_AMD_WAVE_NOAGPR=1 selects the instance that keeps no chain slot in the accumulator registers (what a build whose accumulator base
    was raised to 256 by ntg_amd/build.py would run): the same iteration (identical evaluation counts; the tiers add the chain's terms in
    different orders, so objectives agree to rounding, like the wave kernel and the workgroup kernel do)."""
    import os
    spec = plan_for("M").spec
    nb = 9
    lo, up = cf.kincar_random_bounds(3, nb)
    kw = dict(itlim=50, fixed_iters=1)
    x0, o0 = solve("M", lo, up, np.ones((nb, spec.nC)), **kw)
    os.environ["NTG_AMD_WAVE_NOAGPR"] = "1"
    try:
        assert plan_for("M").solve_kernel(nb, api.default_opts(**kw)) == "sqp_wave_kernel"
        x1, o1 = solve("M", lo, up, np.ones((nb, spec.nC)), **kw)
    finally:
        os.environ.pop("NTG_AMD_WAVE_NOAGPR", None)
    assert same_work(o0["nfev"], o1["nfev"]) and (o1["iters"] == 50).all()
    assert rel(o0["objective"], o1["objective"]) <= 2e-7
    assert np.abs(x0 - x1).max() <= 1e-5 * np.abs(x0).max()


@pytest.mark.parametrize("kw", [dict(itlim=50, fixed_iters=1, hessian=0), dict(itlim=60, fixed_iters=1, hessian=0, qn_memory=20), dict(hessian=0)],
                         ids=["fixed50", "fixed60_memory20_restarts", "to_convergence"])
def test_fat_wave_instance_takes_a_second_problem_per_wave(kw):
    """The headline instance (identity cold start: chain tiers in accumulator registers, LDS and HBM) with more problems than resident waves
    (4 per CU): a persistent wave pops a SECOND and third problem from the queue and must leave nothing of the first behind (chain slots,
    links, head pair, line-search buffers; the queue pop is where round 3's first GPU run hung).  A problem's result may not depend on
    which wave solved it after what: bit-identical to the same problems in a batch of 8, and under a permutation of the batch.
    (the suite's wave-kernel batches were all smaller than the resident waves.)"""
    spec = plan_for("M").spec
    nb = 1536
    lo, up = cf.kincar_random_bounds(3, nb)
    assert plan_for("M").solve_kernel(nb, api.default_opts(**kw)) == "sqp_wave_kernel"
    xb, ob = solve("M", lo, up, np.ones((nb, spec.nC)), **kw)
    xs, os_ = solve("M", lo[:8], up[:8], np.ones((8, spec.nC)), **kw)
    assert np.array_equal(xb[:8], xs) and np.array_equal(ob["nfev"][:8], os_["nfev"]) and np.array_equal(ob["objective"][:8], os_["objective"])
    perm = np.random.default_rng(5).permutation(nb)
    xp, op = solve("M", lo[perm], up[perm], np.ones((nb, spec.nC)), **kw)
    assert np.array_equal(xp, xb[perm]) and np.array_equal(op["iters"], ob["iters"][perm]) and np.array_equal(op["inform"], ob["inform"][perm])
    if not kw.get("fixed_iters"):
        assert (ob["inform"] == 0).all()
