"""-m gpu: the drop-in boundary.  A C program written against include/ntg.h (same call sequence
and problem data as the reference's examples) links against libntg_amd.so and must reach the
known optima with host callbacks + HIP kernels; npsolCostFunction is checked via ctypes."""
import os
import subprocess
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def drv(tmp_path_factory):
    d = tmp_path_factory.mktemp("drv")
    exe = d / "dropin_drv"
    subprocess.check_call(["gcc", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "drivers", "dropin_drv.c"),
                           "-o", str(exe), "-L", os.path.join(ROOT, "ntg_amd"), "-lntg_amd", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "ntg_amd")])
    return exe


def run(exe, which):
    out = subprocess.run([str(exe), which], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    lines = {l.split()[0]: l.split()[1:] for l in out.stdout.splitlines() if l.split() and l.split()[0] in ("RESULT", "INTERP", "ISTATE")}
    res = np.array(lines["RESULT"], dtype=float)
    return int(res[0]), res[1], res[2:], np.array(lines["INTERP"], dtype=float), [int(v) for v in lines["ISTATE"]]


def test_vanderpol_dropin(drv):
    inform, obj, coef, interp, istate = run(drv, "vanderpol")
    assert inform == 0
    assert abs(obj - 1.7022142628309958) <= 1e-9
    np.testing.assert_allclose(coef, [1, 1, 0.3937399093, -0.0369580060, -0.4395320819, -0.7168653229, -0.2449741945], atol=1e-6)
    np.testing.assert_allclose(interp[:2], [1.0, 0.0], atol=1e-8)          # z(0)=1, z'(0)=0
    assert abs(-interp[3] + interp[4] - 1.0) <= 1e-8                       # -z(5)+z'(5)=1
    assert istate == [3, 3, 3]


def test_kincar_dropin(drv):
    inform, obj, coef, interp, istate = run(drv, "kincar")
    assert inform == 0
    assert abs(obj - 2.457581141950512) <= 1e-9
    np.testing.assert_allclose(coef, [0, 5, 10, 20, 30, 35, 40, -2, -2, -2, 0, 2, 2, 2], atol=1e-6)
    np.testing.assert_allclose(interp, [0, 8, 0, 40, 8, 0], atol=1e-7)
    assert istate == [3] * 12


def test_ntg_called_repeatedly_in_one_process(drv):
    """ntg() three times in one process with two different shapes (SURVEY 8b 'Threading': sequential repeated calls, the MPC use):
    each call reaches its own optimum, the third equals the first bit for bit"""
    out = subprocess.run([str(drv), "sequence"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    res = [np.array(l.split()[1:], dtype=float) for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert len(res) == 3
    assert int(res[0][0]) == 0 and int(res[1][0]) == 0 and int(res[2][0]) == 0
    assert abs(res[0][1] - 1.7022142628309958) <= 1e-9 and abs(res[1][1] - 2.457581141950512) <= 1e-9
    assert len(res[0]) == 2 + 7 and len(res[1]) == 2 + 14
    assert np.array_equal(res[0], res[2])


def test_returned_R_factors_the_reduced_hessian(drv):
    """R (ntg.h:64-68): upper triangular, R'R = W^-1.  The shipped kincar problem is a QP with two degrees of freedom
    (14 coefficients, 12 equality rows): after its few majors the quasi-Newton matrix carries the exact curvature there,
    so Z'(R'R)Z must equal the reduced Hessian Z'HZ of the cost (H from central differences of the oracle's gradient)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc, scipy.linalg as sl
    from ntg_amd import configs as cf
    out = subprocess.run([str(drv), "kincarR"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    rl = [l for l in out.stdout.splitlines() if l.startswith("RMAT")][0].split()
    n = int(rl[1]); R = np.array(rl[2:], dtype=float).reshape(n, n)
    assert n == 14 and np.isfinite(R).all()
    assert np.abs(np.tril(R, -1)).max() <= 1e-12 * np.abs(R).max()          # upper triangular
    assert (np.diag(R) > 0).all()
    spec = cf.config_K0()
    X = np.vstack([np.zeros((1, n)), np.eye(n)])
    g = orc.eval_batch(spec, X, 2)["g"]
    H = g[1:] - g[0]; H = 0.5 * (H + H.T)                                    # the cost is quadratic: exact
    lo, up = cf.bounds_K0_shipped()
    Z = sl.null_space(orc.export_tables(spec, lo, up)["A"])
    Hr = Z.T @ H @ Z; Br = Z.T @ (R.T @ R) @ Z
    # here Z'HZ = 1.1948 I: the very first search direction is already the Newton direction, so the solve ends after the one
    # update it needs -- the factor carries the exact curvature along that step and the cold-start identity across it
    eh = np.linalg.eigvalsh(Hr); eb = np.linalg.eigvalsh(Br)
    assert abs(eb[-1] - eh[-1]) <= 1e-6 * eh[-1], (eb, eh)
    assert eb[0] >= 1.0 - 1e-9 and eb[0] <= eh[-1] * (1 + 1e-6)
    # outside null(A) the approximation was never touched: R'R acts as the identity on range(A')
    Pr = np.eye(n) - Z @ Z.T
    assert np.abs(Pr @ (R.T @ R) @ Pr - Pr).max() <= 1e-9


def test_exported_callbacks_all_slots_vs_oracle(tmp_path):
    """npsolCostFunction / npsolConstraintFunction (exported NPfunobj / NPfuncon) with host callbacks in
    all six slots, opened with ntg_open(): f, g, c and the dense column-major cJac against the oracle."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from ntg_amd import configs as cf
    spec = cf.config_T()
    exe = tmp_path / "callbacks_drv"
    subprocess.check_call(["gcc", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "drivers", "callbacks_drv.c"),
                           "-o", str(exe), "-L", os.path.join(ROOT, "ntg_amd"), "-lntg_amd", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "ntg_amd")])
    x = np.random.default_rng(3).normal(size=spec.nC)
    inp = tmp_path / "in.txt"
    with open(inp, "w") as f:
        for M in (spec.lic, spec.ltc, spec.lfc):
            f.write(" ".join("%.17g" % v for v in M.ravel()) + "\n")
        f.write(" ".join("%.17g" % v for v in x) + "\n")
    out = subprocess.run([str(exe), str(inp)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    L = {l.split()[0]: np.array(l.split()[1:], dtype=float) for l in out.stdout.splitlines() if l.split() and l.split()[0] in ("F", "G", "C", "J", "F0", "INFORM", "BADMODE")}
    ref = orc.eval_batch(spec, x[None], 2)
    tol = lambda r: 1e-12 * np.abs(r).max()
    assert abs(L["F"][0] - ref["f"][0]) <= 1e-12 * abs(ref["f"][0])
    assert abs(L["F0"][0] - ref["f"][0]) <= 1e-12 * abs(ref["f"][0])
    assert np.abs(L["G"] - ref["g"][0]).max() <= tol(ref["g"])
    assert np.abs(L["C"] - ref["c"][0]).max() <= tol(ref["c"])
    J = L["J"].reshape(spec.nC, spec.ncnln).T                     # column-major ldJ = ncnln
    assert np.array_equal(J != 0, ref["cJac"][0] != 0)
    assert np.abs(J - ref["cJac"][0]).max() <= tol(ref["cJac"])
    assert int(L["INFORM"][0]) in (0, 1)                          # all bounds are [-1, 1]: every constraint kind at once, solved
    assert int(L["BADMODE"][0]) == -1                              # unknown mode: nstate = -1 (ntg.c:332-333)


def test_obstacle_dropin_nonlinear_inequality(drv):
    """ntg() with a HOST nonlinear trajectory inequality callback: augmented-Lagrangian path of the drop-in
    against the oracle solving the same problem (family 3 == the driver's callback)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from ntg_amd import configs as cf
    inform, obj, coef, interp, istate = run(drv, "obstacle")
    spec = cf.config_O(10)
    zi = cf.kincar_flat_forward([0.0, -2.0, 0.0], [8.0, 0.0]).ravel(); zf = cf.kincar_flat_forward([40.0, 2.0, 0.0], [8.0, 0.0]).ravel()
    lo = np.concatenate([zi, zf, [9.0]]); up = np.concatenate([zi, zf, [1e20]])
    ref = orc.solve_one(spec, lo, up, np.ones(spec.nC))
    assert inform == 0 and ref["inform"] == 0
    assert abs(obj - ref["objective"]) <= 1e-6 * abs(ref["objective"])
    assert np.abs(coef - ref["x"]).max() <= 1e-4 * np.abs(ref["x"]).max()
    c = orc.eval_batch(spec, coef[None], 0)["c"][0]
    assert c.min() >= 9.0 * (1 - 1e-7)                             # stays outside the obstacle
    assert obj > 2.457581141950512                                 # and pays for it relative to the free lane change


def test_linear_inequality_rows_dropin(drv):
    """ntg() with linear INEQUALITY rows (lower < upper on two final-flag rows): against the oracle."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from ntg_amd import configs as cf
    inform, obj, coef, interp, istate = run(drv, "ineq")
    spec = cf.config_K0(); lo, up = cf.bounds_K0_shipped()
    lo = lo.copy(); up = up.copy()
    lo[9], up[9] = -1.0, 1.0; lo[11], up[11] = -0.5, 0.5
    ref = orc.solve_one(spec, lo, up, np.ones(spec.nC))
    assert inform in (0, 1) and ref["inform"] in (0, 1)
    assert abs(obj - ref["objective"]) <= 1e-7 * max(1.0, ref["objective"])
    assert np.abs(coef - ref["x"]).max() <= 1e-5 * np.abs(ref["x"]).max()
    assert obj < 2.457581141950512
    assert istate == list(ref["istate"][spec.nC:spec.nC + 12])
    assert abs(interp[3] - 40.0) <= 1e-7                            # x(T) is still an equality row
