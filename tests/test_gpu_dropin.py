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
