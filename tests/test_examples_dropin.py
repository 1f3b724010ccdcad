"""The reference's example programs, UNCHANGED, compiled against include/ntg.h.

CPU suite: link against the oracle's ABI library and check the printed optimum (this pins the
oracle on the only executable tests the reference has).  The product library is exercised with
the same programs on the GPU box via its own restated drivers (tests/test_gpu_dropin.py),
because /root/reference does not travel.
"""
import os
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_EX = "/root/reference/examples"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_EX), reason="reference tree not present (GPU box)")


def build(tmp_path, name):
    import orc
    orc.build()
    exe = tmp_path / name
    subprocess.check_call(["gcc", "-O1", "-w", "-I", os.path.join(ROOT, "include"), os.path.join(REF_EX, name + ".c"),
                           "-o", str(exe), "-L", os.path.join(ROOT, "oracle"), "-lorc_ntg", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return exe


@pytest.mark.parametrize("name", ["vanderpol", "kincar"])
def test_examples_link_against_the_product_library(tmp_path, name):
    """the unchanged example sources also compile against include/ntg.h and LINK against libntg_amd.so (the product; it needs
    a GPU to run, so this container only links): every symbol the examples use is exported with a compatible prototype"""
    lib = os.path.join(ROOT, "ntg_amd", "libntg_amd.so")
    if not os.path.exists(lib):
        import __graft_entry__ as ge
        ge.build()
    exe = tmp_path / (name + "_amd")
    subprocess.check_call(["gcc", "-O1", "-w", "-I", os.path.join(ROOT, "include"), os.path.join(REF_EX, name + ".c"),
                           "-o", str(exe), "-L", os.path.join(ROOT, "ntg_amd"), "-lntg_amd", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "ntg_amd"), "-Wl,--no-undefined"])
    und = subprocess.run(["nm", "-u", str(exe)], capture_output=True, text=True, check=True).stdout
    for sym in ("ntg", "npsoloption", "linspace") + (("SplineInterp",) if name == "kincar" else ("PrintVector",)):
        assert any(l.split()[-1].split("@")[0] == sym for l in und.splitlines() if l.split()), sym


def test_vanderpol_unchanged(tmp_path):
    exe = build(tmp_path, "vanderpol")
    out = subprocess.run([str(exe)], cwd=tmp_path, capture_output=True, text=True, check=True).stdout
    assert "inform 0" in out
    obj = float(out.split("objective")[1].split()[0])
    assert abs(obj - 1.7022142628309958) < 1e-9
    coef = np.loadtxt(tmp_path / "coef1")                      # vanderpol.c:192 PrintVector("coef1")
    np.testing.assert_allclose(coef, [1, 1, 0.39374, -0.036958, -0.439532, -0.716865, -0.244974], atol=2e-6)


def test_kincar_unchanged(tmp_path):
    exe = build(tmp_path, "kincar")
    out = subprocess.run([str(exe), "-v"], cwd=tmp_path, capture_output=True, text=True, check=True).stdout
    assert "inform 0" in out
    obj = float(out.split("objective")[1].split()[0])
    assert abs(obj - 2.457581141950512) < 1e-9
    rows = [l.split() for l in out.strip().splitlines()[-30:]]
    traj = np.array(rows, dtype=float)                         # time x y theta v delta  (kincar.c:404-405)
    assert traj.shape == (30, 6)
    np.testing.assert_allclose(traj[0, 1:3], [0, -2], atol=1e-3)
    np.testing.assert_allclose(traj[-1, 1:3], [40, 2], atol=1e-3)
    np.testing.assert_allclose(traj[:, 4], 8.0, atol=0.25)    # speed stays near 8 m/s
