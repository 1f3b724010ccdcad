"""Pins for the oracle's restatement of PGS knots/interv/bsplvb/bsplvd (oracle/pgs.c).

The reference holds no vectors for these (the Fortran is not even in its tree), so the pins
are independent: closed forms (SURVEY.md §8a known-answer), partition of unity, and
scipy.interpolate.BSpline evaluated on the same augmented knot vector.
"""
import ctypes as C
import numpy as np
import pytest
import scipy.interpolate as si

import orc
from ntg_amd import configs as cf
from ntg_amd.spec import linspace_c

dp = C.POINTER(C.c_double)


def c_knots(brk, k, m):
    l = len(brk) - 1
    n = l * (k - m) + m
    t = np.zeros(n + k); nn = C.c_int()
    orc.lib().orc_knots(brk.ctypes.data_as(dp), l, k, m, t.ctypes.data_as(dp), C.byref(nn))
    assert nn.value == n
    return t


def c_interv(xt, x):
    left = C.c_int(); mflag = C.c_int()
    orc.lib().orc_interv(xt.ctypes.data_as(dp), len(xt), C.c_double(x), C.byref(left), C.byref(mflag))
    return left.value, mflag.value


def c_bsplvd(t, k, x, left, nderiv):
    a = np.zeros(k * k); db = np.zeros(k * nderiv)
    orc.lib().orc_bsplvd(t.ctypes.data_as(dp), k, C.c_double(x), left, a.ctypes.data_as(dp), db.ctypes.data_as(dp), nderiv)
    return db.reshape(nderiv, k)  # [m][i] = D^m B_{left-k+i}


def test_knots_vanderpol():
    t = c_knots(linspace_c(0, 5, 3), 5, 3)
    assert np.array_equal(t, [0] * 5 + [2.5] * 2 + [5] * 5)   # SURVEY §8a / vanderpol.m:18


def test_interv_rules():
    xt = np.array([0.0, 0, 0, 1, 2, 2, 3, 3, 3])
    assert c_interv(xt, -0.1) == (1, -1)
    assert c_interv(xt, 0.0) == (3, 0)
    assert c_interv(xt, 0.5) == (3, 0)
    assert c_interv(xt, 1.0) == (4, 0)
    assert c_interv(xt, 2.0) == (6, 0)
    assert c_interv(xt, 2.999) == (6, 0)
    # right end and beyond: last non-degenerate interval (2nd-edition rule)
    assert c_interv(xt, 3.0) == (6, 0)
    assert c_interv(xt, 3.0 + 1e-12) == (6, 1)


def test_vanderpol_end_blocks_closed_form():
    t = c_knots(linspace_c(0, 5, 3), 5, 3)
    left, _ = c_interv(t, 0.0)
    b0 = c_bsplvd(t, 5, 0.0, left, 3)
    ref0 = np.array([[1, 0, 0, 0, 0], [-1.6, 1.6, 0, 0, 0], [1.92, -3.84, 1.92, 0, 0]])
    np.testing.assert_allclose(b0, ref0, rtol=0, atol=1e-14)
    left, _ = c_interv(t, 5.0)
    b1 = c_bsplvd(t, 5, 5.0, left, 3)
    ref1 = np.array([[0, 0, 0, 0, 1], [0, 0, 0, -1.6, 1.6], [0, 0, 1.92, -3.84, 1.92]])
    np.testing.assert_allclose(b1, ref1, rtol=0, atol=1e-14)


@pytest.mark.parametrize("spec", [cf.config_A(), cf.config_K0(), cf.config_B(), cf.config_T()], ids=lambda s: s.name)
def test_blocks_vs_scipy_and_structure(spec):
    tab = orc.export_tables(spec)
    pos = 0
    for o in range(spec.nout):
        k, m, l, d, n = spec.order[o], spec.mult[o], spec.kninterv[o], spec.maxderiv[o], spec.ncoef[o]
        P = spec.nbps
        blk = tab["blk"][pos:pos + P * k * d].reshape(P, k, d); pos += P * k * d
        off = tab["off"][o]
        t = c_knots(np.ascontiguousarray(spec.knots[o]), k, m)
        xs = np.clip(spec.bps, t[0], t[-1])
        # dense collocation matrix from blocks
        for r in range(d):
            Mr = np.zeros((P, n))
            for i in range(P):
                Mr[i, off[i]:off[i] + k] = blk[i, :, r]
            Ms = np.zeros((P, n))
            for j in range(n):
                c = np.zeros(n); c[j] = 1.0
                bs = si.BSpline(t, c, k - 1)
                Ms[:, j] = bs.derivative(r)(xs) if r else bs(xs)
            scale = max(1.0, np.abs(Ms).max())
            # values/derivatives below `mult` are continuous across knots, so the piece chosen at
            # a breakpoint sitting on a knot does not matter for r < mult
            if r < m:
                np.testing.assert_allclose(Mr, Ms, rtol=0, atol=5e-12 * scale)
            if r == 0:
                np.testing.assert_allclose(Mr.sum(axis=1), 1.0, atol=1e-13)   # partition of unity
            else:
                np.testing.assert_allclose(Mr.sum(axis=1), 0.0, atol=1e-10 * scale)
        assert off.min() == 0 and off.max() == (l - 1) * (k - m)
        assert np.all(np.diff(off) >= 0)


def test_shipped_offsets():
    tab = orc.export_tables(cf.config_A())
    assert list(tab["off"][0]) == [0] * 10 + [2] * 10       # SURVEY §8a known-answer
