#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own C code (oracle/_ref/libntgref.so).

Run in the build container only (needs /root/reference to build oracle/_ref):
    python tests/golden/make_golden.py

What is reference code here: updateZ, InitialCost/IntegratedCost/FinalCost, IntegrateVector,
NonLinearConstraints, LinearConstraintsMatrix, bounds -- compiled unmodified from
/root/reference/src/{colloc,cost,constraints,integrator,matrix}.c (oracle/Makefile `ref`).
What is NOT reference code: the B-spline block values placed into the reference's Colloc
structs.  CollocMatrix() needs the absent PGS Fortran, so the blocks come from oracle/pgs.c
and are stored in the fixture as INPUTS ("blk", "off").  The fixtures therefore pin
everything between "basis block values" and "what NPSOL would be handed".
User callbacks are the oracle's family functions (oracle/families.c) passed as C pointers.
"""
import ctypes as C
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from ntg_amd import configs as cf  # noqa: E402

dp = C.POINTER(C.c_double)


class FMatrix(C.Structure):
    _fields_ = [("elements", C.POINTER(dp)), ("rows", C.c_int), ("cols", C.c_int)]


class Block(C.Structure):
    _fields_ = [("matrix", C.POINTER(FMatrix)), ("offset", C.c_int)]


class Colloc(C.Structure):
    _fields_ = [("block", C.POINTER(Block)), ("ninterv", C.c_int), ("order", C.c_int), ("mult", C.c_int),
                ("maxderiv", C.c_int), ("nbps", C.c_int), ("rows", C.c_int), ("cols", C.c_int)]


class ConcatColloc(C.Structure):
    _fields_ = [("colloc", C.POINTER(C.POINTER(Colloc))), ("nout", C.c_int), ("nbps", C.c_int),
                ("nz", C.c_int), ("nZ", C.c_int), ("nC", C.c_int),
                ("iZ", C.POINTER(C.c_int)), ("iz", C.POINTER(C.c_int)), ("iC", C.POINTER(C.c_int))]


class AVc(C.Structure):
    _fields_ = [("output", C.c_int), ("deriv", C.c_int)]


def ref_lib():
    path = os.path.join(ROOT, "oracle", "_ref", "libntgref.so")
    if not os.path.exists(path):
        orc.build()
    L = C.CDLL(path)
    L.MakeFMatrix.restype = C.POINTER(FMatrix)
    return L


class RefProblem:
    """The reference's ConcatColloc built by hand (colloc.c:15-117 minus the PGS calls)."""

    def __init__(self, L, spec, tables):
        self.L, self.spec = L, spec
        nout, P = spec.nout, spec.nbps
        self.keep = []
        collocs = (C.POINTER(Colloc) * nout)()
        pos = 0
        for o in range(nout):
            k, d, n = spec.order[o], spec.maxderiv[o], spec.ncoef[o]
            blocks = (Block * P)()
            for i in range(P):
                m = L.MakeFMatrix(d, k)  # rows=maxderiv, cols=order : elements[q][r]
                for q in range(k):
                    for r in range(d):
                        m.contents.elements[q][r] = tables["blk"][pos + (i * k + q) * d + r]
                blocks[i].matrix = m
                blocks[i].offset = int(tables["off"][o, i])
            pos += P * k * d
            c = Colloc(blocks, spec.kninterv[o], k, spec.mult[o], d, P, d * P, n)
            self.keep += [blocks, c]
            collocs[o] = C.pointer(c)
        iZ = (C.c_int * nout)(); iz = (C.c_int * nout)(); iC = (C.c_int * nout)()
        for o in range(1, nout):
            iZ[o] = iZ[o - 1] + spec.maxderiv[o - 1] * P
            iz[o] = iz[o - 1] + spec.maxderiv[o - 1]
            iC[o] = iC[o - 1] + spec.ncoef[o - 1]
        self.cc = ConcatColloc(collocs, nout, P, spec.nz, spec.nz * P, spec.nC, iZ, iz, iC)
        self.keep += [collocs, iZ, iz, iC]

    def avs(self, lst):
        arr = (AVc * max(len(lst), 1))()
        for j, (o, d) in enumerate(lst):
            arr[j].output, arr[j].deriv = o, d
        return arr, len(lst)


def rows_ptr(mat):
    """double** view of a C-contiguous [n, nz] matrix (how the examples pass lic/lfc)."""
    mat = np.ascontiguousarray(mat, dtype=np.float64)
    ptrs = (dp * max(mat.shape[0], 1))()
    for i in range(mat.shape[0]):
        ptrs[i] = mat[i].ctypes.data_as(dp)
    return ptrs, mat


def run_reference(L, spec, x, lowerb, upperb):
    """Drive the reference functions the way NPfunobj/NPfuncon/ntg() do (ntg.c:162-229,274-371)."""
    O = orc.lib()
    O.orc_family_set_nout(spec.nout)
    for nm in ("ucf", "icf", "fcf", "nlicf", "nltcf", "nlfcf"):
        getattr(O, "orc_family_" + nm).restype = C.c_void_p
    fam = spec.family
    tables = orc.export_tables(spec)
    rp = RefProblem(L, spec, tables)
    cc = C.byref(rp.cc)
    n, P, nz = spec.nC, spec.nbps, spec.nz
    bps = np.ascontiguousarray(spec.bps)
    out = dict(blk=tables["blk"], off=tables["off"], x=x, lowerb=lowerb, upperb=upperb)
    fl, gl, cl, Jl, Zl = [], [], [], [], []
    for b in range(x.shape[0]):
        xb = np.ascontiguousarray(x[b])
        Z = np.zeros(spec.nz * P)  # ntg.c:119 calloc
        mode = C.c_int(2); nstate = C.c_int(1)
        I = C.c_double(0); In = C.c_double(0); F = C.c_double(0)
        dI = np.zeros(n); dIn = np.zeros(n); dF = np.zeros(n)
        xp = xb.ctypes.data_as(dp); Zp = Z.ctypes.data_as(dp)
        if spec.nicf:
            av, nav = rp.avs(list(spec.icostav)); L.updateZ(Zp, cc, xp, av, nav, 0)
        if spec.nucf:
            av, nav = rp.avs(list(spec.tcostav)); L.updateZ(Zp, cc, xp, av, nav, 1)
        if spec.nfcf:
            av, nav = rp.avs(list(spec.fcostav)); L.updateZ(Zp, cc, xp, av, nav, 2)
        if spec.nicf:
            L.InitialCost(C.byref(mode), C.byref(nstate), C.byref(I), dI.ctypes.data_as(dp), C.c_void_p(O.orc_family_icf(fam)), cc, Zp)
        if spec.nucf:
            L.IntegratedCost(C.byref(mode), C.byref(nstate), C.byref(In), dIn.ctypes.data_as(dp), bps.ctypes.data_as(dp), C.c_void_p(O.orc_family_ucf(fam)), cc, Zp)
        if spec.nfcf:
            L.FinalCost(C.byref(mode), C.byref(nstate), C.byref(F), dF.ctypes.data_as(dp), C.c_void_p(O.orc_family_fcf(fam)), cc, Zp)
        fl.append(I.value + In.value + F.value)        # ntg.c:328
        gl.append((dI + dIn) + dF)                     # Vector3Add, matrix.c:177-182
        Zcost = Z.copy()
        if spec.ncnln:
            if spec.nnlic:
                av, nav = rp.avs(list(spec.icav)); L.updateZ(Zp, cc, xp, av, nav, 0)
            if spec.nnltc:
                av, nav = rp.avs(list(spec.tcav)); L.updateZ(Zp, cc, xp, av, nav, 1)
            if spec.nnlfc:
                av, nav = rp.avs(list(spec.fcav)); L.updateZ(Zp, cc, xp, av, nav, 2)
            cvec = np.zeros(spec.ncnln)
            J = L.MakeFMatrix(spec.ncnln, n)
            mode = C.c_int(2)
            L.NonLinearConstraints(C.byref(mode), C.byref(nstate), cvec.ctypes.data_as(dp), J,
                                   spec.nnlic, C.c_void_p(O.orc_family_nlicf(fam)),
                                   spec.nnltc, C.c_void_p(O.orc_family_nltcf(fam)),
                                   spec.nnlfc, C.c_void_p(O.orc_family_nlfcf(fam)), cc, Zp)
            Jn = np.array([[J.contents.elements[c][r] for c in range(n)] for r in range(spec.ncnln)])
            cl.append(cvec); Jl.append(Jn)
        Zl.append(np.stack([Zcost, Z.copy()]))
    out.update(f=np.array(fl), g=np.array(gl), Z=np.array(Zl))
    if spec.ncnln:
        out.update(c=np.array(cl), cJac=np.array(Jl))
    # setup-time: A, bl, bu (ntg.c:162-229)
    if spec.nclin:
        A = L.MakeFMatrix(spec.nclin, n)

        def fm(mat):
            if mat.shape[0] == 0:
                return None, None
            ptrs, keep = rows_ptr(mat)
            m = FMatrix(ptrs, nz, mat.shape[0])
            return m, (ptrs, keep)
        mi, k1 = fm(spec.lic); mt, k2 = fm(spec.ltc); mf, k3 = fm(spec.lfc)
        L.LinearConstraintsMatrix(A, C.byref(mi) if mi else None, C.byref(mt) if mt else None,
                                  C.byref(mf) if mf else None, cc)
        out["A"] = np.array([[A.contents.elements[c][r] for c in range(n)] for r in range(spec.nclin)])
    ntot = n + spec.nclin + spec.ncnln
    DBL_MAX = np.finfo(np.float64).max
    bl = np.zeros(ntot); bu = np.zeros(ntot)
    lo0 = np.ascontiguousarray(lowerb[0]); up0 = np.ascontiguousarray(upperb[0])
    args = (n, spec.nlic, spec.nltc, spec.nlfc, spec.nnlic, spec.nnltc, spec.nnlfc, P)
    L.bounds(bu.ctypes.data_as(dp), up0.ctypes.data_as(dp), *args, C.c_double(DBL_MAX))
    L.bounds(bl.ctypes.data_as(dp), lo0.ctypes.data_as(dp), *args, C.c_double(-DBL_MAX))
    out.update(bl=bl, bu=bu)
    # integrator.c on its own
    t = np.cumsum(np.random.default_rng(3).uniform(0.1, 1.0, 33)); fv = np.random.default_rng(4).normal(size=33)
    Iv = C.c_double(0)
    L.IntegrateVector(C.byref(Iv), fv.ctypes.data_as(dp), t.ctypes.data_as(dp), 33, 2)  # TRAPEZOID = 2 (integrator.h:21)
    out.update(int_t=t, int_f=fv, int_I=np.array(Iv.value))
    return out


def cases():
    rng = np.random.default_rng(cf.SEED)
    A = cf.config_A(); loA, upA = cf.bounds_A()
    K0 = cf.config_K0(); loK, upK = cf.bounds_K0_shipped()
    B = cf.config_B(); loB, upB = cf.kincar_random_bounds(1, 2)
    M = cf.config_M(); loM, upM = cf.kincar_random_bounds(3, 1)
    T = cf.config_T()
    loT = np.round(rng.uniform(-2, 0, (1, T.nbounds)), 3); upT = loT + np.round(rng.uniform(0, 2, (1, T.nbounds)), 3)
    yield "A", A, rng.normal(size=(3, A.nC)), loA[None], upA[None]
    yield "K0", K0, rng.normal(size=(3, K0.nC)) * 5, loK[None], upK[None]
    yield "B", B, rng.normal(size=(2, B.nC)) * 5, loB, upB
    yield "M", M, rng.normal(size=(1, M.nC)) * 5, loM, upM
    yield "T", T, rng.normal(size=(3, T.nC)), loT, upT
    # reduced grids of BASELINE configs D (maxderiv 5, order 8) and E (two arms): new draws AFTER the ones above
    D8 = cf.config_D(ninterv=8); loD, upD = cf.quadrotor_bounds(1)
    E8 = cf.config_E(ninterv=8, narms=2); loE, upE = cf.manipulator_bounds(1, narms=2)
    yield "D8", D8, rng.normal(size=(2, D8.nC)) * 2, loD, upD
    yield "E8", E8, rng.normal(size=(2, E8.nC)), loE, upE


def main():
    L = ref_lib()
    for name, spec, x, lo, up in cases():
        out = run_reference(L, spec, x, lo, up)
        np.savez_compressed(os.path.join(HERE, f"ref_{name}.npz"), **out)
        print(name, "f", out["f"][:2], "files ok")


if __name__ == "__main__":
    main()
