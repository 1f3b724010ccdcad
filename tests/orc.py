"""ctypes binding of the CPU oracle (oracle/liborc.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "liborc.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
    return _lib


class AVc(C.Structure):
    _fields_ = [("output", C.c_int), ("deriv", C.c_int)]


class Opts(C.Structure):
    _fields_ = [("itlim", C.c_int), ("opttol", C.c_double), ("steplimit", C.c_double),
                ("ls_mu", C.c_double), ("ls_eta", C.c_double), ("ls_maxfev", C.c_int),
                ("hessian", C.c_int), ("fixed_iters", C.c_int), ("verbose", C.c_int), ("qn_memory", C.c_int), ("banded", C.c_int), ("warm_lam", C.c_void_p)]


class Result(C.Structure):
    _fields_ = [("inform", C.c_int), ("iters", C.c_int), ("nfev", C.c_int),
                ("objective", C.c_double), ("pg_norm", C.c_double), ("feas", C.c_double)]


dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


class BatchSpec(C.Structure):
    _fields_ = [("nout", C.c_int), ("nbps", C.c_int), ("bps", dp), ("kninterv", ip),
                ("knots", C.POINTER(dp)), ("order", ip), ("mult", ip), ("maxderiv", ip),
                ("family", C.c_int),
                ("nlic", C.c_int), ("nltc", C.c_int), ("nlfc", C.c_int),
                ("lic", dp), ("ltc", dp), ("lfc", dp),
                ("nnlic", C.c_int), ("nnltc", C.c_int), ("nnlfc", C.c_int),
                ("nicav", C.c_int), ("ntcav", C.c_int), ("nfcav", C.c_int),
                ("icav", C.POINTER(AVc)), ("tcav", C.POINTER(AVc)), ("fcav", C.POINTER(AVc)),
                ("nicf", C.c_int), ("nucf", C.c_int), ("nfcf", C.c_int),
                ("nicostav", C.c_int), ("ntcostav", C.c_int), ("nfcostav", C.c_int),
                ("icostav", C.POINTER(AVc)), ("tcostav", C.POINTER(AVc)), ("fcostav", C.POINTER(AVc))]


def _d(a):
    return a.ctypes.data_as(dp)


def _i(a):
    return a.ctypes.data_as(ip)


class CSpec:
    """Keeps the numpy buffers alive behind an orc_batch_spec."""

    def __init__(self, spec):
        self.spec = spec
        k = self._keep = {}
        k["bps"] = np.ascontiguousarray(spec.bps, dtype=np.float64)
        k["kninterv"] = np.asarray(spec.kninterv, dtype=np.int32)
        k["order"] = np.asarray(spec.order, dtype=np.int32)
        k["mult"] = np.asarray(spec.mult, dtype=np.int32)
        k["maxderiv"] = np.asarray(spec.maxderiv, dtype=np.int32)
        k["knots"] = [np.ascontiguousarray(x, dtype=np.float64) for x in spec.knots]
        kp = (dp * spec.nout)(*[_d(x) for x in k["knots"]])
        k["kp"] = kp
        for nm in ("lic", "ltc", "lfc"):
            k[nm] = np.ascontiguousarray(getattr(spec, nm), dtype=np.float64).reshape(-1)

        def avs(lst):
            arr = (AVc * max(len(lst), 1))()
            for j, (o, d) in enumerate(lst):
                arr[j].output = o
                arr[j].deriv = d
            return arr
        for nm in ("icav", "tcav", "fcav", "icostav", "tcostav", "fcostav"):
            k[nm] = avs(list(getattr(spec, nm)))
        s = BatchSpec()
        s.nout = spec.nout; s.nbps = spec.nbps; s.bps = _d(k["bps"]); s.kninterv = _i(k["kninterv"])
        s.knots = kp; s.order = _i(k["order"]); s.mult = _i(k["mult"]); s.maxderiv = _i(k["maxderiv"])
        s.family = spec.family
        s.nlic, s.nltc, s.nlfc = spec.nlic, spec.nltc, spec.nlfc
        s.lic, s.ltc, s.lfc = _d(k["lic"]), _d(k["ltc"]), _d(k["lfc"])
        s.nnlic, s.nnltc, s.nnlfc = spec.nnlic, spec.nnltc, spec.nnlfc
        s.nicav, s.ntcav, s.nfcav = len(spec.icav), len(spec.tcav), len(spec.fcav)
        s.icav, s.tcav, s.fcav = k["icav"], k["tcav"], k["fcav"]
        s.nicf, s.nucf, s.nfcf = spec.nicf, spec.nucf, spec.nfcf
        s.nicostav, s.ntcostav, s.nfcostav = len(spec.icostav), len(spec.tcostav), len(spec.fcostav)
        s.icostav, s.tcostav, s.fcostav = k["icostav"], k["tcostav"], k["fcostav"]
        self.c = s


def default_opts(**kw) -> Opts:
    o = Opts()
    lib().orc_sqp_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def export_tables(spec, lowerb=None, upperb=None):
    cs = CSpec(spec)
    nblk = sum(spec.nbps * k * d for k, d in zip(spec.order, spec.maxderiv))
    blk = np.zeros(nblk); off = np.zeros((spec.nout, spec.nbps), dtype=np.int32)
    A = np.zeros((spec.nC, max(spec.nclin, 1)))  # column-major nclin x nC == row-major [nC][nclin]
    ntot = spec.nC + spec.nclin + spec.ncnln
    bl = np.zeros(ntot); bu = np.zeros(ntot)
    lo = None if lowerb is None else np.ascontiguousarray(lowerb, dtype=np.float64)
    up = None if upperb is None else np.ascontiguousarray(upperb, dtype=np.float64)
    lib().orc_spec_export(C.byref(cs.c), _d(lo) if lo is not None else None, _d(up) if up is not None else None,
                          _d(blk), _i(off), _d(A), _d(bl), _d(bu))
    Amat = A.T.copy() if spec.nclin else np.zeros((0, spec.nC))   # [nclin, nC]
    return dict(blk=blk, off=off, A=Amat, bl=bl, bu=bu)


def eval_batch(spec, x, mode=2, nthreads=1):
    """x: [batch, nC] -> dict(f [batch], g [batch,nC], c [batch,ncnln], cJac [batch,ncnln,nC])"""
    cs = CSpec(spec)
    x = np.ascontiguousarray(x, dtype=np.float64)
    batch = x.shape[0]
    f = np.zeros(batch); g = np.zeros((batch, spec.nC))
    c = np.zeros((batch, max(spec.ncnln, 1))); J = np.zeros((batch, spec.nC, max(spec.ncnln, 1)))
    lib().orc_eval_batch(C.byref(cs.c), batch, _d(x), mode, _d(f), _d(g),
                         _d(c) if spec.ncnln else None, _d(J) if spec.ncnln else None, nthreads)
    out = dict(f=f, g=g)
    if spec.ncnln:
        out["c"] = c
        out["cJac"] = np.transpose(J, (0, 2, 1)).copy()  # col-major (ncnln x nC) -> [batch, ncnln, nC]
    return out


def set_scratch_reuse(on: bool):
    """timed CPU baseline only (bench.py): per-thread reuse of the evaluation's dense temporaries instead of calloc / free per call"""
    lib().orc_set_scratch_reuse(1 if on else 0)


def solve_batch(spec, lowerb, upperb, x0, opts=None, nthreads=1):
    cs = CSpec(spec)
    o = opts or default_opts()
    lo = np.ascontiguousarray(lowerb, dtype=np.float64); up = np.ascontiguousarray(upperb, dtype=np.float64)
    x = np.array(x0, dtype=np.float64, order="C", copy=True)
    batch = x.shape[0]
    obj = np.zeros(batch); inform = np.zeros(batch, dtype=np.int32)
    iters = np.zeros(batch, dtype=np.int32); nfev = np.zeros(batch, dtype=np.int32)
    lib().orc_solve_batch(C.byref(cs.c), batch, _d(lo), _d(up), _d(x), C.byref(o), _d(obj), _i(inform), _i(iters), _i(nfev), nthreads)
    return dict(x=x, objective=obj, inform=inform, iters=iters, nfev=nfev)


def solve_one(spec, lowerb, upperb, x0, opts=None, trace_cap=0, want_R=False, warm_lam=None):
    """warm_lam: starting multipliers of the augmented-Lagrangian rows [ncnln + nI], internal sign (= -clambda of those rows)"""
    cs = CSpec(spec)
    o = opts or default_opts()
    if warm_lam is not None:
        wl = np.ascontiguousarray(warm_lam, dtype=np.float64)
        o.warm_lam = wl.ctypes.data
    lo = np.ascontiguousarray(lowerb, dtype=np.float64); up = np.ascontiguousarray(upperb, dtype=np.float64)
    x = np.array(x0, dtype=np.float64, copy=True)
    res = Result()
    ntot = spec.nC + spec.nclin + spec.ncnln
    clam = np.zeros(ntot); ist = np.zeros(ntot, dtype=np.int32)
    R = np.zeros((spec.nC, spec.nC)) if want_R else None
    tr = np.zeros((max(trace_cap, 1), 4))
    lib().orc_solve_one(C.byref(cs.c), _d(lo), _d(up), _d(x), C.byref(o), C.byref(res), _d(clam), _i(ist),
                        _d(R) if want_R else None, _d(tr), trace_cap)
    return dict(x=x, objective=res.objective, inform=res.inform, iters=res.iters, nfev=res.nfev,
                pg_norm=res.pg_norm, feas=res.feas, clambda=clam, istate=ist,
                R=(R.T.copy() if want_R else None), trace=tr[:min(trace_cap, res.iters)])


def thread_cpus(nthreads: int):
    """CPU each OpenMP thread of the batch drivers runs on (sched_getcpu): the affinity note of bench.py's cpu_baseline"""
    import ctypes as C
    arr = (C.c_int * nthreads)()
    lib().orc_thread_cpus(nthreads, arr)
    return list(arr)
