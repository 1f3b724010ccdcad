"""Shared helpers for the -m gpu parity tests (all product calls go through the C ABI)."""
import numpy as np
import torch

import orc
from ntg_amd import api, configs as cf

SPECS = {"A": cf.config_A, "K0": cf.config_K0, "B": cf.config_B, "M": cf.config_M, "T": cf.config_T,
         "D8": lambda: cf.config_D(ninterv=8), "E8": lambda: cf.config_E(ninterv=8, narms=2)}
# shapes without a reference-code fixture (compared with the oracle only): two cars = 4 flat outputs
EXTRA = {"M4": lambda: cf._kincar_spec(2, 6, 3, 20, 101, 5.0, "M4:kincar-4out-k6-l20"),
         "M4b": lambda: cf._kincar_spec(2, 6, 3, 16, 81, 5.0, "M4b:kincar-4out-k6-l16")}   # the wave kernels' second interval count
_plans = {}


def plan_for(name):
    if name not in _plans:
        _plans[name] = api.Plan((SPECS.get(name) or EXTRA[name])(), 0)
    return _plans[name]


def dev(a, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda:0")


def rel(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def same_work(nfev, nfev_ref):
    """Fixed-work runs of two implementations of the same iteration do the same work: identical evaluation counts.  The exception is a
    problem that has converged to rounding level before the forced number of majors is over: its remaining line searches compare values
    that differ by ~1e-16 |F| and are decided by noise (seen: the oracle needs 101 evaluations where every other problem needs 100, one
    implementation 102, another a whole failed search of 20 more).  Rule: where the reference count is the batch's regular one the counts
    must be equal (at most one problem in eight may differ, by at most two evaluations); where the reference itself is irregular only the
    objective is compared (the callers assert it)."""
    import numpy as np
    a, b = np.asarray(nfev), np.asarray(nfev_ref)
    if a.shape != b.shape:
        return False
    vals, cnt = np.unique(b, return_counts=True)
    regular = b == vals[np.argmax(cnt)]
    da = np.abs(a - b)[regular]
    return regular.mean() >= 0.75 and (da != 0).sum() <= max(1, da.size // 8) and (da.max() if da.size else 0) <= 2
