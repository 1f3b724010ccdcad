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
