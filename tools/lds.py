"""Diagnostic (not a test): LDS bytes and workgroup size the plan picks for each config."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntg_amd import api, configs as cf
for name, mk in (("K0", cf.config_K0), ("B", cf.config_B), ("M", cf.config_M), ("O", cf.config_O), ("D", cf.config_D), ("E", cf.config_E)):
    p = api.Plan(mk(), 0)
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    api.lib().ntg_debug_layout(p.h, C.byref(api.default_opts(hessian=1)), C.byref(a), C.byref(b), C.byref(c))
    print(name, "LDS solve", abs(a.value), "(BIG: vectors in HBM)" if a.value < 0 else "", "eval", b.value, "threads", c.value)
