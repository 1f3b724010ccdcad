"""Diagnostic (not a test): LDS bytes per workgroup of the solve and evaluation kernels.  python tools/layout.py M"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntg_amd import api, configs as cf
for name in sys.argv[1:] or ["M"]:
    spec = {"M": cf.config_M, "B": cf.config_B, "D": cf.config_D, "E": cf.config_E, "K0": cf.config_K0}[name]()
    p = api.Plan(spec, 0)
    for kw in (dict(hessian=0, itlim=50, fixed_iters=1), dict(hessian=1), dict(hessian=2)):
        o = api.default_opts(**kw)
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        api.lib().ntg_debug_layout(p.h, C.byref(o), C.byref(a), C.byref(b), C.byref(c))
        print(name, kw, "lds_solve", a.value, "lds_eval", b.value, "nt", c.value)
