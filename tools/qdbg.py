"""one-off: print the LDS layouts (NTG_AMD_DEBUG=1) of a few plans"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntg_amd import api, configs as cf
for name, spec in (("D", cf.config_D()), ("E", cf.config_E())):
    p = api.Plan(spec, 0)
    a = C.c_int(); b = C.c_int(); c = C.c_int()
    print("==", name, flush=True)
    api.lib().ntg_debug_layout(p.h, None, C.byref(a), C.byref(b), C.byref(c))
    print(name, "lds solve", a.value, "lds eval", b.value, "nt", c.value, flush=True)
