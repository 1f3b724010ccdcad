import os, sys, ctypes as C
sys.path.insert(0, "/root/repo")
from ntg_amd import api, configs as cf
for name, spec in (("M", cf.config_M()), ("B", cf.config_B())):
    p = api.Plan(spec, 0)
    a = C.c_int(); b = C.c_int(); c = C.c_int()
    api.lib().ntg_debug_layout(p.h, None, C.byref(a), C.byref(b), C.byref(c))
