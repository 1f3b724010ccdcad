"""Diagnostic (not a test): eval_kernel time with parts switched off (results wrong on purpose)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntg_amd import api, configs as cf
spec = cf.config_M(); plan = api.Plan(spec, 0)
nb = 1 << 18
x = torch.randn((nb, spec.nC), dtype=torch.float64, device="cuda:0")
for dbg, name in ((0, "full"), (1, "no gather"), (2, "no phase1"), (3, "no gather, no phase1"), (7, "neither, no block_sum"), (15, "+ no g store"), (23, "+ no x load"), (31, "+ neither")):
    api.lib().ntg_debug_set(dbg)
    for _ in range(2): plan.eval(x, 2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): plan.eval(x, 2)
    e1.record(); torch.cuda.synchronize()
    print("%-28s %.3f ms" % (name, e0.elapsed_time(e1) / 5))
api.lib().ntg_debug_set(0)
