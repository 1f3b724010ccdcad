"""Diagnostic: sqp_kernel time vs block size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntg_amd import api, configs as cf
spec = cf.config_M(); plan = api.Plan(spec, 0); B = 4096
lo, up = cf.kincar_random_bounds(3, B)
lo = torch.tensor(lo, device="cuda:0"); up = torch.tensor(up, device="cuda:0")
x0 = torch.ones((B, spec.nC), dtype=torch.float64, device="cuda:0"); x = x0.clone()
import ctypes as C
a, b2, c2 = C.c_int(), C.c_int(), C.c_int()
api.lib().ntg_debug_layout(plan.h, C.byref(api.default_opts(itlim=50, fixed_iters=1)), C.byref(a), C.byref(b2), C.byref(c2)); print('LDS solve', a.value, 'eval', b2.value, 'nt', c2.value)
for nt in (128, 256):
    o = api.default_opts(itlim=50, fixed_iters=1, block_threads=nt)
    w = torch.empty(plan.workspace_bytes(B, o), dtype=torch.uint8, device="cuda:0")
    for _ in range(2): x.copy_(x0); plan.solve(lo, up, x, o, work=w)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): x.copy_(x0); plan.solve(lo, up, x, o, work=w)
    torch.cuda.synchronize(); print(nt, "%.3f ms" % ((time.perf_counter() - t) / 5 * 1e3))
