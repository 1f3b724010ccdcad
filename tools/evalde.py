"""Diagnostic: funobj + funcon (banded Jacobian rows) throughput for configs D and E."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from ntg_amd import api, configs as cf
for name, mk, nb in (("D", cf.config_D, 4096), ("E", cf.config_E, 2048)):
    spec = mk(); p = api.Plan(spec, 0)
    x = torch.randn((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    for _ in range(2): p.eval(x, 2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): out = p.eval(x, 2)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    byts = nb * spec.eval_bytes()
    print(name, 'batch', nb, '%.3f ms' % ms, '%.0f evals/s' % (nb / ms * 1e3), 'alg %.1f MB' % (byts / 1e6), '%.0f GB/s' % (byts / ms / 1e6), 'frac %.3f' % (byts / ms / 1e6 / 8000))
