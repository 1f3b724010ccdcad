"""Diagnostic (not a test): time of the standalone evaluation kernel on config M.  NTG_AMD_EVAL_V1=1 selects the breakpoint-lane kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntg_amd import api, configs as cf
spec = cf.config_M() if len(sys.argv) < 2 or sys.argv[1] == "M" else cf.config_B()
nb = 1 << 18
plan = api.Plan(spec, 0)
x = torch.randn((nb, spec.nC), dtype=torch.float64, device="cuda:0")
out = plan.eval(x, 2)
for rep in range(3):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        plan.eval(x, 2, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(spec.name, "v1" if os.environ.get("NTG_AMD_EVAL_V1") else "interval", f"{ms:.4f} ms  {nb * spec.eval_bytes() / ms / 1e6:.0f} GB/s  frac {nb * spec.eval_bytes() / ms / 1e6 / 8000:.3f}")
