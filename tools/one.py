"""Diagnostic driver for rocprofv3 --pmc: a few sqp_kernel and eval_kernel launches, nothing else."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntg_amd import api, configs as cf
spec = cf.config_M(); plan = api.Plan(spec, 0); B = 4096
lo, up = cf.kincar_random_bounds(3, B)
lo = torch.tensor(lo, device="cuda:0"); up = torch.tensor(up, device="cuda:0")
x0 = torch.ones((B, spec.nC), dtype=torch.float64, device="cuda:0"); x = x0.clone()
o = api.default_opts(itlim=50, fixed_iters=1)
w = torch.empty(plan.workspace_bytes(B, o), dtype=torch.uint8, device="cuda:0")
for _ in range(3):
    x.copy_(x0); plan.solve(lo, up, x, o, work=w)
xe = torch.randn((1 << 18, spec.nC), dtype=torch.float64, device="cuda:0")
for _ in range(3):
    plan.eval(xe, 2)
torch.cuda.synchronize()
