"""Diagnostic driver for rocprofv3 (round 2): a few launches of the kernels the round-2 numbers are quoted on, nothing else.
   python3 tools/prof_r2.py [sqp|eval|newtonD|newtonE]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntg_amd import api, configs as cf
which = sys.argv[1] if len(sys.argv) > 1 else "sqp"
dev = "cuda:0"
if which in ("sqp", "eval"):
    spec = cf.config_M(); plan = api.Plan(spec, 0); B = 4096
    lo, up = cf.kincar_random_bounds(3, B)
    lo = torch.tensor(lo, device=dev); up = torch.tensor(up, device=dev)
    if which == "sqp":
        x0 = torch.ones((B, spec.nC), dtype=torch.float64, device=dev); x = x0.clone()
        o = api.default_opts(itlim=50, fixed_iters=1)
        w = torch.empty(plan.workspace_bytes(B, o), dtype=torch.uint8, device=dev)
        for _ in range(3):
            x.copy_(x0); plan.solve(lo, up, x, o, work=w)
    else:
        xe = torch.randn((1 << 18, spec.nC), dtype=torch.float64, device=dev)
        out = plan.eval(xe, 2)
        for _ in range(3):
            plan.eval(xe, 2, out=out)
else:
    spec, bounds, B = (cf.config_D(), cf.quadrotor_bounds, 512) if which == "newtonD" else (cf.config_E(), cf.manipulator_bounds, 1024)
    plan = api.Plan(spec, 0)
    lo, up = bounds(B)
    lo = torch.tensor(lo, device=dev); up = torch.tensor(up, device=dev)
    o = api.default_opts(hessian=2)
    w = torch.empty(plan.workspace_bytes(B, o), dtype=torch.uint8, device=dev)
    for _ in range(2):
        x = torch.ones((B, spec.nC), dtype=torch.float64, device=dev)
        plan.solve(lo, up, x, o, work=w)
torch.cuda.synchronize()
