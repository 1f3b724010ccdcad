"""Diagnostic driver for rocprofv3 (round 3): a few launches of the kernels the round-3 numbers are quoted on, nothing else.
   python3 tools/prof_r3.py [fixed50|conv_h1|conv_h0|eval]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntg_amd import api, configs as cf
which = sys.argv[1] if len(sys.argv) > 1 else "fixed50"
dev = "cuda:0"
if which in ("qp_E", "newton_E"):   # config E, 1024 problems: the QP-based SQP step (hessian = 3) / the structured Newton mode (2); two launches
    spec = cf.config_E(); plan = api.Plan(spec, 0)
    lo_n, up_n = cf.manipulator_bounds(1024)
    lo = torch.tensor(lo_n, device=dev); up = torch.tensor(up_n, device=dev)
    o = api.default_opts(hessian=3 if which == "qp_E" else 2)
    w = torch.empty(plan.workspace_bytes(1024, o), dtype=torch.uint8, device=dev)
    for _ in range(2):
        x = torch.ones((1024, spec.nC), dtype=torch.float64, device=dev); plan.solve(lo, up, x, o, work=w)
    torch.cuda.synchronize()
    sys.exit(0)
spec = cf.config_M(); plan = api.Plan(spec, 0)
B = 65536 if which == "conv_h1_big" else 4096
lo, up = cf.kincar_random_bounds(3, 4096)
import numpy as np
lo = torch.tensor(np.tile(lo, (B // 4096, 1)), device=dev); up = torch.tensor(np.tile(up, (B // 4096, 1)), device=dev)
if which == "eval":
    xe = torch.randn((1 << 18, spec.nC), dtype=torch.float64, device=dev)
    out = plan.eval(xe, 2)
    for _ in range(3):
        plan.eval(xe, 2, out=out)
else:
    o = {"fixed50": api.default_opts(itlim=50, fixed_iters=1, hessian=0), "conv_h1": api.default_opts(hessian=1, itlim=50),
         "conv_h1_big": api.default_opts(hessian=1, itlim=50), "conv_h0": api.default_opts(hessian=0)}[which]
    x0 = torch.ones((B, spec.nC), dtype=torch.float64, device=dev); x = x0.clone()
    w = torch.empty(plan.workspace_bytes(B, o), dtype=torch.uint8, device=dev)
    for _ in range(3):
        x.copy_(x0); plan.solve(lo, up, x, o, work=w)
torch.cuda.synchronize()
