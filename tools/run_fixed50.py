"""the headline launch, N times (for profilers): python tools/run_fixed50.py [N] [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntg_amd import api, configs as cf
api.LIB_PATH = os.environ.get("NTG_AMD_LIB", api.LIB_PATH)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mode = sys.argv[2] if len(sys.argv) > 2 else "fixed50"
dev = torch.device("cuda:0")
spec = cf.config_M(); B = 4096
lo, up = cf.kincar_random_bounds(3, B)
lo = torch.tensor(lo, device=dev); up = torch.tensor(up, device=dev)
plan = api.Plan(spec, 0)
opts = {"fixed50": api.default_opts(itlim=50, fixed_iters=1, hessian=0), "conv_h0": api.default_opts(hessian=0), "conv_h1": api.default_opts(hessian=1, itlim=50)}[mode]
work = torch.empty(plan.workspace_bytes(B, opts), dtype=torch.uint8, device=dev)
x = torch.ones((B, spec.nC), dtype=torch.float64, device=dev)
for _ in range(n):
    x.fill_(1.0); out = plan.solve(lo, up, x, opts, work=work)
torch.cuda.synchronize()
print("done", n, mode, float(out["objective"].sum()))
