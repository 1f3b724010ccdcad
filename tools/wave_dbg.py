"""debug one-off: tiny wave-kernel solves with progress lines"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ntg_amd import api, configs as cf
import orc
which = sys.argv[1]; B = int(sys.argv[2]); itlim = int(sys.argv[3]); fixed = int(sys.argv[4]); hess = int(sys.argv[5])
spec, ncars = {"M": (cf.config_M(), 3), "B": (cf.config_B(), 1)}[which]
dev = torch.device("cuda:0")
lo, up = cf.kincar_random_bounds(ncars, B)
plan = api.Plan(spec, 0)
opts = api.default_opts(itlim=itlim, fixed_iters=fixed, hessian=hess)
x = torch.ones((B, spec.nC), dtype=torch.float64, device=dev)
print("launch", which, B, itlim, fixed, hess, flush=True)
out = plan.solve(torch.tensor(lo, device=dev), torch.tensor(up, device=dev), x, opts, want_lambda=True)
torch.cuda.synchronize()
print("done", {k: v[:4].cpu().numpy() for k, v in out.items() if k != "clambda"}, flush=True)
ref = orc.solve_batch(spec, lo, up, np.ones((B, spec.nC)), orc.default_opts(itlim=itlim, fixed_iters=fixed, hessian=hess), nthreads=8)
print("ref ", {k: np.asarray(ref[k])[:4] for k in ("objective", "inform", "iters", "nfev")}, flush=True)
print("max|dx|", np.abs(x.cpu().numpy() - ref["x"]).max(), flush=True)
