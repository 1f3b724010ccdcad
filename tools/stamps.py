"""Diagnostic (not a test): per-phase cycle shares of sqp_kernel.  NTG_AMD_STAMPS=1 python tools/stamps.py
Needs a variant library built with -DNTG_CLOCK (tools/mkvariant2.sh), passed as NTG_AMD_LIB: the shipped kernels carry no clock."""
import os, sys
os.environ["NTG_AMD_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ntg_amd import api, configs as cf
api.LIB_PATH = os.environ.get("NTG_AMD_LIB", api.LIB_PATH)
cfg = sys.argv[1] if len(sys.argv) > 1 else "M"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
if cfg == "D":
    spec = cf.config_D(); lo, up = cf.quadrotor_bounds(B)
elif cfg == "E":
    spec = cf.config_E(); lo, up = cf.manipulator_bounds(B)
else:
    spec = cf.config_M() if cfg == "M" else cf.config_B()
    lo, up = cf.kincar_random_bounds(spec.nout // 2, B)
dev = torch.device("cuda:0")
plan = api.Plan(spec, 0)
x = torch.ones((B, spec.nC), dtype=torch.float64, device=dev)
for mode in ((dict(hessian=1, qn_memory=int(os.environ.get("QNM", "0"))),) if cfg in "DE" else (dict(itlim=50, fixed_iters=1, hessian=0), dict(hessian=1, itlim=50))):
    x.fill_(1.0)
    out = plan.solve(torch.tensor(lo, device=dev), torch.tensor(up, device=dev), x, api.default_opts(**mode), want_lambda=True)
    torch.cuda.synchronize()
    tk = out["clambda"][:, :8].cpu().numpy()
    names = ["setup", "eval", "project", "history", "w0", "rest", "ev.phase1", "ev.phase2"]
    tot = tk[:, :6].sum(axis=1).mean()
    print(mode, "mean cycles/problem %.0f" % tot, {n: "%.1f%%" % (100 * tk[:, i].mean() / tot) for i, n in enumerate(names)},
          "nfev", float(out["nfev"].float().mean()))
