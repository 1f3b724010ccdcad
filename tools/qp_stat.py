import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from ntg_amd import api, configs as cf
from gpu_common import dev
from test_gpu_newton import _kkt
spec = cf.config_D(); lo, up = cf.quadrotor_bounds(8)
p = api.Plan(spec, 0)
for h in (2, 3):
    x = torch.ones((8, spec.nC), dtype=torch.float64, device="cuda:0")
    out = p.solve(dev(lo), dev(up), x, api.default_opts(hessian=h), want_lambda=True); torch.cuda.synchronize()
    lam = out["clambda"].cpu().numpy()
    try:
        st = _kkt(spec, p, x, lo, up, lam, 1.0)
    except AssertionError as e:
        print("assert", e); continue
    print(h, "stat", st, "iters", out["iters"].cpu().numpy())
    ev = p.eval(x, 2); g = ev["g"].cpu().numpy()
    pos = None
    print("   |g|inf", np.abs(g).max(axis=1))
os.environ["NTG_AMD_STAMPS"] = "3"
for h in (2, 3):
    x = torch.ones((8, spec.nC), dtype=torch.float64, device="cuda:0")
    out = p.solve(dev(lo), dev(up), x, api.default_opts(hessian=h), want_lambda=True); torch.cuda.synchronize()
    print(h, "counters [nfact nfail napply outer iter nfev nsolve ncol over fell]\n", out["clambda"][:, :10].cpu().numpy().astype(int))
del os.environ["NTG_AMD_STAMPS"]
# stationarity on the free coefficients directly: g + J'lam restricted to coefficients no equality row touches
