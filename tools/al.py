"""Diagnostic: state of the augmented-Lagrangian loop at exit (NTG_AMD_STAMPS=2) for config D or E."""
import os, sys
os.environ["NTG_AMD_STAMPS"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ntg_amd import api, configs as cf
cfg = sys.argv[1]; nb = int(sys.argv[2]); qm = int(sys.argv[3]) if len(sys.argv) > 3 else 0
spec = cf.config_D() if cfg == "D" else cf.config_E()
lo, up = (cf.quadrotor_bounds if cfg == "D" else cf.manipulator_bounds)(nb)
p = api.Plan(spec, 0)
x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
out = p.solve(torch.tensor(lo, device="cuda:0"), torch.tensor(up, device="cuda:0"), x, api.default_opts(hessian=1, qn_memory=qm), want_lambda=True)
d = out["clambda"][:, :8].cpu().numpy(); it = out["iters"].cpu().numpy(); inf = out["inform"].cpu().numpy()
print("outer passes: mean %.1f max %d" % (d[:, 2].mean(), d[:, 2].max()), " log10 mu: ", np.bincount(np.log10(d[:, 1]).round().astype(int)).tolist(),
      " iters mean %.0f" % it.mean(), " inform", np.bincount(inf, minlength=5).tolist(), " rv max %.1e" % d[:, 0].max())
print("iters per outer pass: %.0f" % (it.mean() / max(d[:, 2].mean(), 1)))
