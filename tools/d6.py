"""one-off: which config-D problems end with inform != 0 in the Newton mode, and what the oracle says about them"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch, orc
from ntg_amd import api, configs as cf
spec = cf.config_D(); nb = 4096
lo, up = cf.quadrotor_bounds(nb)
p = api.Plan(spec, 0)
x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
out = p.solve(torch.tensor(lo, device="cuda:0"), torch.tensor(up, device="cuda:0"), x, api.default_opts(hessian=2)); torch.cuda.synchronize()
inf = out["inform"].cpu().numpy(); it = out["iters"].cpu().numpy(); obj = out["objective"].cpu().numpy()
bad = np.nonzero(inf != 0)[0]
print("bad", bad, inf[bad], it[bad])
ref = orc.solve_batch(spec, lo[bad], up[bad], np.ones((len(bad), spec.nC)), orc.default_opts(hessian=2), nthreads=8)
print("oracle inform", ref["inform"], "iters", ref["iters"])
print("rel obj diff", np.abs(obj[bad] - ref["objective"]) / np.abs(ref["objective"]))
