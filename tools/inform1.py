"""one-off: inform histograms of the Newton parity cases on both sides (to set the explicit counts in test_gpu_newton.py)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import orc
from ntg_amd import api, configs as cf
from test_gpu_newton import _case
for name, nb in (("O", 24), ("D2", 16), ("E2", 12)):
    spec, bounds = _case(name)
    p = api.Plan(spec, 0)
    lo, up = bounds(nb)
    x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    out = p.solve(torch.tensor(lo, device="cuda:0"), torch.tensor(up, device="cuda:0"), x, api.default_opts(hessian=2))
    ref = orc.solve_batch(spec, lo, up, np.ones((nb, spec.nC)), orc.default_opts(hessian=2), nthreads=8)
    print(name, "gpu inform", out["inform"].cpu().numpy().tolist(), "ref inform", np.asarray(ref["inform"]).tolist(),
          "iters gpu", out["iters"].cpu().numpy().tolist(), "ref", np.asarray(ref["iters"]).tolist(), flush=True)
