"""Diagnostic: the QP-based SQP step on problems the tests and the bench do not touch (the second thousand of the manipulator / quadrotor / obstacle streams)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ntg_amd import api, configs as cf
for name, spec, bnd, n0, n1 in (("E", cf.config_E(), cf.manipulator_bounds, 8192, 9216), ("D", cf.config_D(), cf.quadrotor_bounds, 4096, 5120), ("O", cf.config_O(), cf.obstacle_bounds, 4096, 8192)):
    lo, up = bnd(n1); lo, up = lo[n0:], up[n0:]
    p = api.Plan(spec, 0); nb = n1 - n0
    x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    o = p.solve(torch.tensor(lo, device="cuda:0"), torch.tensor(up, device="cuda:0"), x, api.default_opts(hessian=3)); torch.cuda.synchronize()
    inf = o["inform"].cpu().numpy(); it = o["iters"].cpu().numpy()
    print(name, nb, "problems: inform", np.bincount(inf).tolist(), "majors mean %.1f max %d" % (it.mean(), it.max()), "finite", bool(torch.isfinite(x).all()))
