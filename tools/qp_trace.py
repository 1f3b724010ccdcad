"""Diagnostic: device against oracle, QP-based SQP step, stopped after k majors (itlim = k): where do the two paths part?  python tools/qp_trace.py D 0"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import orc
from ntg_amd import api, configs as cf
from gpu_common import dev
which = sys.argv[1] if len(sys.argv) > 1 else "D"; idx = int(sys.argv[2]) if len(sys.argv) > 2 else 0
spec, bnd = {"O": (cf.config_O(), cf.obstacle_bounds), "D": (cf.config_D(), cf.quadrotor_bounds), "E": (cf.config_E(), cf.manipulator_bounds)}[which]
lo, up = bnd(idx + 1); lo, up = lo[idx:idx + 1], up[idx:idx + 1]
p = api.Plan(spec, 0)
for k in range(1, 10):
    x = torch.ones((1, spec.nC), dtype=torch.float64, device="cuda:0")
    out = p.solve(dev(lo), dev(up), x, api.default_opts(hessian=3, itlim=k)); torch.cuda.synchronize()
    r = orc.solve_batch(spec, lo, up, np.ones((1, spec.nC)), orc.default_opts(hessian=3, itlim=k), nthreads=1)
    print(f"itlim {k}: device inform {out['inform'].item()} iters {out['iters'].item()} nfev {out['nfev'].item()} F {out['objective'].item():.15g} | oracle inform {r['inform'][0]} iters {r['iters'][0]} nfev {r['nfev'][0]} F {r['objective'][0]:.15g} | |dx| {np.abs(x.cpu().numpy() - r['x']).max():.2e}")
