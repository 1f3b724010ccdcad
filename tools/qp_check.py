"""QP-based SQP step (hessian = 3): device against the oracle's statement of the same algorithm.  python tools/qp_check.py [O|D2|E2|D|E ...] [--n N] [--time]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import orc
from ntg_amd import api, configs as cf
from gpu_common import dev

def case(name):
    return {"O": (cf.config_O(), cf.obstacle_bounds), "D2": (cf.config_D(ninterv=10), cf.quadrotor_bounds),
            "E2": (cf.config_E(ninterv=20, narms=2), lambda n: cf.manipulator_bounds(n, narms=2)),
            "D": (cf.config_D(), cf.quadrotor_bounds), "E": (cf.config_E(), cf.manipulator_bounds)}[name]

names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["O", "D2", "E2"]
n_arg = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 0
for name in names:
    spec, bounds = case(name)
    nb = n_arg or {"O": 24, "D2": 16, "E2": 12, "D": 8, "E": 4}[name]
    lo, up = bounds(nb)
    p = api.Plan(spec, 0)
    res = {}
    for h in (3, 2):
        x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
        o = api.default_opts(hessian=h)
        out = p.solve(dev(lo), dev(up), x, o); torch.cuda.synchronize()
        t = time.time()
        x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
        out = p.solve(dev(lo), dev(up), x, o); torch.cuda.synchronize()
        dt = time.time() - t
        res[h] = dict(inform=out["inform"].cpu().numpy(), iters=out["iters"].cpu().numpy(), nfev=out["nfev"].cpu().numpy(), obj=out["objective"].cpu().numpy(), x=x.cpu().numpy())
        print(f"{name} device hessian={h}: {dt*1e3:.2f} ms  inform {np.bincount(res[h]['inform'], minlength=1).tolist()} majors mean {res[h]['iters'].mean():.1f} max {res[h]['iters'].max()} nfev mean {res[h]['nfev'].mean():.1f}", flush=True)
    if "--time" in sys.argv: continue
    nref = min(nb, 16)
    ref = orc.solve_batch(spec, lo[:nref], up[:nref], np.ones((nref, spec.nC)), orc.default_opts(hessian=3), nthreads=8)
    d = res[3]
    print(f"   oracle hessian=3: inform {ref['inform'].tolist()} majors {ref['iters'].tolist()}")
    print(f"   device          : inform {d['inform'][:nref].tolist()} majors {d['iters'][:nref].tolist()}")
    rel = np.abs(d["obj"][:nref] - ref["objective"]) / np.maximum(1.0, np.abs(ref["objective"]))
    print(f"   objective rel diff max {rel.max():.2e}  |dx| max {np.abs(d['x'][:nref] - ref['x']).max():.2e}; vs device hessian=2: rel {(np.abs(d['obj'] - res[2]['obj']) / np.maximum(1, np.abs(res[2]['obj']))).max():.2e}")
