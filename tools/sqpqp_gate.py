"""Gate of VERDICT r3 item 2 (oracle prototype): QP-based SQP step (hessian = 3) against the augmented-Lagrangian Newton mode (hessian = 2)
on the obstacle class, the reduced quadrotor and the reduced two-arm manipulator.  python tools/sqpqp_gate.py [O|D2|E2|all] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import orc
from ntg_amd import configs as cf
which = sys.argv[1] if len(sys.argv) > 1 else "all"
sets = {"O": (cf.config_O(), cf.obstacle_bounds, 16), "D2": (cf.config_D(ninterv=10), cf.quadrotor_bounds, 12),
        "E2": (cf.config_E(ninterv=20, narms=2), lambda n: cf.manipulator_bounds(n, narms=2), 8)}
for name, (spec, bnd, n) in sets.items():
    if which not in ("all", name): continue
    if len(sys.argv) > 2: n = int(sys.argv[2])
    lo, up = bnd(n)
    res = {}
    for h in (2, 3):
        t = time.time()
        r = orc.solve_batch(spec, lo, up, np.ones((n, spec.nC)), orc.default_opts(hessian=h), nthreads=8)
        res[h] = r
        print(f"{name} hessian={h}: majors mean {r['iters'].mean():.1f} max {r['iters'].max()} nfev mean {r['nfev'].mean():.1f} inform {np.bincount(r['inform'])} ({time.time()-t:.1f} s)", flush=True)
    d = np.abs(res[3]["objective"] - res[2]["objective"]) / np.abs(res[2]["objective"])
    print(f"   rel objective difference: max {d.max():.2e}; sqp-qp better (lower) in {(res[3]['objective'] < res[2]['objective'] - 1e-9 * np.abs(res[2]['objective'])).sum()} of {n}; |dx| max {np.abs(res[3]['x'] - res[2]['x']).max():.2e}")
    print("   majors AL:", res[2]["iters"].tolist(), " QP:", res[3]["iters"].tolist())
