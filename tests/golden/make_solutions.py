"""Golden solutions of BASELINE configs D and E at full size: the CPU oracle's structured
Newton solve (oracle/sqp.c, hessian = 2) of the first 8 problems of each config -> tests/golden/sol_{D,E}.npz
(x*, objective, multipliers, inform, majors).  The GPU tests compare the batched HIP solve against these on the GPU box,
where the oracle would need minutes per problem in its BFGS mode.  Run:  python tests/golden/make_solutions.py
`python tests/golden/make_solutions.py qp` writes sol_qp_{D,E}.npz: the same problems (8 of D, 4 of E) by the oracle's QP-based SQP step
(hessian = 3, oracle/sqp.c sqpqp_run; the dense prototype needs ~1 minute per config-E problem)."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
import orc
from ntg_amd import configs as cf

NPROB = 8
if len(sys.argv) > 1 and sys.argv[1] == "qp":
    for name, spec, bounds, nprob in (("D", cf.config_D(), cf.quadrotor_bounds, 8), ("E", cf.config_E(), cf.manipulator_bounds, 4)):
        lo, up = bounds(nprob)
        from concurrent.futures import ThreadPoolExecutor   # (the C call releases the interpreter lock)
        with ThreadPoolExecutor(max_workers=4) as ex:
            rs = list(ex.map(lambda b: orc.solve_one(spec, lo[b], up[b], np.ones(spec.nC), orc.default_opts(hessian=3)), range(nprob)))
        for b, r in enumerate(rs): print(name, b, "inform", r["inform"], "majors", r["iters"], "objective", r["objective"], flush=True)
        np.savez_compressed(os.path.join(HERE, f"sol_qp_{name}.npz"), x=np.array([r["x"] for r in rs]), objective=np.array([r["objective"] for r in rs]),
                            clambda=np.array([r["clambda"] for r in rs]), inform=np.array([r["inform"] for r in rs], dtype=np.int32),
                            iters=np.array([r["iters"] for r in rs], dtype=np.int32), lower=lo, upper=up)
    sys.exit(0)
for name, spec, bounds in (("D", cf.config_D(), cf.quadrotor_bounds), ("E", cf.config_E(), cf.manipulator_bounds)):
    lo, up = bounds(NPROB)
    xs, objs, lams, infs, its = [], [], [], [], []
    for b in range(NPROB):
        r = orc.solve_one(spec, lo[b], up[b], np.ones(spec.nC), orc.default_opts(hessian=2))
        xs.append(r["x"]); objs.append(r["objective"]); lams.append(r["clambda"]); infs.append(r["inform"]); its.append(r["iters"])
        print(name, b, "inform", r["inform"], "majors", r["iters"], "objective", r["objective"], flush=True)
    np.savez_compressed(os.path.join(HERE, f"sol_{name}.npz"), x=np.array(xs), objective=np.array(objs), clambda=np.array(lams),
                        inform=np.array(infs, dtype=np.int32), iters=np.array(its, dtype=np.int32), lower=lo, upper=up)
