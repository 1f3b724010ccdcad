"""Diagnostic (not a test): structured Newton mode (hessian = 2) of sqp_kernel against the oracle.
python tools/newton.py [O|D2|E2|D|E] [batch] [nref]
The phase clock (NTG_AMD_STAMPS=1 or 4) needs a variant library built with -DNTG_CLOCK: tools/mkvariant2.sh clock "fam_quadrotor fam_manip fam_obstacle" -DNTG_CLOCK, then NTG_AMD_LIB=ntg_amd/variants/libntg_clock.so (the shipped kernels carry no clock: 18 registers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import orc
from ntg_amd import api, configs as cf
api.LIB_PATH = os.environ.get("NTG_AMD_LIB", api.LIB_PATH)

which = sys.argv[1] if len(sys.argv) > 1 else "O"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nref = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if which == "O": spec = cf.config_O(); lo, up = cf.obstacle_bounds(B)
if which == "D2": spec = cf.config_D(ninterv=10); lo, up = cf.quadrotor_bounds(B)
if which == "E2": spec = cf.config_E(ninterv=20, narms=2); lo, up = cf.manipulator_bounds(B, narms=2)
if which == "D": spec = cf.config_D(); lo, up = cf.quadrotor_bounds(B)
if which == "E": spec = cf.config_E(); lo, up = cf.manipulator_bounds(B)
dev = torch.device("cuda:0")
plan = api.Plan(spec, 0)
for hess in (2, 1):
    x = torch.ones((B, spec.nC), dtype=torch.float64, device=dev)
    lo_t, up_t = torch.tensor(lo, device=dev), torch.tensor(up, device=dev)
    opts = api.default_opts(hessian=hess)
    out = plan.solve(lo_t, up_t, x, opts)   # warm-up (plan tables, code load)
    torch.cuda.synchronize()
    x.fill_(1.0)
    t0 = time.time()
    out = plan.solve(lo_t, up_t, x, opts)
    torch.cuda.synchronize()
    dt = time.time() - t0
    inf = out["inform"].cpu().numpy(); it = out["iters"].cpu().numpy(); nf = out["nfev"].cpu().numpy()
    print(f"{which} hessian={hess} batch {B}: {dt*1e3:.2f} ms -> {B/dt:.0f} traj/s; inform {np.bincount(inf)} majors mean {it.mean():.1f} max {it.max()} nfev mean {nf.mean():.1f}", flush=True)
    if os.environ.get("NTG_AMD_STAMPS") == "4" and hess == 2:
        x.fill_(1.0)
        o2 = plan.solve(lo_t, up_t, x, opts, want_lambda=True)
        torch.cuda.synchronize()
        tk = o2["clambda"][:, :8].cpu().numpy()
        print("  assemble: K0 copy ticks/problem %.0f ; all phases" % tk[:, 0].mean(), tk.mean(axis=0))
    if os.environ.get("NTG_AMD_STAMPS") == "1":
        x.fill_(1.0)
        o2 = plan.solve(lo_t, up_t, x, opts, want_lambda=True)
        torch.cuda.synchronize()
        tk = o2["clambda"][:, :8].cpu().numpy()
        names = ["setup", "eval", "project", "hist/assemble", "w0/factor+solve", "rest", "ev.phase1", "ev.phase2"]
        if hess == 2: names = ["setup", "eval", "project", "assemble", "solve", "rest", "Bpass", "factor"]
        tot = tk[:, :6].sum(axis=1).mean() + (tk[:, 6:8].sum(axis=1).mean() if hess == 2 else 0.0)
        print("  cycles/problem %.0f (%.1f us at 100 MHz clock)" % (tot, tot / 100.0), {n: "%.1f%%" % (100 * tk[:, i].mean() / tot) for i, n in enumerate(names)})
    if os.environ.get("NTG_AMD_STAMPS") == "2" and hess == 2:
        x.fill_(1.0)
        o2 = plan.solve(lo_t, up_t, x, opts, want_lambda=True)
        torch.cuda.synchronize()
        dd = o2["clambda"][:, :8].cpu().numpy()
        print("  diag [rv, mu, outer, sri, rvprev, inner_inform, bad pivots, F]:")
        for row in dd[:12]: print("   ", " ".join("%.3g" % v for v in row))
    if hess == 2 and nref > 0:
        r = orc.solve_batch(spec, lo[:nref], up[:nref], np.ones((nref, spec.nC)), orc.default_opts(hessian=2), nthreads=8)
        obj = out["objective"].cpu().numpy()[:nref]
        print("  oracle inform", r["inform"], "iters", r["iters"], "gpu iters", it[:nref])
        print("  rel obj diff", np.abs(obj - r["objective"]) / np.abs(r["objective"]))
        print("  max |x - x_orc|", np.abs(x.cpu().numpy()[:nref] - r["x"]).max(axis=1))
