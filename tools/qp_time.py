"""Diagnostic: QP-based SQP step (hessian = 3) against the augmented-Lagrangian Newton mode (2): time, work counters (NTG_AMD_STAMPS=3) and phase
clock (NTG_AMD_STAMPS=1 with a -DNTG_CLOCK variant library: tools/mkvariant2.sh clock fam_manip -DNTG_CLOCK -DNTG_SLIM; NTG_AMD_LIB=...).
python tools/qp_time.py [O|D|E] [batch] [modes, e.g. 32]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ntg_amd import api, configs as cf
which = sys.argv[1] if len(sys.argv) > 1 else "E"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
modes = [int(c) for c in (sys.argv[3] if len(sys.argv) > 3 else "32")]
spec, bnd = {"O": (cf.config_O(), cf.obstacle_bounds), "D": (cf.config_D(), cf.quadrotor_bounds), "E": (cf.config_E(), cf.manipulator_bounds)}[which]
lo, up = bnd(B)
dev = torch.device("cuda:0")
plan = api.Plan(spec, 0)
lo_t, up_t = torch.tensor(lo, device=dev), torch.tensor(up, device=dev)
st = os.environ.get("NTG_AMD_STAMPS")
for h in modes:
    opts = api.default_opts(hessian=h)
    x = torch.ones((B, spec.nC), dtype=torch.float64, device=dev)
    out = plan.solve(lo_t, up_t, x, opts, want_lambda=bool(st)); torch.cuda.synchronize()
    x.fill_(1.0); t0 = time.time()
    out = plan.solve(lo_t, up_t, x, opts, want_lambda=bool(st)); torch.cuda.synchronize()
    dt = time.time() - t0
    inf = out["inform"].cpu().numpy(); it = out["iters"].cpu().numpy(); nf = out["nfev"].cpu().numpy()
    print(f"{which} hessian={h} batch {B}: {dt*1e3:.2f} ms -> {B/dt:.0f} traj/s; inform {np.bincount(inf).tolist()} majors mean {it.mean():.1f} max {it.max()} nfev mean {nf.mean():.1f}", flush=True)
    if st == "3":
        d = out["clambda"][:, :11].cpu().numpy()
        print("   per problem: factorisations %.1f (not PD %.1f) band solves %.1f | passive-set solves %.1f columns %.1f majors with a full working set %.2f (max %d)" % (d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d[:, 6].mean(), d[:, 7].mean(), d[:, 8].mean(), d[:, 8].max()), "| columns from the tables %.1f, problems that fell back %d" % (d[:, 10].mean(), (d[:, 9] > 0).sum()))
    if st == "1":
        tk = out["clambda"][:, :8].cpu().numpy()
        names = ["qp column", "eval", "qp search/solve/step", "assemble", "W g", "rest", "Bpass", "factor"] if h == 3 else ["setup", "eval", "project", "assemble", "solve", "rest", "Bpass", "factor"]
        tot = tk.sum(axis=1).mean()
        print("   ticks/problem %.0f (%.2f ms at 100 MHz)" % (tot, tot / 1e5), {n: "%.1f%%" % (100 * tk[:, i].mean() / tot) for i, n in enumerate(names)})
    if os.environ.get("QP_HIST"):
        print("   majors histogram:", np.bincount(np.minimum(it, 99) // 5).tolist(), "(bins of 5)")
