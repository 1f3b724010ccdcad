"""Diagnostic (not a test): KKT residuals of the structured Newton mode at BASELINE sizes.  python tools/kkt.py [D|E] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from ntg_amd import api, configs as cf
which = sys.argv[1]; B = int(sys.argv[2])
spec, bounds = (cf.config_D(), cf.quadrotor_bounds) if which == "D" else (cf.config_E(), cf.manipulator_bounds)
lo, up = bounds(B)
dev = torch.device("cuda:0")
p = api.Plan(spec, 0)
for hess in (2, 1):
    x = torch.ones((B, spec.nC), dtype=torch.float64, device=dev)
    out = p.solve(torch.tensor(lo, device=dev), torch.tensor(up, device=dev), x, api.default_opts(hessian=hess), want_lambda=True)
    torch.cuda.synchronize()
    lam = out["clambda"].cpu().numpy(); inf = out["inform"].cpu().numpy()
    A = p.tables()["A"]; P = spec.nbps; nl0 = spec.nclin_rows if hasattr(spec, "nclin_rows") else spec.lic.shape[0] + spec.ltc.shape[0] + spec.lfc.shape[0]
    stat, feas, comp, lin = [], [], [], []
    for s in range(0, B, 8):
        xs = x[s:s + 8]
        ev = p.eval(xs, 2, want_dense_jac=True)
        g = ev["g"].cpu().numpy(); J = ev["cJac"].cpu().numpy(); c = ev["c"].cpu().numpy(); xg = xs.cpu().numpy()
        for i in range(xs.shape[0]):
            b = s + i
            ll, ln = lam[b, spec.nC:spec.nC + spec.nclin], lam[b, spec.nC + spec.nclin:]
            r = g[i] - A.T @ ll - J[i].T @ ln
            stat.append(np.abs(r).max() / max(1.0, np.abs(g[i]).max()))
            lin.append(np.abs(A @ xg[i] - lo[b][:spec.nclin]).max())
            f = 0.0; cm = 0.0
            for j in range(spec.nnltc):
                cj = c[i, j * P:(j + 1) * P]; lj = ln[j * P:(j + 1) * P]
                l, u = lo[b, nl0 + j], up[b, nl0 + j]
                f = max(f, (l - cj).max() / (1 + abs(l)), (cj - u).max() / (1 + abs(u)))
                slack = np.minimum(cj - l, u - cj)
                cm = max(cm, np.abs(lj * np.minimum(slack, 1.0)).max() / max(1.0, np.abs(lj).max()))
            feas.append(f); comp.append(cm)
    stat, feas, comp, lin = map(np.array, (stat, feas, comp, lin))
    print(f"{which} hessian={hess}: inform {np.bincount(inf)} stationarity max {stat.max():.2e} median {np.median(stat):.2e}; nonlinear violation max {feas.max():.2e}; "
          f"complementarity max {comp.max():.2e}; linear residual max {lin.max():.2e}; majors max {out['iters'].max().item()}")
