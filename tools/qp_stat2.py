import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import orc
from ntg_amd import api, configs as cf
from gpu_common import dev
spec = cf.config_D(); lo, up = cf.quadrotor_bounds(2)
p = api.Plan(spec, 0)
tb = orc.export_tables(spec, lo[1], up[1]); A = tb["A"]
pinned = np.abs(A).sum(axis=0) > 0
res = {}
for h in (2, 3):
    x = torch.ones((2, spec.nC), dtype=torch.float64, device="cuda:0")
    out = p.solve(dev(lo), dev(up), x, api.default_opts(hessian=h), want_lambda=True); torch.cuda.synchronize()
    lam = out["clambda"].cpu().numpy()[1]
    ev = p.eval(x, 2, want_cjac=True) if "want_cjac" in p.eval.__code__.co_varnames else p.eval(x, 2)
    g = ev["g"].cpu().numpy()[1]
    n, m = spec.nC, spec.nclin
    ref = orc.eval_batch(spec, x.cpu().numpy()[1:2], 2); J = ref["cJac"][0]; g2 = ref["g"][0]
    r = g2 - A.T @ lam[n:n + m] - J.T @ lam[n + m:]
    print(h, "max |r| pinned %.3e free %.3e ; |g| pinned %.3e free %.3e ; |lam_nl| %.3e ; x[:3]" % (np.abs(r[pinned]).max(), np.abs(r[~pinned]).max(), np.abs(g2[pinned]).max(), np.abs(g2[~pinned]).max(), np.abs(lam[n+m:]).max()), x.cpu().numpy()[1][:3])
    res[h] = (x.cpu().numpy()[1], lam)
print("dx", np.abs(res[2][0] - res[3][0]).max(), "dlam_lin rel", np.abs(res[2][1][spec.nC:spec.nC+spec.nclin] - res[3][1][spec.nC:spec.nC+spec.nclin]).max() / np.abs(res[2][1][spec.nC:spec.nC+spec.nclin]).max())
# which point do the reported linear multipliers belong to?
x3, lam3 = res[3]
n, m = spec.nC, spec.nclin
ref = orc.eval_batch(spec, x3[None], 2); g3 = ref["g"][0]
lls = np.linalg.lstsq(A.T[pinned], g3[pinned], rcond=None)[0]
print("mode 3: reported lam_lin vs least squares at the reported x: rel diff %.3e" % (np.abs(lls - lam3[n:n+m]).max() / np.abs(lls).max()))
x2, lam2 = res[2]
ref2 = orc.eval_batch(spec, x2[None], 2); g2 = ref2["g"][0]
lls2 = np.linalg.lstsq(A.T[pinned], g2[pinned], rcond=None)[0]
print("mode 2: same: %.3e ; lam_ls(x3) vs lam_ls(x2): %.3e ; reported lam3 vs lam_ls(x2): %.3e" % (np.abs(lls2 - lam2[n:n+m]).max() / np.abs(lls2).max(), np.abs(lls - lls2).max() / np.abs(lls2).max(), np.abs(lam3[n:n+m] - lls2).max() / np.abs(lls2).max()))
print("A x - b at x3: %.3e ; at x2: %.3e" % (np.abs(A @ x3 - lo[1][:m]).max(), np.abs(A @ x2 - lo[1][:m]).max()))
