import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import orc
from ntg_amd import api, configs as cf
from gpu_common import dev
from test_gpu_linineq import _spec, _bounds
np.set_printoptions(linewidth=200, precision=6)
flags = [0] * 12; flags[9] = 1; flags[11] = 1
spec = _spec("B", flags); p = api.Plan(spec, 0)
nb = 8
lo, up = _bounds("B", nb)
x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
out = p.solve(dev(lo), dev(up), x, api.default_opts(hessian=int(sys.argv[1]), itlim=3000), want_lambda=True)
lam = out["clambda"].cpu().numpy(); xg = x.cpu().numpy(); A = p.tables()["A"]
for i in range(nb):
    ref = orc.solve_one(spec, lo[i], up[i], np.ones(spec.nC), orc.default_opts(hessian=int(sys.argv[1]), itlim=3000))
    g = orc.eval_batch(spec, xg[i][None], 2)["g"][0]
    ll, lr = lam[i, spec.nC:spec.nC + 12], ref["clambda"][spec.nC:spec.nC + 12]
    Ax = A @ xg[i]
    print(i, 'inf', int(out['inform'][i]), ref['inform'], 'it', int(out['iters'][i]), ref['iters'])
    print('  ll', ll[6:]); print('  lr', lr[6:])
    print('  res gpu %.3e  res orc %.3e' % (np.abs(g - A.T @ ll).max(), np.abs(orc.eval_batch(spec, ref['x'][None], 2)['g'][0] - A.T @ lr).max()))
    print('  Ax-lo', (Ax - lo[i])[[9, 11]], 'up-Ax', (up[i] - Ax)[[9, 11]])
    # least-squares multipliers from the GPU point for comparison
    lls = np.linalg.lstsq(A.T, g, rcond=None)[0]
    print('  lsq', lls[6:])
