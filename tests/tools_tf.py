import sys, os, numpy as np, torch
os.environ["NTG_AMD_STAMPS"] = "2"
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import orc
from ntg_amd import api, configs as cf
from gpu_common import dev
spec = cf.config_T(); spec.ltc = np.zeros((0, spec.nz))
p = api.Plan(spec, 0)
rng = np.random.default_rng(5)
nb = 6
tab = orc.export_tables(spec)
blin = (rng.normal(size=(nb, spec.nC)) * 0.3) @ tab["A"].T
lo = np.zeros((nb, spec.nbounds)); up = np.zeros((nb, spec.nbounds))
lo[:, 0:4] = up[:, 0:4] = blin
lo[:, 4], up[:, 4] = 0.2, 3.0
lo[:, 5], up[:, 5] = -1e20, 40.0
lo[:, 6], up[:, 6] = -6.0, 6.0
lo[:, 7] = up[:, 7] = 0.5
x0 = np.ones((nb, spec.nC))
P = spec.nbps
x = dev(x0)
out = p.solve(dev(lo), dev(up), x, api.default_opts(hessian=0, itlim=3000), want_lambda=True)
xg = x.cpu().numpy(); lam = out['clambda'].cpu().numpy()
for i in range(nb):
    ev = orc.eval_batch(spec, xg[i][None], 0)["c"][0]
    print(i, 'inf', int(out['inform'][i]), 'it', int(out['iters'][i]), 'c1max %.6f' % np.abs(ev[1+P:1+2*P]).max(), 'c0max %.4f' % ev[1:1+P].max(), 'ci %.6f cf %.6f' % (ev[0], ev[-1]),
          'rv %.3e mu %.1e outer %d sri %.2e rvprev %.2e inner %d mfres %.3e F %.6f' % tuple(lam[i, :8]))
