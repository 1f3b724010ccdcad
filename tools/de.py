"""Diagnostic: time the D / E solves (to convergence, collocation preconditioner) at a given batch."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from ntg_amd import api, configs as cf
from gpu_common import dev
for name, nb in (("D", int(sys.argv[1])), ("E", int(sys.argv[2]))):
    spec = cf.config_D() if name == "D" else cf.config_E()
    lo, up = (cf.quadrotor_bounds if name == "D" else cf.manipulator_bounds)(nb)
    p = api.Plan(spec, 0)
    lo_d, up_d = dev(lo), dev(up)
    o = api.default_opts(hessian=1)
    work = torch.empty(p.workspace_bytes(nb, o), dtype=torch.uint8, device="cuda:0")
    for rep in range(2):
        x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize(); t = time.time()
        out = p.solve(lo_d, up_d, x, o, work=work)
        torch.cuda.synchronize(); dt = time.time() - t
    inf = out["inform"].cpu().numpy(); it = out["iters"].cpu().numpy(); nf = out["nfev"].cpu().numpy()
    print(name, 'batch', nb, 'time %.3f s' % dt, '%.1f traj/s' % (nb / dt), 'inform', np.bincount(inf, minlength=10).tolist(),
          'iters mean %.0f max %d' % (it.mean(), it.max()), 'nfev mean %.0f' % nf.mean(), 'work %.1f GB' % (work.numel() / 1e9), flush=True)
