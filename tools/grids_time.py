"""one-off: wall time of ntg_plan_set_grids for 16384 grids (device-side algebra) and the per-problem-grid evaluation"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntg_amd import api, configs as cf
dev = "cuda:0"; spec = cf.config_M(); nbg = 16384
k0 = np.asarray(spec.knots[0]); rngg = np.random.default_rng(3)
scale = rngg.uniform(0.6, 1.6, nbg)[:, None]
kn = k0[None, :] * scale
jj = np.minimum(np.searchsorted(k0, spec.bps, side="right") - 1, spec.kninterv[0] - 1)
fr = (np.asarray(spec.bps) - k0[jj]) / (k0[jj + 1] - k0[jj])
bpg = kn[:, jj] + fr[None, :] * (kn[:, jj + 1] - kn[:, jj])
inner = jj < spec.kninterv[0] - 1
bpg = np.maximum(bpg, kn[:, jj]); bpg[:, inner] = np.minimum(bpg[:, inner], np.nextafter(kn[:, jj + 1][:, inner], -np.inf))
bpg[:, -1] = kn[:, -1]
pg = api.Plan(spec, 0)
knd = torch.tensor(np.ascontiguousarray(kn), device=dev); bpd = torch.tensor(np.ascontiguousarray(bpg), device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    pg.set_grids(knd, bpd, with_precond=False)
    torch.cuda.synchronize(); print(f"set_grids({nbg}, no preconditioner): {1e3 * (time.perf_counter() - t):.2f} ms", flush=True)
t = time.perf_counter(); pg.set_grids(knd[:2048], bpd[:2048], with_precond=True); torch.cuda.synchronize()
print(f"set_grids(2048, with preconditioner blocks (device: grid_prec_kernel; NTG_AMD_HOST_PRECOND=1: host threads)): {1e3 * (time.perf_counter() - t):.1f} ms", flush=True)
pg.set_grids(knd, bpd, with_precond=False)
xg = torch.randn((nbg, spec.nC), dtype=torch.float64, device=dev)
og = pg.eval(xg, 2); pg.eval(xg, 2, out=og); torch.cuda.synchronize()
g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g0.record()
for _ in range(10):
    pg.eval(xg, 2, out=og)
g1.record(); torch.cuda.synchronize()
gms = g0.elapsed_time(g1) / 10
gb = nbg * 11712
print(f"per-problem-grid evaluation: {gms:.4f} ms per {nbg}: {gb / (gms * 1e-3) / 1e9:.0f} GB/s = {gb / (gms * 1e-3) / 1e9 / 8000:.3f} of the HBM spec", flush=True)
