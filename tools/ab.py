"""A/B timing of library variants on ONE box:  python tools/ab.py libA.so libB.so ...   (each in its own process; M, 4096 problems)
modes: fixed50 (headline), conv_h0 (cold start to convergence), conv_h1 (preconditioned)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from ntg_amd import api, configs as cf
api.LIB_PATH = os.environ.get("NTG_AMD_LIB", api.LIB_PATH)
dev = torch.device("cuda:0")
spec = cf.config_M(); B = 4096
lo, up = cf.kincar_random_bounds(3, B)
lo = torch.tensor(lo, device=dev); up = torch.tensor(up, device=dev)
plan = api.Plan(spec, 0)
res = []
for name, opts in (("fixed50", api.default_opts(itlim=50, fixed_iters=1, hessian=0)), ("conv_h0", api.default_opts(hessian=0)), ("conv_h1", api.default_opts(hessian=1, itlim=50))):
    work = torch.empty(plan.workspace_bytes(B, opts), dtype=torch.uint8, device=dev)
    x = torch.ones((B, spec.nC), dtype=torch.float64, device=dev)
    for _ in range(3):
        x.fill_(1.0); out = plan.solve(lo, up, x, opts, work=work)
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        x.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = plan.solve(lo, up, x, opts, work=work); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    res.append(f"{name} med {ts[len(ts)//2]:.3f} min {ts[0]:.3f} ms nfev {out['nfev'].float().mean().item():.2f} F {out['objective'].sum().item():.10e}")
print(" | ".join(res))
''' % ROOT
for rep in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ); env["NTG_AMD_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"{os.path.basename(lib):28s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
