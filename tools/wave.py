"""Profiling one-off (not a test): the wave kernel against sqp_kernel on the headline workload -- time, iteration counts, agreement."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntg_amd import api, configs as cf
api.LIB_PATH = os.environ.get("NTG_AMD_LIB", api.LIB_PATH)   # variant builds (tools/mkvariant.sh)

def run(spec, ncars, B, opts, reps=5):
    dev = torch.device("cuda:0")
    lo, up = cf.kincar_random_bounds(ncars, B)
    lo = torch.tensor(lo, device=dev); up = torch.tensor(up, device=dev)
    plan = api.Plan(spec, 0)
    work = torch.empty(plan.workspace_bytes(B, opts), dtype=torch.uint8, device=dev)
    x = torch.ones((B, spec.nC), dtype=torch.float64, device=dev)
    for _ in range(2):
        x.fill_(1.0); out = plan.solve(lo, up, x, opts, work=work)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = 0.0
    for _ in range(reps):
        x.fill_(1.0); e0.record(); out = plan.solve(lo, up, x, opts, work=work); e1.record(); torch.cuda.synchronize()
        ms += e0.elapsed_time(e1)
    return ms / reps, x.clone(), {k: v.clone() for k, v in out.items()}

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "M"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    spec, ncars = {"M": (cf.config_M(), 3), "B": (cf.config_B(), 1), "M4": (cf._kincar_spec(2, 6, 3, 20, 101, 5.0, "M4"), 2)}[which]
    for name, opts in (("fixed50", api.default_opts(itlim=50, fixed_iters=1, hessian=0)), ("conv_h1", api.default_opts(hessian=1, itlim=50)),
                       ("conv_h0", api.default_opts(hessian=0))):
        res = {}
        for mode in ("wave", "wg"):
            if mode == "wg":
                os.environ["NTG_AMD_NOWAVE"] = "1"
            else:
                os.environ.pop("NTG_AMD_NOWAVE", None)
            ms, x, out = run(spec, ncars, B, opts)
            res[mode] = (ms, x, out)
            inf = out["inform"].cpu().numpy()
            print(f"{which} B={B} {name:8s} {mode:4s}: {ms:8.3f} ms  {B / ms * 1e3:12.0f} traj/s  iters mean {out['iters'].float().mean().item():.2f} "
                  f"nfev mean {out['nfev'].float().mean().item():.2f} inform {dict(zip(*np.unique(inf, return_counts=True)))}", flush=True)
        xw, xg = res["wave"][1], res["wg"][1]
        ow, og = res["wave"][2], res["wg"][2]
        print(f"   max |dx| {float((xw - xg).abs().max()):.3e}  max rel dF {float(((ow['objective'] - og['objective']).abs() / og['objective'].abs()).max()):.3e} "
              f"nfev equal {bool((ow['nfev'] == og['nfev']).all())} iters equal {bool((ow['iters'] == og['iters']).all())}", flush=True)
