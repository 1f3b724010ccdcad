import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import orc
from ntg_amd import api, configs as cf
from gpu_common import plan_for, dev
name, ncars = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("M4b", 2)
itl = int(sys.argv[3]) if len(sys.argv) > 3 else 50
p = plan_for(name); spec = p.spec
nb = 16
lo, up = cf.kincar_random_bounds(ncars, nb)
ref = orc.solve_batch(spec, lo, up, np.ones((nb, spec.nC)), orc.default_opts(itlim=itl, fixed_iters=1), nthreads=8)
for mode in ("wave", "wg", "noagpr"):
    os.environ.pop("NTG_AMD_NOWAVE", None); os.environ.pop("NTG_AMD_WAVE_NOAGPR", None)
    if mode == "wg": os.environ["NTG_AMD_NOWAVE"] = "1"
    if mode == "noagpr": os.environ["NTG_AMD_WAVE_NOAGPR"] = "1"
    x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    out = p.solve(dev(lo), dev(up), x, api.default_opts(itlim=itl, fixed_iters=1))
    torch.cuda.synchronize()
    nf = out["nfev"].cpu().numpy(); F = out["objective"].cpu().numpy()
    print(mode, p.solve_kernel(nb, api.default_opts(itlim=itl, fixed_iters=1)), "nfev", nf.tolist(), "ref", ref["nfev"].tolist())
    print("   rel dF", np.abs(F - ref["objective"]) / np.abs(ref["objective"]))
