"""one-off: timing of funobj + funcon with banded Jacobian rows (configs D and E)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntg_amd import api, configs as cf
api.LIB_PATH = os.environ.get("NTG_AMD_LIB", api.LIB_PATH)
dev = "cuda:0"
for key, mk, nbJ in (("D", cf.config_D, 4096), ("E", cf.config_E, 2048)):
    specJ = mk(); planJ = api.Plan(specJ, 0)
    xJ = torch.randn((nbJ, specJ.nC), dtype=torch.float64, device=dev)
    oJ = planJ.eval(xJ, 2)
    planJ.eval(xJ, 2, out=oJ); torch.cuda.synchronize()
    j0, j1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    j0.record()
    for _ in range(5):
        planJ.eval(xJ, 2, out=oJ)
    j1.record(); torch.cuda.synchronize()
    msJ = j0.elapsed_time(j1) / 5
    bJ = nbJ * specJ.eval_bytes()
    print(f"jacobian assembly {key}: {msJ:.4f} ms per {nbJ}; {bJ / (msJ * 1e-3) / 1e9:.0f} GB/s = {bJ / (msJ * 1e-3) / 1e9 / 8000:.3f} of peak", flush=True)
