"""one-off: phase clock of eval_kernel (variant library built with -DNTG_EVAL_CLOCK) on configs D and E"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntg_amd import api, configs as cf
api.LIB_PATH = os.environ.get("NTG_AMD_LIB", api.LIB_PATH)
for key, mk, nb in (("D", cf.config_D, 4096), ("E", cf.config_E, 2048)):
    spec = mk(); plan = api.Plan(spec, 0)
    x = torch.randn((nb, spec.nC), dtype=torch.float64, device="cuda:0")
    o = plan.eval(x, 2); torch.cuda.synchronize()
    print(key, "full grid", flush=True)
    plan.eval(x, 2, out=o); torch.cuda.synchronize()
    os.environ["NTG_AMD_EVAL_GRID"] = "8"
    print(key, "grid 8", flush=True)
    plan.eval(x, 2, out=o); torch.cuda.synchronize()
    del os.environ["NTG_AMD_EVAL_GRID"]
