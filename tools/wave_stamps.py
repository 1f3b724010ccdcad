"""Profiling one-off: phase clock of the wave kernel (NTG_AMD_STAMPS=1 -> clambda[b][0..7] = s_memtime ticks per phase).
Needs a variant library built with -DNTGW_STAMPS (tools/mkvariant.sh stamps -DNTGW_STAMPS), passed as NTG_AMD_LIB: the shipped wave kernel carries no clock."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NTG_AMD_STAMPS"] = "1"
from ntg_amd import api, configs as cf
api.LIB_PATH = os.environ.get("NTG_AMD_LIB", api.LIB_PATH)   # variant builds (tools/mkvariant.sh)
which = sys.argv[1] if len(sys.argv) > 1 else "M"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
mode = sys.argv[3] if len(sys.argv) > 3 else "fixed50"
spec, ncars = {"M": (cf.config_M(), 3), "B": (cf.config_B(), 1)}[which]
opts = {"fixed50": api.default_opts(itlim=50, fixed_iters=1, hessian=0), "conv_h1": api.default_opts(hessian=1, itlim=50), "conv_h0": api.default_opts(hessian=0)}[mode]
dev = torch.device("cuda:0")
lo, up = cf.kincar_random_bounds(ncars, B)
plan = api.Plan(spec, 0)
x = torch.ones((B, spec.nC), dtype=torch.float64, device=dev)
for _ in range(2):
    x.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = plan.solve(torch.tensor(lo, device=dev), torch.tensor(up, device=dev), x, opts, want_lambda=True); e1.record(); torch.cuda.synchronize()
tk = out["clambda"][:, :8].cpu().numpy()
names = ["setup", "eval", "project", "sweep", "W0", "rest", "reduce3", "lsstep"]
tot = tk.sum(axis=1).mean()
print(f"{which} B={B} {mode}: launch {e0.elapsed_time(e1):.3f} ms; ticks per problem {tot:.0f}")
for n, v in zip(names, tk.mean(axis=0)):
    print(f"  {n:8s} {v:12.0f}  {100 * v / tot:5.1f} %")
