"""one-off: funobj + funcon with banded Jacobian rows at reduced grids (is the emission bound per CU or chip-wide?)"""
import os, sys, subprocess
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ntg_amd import api, configs as cf
    for key, mk, nb in (("D", cf.config_D, 4096), ("E", cf.config_E, 2048)):
        spec = mk(); plan = api.Plan(spec, 0)
        x = torch.randn((nb, spec.nC), dtype=torch.float64, device="cuda:0")
        o = plan.eval(x, 2); plan.eval(x, 2, out=o); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): plan.eval(x, 2, out=o)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"grid {os.environ.get('NTG_AMD_EVAL_GRID')} {key}: {ms:.4f} ms per {nb}", flush=True)
else:
    for g in (1024, 256, 128, 64, 32, 8):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, NTG_AMD_EVAL_GRID=str(g)))
