"""Driver for rocprofv3 --pmc (matrix-core counters): a few preconditioned solves of configs M and E, nothing else."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntg_amd import api, configs as cf
for mk, bn, nb in ((cf.config_M, lambda n: cf.kincar_random_bounds(3, n), 4096), (cf.config_E, cf.manipulator_bounds, 256)):
    spec = mk(); p = api.Plan(spec, 0)
    lo, up = bn(nb)
    lo = torch.tensor(lo, device="cuda:0"); up = torch.tensor(up, device="cuda:0")
    o = api.default_opts(hessian=1)
    w = torch.empty(p.workspace_bytes(nb, o), dtype=torch.uint8, device="cuda:0")
    for _ in range(3):
        x = torch.ones((nb, spec.nC), dtype=torch.float64, device="cuda:0")
        p.solve(lo, up, x, o, work=w)
    torch.cuda.synchronize()
