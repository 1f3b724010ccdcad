import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from ntg_amd import api, configs as cf
spec=cf.config_E(); lo,up=cf.manipulator_bounds(1024)
p=api.Plan(spec,0); x=torch.ones((1024,spec.nC),dtype=torch.float64,device='cuda:0')
o=p.solve(torch.tensor(lo,device='cuda:0'),torch.tensor(up,device='cuda:0'),x,api.default_opts(hessian=3)); torch.cuda.synchronize()
it=o['iters'].cpu().numpy(); idx=np.argsort(-it)[:8]; print('idx',idx.tolist(),'iters',it[idx].tolist())
