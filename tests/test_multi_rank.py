"""World-size-2 rehearsal of the sharding used by bench.py --gpus N (gloo, CPU): every rank takes
its slice of one global problem stream, solves it (here with the CPU oracle standing in for the
kernel, which needs a GPU) and the results are gathered once at the end.  Checks that the slices
tile the stream without overlap and that the gathered result equals the single-rank result."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from ntg_amd import configs as cf
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = cf.config_K0()
    lo_all, up_all = cf.kincar_random_bounds(1, B * world)           # one global stream, as in bench.py
    sl = slice(rank * B, (rank + 1) * B)
    r = orc.solve_batch(spec, lo_all[sl], up_all[sl], np.ones((B, spec.nC)), orc.default_opts())
    x = torch.tensor(r["x"]); obj = torch.tensor(r["objective"])
    gx = torch.empty((world * B, spec.nC), dtype=torch.float64); go = torch.empty(world * B, dtype=torch.float64)
    dist.all_gather_into_tensor(gx, x)                               # the only collective of the path
    dist.all_gather_into_tensor(go, obj)
    t = torch.tensor([float(rank + 1)]); dist.all_reduce(t, op=dist.ReduceOp.MAX)   # max-over-ranks timing
    if rank == 0:
        ret["x"] = gx.numpy(); ret["obj"] = go.numpy(); ret["tmax"] = float(t.item())
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from ntg_amd import configs as cf
    world, B = 2, 6
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, 29000 + os.getpid() % 2000, B, ret), nprocs=world, join=True)
        spec = cf.config_K0()
        lo, up = cf.kincar_random_bounds(1, B * world)
        ref = orc.solve_batch(spec, lo, up, np.ones((B * world, spec.nC)), orc.default_opts())
        assert np.array_equal(ret["x"], ref["x"]) and np.array_equal(ret["obj"], ref["objective"])
        assert ret["tmax"] == 2.0
