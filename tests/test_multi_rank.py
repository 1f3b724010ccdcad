"""World-size-2 rehearsal of the sharding used by bench.py --gpus N (gloo, CPU): every rank takes its slice of one global problem
stream -- through ntg_amd.shard, the SAME functions bench.py calls -- solves it (here with the CPU oracle standing in for the kernel,
which needs a GPU) and the results are gathered once at the end.  Checks that the slices tile the stream without overlap, equal and
ragged, and that the gathered result equals the single-rank result."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ntg_amd import shard


def _worker(rank, world, port, total, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from ntg_amd import configs as cf, shard as sh
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = cf.config_K0()
    lo_all, up_all = cf.kincar_random_bounds(1, total)               # one global stream, as in bench.py
    sl = sh.rank_slice(total, world, rank)
    n = sl.stop - sl.start
    r = orc.solve_batch(spec, lo_all[sl], up_all[sl], np.ones((n, spec.nC)), orc.default_opts())
    gx, go, gi, gk = sh.gather_results(torch.tensor(r["x"]), torch.tensor(r["objective"]), total, world,
                                       inform=torch.tensor(r["inform"], dtype=torch.int32), iters=torch.tensor(r["iters"], dtype=torch.int32))   # the only collective of the path
    tmax = sh.max_over_ranks(float(rank + 1), world)                  # max-over-ranks timing
    if rank == 0:
        ret["x"] = gx.numpy(); ret["obj"] = go.numpy(); ret["tmax"] = tmax
        ret["inform"] = gi.numpy(); ret["iters"] = gk.numpy(); ret["int_dtype"] = str(gi.dtype)
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("total", [12, 7])   # equal slices (all_gather_into_tensor) and ragged ones (all_gather)
def test_two_rank_sharding_and_gather(total):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from ntg_amd import configs as cf
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, 29000 + (os.getpid() + total) % 2000, total, ret), nprocs=world, join=True)
        spec = cf.config_K0()
        lo, up = cf.kincar_random_bounds(1, total)
        ref = orc.solve_batch(spec, lo, up, np.ones((total, spec.nC)), orc.default_opts())
        assert np.array_equal(ret["x"], ref["x"]) and np.array_equal(ret["obj"], ref["objective"])
        assert np.array_equal(ret["inform"], ref["inform"]) and np.array_equal(ret["iters"], ref["iters"]) and ret["int_dtype"] == "torch.int32"
        assert ret["iters"].min() >= 1
        assert ret["tmax"] == 2.0


def test_slices_tile_the_stream():
    for total in (1, 7, 4096, 8192, 8191):
        for world in (1, 2, 3, 4, 8):
            owned = np.zeros(total, dtype=int)
            for r in range(world):
                owned[shard.rank_slice(total, world, r)] += 1
            assert (owned == 1).all()
            c = shard.per_rank_counts(total, world)
            assert sum(c) == total and max(c) - min(c) <= 1


def test_imbalance_figure():
    w = np.r_[np.full(8, 3.0), np.full(8, 9.0)]
    im = shard.imbalance(w, 2)
    assert im["per_rank_total"] == [24.0, 72.0] and abs(im["max_over_mean"] - 1.5) < 1e-15
    assert shard.imbalance(np.ones(64), 8)["max_over_mean"] == 1.0
