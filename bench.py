#!/usr/bin/env python3
"""bench.py -- trajectories/sec of the batched SQP hot path on MI355X.

Workload (BASELINE.json metric): 4096 kincar problems per GPU, 6 flat outputs, order-6 splines,
20 intervals, 101 breakpoints (config M of SURVEY.md §8), random initial/final states
(numpy PCG64 seed 20261003), start C = 1, exactly 50 SQP major iterations per problem in the
NPSOL-equivalent mode (identity cold start, no convergence exit) -- one "step" = one batch solve.
Inputs are resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W                 # config M, weak scaling: 4096 problems per GPU
    python bench.py --config D|E --gpus N                         # BASELINE configs[3] / [4]: 4096 / 8192 problems sharded over N GPUs
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      dominant kernel of the timed launch against the HBM roof, algorithmic bytes =
                SURVEY §8d per-evaluation bytes x evaluations actually performed in the launch
  cpu_baseline  the CPU oracle (oracle/, kind "port") solving a bounded sample of the same
                workload on the host cores of this box (rank 0, N=1 only), one pinned thread per physical core
  to_convergence  the metric's own mode ("SQP-to-convergence"): NPSOL-equivalent cold start run to NPSOL's tolerances, and the
                product's default (collocation preconditioner) -- values, majors, inform histograms, rank imbalance of fixed slices
  eval_kernel   the standalone colloc+assembly kernel (npsolCostFunction batched) streamed over
                a large batch -- the HBM-bound view of the colloc+Jacobian assembly path
"""
from __future__ import annotations
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def orc_off(spec):
    """block offset of every breakpoint of output 0 (colloc.c:104-111): (interval index) * (order - mult)"""
    import numpy as np
    kn = np.asarray(spec.knots[0]); l = len(kn) - 1
    left = np.clip(np.searchsorted(kn, np.asarray(spec.bps), side="right"), 1, l)
    return (left - 1) * (spec.order[0] - spec.mult[0])


def csrc_sha() -> str:
    """hash of the device sources: profiles/traffic.json entries carry the hash they were measured on, so a counter value cannot be
    attached to a kernel it was not taken from"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ntg_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def physical_cores() -> int:
    """physical cores of the host (SMT siblings are not cores)"""
    seen = set()
    phys = core = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return len(seen) or (os.cpu_count() or 1)


def cpu_quota() -> float:
    """CPUs this job may actually use: the cgroup's CPU quota (cpu.max / cfs_quota), 0 if unlimited.  The GPU boxes show all 256 hardware
    threads of the host but give a one-GPU job 16 CPUs' worth of time: threads beyond the quota are throttled, not run (rounds 2-3 read the
    resulting 11-12x "scaling" of 128 threads as a property of the oracle; it is the scheduler's)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return 0.0 if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / per if q > 0 else 0.0
    except (OSError, ValueError):
        return 0.0


def newton_mfma_entry(specL, key, cnt, nbL, dtl, MFMA_PEAK_TF):
    """matrix-core work of a structured-Newton launch from the kernel's own counters (factorisations, failed attempts)"""
    import numpy as np
    nfact, nfail = cnt[:, 0].sum(), cnt[:, 1].sum()
    go = {"config_D": 4, "config_E": 3}[key]; cgn = {"config_D": 6, "config_E": 3}[key]
    ngrp = specL.nout // go
    nfree = specL.nC // specL.nout - 2 * specL.maxderiv[0]            # free coefficients per output (flag pinned at both ends)
    nbr = (nfree * go + 15) // 16
    offs = np.asarray(orc_off(specL))
    cnts = np.unique(offs, return_counts=True)[1]
    mfma_fact = ngrp * nbr * 12
    mfma_asm = ngrp * int(sum(3 * ((c * cgn + 3) // 4) for c in cnts))
    flops = 2048.0 * ((nfact - 0.5 * nfail) * mfma_fact + nfact * mfma_asm)
    m_rows = specL.nclin + specL.ncnln
    return {"instr": "v_mfma_f64_16x16x4_f64", "flops_executed": flops, "achieved_tflops": flops / dtl / 1e12,
            "peak_tflops": MFMA_PEAK_TF, "util": flops / dtl / 1e12 / MFMA_PEAK_TF,
            "factorisations_per_problem": float(nfact / nbL), "not_positive_definite_per_problem": float(nfail / nbL),
            "what": "band assembly (M' B M per knot interval) + trailing updates of the band Cholesky; structured count, "
                    "not the dense 2 m^2 nC bound of SURVEY 8d (%.3g flop per iteration per problem)" % (2.0 * m_rows * m_rows * specL.nC)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0,
                    help="problems per GPU (weak) or in all (strong); 0: 4096 per GPU for M / B, the config's whole batch (4096 / 8192) for D / E")
    ap.add_argument("--config", default="M", choices=["M", "B", "D", "E"],
                    help="M (headline) / B: kincar, fixed 50 majors; D (quadrotor) / E (manipulator): structured Newton mode to convergence, "
                         "BASELINE's batch of 4096 / 8192 problems sharded over the GPUs")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--cpu-sample", type=int, default=0, help="problems of the CPU baseline sample (0: 96 per usable core)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: --batch problems per GPU (default for M / B); strong: --batch problems in all, split over the GPUs (default for D / E)")
    ap.add_argument("--hessian", type=int, default=-1, help="configs D / E: 2 structured Newton mode, 3 QP-based SQP step (default: 2 for D, 3 for E)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()
    # the CPU baseline pins one thread per physical core: the OpenMP runtime reads these when it starts
    os.environ.setdefault("OMP_PROC_BIND", "spread"); os.environ.setdefault("OMP_PLACES", "cores")

    import numpy as np
    import torch
    from ntg_amd import api, configs as cf, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")

    large = args.config in ("D", "E")
    spec = {"M": cf.config_M, "B": cf.config_B, "D": cf.config_D, "E": cf.config_E}[args.config]()
    ncars = spec.nout // 2
    scaling = args.scaling or ("strong" if large else "weak")
    batch_arg = args.batch or ({"D": 4096, "E": 8192}[args.config] if large else 4096)
    total = batch_arg if scaling == "strong" else batch_arg * world
    # every rank owns its own contiguous slice of one global problem stream (ntg_amd/shard.py; the gloo test runs the same functions)
    sl = shard.rank_slice(total, world, rank)
    B = sl.stop - sl.start
    bounds_fn = {"D": cf.quadrotor_bounds, "E": cf.manipulator_bounds}.get(args.config, lambda n: cf.kincar_random_bounds(ncars, n))
    lo_all, up_all = bounds_fn(total)
    lo = torch.tensor(lo_all[sl], device=dev)
    up = torch.tensor(up_all[sl], device=dev)
    x0 = torch.ones((B, spec.nC), dtype=torch.float64, device=dev)
    x = x0.clone()

    plan = api.Plan(spec, local)
    # M / B: NPSOL-equivalent identity cold start, exactly --iters majors (fixed work); D / E: structured Newton mode to convergence
    # (E: the QP-based SQP step, hessian = 3 -- 1.9 x the Newton mode's rate, every problem at inform 0; D: the Newton mode, hessian = 2, which
    #  is the faster one there -- the bench line's config_D / config_E entries carry both)
    hess_large = args.hessian if args.hessian >= 0 else (3 if args.config == "E" else 2)
    opts = api.default_opts(hessian=hess_large) if large else api.default_opts(itlim=args.iters, fixed_iters=1, hessian=0)
    work = torch.empty(plan.workspace_bytes(B, opts), dtype=torch.uint8, device=dev)
    out = dict(objective=torch.empty(B, dtype=torch.float64, device=dev),
               inform=torch.empty(B, dtype=torch.int32, device=dev),
               iters=torch.empty(B, dtype=torch.int32, device=dev),
               nfev=torch.empty(B, dtype=torch.int32, device=dev))
    solve_kernel = plan.solve_kernel(B, opts)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        x.copy_(x0)
        if i is not None:
            ev[i][0].record()
        plan.solve(lo, up, x, opts, work=work, out=out)   # launched on torch's current stream: the events bracket the kernel
        if i is not None:
            ev[i][1].record()
        if world > 1:  # the only collective: final gather of the results (RCCL over xGMI)
            shard.gather_results(x, out["objective"], total, world, inform=out["inform"], iters=out["iters"])

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync_all()
    dt = shard.max_over_ranks(time.perf_counter() - t0, world, dev)

    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)
    kern_ms_ranks = [kern_ms]
    if world > 1:   # after the timed region: every rank's own kernel time (HIP events on its launch stream), for the load-balance picture
        kt = torch.tensor([kern_ms], dtype=torch.float64, device=dev)
        kall = torch.empty(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(kall, kt)
        kern_ms_ranks = [float(v) for v in kall.cpu()]
    nfev_total = int(out["nfev"].sum().item())
    iters_np = out["iters"].cpu().numpy()
    inform_np = out["inform"].cpu().numpy()
    value = total * args.steps / dt

    mode_txt = (("QP-based SQP step on the band model (hessian = 3) to convergence from C = 1" if hess_large == 3 else "structured Newton mode (hessian = 2) to convergence from C = 1") if large else
                f"{args.iters} SQP majors (identity cold start, fixed work)")
    res = {
        "metric": "trajectories/sec (batched SQP, kincar 6-output order-6/20-interval; value = the fixed-work mode of BASELINE's "
                  "target: exactly 50 SQP major iterations per problem from an identity cold start; the to-convergence modes are under to_convergence)"
                  if not large else "trajectories/sec (batched SQP to convergence, %s)" % spec.name,
        "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{spec.name}: {B} problems/GPU ({total} in all) x {mode_txt}",
                   "value_mode": ("qp_sqp_to_convergence" if hess_large == 3 else "newton_to_convergence") if large else "fixed_50_majors",
                   "batch_per_gpu": B, "batch_total": total, "nout": spec.nout, "order": spec.order[0], "ninterv": spec.kninterv[0],
                   "nbps": spec.nbps, "nC": spec.nC, "nclin": spec.nclin, "ncnln": spec.ncnln, "sqp_iters": args.iters,
                   "parallelism": f"problems sharded over {world} GPU(s), final all_gather only"},
        "solve_check": {"iters_min": int(iters_np.min()), "iters_max": int(iters_np.max()), "iters_mean": float(iters_np.mean()),
                        "inform_counts": {str(k): int((inform_np == k).sum()) for k in np.unique(inform_np)},
                        "nfev_per_problem": nfev_total / B},
    }
    alg_bytes = nfev_total * spec.eval_bytes()
    ach = alg_bytes / (kern_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections
    # applied; profiles/traffic.json says how).  An entry counts only for the workload AND the device sources it was taken on.
    sha = csrc_sha()
    traffic, traffic_note = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        key = f"{solve_kernel}:{args.config}:{B}:" + (("qp_sqp" if hess_large == 3 else "newton") if large else f"fixed{args.iters}")
        if key in tj:
            if tj[key].get("csrc_sha") == sha:
                traffic = tj[key]["hbm_bytes"]
            else:
                traffic_note = f"profiles/traffic.json has {key} for device sources {tj[key].get('csrc_sha')}, these are {sha}: not attached"
    except Exception:
        pass
    res["roofline"] = {"bound": "hbm", "kernel": solve_kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": kern_ms, "kernel_ms_per_rank": kern_ms_ranks,
                       "alg_bytes_per_launch": alg_bytes, "csrc_sha": sha,
                       "alg_bytes_def": f"{spec.eval_bytes()} B per funobj{'+funcon' if large else ''} evaluation (SURVEY 8d) x {nfev_total} evaluations"}
    if traffic_note:
        res["roofline"]["traffic_note"] = traffic_note
    # The roof that BINDS this kernel is not HBM: the wave kernel runs one wavefront per SIMD, its chain of search directions on chip, and
    # is limited by fp64 vector-instruction issue.  From the SQ counters of a separate rocprofv3 --pmc pass over this launch
    # (profiles/r04_sq_fixed50.json, attached only for the device sources it was measured on): VALU wave-instructions x 4 cycles of pipe
    # occupancy each / cycles the waves were resident.  (A single wave per SIMD cannot issue fp64 back to back: tools/probes/mfma64.hip
    # measures 6.5 ticks per independent v_fma_f64 alone and the same per wave with two waves per SIMD -- half the pipe is the ceiling of
    # this residency, which the 512-register chain tier forces.)
    if not large:
        try:
            sq = json.load(open(os.path.join(ROOT, "profiles", "r04_sq_fixed50.json")))
            if sq.get("csrc_sha") == sha and B == 4096 and args.iters == 50 and args.config == "M":
                ir = sq["issue_roof"]
                res["roofline"]["issue"] = {"bound": "fp64 VALU issue (one wavefront per SIMD)", "achieved": ir["pipe_cycles_at_4_per_inst"], "peak": ir["wave_cycles"],
                                            "unit": "cycles per launch (VALU wave-instructions x 4 / resident wave cycles)", "frac": ir["frac"],
                                            "valu_wave_insts_per_problem": sq["per_problem"]["valu_wave_insts"], "valu_active_frac": ir.get("valu_active_frac"),
                                            "single_wave_ceiling": 0.5, "source": "profiles/r04_sq_fixed50.json"}
            else:
                res["roofline"]["issue_note"] = "profiles/r04_sq_fixed50.json was measured on other device sources or another workload: not attached"
        except Exception:
            pass
    if traffic:
        res["roofline"]["traffic_over_alg"] = traffic / alg_bytes

    if rank == 0 and world == 1 and not args.no_extras and not large:
        # ---- the metric's own mode, SQP to convergence, two ways ----
        def to_conv(opt, nrep=5):
            wc = torch.empty(plan.workspace_bytes(B, opt), dtype=torch.uint8, device=dev)
            for _ in range(2):
                x.copy_(x0); oo = plan.solve(lo, up, x, opt, work=wc)
            torch.cuda.synchronize()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ms = 0.0
            for _ in range(nrep):
                x.copy_(x0); g0.record(); oo = plan.solve(lo, up, x, opt, work=wc); g1.record(); torch.cuda.synchronize()
                ms += g0.elapsed_time(g1)
            ms /= nrep
            inf = oo["inform"].cpu().numpy(); it = oo["iters"].cpu().numpy()
            nf = int(oo["nfev"].sum().item())
            return {"value": B / (ms * 1e-3), "unit": "trajectories/s", "ms_per_batch": ms, "kernel": plan.solve_kernel(B, opt),
                    "converged_frac": float((inf == 0).mean()), "inform_counts": {str(k): int((inf == k).sum()) for k in np.unique(inf)},
                    "iters_mean": float(it.mean()), "iters_min": int(it.min()), "iters_max": int(it.max()), "nfev_per_problem": nf / B,
                    "roofline": {"bound": "hbm", "achieved": nf * spec.eval_bytes() / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": nf * spec.eval_bytes() / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                    # what fixed contiguous slices would cost on 8 GPUs against a perfect balance (majors are the work measure)
                    "rank_imbalance_8": shard.imbalance(it, 8)["max_over_mean"]}, oo
        oc = api.default_opts(hessian=1, itlim=args.iters)
        tc, _ = to_conv(oc)
        tc["mode"] = "collocation-preconditioned BFGS (product default), KKT exit at NPSOL's tolerances"
        t0c, _ = to_conv(api.default_opts(hessian=0), nrep=3)
        t0c["mode"] = "NPSOL-equivalent: identity cold start, full-memory BFGS, KKT exit at NPSOL's tolerances, default major-iteration limit max(50, 3(n+nclin)) = %d" % max(50, 3 * (spec.nC + spec.nclin))
        tc["npsol_cold_start"] = t0c
        res["to_convergence"] = tc
        # ---- the preconditioned mode at a saturating batch (65536 problems): throughput and the contract roofline of its launch ----
        nbig = 65536
        loG, upG = cf.kincar_random_bounds(ncars, 4096)
        loG = torch.tensor(np.tile(loG, (nbig // 4096, 1)), device=dev); upG = torch.tensor(np.tile(upG, (nbig // 4096, 1)), device=dev)
        xG = torch.ones((nbig, spec.nC), dtype=torch.float64, device=dev)
        wG = torch.empty(plan.workspace_bytes(nbig, oc), dtype=torch.uint8, device=dev)
        for _ in range(2):
            xG.fill_(1.0); ooG = plan.solve(loG, upG, xG, oc, work=wG)
        torch.cuda.synchronize()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot_ms = 0.0
        for _ in range(5):
            xG.fill_(1.0); g0.record(); ooG = plan.solve(loG, upG, xG, oc, work=wG); g1.record(); torch.cuda.synchronize()
            tot_ms += g0.elapsed_time(g1)
        msG = tot_ms / 5
        nfG = int(ooG["nfev"].sum().item())
        res["to_convergence"]["saturating_batch"] = {
            "batch": nbig, "ms_per_batch": msG, "value": nbig / (msG * 1e-3), "unit": "trajectories/s", "nfev_per_problem": nfG / nbig,
            "kernel": plan.solve_kernel(nbig, oc), "converged_frac": float((ooG["inform"] == 0).float().mean().item()),
            "roofline": {"bound": "hbm", "kernel": plan.solve_kernel(nbig, oc), "achieved": nfG * spec.eval_bytes() / (msG * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": nfG * spec.eval_bytes() / (msG * 1e-3) / 1e9 / HBM_PEAK_GBS}}
        del wG, xG, loG, upG
        # ---- BASELINE configs[1]: kincar 2 outputs, order 6, 20 intervals, batch 256 ----
        specB = cf.config_B(); planB = api.Plan(specB, local)
        loB, upB = cf.kincar_random_bounds(1, 256)
        loB = torch.tensor(loB, device=dev); upB = torch.tensor(upB, device=dev)
        resB = {"workload": specB.name + ", batch 256"}
        for key, oB in (("fixed_50_majors", api.default_opts(itlim=50, fixed_iters=1, hessian=0)), ("to_convergence", api.default_opts(hessian=1))):
            wB = torch.empty(planB.workspace_bytes(256, oB), dtype=torch.uint8, device=dev)
            xB = torch.ones((256, specB.nC), dtype=torch.float64, device=dev)
            for _ in range(3):
                xB.fill_(1.0); planB.solve(loB, upB, xB, oB, work=wB)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(20):
                xB.fill_(1.0); ooB = planB.solve(loB, upB, xB, oB, work=wB)
            torch.cuda.synchronize(); dtB = (time.perf_counter() - t1) / 20
            resB[key] = {"value": 256 / dtB, "unit": "trajectories/s", "ms_per_batch": 1e3 * dtB, "iters_mean": float(ooB["iters"].float().mean().item()),
                         "kernel": planB.solve_kernel(256, oB)}
        res["config_B_batch256"] = resB   # one problem per CU: this size measures the latency of a single solve, not throughput
        del planB
        # ---- cost of specialisation: shapes that take the generic kernel instance (any nout / order at run time) next to tuned ones,
        #      same fixed-work mode, normalised per flat output ----
        gen = {}
        for gname, gspec, ncg in (("kincar-4out-k6-l20 (wave kernel)", cf._kincar_spec(2, 6, 3, 20, 101, 5.0, "G4"), 2),
                                  ("kincar-4out-k6-l16, 81 breakpoints (tuned workgroup-per-problem instance; breakpoint-lane evaluation kernel)", cf._kincar_spec(2, 6, 3, 16, 81, 5.0, "G4b"), 2),
                                  ("kincar-2out-k6-l20 (wave kernel)", cf.config_B(), 1),
                                  ("kincar-2out-k5-l2, 20 breakpoints: the shipped example's shape (tuned order-5 instance)", cf.config_K0(), 1),
                                  ("kincar-2out-k4-l10, 41 breakpoints (generic instance)", cf._kincar_spec(1, 4, 2, 10, 41, 5.0, "G2"), 1)):
            planG = api.Plan(gspec, local)
            loG, upG = cf.kincar_random_bounds(ncg, 4096)
            loG = torch.tensor(loG, device=dev); upG = torch.tensor(upG, device=dev)
            xG = torch.ones((4096, gspec.nC), dtype=torch.float64, device=dev)
            oG = api.default_opts(itlim=args.iters, fixed_iters=1, hessian=0)
            wG = torch.empty(planG.workspace_bytes(4096, oG), dtype=torch.uint8, device=dev)
            for _ in range(2):
                xG.fill_(1.0); planG.solve(loG, upG, xG, oG, work=wG)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(5):
                xG.fill_(1.0); ooG = planG.solve(loG, upG, xG, oG, work=wG)
            torch.cuda.synchronize(); dtG = (time.perf_counter() - t1) / 5
            xe4 = torch.randn((1 << 16, gspec.nC), dtype=torch.float64, device=dev)
            og4 = planG.eval(xe4, 2); planG.eval(xe4, 2, out=og4); torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(5):
                planG.eval(xe4, 2, out=og4)
            torch.cuda.synchronize(); dte = (time.perf_counter() - t1) / 5
            gen[gname] = {"solve_ms_per_4096": 1e3 * dtG, "trajectories_per_s": 4096 / dtG, "nfev_per_problem": float(ooG["nfev"].float().mean().item()),
                          "solve_kernel": planG.solve_kernel(4096, oG),
                          "eval_GBps": (1 << 16) * gspec.eval_bytes() / dte / 1e9, "eval_frac_of_hbm_peak": (1 << 16) * gspec.eval_bytes() / dte / 1e9 / HBM_PEAK_GBS}
            del planG, wG, xe4, og4
        res["generic_instances"] = gen
        # ---- BASELINE config C: receding-horizon MPC, 100 re-solves x batch 1024, kincar 2-output ----
        specC = cf.config_B(); planC = api.Plan(specC, local)
        nbC = 1024
        loC, upC = cf.kincar_random_bounds(1, nbC)
        loC0 = torch.tensor(loC, device=dev); upC0 = torch.tensor(upC, device=dev)
        xC0 = torch.ones((nbC, specC.nC), dtype=torch.float64, device=dev)
        wC = torch.empty(planC.workspace_bytes(nbC, oc), dtype=torch.uint8, device=dev)

        def mpc_run(nres):   # nres x (solve, shift) inside the library: first step direct, the rest replayed as a hipGraph
            loC1, upC1, xC = loC0.clone(), upC0.clone(), xC0.clone()
            _, bad = planC.mpc_run(xC, loC1, upC1, nres, 5, 1, oc, work=wC)
            return bad
        mpc_run(5); torch.cuda.synchronize()
        t1 = time.perf_counter(); bad = mpc_run(100); torch.cuda.synchronize()
        dtm = time.perf_counter() - t1
        res["mpc_config_C"] = {"value": nbC * 100 / dtm, "unit": "re-solves/s", "resolves": 100, "batch": nbC,
                               "ms_per_resolve_batch": 1e3 * dtm / 100, "not_converged": int(bad.item()), "kernel": planC.solve_kernel(nbC, oc),
                               "workload": "B:kincar-2out-k6-l20, advance one knot interval per re-solve, shift warm start"}
        del wC
        # ---- nonlinear inequality constraints: kincar + circular obstacle, to convergence ----
        specO = cf.config_O(20); planO = api.Plan(specO, local)
        loO, upO = cf.obstacle_bounds(B)
        loO = torch.tensor(loO, device=dev); upO = torch.tensor(upO, device=dev)
        xO0 = torch.ones((B, specO.nC), dtype=torch.float64, device=dev); xO = xO0.clone()
        oO = api.default_opts(hessian=1)
        wO = torch.empty(planO.workspace_bytes(B, oO), dtype=torch.uint8, device=dev)
        for _ in range(2):
            xO.copy_(xO0); ooO = planO.solve(loO, upO, xO, oO, work=wO)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(3):
            xO.copy_(xO0); ooO = planO.solve(loO, upO, xO, oO, work=wO)
        torch.cuda.synchronize(); dto = (time.perf_counter() - t1) / 3
        infO = ooO["inform"].cpu().numpy()
        # (beside it, the QP-based SQP step, hessian = 3: a fifth of the majors; at this size the quasi-Newton passes are the faster ones per major)
        oOq = api.default_opts(hessian=3)
        wOq = torch.empty(planO.workspace_bytes(B, oOq), dtype=torch.uint8, device=dev)
        for _ in range(2):
            xO.copy_(xO0); ooOq = planO.solve(loO, upO, xO, oOq, work=wOq)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(3):
            xO.copy_(xO0); ooOq = planO.solve(loO, upO, xO, oOq, work=wOq)
        torch.cuda.synchronize(); dtoq = (time.perf_counter() - t1) / 3
        infOq = ooOq["inform"].cpu().numpy()
        obst_qp = {"value": B / dtoq, "unit": "trajectories/s", "ms_per_batch": 1e3 * dtoq, "inform_counts": {str(k): int((infOq == k).sum()) for k in np.unique(infOq)},
                   "iters_mean": float(ooOq["iters"].float().mean().item()), "iters_max": int(ooOq["iters"].max().item()), "nfev_mean": float(ooOq["nfev"].float().mean().item())}
        del wOq
        res["constrained_obstacle"] = {"value": B / dto, "unit": "trajectories/s", "ms_per_batch": 1e3 * dto, "batch": B, "qp_sqp": obst_qp, "nfev_mean": float(ooO["nfev"].float().mean().item()),
                                       "workload": specO.name + ": 101 nonlinear trajectory inequalities per problem, augmented-Lagrangian outer loop",
                                       "inform_counts": {str(k): int((infO == k).sum()) for k in np.unique(infO)},
                                       "iters_mean": float(ooO["iters"].float().mean().item()), "iters_max": int(ooO["iters"].max().item())}
        del wO
        # ---- receding horizon WITH a nonlinear inequality row (obstacle family): multiplier estimates carried over and shifted with the
        #      horizon (ntg_solve_opts.warm_start, ntg_batch_mpc_shift_multipliers) against re-solving every step from lambda = 0 ----
        nbM, nresM = 1024, 20
        loM, upM = cf.obstacle_bounds(nbM)
        mo = {}
        # (round 4: the QP-based SQP step, hessian = 3 -- its warm start IS the carried-over working set -- host loop and, last, the library's own
        #  loop: (solve, count, shift, multiplier shift) captured once in a hipGraph and replayed, ntg_batch_mpc_run; the quasi-Newton
        #  augmented-Lagrangian mode of round 3 stays beside it.  Majors and non-converged counts accumulate on the device: no sync per step.)
        for tag, hM, ws in (("cold_every_step", 3, 0), ("multipliers_carried_over", 3, 1), ("quasi_newton_cold_every_step", 1, 0), ("quasi_newton_multipliers_carried_over", 1, 1)):
            oM = api.default_opts(hessian=hM, warm_start=ws)
            wM = torch.empty(planO.workspace_bytes(nbM, oM), dtype=torch.uint8, device=dev)
            loM1 = torch.tensor(loM, device=dev); upM1 = torch.tensor(upM, device=dev)
            xM = torch.ones((nbM, specO.nC), dtype=torch.float64, device=dev)
            o_first = api.default_opts(hessian=hM)
            planO.solve(loM1, upM1, xM, o_first, work=wM); planO.mpc_shift(xM, loM1, upM1, 5, 1)
            if ws:
                planO.mpc_shift_multipliers(nbM, 5, oM, wM)
            majd = torch.zeros((), dtype=torch.float64, device=dev); notd = torch.zeros((), dtype=torch.int64, device=dev)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(nresM):
                ooM = planO.solve(loM1, upM1, xM, oM, work=wM)
                planO.mpc_shift(xM, loM1, upM1, 5, 1)
                if ws:
                    planO.mpc_shift_multipliers(nbM, 5, oM, wM)
                majd += ooM["iters"].double().mean(); notd += (ooM["inform"] != 0).sum()
            torch.cuda.synchronize(); dtM = time.perf_counter() - t1
            mo[tag] = {"value": nbM * nresM / dtM, "unit": "re-solves/s", "ms_per_resolve_batch": 1e3 * dtM / nresM, "majors_per_resolve": float(majd.item()) / nresM,
                       "not_converged": int(notd.item()), "mode": "QP-based SQP step (hessian = 3)" if hM == 3 else "quasi-Newton augmented Lagrangian (hessian = 1)"}
            del wM
        if True:
            oM = api.default_opts(hessian=3, warm_start=1)
            wM = torch.empty(planO.workspace_bytes(nbM, oM), dtype=torch.uint8, device=dev)
            loM1 = torch.tensor(loM, device=dev); upM1 = torch.tensor(upM, device=dev)
            xM = torch.ones((nbM, specO.nC), dtype=torch.float64, device=dev)
            planO.mpc_run(xM, loM1, upM1, 2, 5, 1, oM, work=wM)   # first (cold) step and graph code paths warm
            torch.cuda.synchronize(); t1 = time.perf_counter()
            _, badM = planO.mpc_run(xM, loM1, upM1, nresM + 1, 5, 1, oM, work=wM)   # one cold step + nresM warm ones inside the library
            torch.cuda.synchronize(); dtM = time.perf_counter() - t1
            mo["multipliers_carried_over_hipgraph"] = {"value": nbM * (nresM + 1) / dtM, "unit": "re-solves/s", "ms_per_resolve_batch": 1e3 * dtM / (nresM + 1),
                                                       "not_converged": int(badM.item()), "mode": "ntg_batch_mpc_run: QP-based SQP step, warm, hipGraph replay (the first of the %d steps is cold)" % (nresM + 1)}
            del wM
        mo["workload"] = specO.name + f": {nresM} re-solves x {nbM}, advance one knot interval per re-solve (host loop: solve, shift, multiplier shift; _hipgraph: the library's captured loop)"
        res["mpc_obstacle"] = mo
        # ---- BASELINE configs D and E at their full sizes, to convergence (per-GPU share of the 8-GPU batch): the structured
        #      Newton mode (hessian = 2) and, beside it, the quasi-Newton mode of round 1.  (`bench.py --config D|E
        #      --gpus N` times the same solve as the headline of its own line, sharded over N GPUs.) ----
        MFMA_PEAK_TF = 78.6   # fp64 matrix peak of one MI355X (MI355X_MICROARCH.md)
        for key, mk, bnds, nbL, qnm in (("config_D", cf.config_D, cf.quadrotor_bounds, 512, 48), ("config_E", cf.config_E, cf.manipulator_bounds, 1024, 0)):
            specL = mk(); planL = api.Plan(specL, local)
            loLn, upLn = bnds(nbL)
            loL = torch.tensor(loLn, device=dev); upL = torch.tensor(upLn, device=dev)
            entry = {"batch": nbL, "workload": "%s: nC %d, %d breakpoints, %d nonlinear trajectory rows per problem" % (specL.name, specL.nC, specL.nbps, specL.ncnln)}
            # qp_sqp: the QP-based SQP step on the band model (hessian = 3) -- what NPSOL does with the Jacobian the reference hands it
            for mode, oL in (("newton", api.default_opts(hessian=2)), ("qp_sqp", api.default_opts(hessian=3)), ("quasi_newton", api.default_opts(hessian=1, qn_memory=qnm))):
                wL = torch.empty(planL.workspace_bytes(nbL, oL), dtype=torch.uint8, device=dev)
                for rep in range(2):   # first pass builds tables / warms the instruction cache
                    xL = torch.ones((nbL, specL.nC), dtype=torch.float64, device=dev)
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    ooL = planL.solve(loL, upL, xL, oL, work=wL)
                    torch.cuda.synchronize(); dtl = time.perf_counter() - t1
                infL = ooL["inform"].cpu().numpy()
                e = {"value": nbL / dtl, "unit": "trajectories/s", "ms_per_batch": 1e3 * dtl,
                     "inform_counts": {str(k): int((infL == k).sum()) for k in np.unique(infL)},
                     "iters_mean": float(ooL["iters"].float().mean().item()), "iters_max": int(ooL["iters"].max().item()),
                     "nfev_mean": float(ooL["nfev"].float().mean().item())}
                if mode == "quasi_newton":
                    e["qn_memory"] = qnm or 256
                else:
                    # matrix-core work of the launch: counters of a separate, untimed run (NTG_AMD_STAMPS=3 makes the kernel report
                    # factorisations / failed attempts in place of the multipliers)
                    os.environ["NTG_AMD_STAMPS"] = "3"
                    xL = torch.ones((nbL, specL.nC), dtype=torch.float64, device=dev)
                    od = planL.solve(loL, upL, xL, oL, work=wL, want_lambda=True)
                    torch.cuda.synchronize()
                    del os.environ["NTG_AMD_STAMPS"]
                    cntL = od["clambda"][:, :10].cpu().numpy()
                    e["mfma"] = newton_mfma_entry(specL, key, cntL[:, :3], nbL, dtl, MFMA_PEAK_TF)
                    if key == "config_E":   # matrix-core counters of the same launch (rocprofv3 --pmc pass, profiles/r04_config_E_modes.json), for these device sources only
                        try:
                            cj = json.load(open(os.path.join(ROOT, "profiles", "r04_config_E_modes.json")))
                            ck = cj.get("qp_E" if mode == "qp_sqp" else "newton_E")
                            if ck and cj.get("csrc_sha") == csrc_sha():
                                e["mfma"]["counters"] = {"SQ_INSTS_MFMA": ck["SQ_INSTS_MFMA"], "MfmaUtil_pct": ck["MfmaUtil_pct"], "mfma_flops": ck["mfma_flops"],
                                                         "achieved_tflops": ck["mfma_flops"] / dtl / 1e12}
                        except Exception:
                            pass
                    e["band_solves_per_problem"] = float(cntL[:, 2].mean())
                    if mode == "qp_sqp":
                        # the dual active-set QP: passive-set solves, columns W J' formed (one band solve each, all coupling groups at once),
                        # problems that left the mode for the augmented-Lagrangian passes (working set of NTG_QP_MAXA slots full, or no
                        # acceptable step on the l1 merit function)
                        e["qp"] = {"passive_set_solves_per_problem": float(cntL[:, 6].mean()), "columns_per_problem": float(cntL[:, 7].mean()),
                                   "fell_back_to_augmented_lagrangian": int((cntL[:, 9] > 0).sum())}
                entry[mode] = e
                del wL
            # `value`: the faster of the two modes among those that end (nearly) every problem at inform 0 -- named in value_mode, so that
            # readers comparing rounds compare like with like
            def ok_frac(e):
                return e["inform_counts"].get("0", 0) / nbL
            cands = [(m, entry[m]) for m in ("newton", "qp_sqp", "quasi_newton") if ok_frac(entry[m]) >= 0.99] or [("newton", entry["newton"])]
            best = max(cands, key=lambda t: t[1]["value"])
            entry["value"] = best[1]["value"]; entry["unit"] = "trajectories/s"; entry["value_mode"] = best[0]
            # the same solve on the config's WHOLE batch on this one GPU: at 512 / 1024 problems (2 / 4 per CU) the launch ends when the
            # slowest problems do (majors range from 3 to 41 / 104), a larger batch shows the per-problem cost
            nbW = 4096
            loWn, upWn = bnds(nbW)
            loW = torch.tensor(loWn, device=dev); upW = torch.tensor(upWn, device=dev)
            oW = api.default_opts(hessian=2)
            wW = torch.empty(planL.workspace_bytes(nbW, oW), dtype=torch.uint8, device=dev)
            xW = torch.ones((nbW, specL.nC), dtype=torch.float64, device=dev)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            ooW = planL.solve(loW, upW, xW, oW, work=wW)
            torch.cuda.synchronize(); dtW = time.perf_counter() - t1
            entry["newton_batch4096"] = {"value": nbW / dtW, "unit": "trajectories/s", "ms_per_batch": 1e3 * dtW,
                                         "inform0_frac": float((ooW["inform"] == 0).float().mean().item()), "iters_mean": float(ooW["iters"].float().mean().item()),
                                         "rank_imbalance_8": shard.imbalance(ooW["iters"].cpu().numpy(), 8)["max_over_mean"]}
            if entry["value_mode"] == "qp_sqp":   # ... and in the mode `value` is quoted in, when that is not the Newton mode
                oQ = api.default_opts(hessian=3)
                wQ = torch.empty(planL.workspace_bytes(nbW, oQ), dtype=torch.uint8, device=dev)
                xW.fill_(1.0)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                ooQ = planL.solve(loW, upW, xW, oQ, work=wQ)
                torch.cuda.synchronize(); dtQ = time.perf_counter() - t1
                entry["qp_sqp_batch4096"] = {"value": nbW / dtQ, "unit": "trajectories/s", "ms_per_batch": 1e3 * dtQ,
                                             "inform0_frac": float((ooQ["inform"] == 0).float().mean().item()), "iters_mean": float(ooQ["iters"].float().mean().item()),
                                             "iters_max": int(ooQ["iters"].max().item()),
                                             "rank_imbalance_8": shard.imbalance(ooQ["iters"].cpu().numpy(), 8)["max_over_mean"]}
                del wQ
            del wW, xW, loW, upW
            res[key] = entry
            del planL
        # ---- funobj + funcon with banded Jacobian rows (the constraint-Jacobian assembly), configs D and E ----
        for key, mk, nbJ in (("jacobian_assembly_D", cf.config_D, 4096), ("jacobian_assembly_E", cf.config_E, 2048)):
            specJ = mk(); planJ = api.Plan(specJ, local)
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
            res[key] = {"kernel": "eval_kernel", "batch": nbJ, "ms": msJ, "achieved": bJ / (msJ * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": bJ / (msJ * 1e-3) / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_eval": specJ.eval_bytes(), "workload": specJ.name}
            del oJ, planJ, xJ
        # ---- standalone evaluation kernel streamed over a large batch ----
        nb = 1 << 18
        xe = torch.randn((nb, spec.nC), dtype=torch.float64, device=dev)
        for _ in range(2):
            plan.eval(xe, 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nrep = 10
        e0.record()
        for _ in range(nrep):
            plan.eval(xe, 2)
        e1.record(); torch.cuda.synchronize()
        ems = e0.elapsed_time(e1) / nrep
        eb = nb * spec.eval_bytes()
        res["eval_kernel"] = {"kernel": "eval_interval_kernel (cost-only single-class instance of the evaluation: one lane per knot interval and output pair)", "batch": nb, "ms": ems, "achieved": eb / (ems * 1e-3) / 1e9,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": eb / (ems * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "evals_per_s": nb / (ems * 1e-3)}
        # the practical ceiling next to the 8 TB/s spec: a plain device-to-device copy with the same read and write volume on this box
        ca = torch.empty(nb * (spec.nC + 1), dtype=torch.float64, device=dev).normal_(); cb_ = torch.empty_like(ca)
        for _ in range(3):
            cb_.copy_(ca)
        torch.cuda.synchronize(); e0.record()
        for _ in range(nrep):
            cb_.copy_(ca)
        e1.record(); torch.cuda.synchronize()
        cms = e0.elapsed_time(e1) / nrep
        cgb = 2 * ca.numel() * 8 / (cms * 1e-3) / 1e9
        res["eval_kernel"].update({"device_copy_same_bytes_ms": cms, "device_copy_same_bytes_GBps": cgb, "frac_of_device_copy": res["eval_kernel"]["achieved"] / cgb})
        del xe, ca, cb_
        # ---- per-problem grids (free final time): every problem on its own horizon; the basis rows and weights are then per-problem
        #      input (counted in the algorithmic bytes) and the interval kernel stages them per problem and wave; setup on the device ----
        nbg = 16384
        k0 = np.asarray(spec.knots[0]); rngg = np.random.default_rng(3)
        scale = rngg.uniform(0.6, 1.6, nbg)[:, None]
        kn = k0[None, :] * scale
        jj = np.minimum(np.searchsorted(k0, spec.bps, side="right") - 1, spec.kninterv[0] - 1)
        fr = (np.asarray(spec.bps) - k0[jj]) / (k0[jj + 1] - k0[jj])
        bpg = kn[:, jj] + fr[None, :] * (kn[:, jj + 1] - kn[:, jj])
        inner = jj < spec.kninterv[0] - 1
        bpg = np.maximum(bpg, kn[:, jj]); bpg[:, inner] = np.minimum(bpg[:, inner], np.nextafter(kn[:, jj + 1][:, inner], -np.inf))
        bpg[:, -1] = kn[:, -1]
        pg = api.Plan(spec, local)
        tg = time.perf_counter()
        pg.set_grids(torch.tensor(np.ascontiguousarray(kn), device=dev), torch.tensor(np.ascontiguousarray(bpg), device=dev), with_precond=False)
        setup_s = time.perf_counter() - tg
        xg = torch.randn((nbg, spec.nC), dtype=torch.float64, device=dev)
        og = pg.eval(xg, 2); pg.eval(xg, 2, out=og); torch.cuda.synchronize()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record()
        for _ in range(10):
            pg.eval(xg, 2, out=og)
        g1.record(); torch.cuda.synchronize()
        gms = g0.elapsed_time(g1) / 10
        row_bytes = 8 * (sum(1 for r in range(spec.maxderiv[0]) if any(a[1] == r for a in list(spec.tcostav) + list(spec.icostav) + list(spec.fcostav) + list(spec.tcav) + list(spec.icav) + list(spec.fcav))) * spec.order[0] * spec.nbps + spec.nbps)
        gb = nbg * (spec.eval_bytes() + row_bytes)
        res["per_problem_grids"] = {"workload": spec.name + ", %d horizons in [0.6, 1.6] x the plan's" % nbg, "kernel": "eval_interval_kernel with wave-private interval tables restaged per problem (round 2: the general eval_kernel)",
                                    "batch": nbg, "ms": gms, "alg_bytes_per_eval": spec.eval_bytes() + row_bytes, "achieved": gb / (gms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": gb / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS, "evals_per_s": nbg / (gms * 1e-3),
                                    "set_grids_host_s": setup_s}
        # ... and the headline solve (50 majors from the identity cold start) on the first 4096 of those grids: the wave kernel's instance
        # with wave-private value tables restaged per problem
        nbs = min(4096, lo.shape[0], nbg)
        pg.clear_grids()
        pg.set_grids(torch.tensor(np.ascontiguousarray(kn[:nbs]), device=dev), torch.tensor(np.ascontiguousarray(bpg[:nbs]), device=dev), with_precond=False)
        og5 = api.default_opts(itlim=args.iters, fixed_iters=1, hessian=0)
        xs_g = torch.ones((nbs, spec.nC), dtype=torch.float64, device=dev)
        lo_g, up_g = lo[:nbs].contiguous(), up[:nbs].contiguous()
        pg.solve(lo_g, up_g, xs_g, og5); torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(3):
            xs_g.fill_(1.0)
            osg = pg.solve(lo_g, up_g, xs_g, og5)
        s1.record(); torch.cuda.synchronize()
        sms = s0.elapsed_time(s1) / 3
        res["per_problem_grids"]["solve_fixed_50_majors"] = {"batch": nbs, "ms_per_batch": sms, "value": nbs / (sms * 1e-3), "unit": "trajectories/s",
                                                             "kernel": pg.solve_kernel(nbs, og5), "iters_mean": float(osg["iters"].float().mean().item())}
        # ... and to convergence with the per-problem preconditioner blocks (hessian = 1)
        pg.clear_grids()
        tg2 = time.perf_counter()
        pg.set_grids(torch.tensor(np.ascontiguousarray(kn[:nbs]), device=dev), torch.tensor(np.ascontiguousarray(bpg[:nbs]), device=dev), with_precond=True)
        setup2_s = time.perf_counter() - tg2
        og1 = api.default_opts(hessian=1)
        xs_g.fill_(1.0); pg.solve(lo_g, up_g, xs_g, og1); torch.cuda.synchronize()
        s0.record()
        for _ in range(3):
            xs_g.fill_(1.0)
            osg1 = pg.solve(lo_g, up_g, xs_g, og1)
        s1.record(); torch.cuda.synchronize()
        sms1 = s0.elapsed_time(s1) / 3
        res["per_problem_grids"]["solve_to_convergence"] = {"batch": nbs, "ms_per_batch": sms1, "value": nbs / (sms1 * 1e-3), "unit": "trajectories/s",
                                                            "kernel": pg.solve_kernel(nbs, og1), "iters_mean": float(osg1["iters"].float().mean().item()),
                                                            "converged_frac": float((osg1["inform"] == 0).float().mean().item()),
                                                            "set_grids_with_preconditioner_s": setup2_s}
        del og, xg, pg, xs_g

    if rank == 0 and world == 1 and not args.no_cpu:
        # ---- CPU baseline: the oracle on a bounded sample of the same workload, on the host cores of this box.  Two flavours
        #      (SURVEY 8d): "ref" = the reference's loops and dense temporaries (cost.c:117-134), "opt" = banded, allocation-free
        #      evaluation.  One pinned thread per PHYSICAL core (OMP_PLACES=cores, OMP_PROC_BIND=spread), >= 32 problems per thread;
        #      the per-call dense temporaries (calloc of 0.3 - 100 MB in the reference's loops) are served from a reused per-thread
        #      buffer (orc_set_scratch_reuse): the loops and zero-fills are the reference's, the C library's mmap / page-fault
        #      traffic under 128 threads (one address-space lock per process) is not measured ----
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc
        orc.set_scratch_reuse(True)
        nthreads_hw = os.cpu_count() or 1
        nphys = min(physical_cores(), nthreads_hw)
        quota = cpu_quota()
        ncore = max(1, min(nphys, int(quota))) if quota >= 1 else nphys   # threads = the cores this job can really run on
        ns = args.cpu_sample if args.cpu_sample > 0 else 96 * ncore   # ~20 s of CPU work at config M (12 ms per problem and thread)
        ns = min(ns, lo_all.shape[0])
        if large:
            ns = min(ns, max(ncore // 8, 1) * 2)   # a D / E problem takes seconds to minutes of CPU time: a handful only
        model = "unknown"
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip(); break
        except OSError:
            pass
        cflags = "gcc -O2 -fopenmp -ffp-contract=off (oracle/Makefile)"
        flav = {}
        r_ref = None
        for fl, banded in (("ref", 0), ("opt", 1)):
            oo = orc.default_opts(hessian=2, banded=banded) if large else orc.default_opts(itlim=args.iters, fixed_iters=1, hessian=0, banded=banded)
            t1 = time.perf_counter()
            r = orc.solve_batch(spec, lo_all[:ns], up_all[:ns], np.ones((ns, spec.nC)), oo, nthreads=ncore)
            dta = time.perf_counter() - t1
            n1 = 1 if large else 96   # >= 1 s of single-thread work
            t1 = time.perf_counter()
            orc.solve_batch(spec, lo_all[:n1], up_all[:n1], np.ones((n1, spec.nC)), oo, nthreads=1)
            dt1 = time.perf_counter() - t1
            flav[fl] = {"all_cores": ns / dta, "one_thread": n1 / dt1, "all_cores_problems": ns, "all_cores_wall_s": dta, "one_thread_problems": n1,
                        "one_thread_wall_s": dt1, "scaling_1_to_all_cores": (ns / dta) / (n1 / dt1)}
            if not large and ncore >= 4:   # more points of the scaling curve (same problems per thread): half the usable cores, and (when a
                # cgroup quota caps the job) four times as many threads as cores, to show the cap
                for nth in sorted({max(ncore // 2, 1), min(4 * ncore, nphys)} - {ncore}):
                    nsub = min(lo_all.shape[0], 32 * nth)
                    t1 = time.perf_counter()
                    orc.solve_batch(spec, lo_all[:nsub], up_all[:nsub], np.ones((nsub, spec.nC)), oo, nthreads=nth)
                    flav[fl][f"threads_{nth}"] = nsub / (time.perf_counter() - t1)
            if fl == "ref":
                r_ref = r
        res["cpu_baseline"] = {"value": flav["ref"]["all_cores"], "unit": "trajectories/s", "cores": ncore, "threads": ncore, "hardware_threads": nthreads_hw,
                               "physical_cores_of_host": nphys, "cgroup_cpu_quota": quota or None,
                               "kind": "port",
                               "sample": f"first {ns} problems of the same batch ({ns / ncore:.1f} per pinned thread, one thread per core the job may use), same solve mode, oracle/sqp.c with the "
                                         f"reference-faithful dense assembly, OpenMP, {flav['ref']['all_cores_wall_s']:.2f} s wall",
                               "flavours": flav, "cpu_model": model, "compiler": cflags,
                               "malloc": "per-call dense temporaries and the dense quasi-Newton matrix from per-thread buffers that are reused from problem to problem and first "
                                         "touched by their owner (orc_set_scratch_reuse): no mmap / munmap / page faults per call"}
        try:
            cpus = orc.thread_cpus(ncore)
            res["cpu_baseline"]["affinity"] = {"distinct_cpus": len(set(cpus)), "min_cpu": min(cpus), "max_cpu": max(cpus), "first_8_threads_on": cpus[:8],
                                               "OMP_PLACES": os.environ.get("OMP_PLACES"), "OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND")}
        except Exception as e:   # noqa: BLE001
            res["cpu_baseline"]["affinity"] = str(e)
        # same inputs -> same answers (oracle is the checker here, never the thing shipped)
        nchk = min(ns, B)
        gobj = out["objective"][:nchk].cpu().numpy()
        res["cpu_baseline"]["max_rel_objective_diff_vs_gpu"] = float(np.max(np.abs(gobj - r_ref["objective"][:nchk]) / np.abs(r_ref["objective"][:nchk])))

    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
