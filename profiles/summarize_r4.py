"""Summarise gpurun_out/r4prof (made by tools/prof_r4.sh on the GPU box) into profiles/r04_*.  python profiles/summarize_r3.py
Remove gpurun_out/r4prof before the gpurun call: gpurun MERGES the box's output into the local directory, and run directories of an earlier
call (other process ids) would be summarised instead of, or averaged with, the new ones."""
import csv, glob, json, os, re, collections, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "gpurun_out", "r4prof"); DST = os.path.join(ROOT, "profiles")
import bench as bench_mod
SHA = bench_mod.csrc_sha()


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").strip()[:170]


# ---- kernel statistics of the bench command ----
rows = list(csv.DictReader(open(glob.glob(os.path.join(SRC, "stats", "*", "*_kernel_stats.csv"))[0])))
keep = [r for r in rows if any(k in r["Name"] for k in ("sqp_", "eval_", "basis_kernel", "mpc_", "interp", "linrows", "bounds_kernel", "count_notconv", "grid_"))]
with open(os.path.join(DST, "r04_bench_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for r in keep:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]])
tr = list(csv.DictReader(open(glob.glob(os.path.join(SRC, "stats", "*", "*_kernel_trace.csv"))[0])))
dur = collections.defaultdict(list)
for r in tr:
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
bench = json.loads(open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(DST, "r04_bench.json"), "w"), indent=1)
summ = {"how": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu (tools/prof_r4.sh); durations in ms; device sources " + SHA}
for k, d in sorted(dur.items()):
    if any(s in k for s in ("sqp_", "eval_", "grid_")):
        d = sorted(d)
        summ[k] = {"calls": len(d), "avg_ms": sum(d) / len(d), "min_ms": d[0], "max_ms": d[-1]}
# the headline launches: FAT instance of the wave kernel for 6 outputs (MINW = 1, NLDS 10): 1 warmup + 5 timed + extras' launches of the same instance
head = [k for k in dur if re.search(r"sqp_wave_kernel<0, 6, 2, 6, 4, 20, 4, 1, \d+, 10,", k)]
if head:
    d = sorted(sum((dur[k] for k in head), []))
    fixed = [v for v in d if abs(v - bench["roofline"]["kernel_ms"]) < 0.35 * bench["roofline"]["kernel_ms"]]
    summ["headline: sqp_wave_kernel FAT instance, launches within 35 % of bench's own HIP-event figure"] = {
        "calls": len(fixed), "avg_ms": sum(fixed) / max(len(fixed), 1), "min_ms": min(fixed) if fixed else None, "max_ms": max(fixed) if fixed else None,
        "bench_roofline_kernel_ms (HIP events, same run)": bench["roofline"]["kernel_ms"]}


# ---- counters ----
def counters(tag, kernel_sub):
    out = collections.defaultdict(list)
    for fcsv in glob.glob(os.path.join(SRC, tag, "*", "*_counter_collection.csv")):
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(fcsv)):
            if kernel_sub in r["Kernel_Name"]:
                per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for _, c in per.items():
            for n, v in c.items():
                out[n].append(v)
    return {n: sum(v) / len(v) for n, v in out.items()}


tpath = os.path.join(DST, "traffic.json")
traffic = json.load(open(tpath))
traffic["how_r4"] = ("round 4: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/prof_r3.py (3 launches each, averaged per launch); "
                     "KiB units; FETCH_SIZE x 2 on gfx950 (128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as read; every entry "
                     "carries the hash of the device sources it was measured on (bench.py attaches it only on a match)")
cfgM_bytes = 6056
for what, key, evals in (("fixed50", "sqp_wave_kernel:M:4096:fixed50", 4096 * 101), ("conv_h1_big", "sqp_wave_kernel:M:65536:conv_h1", None)):
    fe = counters(f"pmc_{what}_fetch", "sqp_wave_kernel"); wr = counters(f"pmc_{what}_write", "sqp_wave_kernel")
    if "FETCH_SIZE" in fe and "WRITE_SIZE" in wr:
        fb, wb = fe["FETCH_SIZE"] * 1024 * 2, wr["WRITE_SIZE"] * 1024
        e = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb, "raw_KiB": {"FETCH_SIZE": fe["FETCH_SIZE"], "WRITE_SIZE": wr["WRITE_SIZE"]}, "csrc_sha": SHA}
        if evals:
            e["algorithmic_bytes"] = evals * cfgM_bytes; e["traffic_over_algorithmic"] = (fb + wb) / (evals * cfgM_bytes)
        traffic[key] = e
json.dump(traffic, open(tpath, "w"), indent=1)
sq = counters("pmc_fixed50_sq", "sqp_wave_kernel")
sq2 = counters("pmc_fixed50_sq2", "sqp_wave_kernel")
if sq:
    sq.update({k: v for k, v in sq2.items() if k not in sq})
    sq["csrc_sha"] = SHA
    sq["per_problem"] = {"valu_wave_insts": sq.get("SQ_INSTS_VALU", 0) / 4096, "lds_wave_insts": sq.get("SQ_INSTS_LDS", 0) / 4096,
                         "salu_wave_insts": sq.get("SQ_INSTS_SALU", 0) / 4096, "wave_cycles": 4 * sq.get("SQ_WAVE_CYCLES", 0) / 4096}
    # the binding roof of this kernel: fp64 VALU issue.  One wave-instruction occupies the SIMD's vector pipe for 4 cycles (fp64 FMA: 16 lanes
    # per clock; 32-bit operations as issued by a single wave); SQ_WAVE_CYCLES counts quad-cycles of resident waves (one wave per SIMD here)
    if sq.get("SQ_WAVE_CYCLES"):
        sq["issue_roof"] = {"valu_wave_insts": sq["SQ_INSTS_VALU"], "pipe_cycles_at_4_per_inst": 4 * sq["SQ_INSTS_VALU"], "wave_cycles": 4 * sq["SQ_WAVE_CYCLES"],
                            "frac": sq["SQ_INSTS_VALU"] / sq["SQ_WAVE_CYCLES"], "valu_active_frac": sq.get("SQ_ACTIVE_INST_VALU", 0) / sq["SQ_WAVE_CYCLES"]}
    sq["note"] = ("headline launch (4096 x config M x 50 majors, FAT instance, one wave per SIMD): SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles per wave; "
                  "per evaluation = / (4096 * 101)")
    json.dump(sq, open(os.path.join(DST, "r04_sq_fixed50.json"), "w"), indent=1)
mf = counters("pmc_conv_h1_big_mfma", "sqp_wave_kernel")
if mf:
    mf["note"] = "to-convergence launch (65536 x config M, collocation preconditioner): W0 v on v_mfma_f64_16x16x4_f64, 64 instructions per product, ~4 products per problem"
    json.dump(mf, open(os.path.join(DST, "r04_mfma_conv.json"), "w"), indent=1)
# config E, 1024 problems: matrix-core counters of the QP-based SQP step (hessian = 3) and of the structured Newton mode (2), HBM traffic of the former
qe = {}
for what in ("qp_E", "newton_E"):
    c = counters(f"pmc_{what}_mfma", "sqp_kernel")
    if c:
        # SQ_VALU_MFMA_BUSY_CYCLES: matrix-pipe cycles summed over the chip's 1024 SIMDs (64 per v_mfma_f64_16x16x4_f64: checked against SQ_INSTS_MFMA);
        # GRBM_GUI_ACTIVE: active cycles summed over the 8 XCDs.  MfmaUtil = busy / (elapsed x SIMDs), the gfx94x formula of rocprof's derived metrics.
        el = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        c["elapsed_cycles"] = el
        c["MfmaUtil_pct"] = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(el * 1024.0, 1.0)
        c["mfma_flops"] = 2048.0 * c.get("SQ_INSTS_MFMA", 0.0)
        qe[what] = c
fe = counters("pmc_qp_E_fetch", "sqp_kernel"); wr = counters("pmc_qp_E_write", "sqp_kernel")
if "FETCH_SIZE" in fe and "WRITE_SIZE" in wr:
    qe["qp_E_traffic"] = {"fetch_bytes": fe["FETCH_SIZE"] * 2048, "write_bytes": wr["WRITE_SIZE"] * 1024, "hbm_bytes": fe["FETCH_SIZE"] * 2048 + wr["WRITE_SIZE"] * 1024,
                          "note": "per launch of 1024 problems; the evaluations' algorithmic bytes: 19 evaluations x 738 280 B x 1024 = 14.4 GB"}
if qe:
    qe["csrc_sha"] = SHA
    qe["note"] = "config E (12-output manipulator), 1024 problems per launch, averaged over 2 launches; sqp_kernel<manipulator, 12, 6, 512, 5, BIG, ..., NWT[, QPM]>"
    json.dump(qe, open(os.path.join(DST, "r04_config_E_modes.json"), "w"), indent=1)
json.dump(summ, open(os.path.join(DST, "r04_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summ.items() if "wave" in k or "headline" in k}, indent=1)[:3000])
print({k: traffic[k] for k in traffic if "wave" in k})
print(sq); print(mf)
