"""Summarise gpurun_out/r2prof (made by tools/prof_r2.sh on the GPU box) into profiles/r02_*.  python profiles/summarize_r2.py"""
import csv, glob, json, os, re, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r2prof"); DST = os.path.join(ROOT, "profiles")

def short(name):
    name = re.sub(r"\(.*", "", name)
    return name[:150]

# ---- kernel statistics of the bench command ----
rows = list(csv.DictReader(open(glob.glob(os.path.join(SRC, "stats", "*", "*_kernel_stats.csv"))[0])))
keep = [r for r in rows if any(k in r["Name"] for k in ("sqp_kernel", "eval_", "basis_kernel", "mpc_", "interp", "linrows", "bounds_kernel", "count_notconv"))]
with open(os.path.join(DST, "r02_bench_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for r in keep:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]])
# per-dispatch durations of the headline kernel (fixed-50 launches are the long cluster)
tr = list(csv.DictReader(open(glob.glob(os.path.join(SRC, "stats", "*", "*_kernel_trace.csv"))[0])))
dur = collections.defaultdict(list)
for r in tr:
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
bench = json.loads(open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(DST, "r02_bench.json"), "w"), indent=1)
summ = {"how": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu (tools/prof_r2.sh); durations in ms"}
label = {"sqp_kernel<0, 6, 6, 128, 4, false, false, 4, false>": "headline sqp_kernel: 4096 x config M, 50 fixed majors (identity cold start)",
         "sqp_kernel<0, 6, 6, 128, 4, false, true, 4, false>": "sqp_kernel config M to convergence (4096 and 65536 problems mixed)",
         "sqp_kernel<4, 4, 8, 256, 4, false, true, 0, true>": "sqp_kernel config D, 512 problems, structured Newton mode",
         "sqp_kernel<4, 4, 8, 256, 4, false, true, 0, false>": "sqp_kernel config D, 512 problems, quasi-Newton mode (48 pairs)",
         "sqp_kernel<5, 12, 6, 512, 5, true, true, 5, true>": "sqp_kernel config E, 1024 problems, structured Newton mode",
         "sqp_kernel<5, 12, 6, 512, 5, true, true, 5, false>": "sqp_kernel config E, 1024 problems, quasi-Newton mode",
         "eval_interval_kernel<0, 6, 2, 6, 4, 256, 4, 20, true>": "eval_interval_kernel: 2^18 evaluations of config M",
         "eval_kernel<4, 4, 8, 256, 4, 0>": "eval_kernel config D with banded Jacobian rows, 4096 evaluations (two workgroups of 256 per CU)",
         "eval_kernel<5, 12, 6, 512, 5, 0>": "eval_kernel config E with banded Jacobian rows, 2048 evaluations"}
for k, d in dur.items():
    kk = k.replace("void ", "").strip()
    if kk in label:
        d = sorted(d); summ[label[kk]] = {"kernel": kk, "calls": len(d), "avg_ms": sum(d) / len(d), "min_ms": d[0], "max_ms": d[-1]}
summ["bench_roofline_kernel_ms (HIP events, same run, headline kernel)"] = bench["roofline"]["kernel_ms"]

# ---- counters ----
def counters(tag, kernel_sub):
    out = collections.defaultdict(list)
    for fcsv in glob.glob(os.path.join(SRC, tag, "*", "*_counter_collection.csv")):
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(fcsv)):
            if kernel_sub in r["Kernel_Name"]:
                per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for d, c in per.items():
            for n, v in c.items():
                out[n].append(v)
    return {n: sum(v) / len(v) for n, v in out.items()}, {n: len(v) for n, v in out.items()}

traffic = {"how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/prof_r2.py (3 launches each, averaged per launch); "
                  "KiB units; FETCH_SIZE x 2 on gfx950 (128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as read"}
for what, sub, key in (("sqp", "sqp_kernel", "sqp_kernel:M:4096:fixed50"), ("eval", "eval_interval_kernel", "eval_interval_kernel:M:262144"), ("newtonE", "sqp_kernel", "sqp_kernel:E:1024:newton")):
    fe, _ = counters(f"pmc_{what}_fetch", sub); wr, _ = counters(f"pmc_{what}_write", sub)
    if "FETCH_SIZE" in fe and "WRITE_SIZE" in wr:
        fb, wb = fe["FETCH_SIZE"] * 1024 * 2, wr["WRITE_SIZE"] * 1024
        traffic[key] = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb, "raw_KiB": {"FETCH_SIZE": fe["FETCH_SIZE"], "WRITE_SIZE": wr["WRITE_SIZE"]}}
cfgM_bytes = 6056
if "eval_interval_kernel:M:262144" in traffic:
    t = traffic["eval_interval_kernel:M:262144"]; t["algorithmic_bytes"] = 262144 * cfgM_bytes; t["traffic_over_algorithmic"] = t["hbm_bytes"] / t["algorithmic_bytes"]
if "sqp_kernel:M:4096:fixed50" in traffic:
    t = traffic["sqp_kernel:M:4096:fixed50"]; t["algorithmic_bytes"] = bench["roofline"]["alg_bytes_per_launch"]; t["traffic_over_algorithmic"] = t["hbm_bytes"] / t["algorithmic_bytes"]
json.dump(traffic, open(os.path.join(DST, "traffic.json"), "w"), indent=1)

lds, n = counters("pmc_eval_lds", "eval_interval_kernel")
if lds:
    lds_s = {"how": "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY over "
                    "tools/prof_r2.py eval (eval_interval_kernel, 2^18 evaluations of config M per launch, averaged per launch)", "raw": lds}
    if lds.get("SQ_ACTIVE_INST_LDS"):
        lds_s["bank_conflict_share_of_lds_active"] = lds["SQ_LDS_BANK_CONFLICT"] / lds["SQ_ACTIVE_INST_LDS"]
    if lds.get("SQ_WAVE_CYCLES"):
        lds_s["wait_inst_lds_share_of_wave_cycles"] = lds.get("SQ_WAIT_INST_LDS", 0) / lds["SQ_WAVE_CYCLES"]
        lds_s["valu_active_share_of_wave_cycles"] = lds.get("SQ_ACTIVE_INST_VALU", 0) / lds["SQ_WAVE_CYCLES"]
        lds_s["wait_any_share_of_wave_cycles"] = lds.get("SQ_WAIT_ANY", 0) / lds["SQ_WAVE_CYCLES"]
    lds_s["lds_instructions_per_evaluation"] = lds.get("SQ_INSTS_LDS", 0) / 262144
    lds_s["valu_instructions_per_evaluation"] = lds.get("SQ_INSTS_VALU", 0) / 262144
    json.dump(lds_s, open(os.path.join(DST, "r02_eval_lds.json"), "w"), indent=1)

mf = {"how": "rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over tools/prof_r2.py newtonD / newtonE "
             "(structured Newton solves at the bench batch, averaged per launch); MfmaFlopsF64 = MOPS_F64 x 512; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs)"}
for what, key in (("newtonD", "D:512:newton"), ("newtonE", "E:1024:newton")):
    c, _ = counters(f"pmc_{what}_mfma", "sqp_kernel")
    if c:
        e = {"mfma_instructions": c.get("SQ_INSTS_MFMA"), "mfma_flops_f64": c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) * 512, "raw": c}
        if c.get("GRBM_GUI_ACTIVE"):
            e["mfma_util_percent"] = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
        mf[key] = e
json.dump(mf, open(os.path.join(DST, "r02_mfma.json"), "w"), indent=1)
json.dump(summ, open(os.path.join(DST, "r02_summary.json"), "w"), indent=1)
print(json.dumps(summ, indent=1)[:3000]); print(json.dumps(traffic, indent=1)[:2500])
print(json.dumps(json.load(open(os.path.join(DST, "r02_eval_lds.json"))), indent=1)[:1500] if os.path.exists(os.path.join(DST, "r02_eval_lds.json")) else "no lds")
print(json.dumps(mf, indent=1)[:2000])
