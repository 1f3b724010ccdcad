"""Diagnostic (not a test): build tuning variants of one translation unit and time the headline solve with each.
  here:        python tools/variants.py build fam_kincar_chm.hip tag1="-DX=1 -DY=2" tag2="..."
  on the box:  python tools/variants.py run [M|D|E] [batch]      (one child process per variant library)"""
import os, subprocess, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "build", "variants")
CSRC = os.path.join(ROOT, "ntg_amd", "csrc")

CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from ntg_amd import api, configs as cf
cfg, B = sys.argv[1], int(sys.argv[2])
if cfg.startswith("eval"):
    spec = {"evalD": cf.config_D, "evalE": cf.config_E, "evalM": cf.config_M}[cfg]()
    p = api.Plan(spec, 0)
    x = torch.randn((B, spec.nC), dtype=torch.float64, device="cuda:0")
    o = p.eval(x, 2)
    for _ in range(3): p.eval(x, 2, out=o)
    torch.cuda.synchronize()
    tms = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); p.eval(x, 2, out=o); e1.record(); torch.cuda.synchronize(); tms.append(e0.elapsed_time(e1))
    ms = float(np.median(tms)); byts = B * spec.eval_bytes()
    print(os.environ.get("NTG_AMD_LIB", "default").split("/")[-1], cfg, "B", B, "med %%.3f ms  %%.0f GB/s  frac %%.3f" %% (ms, byts / ms / 1e6, byts / ms / 1e6 / 8000), flush=True)
    raise SystemExit(0)
dev = torch.device("cuda:0")
if cfg == "M":
    spec = cf.config_M(); lo, up = cf.kincar_random_bounds(3, B); modes = [dict(itlim=50, fixed_iters=1, hessian=0), dict(hessian=1)]
elif cfg == "D":
    spec = cf.config_D(); lo, up = cf.quadrotor_bounds(B); modes = [dict(hessian=2), dict(hessian=1)]
else:
    spec = cf.config_E(); lo, up = cf.manipulator_bounds(B); modes = [dict(hessian=2), dict(hessian=1)]
plan = api.Plan(spec, 0)
lo_d, up_d = torch.tensor(lo, device=dev), torch.tensor(up, device=dev)
x = torch.ones((B, spec.nC), dtype=torch.float64, device=dev)
res = []
for m in modes:
    o = api.default_opts(**m)
    ws = plan.workspace(B, o) if hasattr(plan, "workspace") else None
    tms = []
    for rep in range(12):
        x.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = plan.solve(lo_d, up_d, x, o); e1.record(); torch.cuda.synchronize()
        tms.append(e0.elapsed_time(e1))
    ms = float(np.median(tms[2:]))
    res.append("%%s: med %%.3f min %%.3f ms inform0" %% (m, ms, min(tms)) + " %%.3f it %%.1f obj %%.10g" %% (float((out["inform"] == 0).float().mean()), float(out["iters"].float().mean()), float(out["objective"].sum())))
    continue
    res.append("%%s: %%.3f ms inform0 %%.3f it %%.1f obj %%.10g" %% (m, ms, float((out["inform"] == 0).float().mean()), float(out["iters"].float().mean()), float(out["objective"].sum())))
print(os.environ.get("NTG_AMD_LIB", "default").split("/")[-1], " | ".join(res), flush=True)
''' % ROOT

if sys.argv[1] == "build":
    src = sys.argv[2]
    os.makedirs(VDIR, exist_ok=True)
    sys.path.insert(0, ROOT)
    from ntg_amd import build as B
    B.build()
    others = [os.path.join(CSRC, os.path.splitext(s)[0] + ".o") for s in B.SOURCES if s != src]
    procs = []
    for spec in sys.argv[3:]:
        tag, flags = spec.split("=", 1)
        obj = os.path.join(VDIR, tag + ".o")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj,
               "-I", os.path.join(ROOT, "include"), "-Wno-unused-result", "-Wno-unused-value", "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage"] + flags.split()
        procs.append((tag, obj, subprocess.Popen(cmd, stderr=open(obj + ".log", "w"))))
    for tag, obj, pr in procs:
        if pr.wait():
            print(open(obj + ".log").read()[-3000:]); raise SystemExit("compile failed: " + tag)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VDIR, "lib_" + tag + ".so"), obj] + others)
        print("built", tag)
else:
    cfg = sys.argv[2] if len(sys.argv) > 2 else "M"
    B = sys.argv[3] if len(sys.argv) > 3 else "4096"
    for lib in [None] + sorted(glob.glob(os.path.join(VDIR, "lib_*.so"))):
        env = dict(os.environ)
        if lib:
            env["NTG_AMD_LIB"] = lib
        subprocess.run([sys.executable, "-c", CHILD, cfg, B], env=env, timeout=300)
