#!/usr/bin/env python3
"""Resource usage of every wave-kernel instance (hipcc -Rpass-analysis=kernel-resource-usage on fam_kincar_wave.hip): tools/wave_res.py [-DFLAG ...]"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-c", "--cuda-device-only", os.path.join(root, "ntg_amd/csrc/fam_kincar_wave.hip"), "-o", "/dev/null",
       "-I", os.path.join(root, "include"), "-Wno-unused-result", "-Wno-unused-value", "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[[^\]]*\])?: (.*?)\s*\[-Rpass", line)
    if not m:
        if "error" in line: print(line)
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        mm = re.search(r"sqp_wave_kernelI(.*?)EEv", v)
        cur = mm.group(1).replace("ELi", ",").replace("ELb", ",b").replace("Li", "") if mm else v[:40]
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
print(f"{'FAM,NOUT,OPL,K,CHM,NINT,NWV,MINW,NREG,NLDS,HESS,XLDS,PPG':52s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scr':>5s} {'occ':>4s} {'sSpill':>7s} {'vSpill':>7s} {'LDS':>6s}")
for k, r in rows.items():
    print(f"{k:52s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('SGPRs','?'):>5s} {r.get('ScratchSize','?'):>5s} {r.get('Occupancy','?'):>4s} {r.get('SGPRs Spill','?'):>7s} {r.get('VGPRs Spill','?'):>7s} {r.get('LDS Size','?'):>6s}")
