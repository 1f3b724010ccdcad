"""Build libntg_amd.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the repo)."""
from __future__ import annotations
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libntg_amd.so")
SOURCES = ["kernels.hip", "grids.hip", "fam_kincar.hip", "fam_kincar_chm.hip", "fam_kincar_wave.hip", "fam_vanderpol.hip", "fam_testfam.hip", "fam_obstacle.hip", "fam_quadrotor.hip",
           "fam_manip.hip", "plan.cpp", "ntg_host.cpp"]
HEADERS = ["ntg_dev.hpp", "solve_impl.hpp", "newton.hpp", "eval_fast.hpp", "solve_wave.hpp", "families.hpp", "linesearch.hpp", "plan.hpp", "../../include/ntg_amd.h", "../../include/ntg.h"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS if os.path.exists(os.path.join(CSRC, f)))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, f)) for f in HEADERS if os.path.exists(os.path.join(CSRC, f)))
    for src in SOURCES:   # one hipcc per translation unit, all at once (the family units are independent)
        path = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_t):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", path, "-o", obj,
               "-I", os.path.join(HERE, "..", "include"), "-Wno-unused-result", "-Wno-unused-value", "-Wno-pass-failed"]
        if src.endswith(".hip"):
            cmd += ["-save-temps=obj", "-Wno-unused-command-line-argument"]   # keeps the device assembly next to the object: audited below
        if verbose:
            cmd += ["-Rpass-analysis=kernel-resource-usage"]
        cmd += os.environ.get("NTG_AMD_CXXFLAGS", "").split()   # e.g. -DNTG_HIST_G=8 for tuning experiments
        log = open(obj + ".log", "w")
        procs.append((src, obj, subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT), log))
    failed = []
    for src, obj, pr, log in procs:
        rc = pr.wait()
        log.close()
        if rc != 0:
            failed.append(src)
            sys.stderr.write(open(obj + ".log").read())
    if failed:
        raise RuntimeError("hipcc failed for " + ", ".join(failed))
    # call boundaries of the device code (no generic pointers into the private segment, no FLAT in out-of-line functions, no dynamic
    # stack: ntg_amd/call_audit.py says why); the other intermediate files of -save-temps are removed
    from . import call_audit
    bad = []
    for src in SOURCES:
        stem = os.path.splitext(src)[0]
        asm = os.path.join(CSRC, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        if os.path.exists(asm):
            bad += call_audit.audit(asm)
        for f in os.listdir(CSRC):
            if f.startswith(stem + "-hip-") or f.startswith(stem + "-host-") or f.startswith(stem + ".hip-"):
                if not f.endswith("gfx950.s"):
                    os.remove(os.path.join(CSRC, f))
    if bad:
        raise RuntimeError("call-boundary audit failed:\n  " + "\n  ".join(bad[:20]))
    # the wave kernels address accumulator registers by hand: the compiler's own code must stay below their base
    from . import isa_audit
    wave_asm = os.path.join(CSRC, "fam_kincar_wave-hip-amdgcn-amd-amdhsa-gfx950.s")
    bad = isa_audit.audit(hipcc, os.path.join(CSRC, "fam_kincar_wave.hip"), os.path.join(HERE, "..", "include"),
                          os.environ.get("NTG_AMD_CXXFLAGS", "").split(), isa_audit.agpr_base(os.path.join(CSRC, "solve_wave.hpp")),
                          asm_path=wave_asm if os.path.exists(wave_asm) else None)
    if bad:
        raise RuntimeError("ISA audit of fam_kincar_wave.hip failed (compiler-generated code in the hand-managed AGPR range, or spills):\n  " + "\n  ".join(bad[:20]))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
