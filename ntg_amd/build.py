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
HEADERS = ["ntg_dev.hpp", "solve_impl.hpp", "newton.hpp", "qpdual.hpp", "eval_fast.hpp", "solve_wave.hpp", "families.hpp", "linesearch.hpp", "plan.hpp", "../../include/ntg_amd.h", "../../include/ntg.h"]


BASES = os.path.join(CSRC, "fam_kincar_wave.abase")   # accumulator bases (main, alt) the wave-kernel object on disk was compiled with


def _flag_value(flags, name, default):
    for f in flags:
        if f.startswith("-D" + name + "="):
            return int(f.split("=", 1)[1])
    return default


def _read_bases():
    try:
        a, b = open(BASES).read().split()
        return {"main": int(a), "alt": int(b)}
    except Exception:
        return None


def _compile_wave(hipcc, src, obj, base, extra):
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", src, "-o", obj, "-I", os.path.join(HERE, "..", "include"),
           "-Wno-unused-result", "-Wno-unused-value", "-Wno-pass-failed", "-save-temps=obj", "-Wno-unused-command-line-argument",
           "-DNTGW_ABASE=%d" % base["main"], "-DNTGW_ABASE_ALT=%d" % base["alt"]] + [f for f in extra if not f.startswith("-DNTGW_ABASE")]
    with open(obj + ".log", "w") as log:
        if subprocess.call(cmd, stdout=log, stderr=subprocess.STDOUT) != 0:
            sys.stderr.write(open(obj + ".log").read())
            raise RuntimeError("hipcc failed for fam_kincar_wave.hip")
    stem = "fam_kincar_wave"
    for f in os.listdir(CSRC):
        if (f.startswith(stem + "-hip-") or f.startswith(stem + "-host-") or f.startswith(stem + ".hip-")) and not f.endswith("gfx950.s"):
            os.remove(os.path.join(CSRC, f))


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
    extra = os.environ.get("NTG_AMD_CXXFLAGS", "").split()
    wave_base = _read_bases() or {"main": _flag_value(extra, "NTGW_ABASE", 16), "alt": _flag_value(extra, "NTGW_ABASE_ALT", 16)}
    if os.environ.get("NTG_AMD_WAVE_ABASE"):      # e.g. NTG_AMD_WAVE_ABASE=256: build the fallback on purpose; =16: try the full register tier again
        wave_base = {"main": int(os.environ["NTG_AMD_WAVE_ABASE"]), "alt": int(os.environ["NTG_AMD_WAVE_ABASE"])}
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
        if src == "fam_kincar_wave.hip":   # accumulator bases: where the last build ended (NTG_AMD_WAVE_ABASE=16 tries the full register tier again)
            cmd = [c for c in cmd if not c.startswith("-DNTGW_ABASE")] + ["-DNTGW_ABASE=%d" % wave_base["main"], "-DNTGW_ABASE_ALT=%d" % wave_base["alt"]]
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
    # The wave kernels address accumulator registers by hand: the compiler's own code must stay below their base.  If it does not
    # (register pressure moves with every edit and every toolchain), the base of the failing class of instances is RAISED and the unit
    # compiled again -- fewer chain slots in registers, in the limit none (base 256 = the NREG = 0 instances, nothing hand-managed left):
    # the build degrades, it does not break.  The bases in force are recorded next to the library.
    from . import isa_audit
    wave_src, wave_obj = os.path.join(CSRC, "fam_kincar_wave.hip"), os.path.join(CSRC, "fam_kincar_wave.o")
    wave_asm = os.path.join(CSRC, "fam_kincar_wave-hip-amdgcn-amd-amdhsa-gfx950.s")
    steps = [16, 24, 32, 48, 64, 96, 128, 256]
    base = dict(wave_base)
    if os.environ.get("NTG_AMD_WAVE_ABASE") and not any(src == "fam_kincar_wave.hip" for src, *_ in procs):
        _compile_wave(hipcc, wave_src, wave_obj, base, extra)   # asked for other bases than the object on disk has
    while True:
        bad = isa_audit.audit(hipcc, wave_src, os.path.join(HERE, "..", "include"), [], 16, asm_path=wave_asm)
        if not bad:
            break
        cls = isa_audit.failing_classes(bad)
        sys.stderr.write("ISA audit of fam_kincar_wave.hip: compiler-generated code in the hand-managed AGPR range (or spills) for the %s instances, e.g.\n  %s\n" % (" and ".join(sorted(cls)), bad[0][:200]))
        for c in cls:
            if base[c] >= 256:
                raise RuntimeError("ISA audit of fam_kincar_wave.hip still fails with no hand-managed accumulator registers left:\n  " + "\n  ".join(bad[:20]))
            base[c] = steps[steps.index(base[c]) + 1] if base[c] in steps else 256
        sys.stderr.write("  -> rebuilding with NTGW_ABASE=%d NTGW_ABASE_ALT=%d\n" % (base["main"], base["alt"]))
        _compile_wave(hipcc, wave_src, wave_obj, base, extra)
    with open(BASES, "w") as f:
        f.write("%d %d\n" % (base["main"], base["alt"]))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
