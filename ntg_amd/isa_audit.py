"""ISA audit of the wave kernels (fam_kincar_wave.hip): the instances that keep the chain of search directions in the accumulator
registers address a0..a251 by hand (inline asm).  That is only safe if the compiler itself never touches an AGPR in those kernels
(guide: cdna_hip_programming.md 5.7 item 4) -- no spill to AGPRs, no AV-class allocation -- and never spills to scratch.  Run by
ntg_amd/build.py after every build; a violation fails the build."""
from __future__ import annotations
import re
import subprocess
import sys


maxidx: dict = {}
warnings: list = []   # spills in kernels that do not address AGPRs by hand: slow, not wrong


def audit(hipcc: str, src: str, include: str, flags: list[str], reserve_from: int = 0, asm_path: str | None = None) -> list[str]:
    asm = open(asm_path, errors="replace").read() if asm_path else subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-S", "--cuda-device-only", src, "-o", "-",
                          "-I", include, "-Wno-unused-result", "-Wno-unused-value", "-Wno-pass-failed", "-Wno-unused-command-line-argument"] + flags,
                         check=True, capture_output=True, text=True).stdout
    problems, cur, inasm, uses_manual, meta_name = [], None, False, {}, None
    for ln, line in enumerate(asm.splitlines(), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
        if ";;#ASMSTART" in line:
            inasm = True
            continue
        if ";;#ASMEND" in line:
            inasm = False
            continue
        code = line.split(";")[0]
        if inasm and "accvgpr" in code and cur:
            uses_manual[cur] = True
        if not inasm and cur and not code.strip().startswith("."):
            for mm in re.finditer(r"\ba\[?(\d+)(?::(\d+))?\]?", code):
                hi = int(mm.group(2) or mm.group(1))
                maxidx[cur] = max(maxidx.get(cur, -1), hi)
                if hi >= kernel_base(cur, reserve_from):
                    problems.append((cur, ln, code.strip()))
        m = re.match(r"\s*\.name:\s*(\S+)", line)
        if m:
            meta_name = m.group(1)
        m = re.match(r"\s*\.(vgpr_spill_count|private_segment_fixed_size):\s*(\d+)", line)
        if m and int(m.group(2)) != 0:
            problems.append((meta_name, ln, line.strip()))
    out = []
    for cur, ln, code in problems:
        if uses_manual.get(cur):
            out.append(f"{cur}: line {ln}: {code}")
        else:
            warnings.append(f"{cur}: line {ln}: {code}")
    return out


def kernel_base(mangled: str, default: int) -> int:
    """first hand-managed accumulator register of a wave-kernel instance: its last template argument (ABASE, solve_wave.hpp)"""
    m = re.search(r"sqp_wave_kernelI.*?ELi(\d+)EEEv", mangled or "")
    return int(m.group(1)) if m else default


def failing_classes(problems: list[str]) -> set[str]:
    """which class of instances the audit's findings belong to: 'alt' = 16 knot intervals or per-problem grids (NTGW_ABASE_ALT), 'main' = the rest"""
    out = set()
    for pr in problems:
        m = re.search(r"sqp_wave_kernelILi\d+ELi\d+ELi\d+ELi\d+ELi\d+ELi(\d+)ELi\d+ELi\d+ELi\d+ELi\d+ELb[01]ELb[01]ELb([01])E", pr)
        out.add("alt" if (m and (m.group(1) != "20" or m.group(2) == "1")) else "main")
    return out


def agpr_base(header: str) -> int:
    """NTGW_ABASE of solve_wave.hpp: first accumulator register of the hand-managed range"""
    m = re.search(r"#define NTGW_ABASE (\d+)", open(header).read())
    return int(m.group(1))


if __name__ == "__main__":
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    rf = agpr_base(os.path.join(here, "csrc", "solve_wave.hpp"))
    bad = audit(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), os.path.join(here, "csrc", "fam_kincar_wave.hip"), os.path.join(here, "..", "include"), sys.argv[1:], rf)
    for k, v in maxidx.items():
        print("highest AGPR touched by compiler-generated code:", v, "in", k[:70])
    for w in warnings[:10]:
        print("warning:", w)
    print(f"{len(bad)} problem(s)")
    for b in bad[:30]:
        print(" ", b)
    sys.exit(1 if bad else 0)
