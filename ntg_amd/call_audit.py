"""Audit of the out-of-line call boundaries in the device code (run on the assembly of every translation unit).

Why: round 2 met an HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in a build where nwt_assemble / nwt_factor_wave / nwt_solve_wave were
all out of line.  What that build did and the shipped one does not: it passed `const NtgDims &` / `const NtgTables &`
-- the kernel's PRIVATE (scratch) copy of its by-value arguments -- to a callee as generic pointers, which the callee read with FLAT
loads through the private aperture, under a 2.6 KB scratch frame with ~300 VGPR and ~360 SGPR spills around the calls.  The rules
below keep every kernel out of that regime, by construction:
  R1  no kernel or device function ever forms a generic pointer into the private segment (no `src_private_base`);
  R2  out-of-line device functions contain no FLAT memory instruction (their pointer parameters carry an address space);
  R3  no dynamic stack, and the kernel descriptor's private segment covers the largest callee frame (the assembler's own max()).
"""
from __future__ import annotations
import re
import sys


def audit(path: str) -> list[str]:
    problems: list[str] = []
    cur, kind = None, {}
    flat_in_fn: dict[str, int] = {}
    for ln, line in enumerate(open(path, errors="replace"), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            kind[m.group(1)] = "kernel"
        code = line.split(";")[0].strip()
        if not code or cur is None:
            continue
        if "src_private_base" in code:
            problems.append(f"R1 {cur[:60]}: line {ln}: generic pointer into the private segment: {code}")
        if re.match(r"flat_(load|store|atomic)", code):
            flat_in_fn[cur] = flat_in_fn.get(cur, 0) + 1
        m = re.match(r"\.amdhsa_uses_dynamic_stack\s+(\d+)", code)
        if m and int(m.group(1)):
            problems.append(f"R3 {cur[:60]}: dynamic stack")
    for fn, n in flat_in_fn.items():
        if kind.get(fn) != "kernel":
            problems.append(f"R2 {fn[:70]}: {n} FLAT instruction(s) in an out-of-line device function")
    return problems


if __name__ == "__main__":
    bad = []
    for p in sys.argv[1:]:
        b = audit(p)
        print(f"{p}: {len(b)} problem(s)")
        bad += b
    for b in bad[:40]:
        print("  ", b)
    sys.exit(1 if bad else 0)
