#!/bin/bash
# build a variant of libntg_amd.so that differs in the wave-kernel translation unit only:  tools/mkvariant.sh NAME [-DFLAG ...]
# The variant's device code goes through the same audits as the shipped unit (ntg_amd/isa_audit.py: no compiler-generated code in the
# hand-managed accumulator range, no spills in those instances; ntg_amd/call_audit.py) BEFORE anything is linked: other -D flags move the
# register pressure, and a variant that fails them can corrupt the chain or fault on the GPU.  ALLOW_ALT=1 tolerates findings in the
# instances the variant will not run (16 knot intervals / per-problem grids) and says so.
set -e
name=$1; shift
cd /root/repo/ntg_amd/csrc
mkdir -p ../variants
bases=$(cat fam_kincar_wave.abase 2>/dev/null || echo "16 16")
set -- -DNTGW_ABASE=${bases% *} -DNTGW_ABASE_ALT=${bases#* } "$@"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c fam_kincar_wave.hip -o ../variants/wave_$name.o -I ../../include -Wno-unused-result -Wno-unused-value -Wno-pass-failed -save-temps=obj -Wno-unused-command-line-argument "$@"
mv ../variants/fam_kincar_wave-hip-amdgcn-amd-amdhsa-gfx950.s ../variants/wave_$name.s
asm=../variants/wave_$name.s
rm -f ../variants/fam_kincar_wave-hip-* ../variants/fam_kincar_wave-host-* ../variants/fam_kincar_wave.hip-*
cd /root/repo
python - "$asm" <<'PY'
import os, sys
sys.path.insert(0, "/root/repo")
from ntg_amd import isa_audit, call_audit
asm = os.path.join("/root/repo/ntg_amd/csrc", sys.argv[1])
bad = isa_audit.audit("hipcc", "", "", [], 16, asm_path=asm)
cls = isa_audit.failing_classes(bad)
calls = call_audit.audit(asm)
if calls or "main" in cls or (cls and not os.environ.get("ALLOW_ALT")):
    print("variant REFUSED: audit failed (%s)" % (", ".join(sorted(cls)) or "call boundaries"))
    for b in (bad + calls)[:6]: print("  ", b[:220])
    sys.exit(1)
if cls: print("note: audit findings in the alt instances (16 intervals / per-problem grids) only -- do NOT run those with this variant")
PY
cd /root/repo/ntg_amd/csrc
objs=$(ls *.o | grep -v fam_kincar_wave.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libntg_$name.so $objs ../variants/wave_$name.o
echo built ../variants/libntg_$name.so
