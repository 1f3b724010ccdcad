#!/bin/bash
# build a variant of libntg_amd.so that differs in the wave-kernel translation unit only:  tools/mkvariant.sh NAME [-DFLAG ...]
set -e
name=$1; shift
cd /root/repo/ntg_amd/csrc
mkdir -p ../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c fam_kincar_wave.hip -o ../variants/wave_$name.o -I ../../include -Wno-unused-result -Wno-unused-value -Wno-pass-failed "$@"
objs=$(ls *.o | grep -v fam_kincar_wave.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libntg_$name.so $objs ../variants/wave_$name.o
echo built ../variants/libntg_$name.so
