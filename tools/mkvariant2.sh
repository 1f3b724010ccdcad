#!/bin/bash
# variant of libntg_amd.so that differs in the listed translation units:  tools/mkvariant2.sh NAME "fam_quadrotor fam_manip" [-DFLAG ...]
set -e
name=$1; tus=$2; shift 2
cd /root/repo/ntg_amd/csrc
mkdir -p ../variants
excl=""
for tu in $tus; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c $tu.hip -o ../variants/${tu}_$name.o -I ../../include -Wno-unused-result -Wno-unused-value -Wno-pass-failed "$@" &
  excl="$excl|$tu.o"
done
wait
objs=$(ls *.o | grep -v -E "^(${excl#|})$")
vobjs=""; for tu in $tus; do vobjs="$vobjs ../variants/${tu}_$name.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libntg_$name.so $objs $vobjs
echo built ../variants/libntg_$name.so
