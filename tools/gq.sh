#!/bin/bash
# tools/gq.sh TAG 'commands'  -- run commands on the GPU box with gpurun_out/r4 present; output of the commands lands in gpurun_out/r4/TAG.txt
tag=$1; shift
gpurun --timeout ${GQ_TIMEOUT:-900} -- "mkdir -p gpurun_out/r4 && ( $* ) > gpurun_out/r4/$tag.txt 2>&1; rc=\$?; grep -v amdgpu.ids gpurun_out/r4/$tag.txt | tail -${GQ_TAIL:-60}; exit \$rc" 2>&1 | grep -v "^\[gpurun\] sending\|merged" | tail -${GQ_TAIL:-70}
