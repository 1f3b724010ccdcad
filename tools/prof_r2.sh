#!/bin/bash
# Round-2 profiles on the GPU box (run from the repo root through gpurun): kernel-trace statistics of the bench command, then
# counter passes (each with --kernel-trace/--stats-free --pmc only) over tools/prof_r2.py.  Raw output under gpurun_out/r2prof/.
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r2prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu > $OUT/bench.json 2> $OUT/bench.err || exit 1
pmc() { name=$1; shift; what=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_${what}_${name} -- python3 $ROOT/tools/prof_r2.py $what > /dev/null 2> $OUT/pmc_${what}_${name}.err || exit 1; }
pmc fetch sqp FETCH_SIZE
pmc write sqp WRITE_SIZE
pmc fetch eval FETCH_SIZE
pmc write eval WRITE_SIZE
pmc lds eval SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY
pmc mfma newtonD SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc mfma newtonE SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc fetch newtonE FETCH_SIZE
pmc write newtonE WRITE_SIZE
echo profiles done
