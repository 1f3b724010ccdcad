#!/bin/bash
# (locally: rm -rf gpurun_out/r4prof first -- gpurun merges into it, and stale run directories would be picked up by the summariser)
# Round-4 profiles on the GPU box (run from the repo root through gpurun): kernel-trace statistics of the bench command, then
# counter passes (--pmc only, one group per pass) over tools/prof_r3.py.  Raw output under gpurun_out/r4prof/.
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r4prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo stats done
pmc() { name=$1; shift; what=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_${what}_${name} -- python3 $ROOT/tools/prof_r3.py $what > /dev/null 2> $OUT/pmc_${what}_${name}.err || exit 1; echo pmc $what $name done; }
pmc fetch fixed50 FETCH_SIZE
pmc write fixed50 WRITE_SIZE
pmc fetch conv_h1_big FETCH_SIZE
pmc write conv_h1_big WRITE_SIZE
pmc sq fixed50 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY
pmc sq2 fixed50 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
pmc mfma conv_h1_big SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc mfma qp_E SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc mfma newton_E SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc fetch qp_E FETCH_SIZE
pmc write qp_E WRITE_SIZE
echo profiles done
