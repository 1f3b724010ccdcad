#!/bin/bash
# SQ counters of the Jacobian-assembly evaluation (configs D and E): tools/jac.py under rocprofv3 --pmc
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3jac; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY --output-format csv -d $OUT/sq -- python3 $ROOT/tools/jac.py > $OUT/a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/sq2 -- python3 $ROOT/tools/jac.py > $OUT/b.log 2>&1 || exit 1
echo done
