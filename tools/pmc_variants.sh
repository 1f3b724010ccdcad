#!/bin/bash
# instruction counts of the headline launch per library variant (doubled-phase ablations): tools/pmc_variants.sh lib1.so lib2.so ...
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  NTG_AMD_LIB=$ROOT/$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/$n -- python3 $ROOT/tools/run_fixed50.py 3 > /dev/null 2> $OUT/$n.err || { echo "$n failed"; tail -3 $OUT/$n.err; exit 1; }
  python3 - $OUT/$n $n <<'PY'
import csv, glob, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sqp_wave_kernel" in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
n = len(per); tot = collections.defaultdict(float)
for d in per.values():
    for k, v in d.items(): tot[k] += v / n
print(sys.argv[2], {k: round(v / 4096) for k, v in sorted(tot.items())}, "per problem,", n, "launches")
PY
done
