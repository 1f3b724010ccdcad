#!/bin/bash
# end-of-round collection on the GPU box (run from the repo root through gpurun): GPU tests, the round's profiles, the default bench line
# (with the CPU baseline) and the lines of configs D / E.  Locally first: rm -rf gpurun_out/r4prof gpurun_out/final
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/final; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $OUT/gputests.log
bash tools/prof_r4.sh > $OUT/prof.log 2>&1; echo "profiles rc=$?"
cd $ROOT
timeout -k 10 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
timeout -k 10 600 python bench.py --config E --steps 3 --warmup 1 --no-cpu --no-extras > $OUT/bench_config_E.json 2> $OUT/bench_config_E.err; echo "bench E rc=$?"
timeout -k 10 600 python bench.py --config D --steps 3 --warmup 1 --no-cpu --no-extras > $OUT/bench_config_D.json 2> $OUT/bench_config_D.err; echo "bench D rc=$?"
