#!/bin/bash
# Runs ON the GPU box (through gpurun): rocprofv3 kernel trace of bench.py for a non-headline config (4 / 5) + its bench lines.
# usage: tools/collect_profiles_cfg.sh TAG CONFIG   -> gpurun_out/prof_TAG/...
set -e
TAG=$1; CFG=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --config $CFG --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.err
python3 $ROOT/bench.py --config $CFG --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
python3 $ROOT/bench.py --config $CFG --steps 30 --warmup 5 --profile-all --no-cpu-baseline > $OUT/bench_all_kernels.json 2> $OUT/bench_all.err
