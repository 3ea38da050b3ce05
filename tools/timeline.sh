#!/bin/bash
# Runs ON the GPU box: kernel timeline (all queues, gaps per queue) of one steady-state training step of bench.py --config CFG.
# usage: tools/timeline.sh TAG CONFIG [step]      -> gpurun_out/timeline_TAG.txt
TAG=$1; CFG=${2:-2}; STEP=${3:-40}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --config $CFG --steps 30 --warmup 10 --no-cpu-baseline --no-variants --repeats 1 > $OUT/bench.json 2> $OUT/err.txt
python3 $ROOT/tools/step_timeline.py $OUT $STEP 1 > $ROOT/gpurun_out/timeline_$TAG.txt 2>> $OUT/err.txt
tail -3 $OUT/err.txt
find $OUT -name "*_kernel_trace.csv" -delete
