#!/bin/bash
# one steady-state step of bench.py under rocprofv3 --kernel-trace as a timeline (runs ON the GPU box).  usage: tools/timeline_only.sh TAG [CONFIG]
TAG=$1; CFG=${2:-2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tl_${TAG}; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --config $CFG --steps 30 --warmup 10 --no-cpu-baseline --no-variants --no-facade --repeats 1 > $OUT/bench.json 2> $OUT/err.txt)
cd $ROOT
python3 tools/step_timeline.py $OUT $([ "$CFG" = "2" ] && echo 200 || echo 60) 1 > gpurun_out/${TAG}_step_timeline_config${CFG}.txt 2>> $OUT/err.txt
find $OUT -name "*_kernel_trace.csv" -delete
cat gpurun_out/${TAG}_step_timeline_config${CFG}.txt
