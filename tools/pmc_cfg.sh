#!/bin/bash
# Runs ON the GPU box: one rocprofv3 --pmc pass of a short bench run of a given config; counters as arguments.
# usage: tools/pmc_cfg.sh TAG CONFIG COUNTER...   -> gpurun_out/pmc_TAG/
set -e
TAG=$1; CFG=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 $ROOT/bench.py --config $CFG --steps 12 --warmup 4 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt
