#!/bin/bash
# Runs ON the GPU box (through gpurun): rocprofv3 kernel trace + the two HBM-traffic PMC passes of the bench command.
# usage: tools/collect_profiles.sh TAG        -> gpurun_out/prof_TAG/{trace,fetch,write}/...
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-facade"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_under_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > /dev/null 2> $OUT/write.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -- $CMD > /dev/null 2> $OUT/sq.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq2 -- $CMD > /dev/null 2> $OUT/sq2.err
rocprofv3 --pmc SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/sq3 -- $CMD > /dev/null 2> $OUT/sq3.err
python3 $ROOT/bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
python3 $ROOT/bench.py --steps 30 --warmup 5 --profile-all --no-cpu-baseline --no-facade > $OUT/bench_all_kernels.json 2> $OUT/bench_all.err
ls -R $OUT | head -40
