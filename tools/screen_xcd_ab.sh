#!/bin/bash
# Runs ON the GPU box: the screening pass's block -> XCD map (XQ_SCREEN_XCD = 0: an XCD takes sample panels / unset: half of the row ranges x a
# quarter of the panels): HBM traffic of screen_top2_kernel (FETCH_SIZE / WRITE_SIZE passes, gfx950 correction as in summarize_profiles.py) and step time.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for mode in 0 2; do
  if [ $mode = 0 ]; then export XQ_SCREEN_XCD=0; else unset XQ_SCREEN_XCD; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    OUT=$ROOT/gpurun_out/xcd_${mode}_$c; rm -rf $OUT; mkdir -p $OUT
    rocprofv3 --pmc $c --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-variants --no-facade --no-chain --repeats 1 > /dev/null 2> $OUT/err.txt
  done
  python3 - <<PY
import csv, glob, collections
def mean(c):
    v=[float(r["Counter_Value"]) for f in glob.glob("$ROOT/gpurun_out/xcd_${mode}_%s/**/*_counter_collection.csv" % c, recursive=True) for r in csv.DictReader(open(f)) if "screen_top2_kernel" in r["Kernel_Name"]]
    return sum(v)/max(len(v),1), len(v)
f,nf=mean("FETCH_SIZE"); w,nw=mean("WRITE_SIZE")
print("mode $mode: screen_top2_kernel HBM bytes per launch = (2 x %.0f + %.0f) KB = %.1f MB  (%d / %d launches)" % (f, w, (2*f+w)*1024/1e6, nf, nw))
PY
  rm -rf $ROOT/gpurun_out/xcd_${mode}_FETCH_SIZE $ROOT/gpurun_out/xcd_${mode}_WRITE_SIZE
done
cd $ROOT
unset XQ_SCREEN_XCD
tools/ab_env.sh r5_xcd XQ_SCREEN_XCD=0 3 | tail -6
