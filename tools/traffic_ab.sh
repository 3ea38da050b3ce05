#!/bin/bash
# Runs ON the GPU box: HBM traffic per kernel (FETCH_SIZE / WRITE_SIZE passes; bytes = (2 FETCH + WRITE) KB as in summarize_profiles.py) of this tree's
# library against tools/_build/prev/libxqhip.so, then the same-box step-time A/B.   usage: tools/traffic_ab.sh [CONFIG] [rounds]
CFG=${1:-2}; R=${2:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in new prev; do
  if [ $lib = prev ]; then export XQ_LIBXQHIP=$ROOT/tools/_build/prev/libxqhip.so; else unset XQ_LIBXQHIP; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    OUT=$ROOT/gpurun_out/traf_${lib}_$c; rm -rf $OUT; mkdir -p $OUT
    rocprofv3 --pmc $c --output-format csv -d $OUT -- python3 $ROOT/bench.py --config $CFG --steps 12 --warmup 4 --no-cpu-baseline --no-variants --no-facade --no-chain --repeats 1 > /dev/null 2> $OUT/err.txt
  done
  python3 - <<PY
import csv, glob, collections
def means(c):
    agg=collections.defaultdict(list)
    for f in glob.glob("$ROOT/gpurun_out/traf_${lib}_%s/**/*_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "xq::" in r["Kernel_Name"]: agg[r["Kernel_Name"].split("(")[0][:70]].append(float(r["Counter_Value"]))
    return {k: sum(v)/len(v) for k,v in agg.items()}
f=means("FETCH_SIZE"); w=means("WRITE_SIZE")
print("$lib:", "  ".join("%s %.1f MB" % (k.replace("void xq::",""), (2*f[k]+w.get(k,0))*1024/1e6) for k in sorted(f)))
PY
  rm -rf $ROOT/gpurun_out/traf_${lib}_FETCH_SIZE $ROOT/gpurun_out/traf_${lib}_WRITE_SIZE
done
cd $ROOT; unset XQ_LIBXQHIP
tools/ab_builds.sh traf $CFG $R 300 | tail -$((2*R))
