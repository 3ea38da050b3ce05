#!/bin/bash
# same-box A/B: plain single-GPU step vs the same step with a one-rank communicator attached (the whole N > 1 code path of the
# library: unfused slab reductions, two all-reduce calls on the producers' streams).  Usage: tools/ab_comm1.sh TAG [rounds]
set -e
TAG=${1:-ab}; R=${2:-3}
mkdir -p gpurun_out
for i in $(seq 1 $R); do
  python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants --repeats 3 > gpurun_out/${TAG}_plain_$i.json 2> gpurun_out/${TAG}_plain_$i.err
  XQ_BENCH_COMM1=1 python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants --repeats 3 > gpurun_out/${TAG}_comm1_$i.json 2> gpurun_out/${TAG}_comm1_$i.err
done
python3 - <<PY
import json, glob
for kind in ("plain", "comm1"):
    for f in sorted(glob.glob("gpurun_out/${TAG}_%s_*.json" % kind)):
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(kind, f, "ms_per_step", round(d["ms_per_step"], 4), [round(x, 4) for x in d["ms_per_step_samples"]], d["config"]["parallelism"])
PY
