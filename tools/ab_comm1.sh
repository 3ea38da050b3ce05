#!/bin/bash
# same-box A/B: plain single-GPU step vs the same step with a one-rank communicator attached (the whole N > 1 code path of the
# library: unfused slab reductions, two all-reduce calls), for this build and — when tools/_build/r02/libxqhip.so exists — for the
# round-2 build, whose buckets ran on a communicator stream of their own.  Usage: tools/ab_comm1.sh TAG [rounds]
set -e
TAG=${1:-ab}; R=${2:-3}
mkdir -p gpurun_out
OLD=tools/_build/r02/libxqhip.so
for i in $(seq 1 $R); do
  python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants --repeats 3 > gpurun_out/${TAG}_plain_$i.json 2> gpurun_out/${TAG}_plain_$i.err
  XQ_BENCH_COMM1=1 python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants --repeats 3 > gpurun_out/${TAG}_comm1_$i.json 2> gpurun_out/${TAG}_comm1_$i.err
  if [ -f $OLD ]; then
    XQ_LIBXQHIP=$OLD python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants --repeats 3 > gpurun_out/${TAG}_r02plain_$i.json 2> gpurun_out/${TAG}_r02plain_$i.err
    XQ_LIBXQHIP=$OLD XQ_BENCH_COMM1=1 python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants --repeats 3 > gpurun_out/${TAG}_r02comm1_$i.json 2> gpurun_out/${TAG}_r02comm1_$i.err
  fi
done
python3 - <<PY
import json, glob
out = {}
for kind in ("plain", "comm1", "r02plain", "r02comm1"):
    for f in sorted(glob.glob("gpurun_out/${TAG}_%s_[0-9]*.json" % kind)):
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        out.setdefault(kind, []).append({"ms_per_step": d["ms_per_step"], "samples": d["ms_per_step_samples"], "env_steps_per_s": d["value"]})
        print(kind, f, "ms_per_step", round(d["ms_per_step"], 4), [round(x, 4) for x in d["ms_per_step_samples"]])
json.dump(out, open("gpurun_out/${TAG}_summary.json", "w"), indent=1)
PY
