#!/bin/bash
# same-box A/B of one environment variable: alternates `bench.py ARGS` and `VAR=VALUE bench.py ARGS`, R rounds, 3 x 300 steps each.
# Usage: tools/ab_env.sh TAG VAR=VALUE [rounds] [bench args...]
set -e
TAG=$1; KV=$2; R=${3:-3}; shift; shift; shift || true
mkdir -p gpurun_out
for i in $(seq 1 $R); do
  python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants --repeats 3 "$@" > gpurun_out/${TAG}_base_$i.json 2> gpurun_out/${TAG}_base_$i.err
  export "$KV"
  python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants --repeats 3 "$@" > gpurun_out/${TAG}_flag_$i.json 2> gpurun_out/${TAG}_flag_$i.err
  unset "${KV%%=*}"
done
python3 - <<PY
import json, glob
for kind in ("base", "flag"):
    for f in sorted(glob.glob("gpurun_out/${TAG}_%s_[0-9]*.json" % kind)):
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(kind, "ms_per_step", round(d["ms_per_step"], 4), [round(x, 4) for x in d["ms_per_step_samples"]])
PY
