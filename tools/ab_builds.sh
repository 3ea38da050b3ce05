#!/bin/bash
# same-box A/B of two builds of libxqhip.so on bench.py: this tree's against tools/_build/prev/libxqhip.so (XQ_LIBXQHIP).
# usage: tools/ab_builds.sh TAG CONFIG [rounds] [steps]
TAG=${1:-ab}; CFG=${2:-2}; R=${3:-3}; STEPS=${4:-300}
for i in $(seq 1 $R); do
  python3 bench.py --config $CFG --steps $STEPS --warmup 20 --no-cpu-baseline --no-variants --no-facade --no-chain --repeats 3 > gpurun_out/${TAG}_new_$i.json 2>/dev/null
  XQ_LIBXQHIP=tools/_build/prev/libxqhip.so python3 bench.py --config $CFG --steps $STEPS --warmup 20 --no-cpu-baseline --no-variants --no-facade --no-chain --repeats 3 > gpurun_out/${TAG}_prev_$i.json 2>/dev/null
done
python3 - <<PY
import json, glob
for kind in ("new", "prev"):
    for f in sorted(glob.glob("gpurun_out/${TAG}_%s_[0-9]*.json" % kind)):
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(kind, f, round(d["ms_per_step"], 4), [round(x, 4) for x in d["ms_per_step_samples"]])
PY
