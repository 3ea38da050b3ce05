#!/bin/bash
# same-box A/B of one bench.py flag: alternates `bench.py ARGS` and `bench.py ARGS FLAG`, R rounds, 3 x 300 steps each.
# Usage: tools/ab_flag.sh TAG FLAG [rounds] [bench args...]      e.g.  tools/ab_flag.sh tail --no-td-tail 3 --config 4
set -e
TAG=$1; FLAG=$2; R=${3:-3}; shift; shift; shift || true
mkdir -p gpurun_out
for i in $(seq 1 $R); do
  python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants --repeats 3 "$@" > gpurun_out/${TAG}_base_$i.json 2> gpurun_out/${TAG}_base_$i.err
  python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants --repeats 3 "$@" $FLAG > gpurun_out/${TAG}_flag_$i.json 2> gpurun_out/${TAG}_flag_$i.err
done
python3 - <<PY
import json, glob
out = {"flag_tested": "$FLAG", "args": "$*"}
for kind in ("base", "flag"):
    for f in sorted(glob.glob("gpurun_out/${TAG}_%s_[0-9]*.json" % kind)):
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        out.setdefault(kind, []).append({"ms_per_step": d["ms_per_step"], "samples": d["ms_per_step_samples"], "value": d["value"]})
        print(kind, "ms_per_step", round(d["ms_per_step"], 4), [round(x, 4) for x in d["ms_per_step_samples"]])
json.dump(out, open("gpurun_out/${TAG}_summary.json", "w"), indent=1)
PY
