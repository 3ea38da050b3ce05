#!/bin/bash
# plain / one-rank communicator early start / one-rank communicator late start, same box
for i in 1 2; do
  python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants --no-facade --no-chain --repeats 3 > gpurun_out/c1_plain_$i.json 2>/dev/null
  XQ_BENCH_COMM1=1 python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants --no-facade --no-chain --repeats 3 --exchange-overlap 0 > gpurun_out/c1_early_$i.json 2>/dev/null
  XQ_BENCH_COMM1=1 python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants --no-facade --no-chain --repeats 3 --exchange-overlap 1 > gpurun_out/c1_late_$i.json 2>/dev/null
done
python3 - <<PY
import json, glob
for kind in ("plain", "early", "late"):
    for f in sorted(glob.glob("gpurun_out/c1_%s_[0-9].json" % kind)):
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(kind, round(d["ms_per_step"], 4), [round(x, 4) for x in d["ms_per_step_samples"]])
PY
