#!/bin/bash
# per-kernel HIP-event brackets (bench.py's roofline_chain) of this tree's library against tools/_build/prev/libxqhip.so, same box
# usage: tools/chain_ab.sh [bench args...]
mkdir -p gpurun_out
for lib in new prev; do
  if [ $lib = prev ]; then export XQ_LIBXQHIP=tools/_build/prev/libxqhip.so; else unset XQ_LIBXQHIP; fi
  python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-variants --no-facade --repeats 1 "$@" > gpurun_out/chain_ab_$lib.json 2>/dev/null
  python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/chain_ab_$lib.json") if l.startswith("{")][-1])
rc=d["roofline_chain"]
print("$lib", "step ms", round(d["ms_per_step"],4), [(e["kernel"], round(e["avg_us"],1)) for e in rc["handle_stream"]], [(e["kernel"], round(e["avg_us"],1)) for e in rc.get("collect_stream", [])])
PY
done
