#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   -> ms_per_step of bench.py for each value of the env var (diagnostic)
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value']), d['ms_per_step'])"
done
