#!/bin/bash
CMD="examples/_build/train_selfplay 400000 /tmp/model.bin 8192 --replay 1048576 --minibatch 8192 --hidden 256,256 --save-every 0 --prefill 300 --seed 0x5EED --json --derive"
for kv in "NONE=1" "XQ_EVENT_SYSFENCE=1" "XQ_FORK_STOP_EVENT=0" "XQ_SGD_SCALAR=1" "XQ_SCREEN_XCD=0" "NONE=2"; do
  export "$kv"
  echo "$kv: $($CMD 2>/dev/null | python3 -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d.get("env_steps_per_s"), d.get("updates"), d.get("loop_seconds"))')"
  unset "${kv%%=*}"
done
