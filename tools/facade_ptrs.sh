#!/bin/bash
export XQ_DEBUG_PTRS=1
CMD="examples/_build/train_selfplay 400000 /tmp/model.bin 8192 --replay 1048576 --minibatch 8192 --hidden 256,256 --save-every 0 --prefill 300 --seed 0x5EED --json --derive"
for i in $(seq 1 12); do
  $CMD 2>/tmp/err.txt | python3 -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("%.4f ms per update" % (1e3*d["loop_seconds"]/d["updates"]), end="  ")'
  grep "\[xq\]" /tmp/err.txt | tail -1
done
