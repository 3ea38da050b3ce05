#!/bin/bash
# the C++ entry point several times (time-seeded initial weights, like upstream): time per update beside the screening guard's counters
CMD="examples/_build/train_selfplay 400000 /tmp/model.bin 8192 --replay 1048576 --minibatch 8192 --hidden 256,256 --save-every 0 --prefill 300 --seed 0x5EED --json --derive"
for i in $(seq 1 ${1:-8}); do
  $CMD 2>/dev/null | python3 -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("%.2f M env steps/s, %d updates, %.4f ms per update, screened %d, guard fallbacks %d, %.2f candidate groups per sample (%.2f whole)" % (d["env_steps_per_s"]/1e6, d["updates"], 1e3*d["loop_seconds"]/d["updates"], d["screened_steps"], d["guard_fallbacks"], d["candidate_groups_per_sample"], d["whole_groups_per_sample"]))'
done
