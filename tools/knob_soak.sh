#!/bin/bash
export KNOB_PROBE_GAMES=8192 KNOB_PROBE_UPDATES=5000
for kv in "NONE=1" "XQ_EVENT_SYSFENCE=1" "NONE=2" "XQ_FORK_STOP_EVENT=0"; do
  export "$kv"; echo "$kv $(python3 tests/knob_probe.py | grep KNOB_PROBE)"; unset "${kv%%=*}"
done
