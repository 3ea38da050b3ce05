#!/usr/bin/env python3
"""Percentiles of every kernel's duration from a rocprofv3 --kernel-trace CSV directory (tools/timeline.sh leaves none; run
rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py ... first).  usage: kernel_percentiles.py DIR [min_calls]"""
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 50
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < mn: continue
    v.sort()
    q = lambda p: v[min(len(v) - 1, int(p * len(v)))]
    print("%-64s n=%5d  p10 %6.1f  p50 %6.1f  p90 %6.1f  p99 %6.1f  mean %6.1f" % (k[:64], len(v), q(0.1), q(0.5), q(0.9), q(0.99), sum(v) / len(v)))
