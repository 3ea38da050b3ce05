#!/usr/bin/env python3
"""Prints the kernel timeline of one training step from a rocprofv3 --kernel-trace CSV (diagnostic).
usage: timeline.py DIR [--all]     (--all: also kernels outside xq::, e.g. runtime fill/copy kernels)"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
keep_all = '--all' in sys.argv
rows = [r for r in csv.DictReader(open(f)) if keep_all or 'xq::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'env_kernel' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp']); prev_end = None
for r in rows[a:b]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    gap = (s - prev_end) if prev_end is not None else 0
    name = r['Kernel_Name'].replace('void xq::', '').replace('xq::', '')[:44]
    print(f"{s/1000:8.1f} {e/1000:8.1f} dur {(e-s)/1000:7.1f} gap {gap/1000:6.1f}  q{r['Queue_Id']} {name}")
    prev_end = max(prev_end or 0, e)
print('step', (int(rows[b]['Start_Timestamp']) - t0) / 1000)
