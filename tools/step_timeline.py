#!/usr/bin/env python3
"""Kernel timeline of whole training steps (sgd kernel to sgd kernel) from a rocprofv3 --kernel-trace CSV, all queues.
usage: step_timeline.py DIR [first_step] [n_steps]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True))[-1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 40
count = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sg = [i for i, r in enumerate(rows) if 'sgd_segments' in r['Kernel_Name']]
for k in range(first, first + count):
    a, b = sg[k], sg[k + 1]
    t0 = int(rows[a]['End_Timestamp'])
    last_end = {}
    for r in rows[a + 1:b + 1]:
        s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
        q = r['Queue_Id']
        gap = s - last_end.get(q, s)
        last_end[q] = e
        name = r['Kernel_Name'].replace('void xq::', '').replace('xq::', '')[:46]
        print(f"{s/1000:8.1f} {e/1000:8.1f} dur {(e-s)/1000:6.1f} gap {gap/1000:5.1f} q{q} {name}")
    print('step', (int(rows[b]['End_Timestamp']) - t0) / 1000, 'us')
