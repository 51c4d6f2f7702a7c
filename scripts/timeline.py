#!/usr/bin/env python3
"""Prints the kernel timeline of the last full step from a rocprofv3 --kernel-trace CSV (dev tool)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
tr = list(csv.DictReader(open(f)))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
# a step starts with pack_xyz4_kernel (the early schedule issues two Adam launches per step)
pk = [i for i, r in enumerate(tr) if r['Kernel_Name'].startswith('pack_xyz4_kernel')]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 0           # steps back from the last one (bench.py ends with two SERIAL stand-alone steps: 3 = the last two-stream step)
a, b = pk[-2 - back] - 1, pk[-1 - back] - 1
t0 = int(tr[a + 1]['Start_Timestamp'])
minms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
for r in tr[a + 1:b + 1]:
    s = (int(r['Start_Timestamp']) - t0) / 1e6; e = (int(r['End_Timestamp']) - t0) / 1e6
    if e - s >= minms:
        print(f"{s:8.3f} {e:8.3f} {e-s:7.3f} q{r['Queue_Id']} {r['Kernel_Name'][:70]}")
