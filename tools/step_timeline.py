#!/usr/bin/env python3
"""One search step as the GPU saw it: reads a rocprofv3 --kernel-trace CSV, finds the last complete step (from one
fused_query prep kernel to the next) and prints every kernel with its start offset, duration, queue, and the idle time of
the whole GPU in front of it.  Usage: step_timeline.py <kernel_trace.csv> [anchor kernel substring]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
anchor = sys.argv[2] if len(sys.argv) > 2 else "finish_survivors"
ends = [i for i, r in enumerate(rows) if anchor in r[2]]
# a step = the kernels after the last-but-one anchor run up to and including the last anchor run
last = ends[-1]
prev = max(i for i in ends if rows[last][0] - rows[i][0] > 2_000_000) if any(rows[last][0] - rows[i][0] > 2_000_000 for i in ends) else -1
step = rows[prev + 1:last + 1]
t0 = step[0][0]
busy_until = t0
print(f"{'start us':>10} {'dur us':>9} {'gpu idle before':>16} queue  kernel")
for s, e, name, q in step:
    idle = max(0, s - busy_until)
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {idle / 1e3:16.1f} {q:>5}  {short}")
    busy_until = max(busy_until, e)
print(f"step span {(step[-1][1] - t0) / 1e3:.1f} us")
