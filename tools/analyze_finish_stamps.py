#!/usr/bin/env python3
"""Reads an ORR_FINISH_STAMPS file: per launch, the workgroups' phase durations (us; the stamps tick at 100 MHz) and when the
workgroups started and ended relative to the launch's first stamp.  Usage: analyze_finish_stamps.py file [launch index, default last]"""
import sys
import numpy as np

launches, cur = [], None
for line in open(sys.argv[1]):
    if line.startswith("launch"):
        cur = {"head": line.strip(), "rows": []}
        launches.append(cur)
    elif cur is not None:
        cur["rows"].append([int(x) for x in line.split()])
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
L = launches[which]
a = np.array(L["rows"], dtype=np.float64)
print(len(launches), "launches; showing", L["head"], "workgroups with stamps:", len(a))
t0 = a[:, 1].min()
names = ["start", "Q staged + metadata", "rows added", "slow rows redone", "scored", "sorted + ticket", "finisher end"]
for k in range(1, 8):
    col = a[:, k]
    have = col > 0
    if not have.any():
        continue
    rel = (col[have] - t0) / 100.0
    print(f"{names[k - 1]:>22}: n={have.sum():5d}  at us: min {rel.min():8.1f} median {np.median(rel):8.1f} p95 {np.percentile(rel, 95):8.1f} max {rel.max():8.1f}")
for k in range(2, 8):
    have = (a[:, k] > 0) & (a[:, k - 1] > 0)
    d = (a[have, k] - a[have, k - 1]) / 100.0
    if have.sum():
        print(f"phase -> {names[k - 1]:>22}: n={have.sum():5d}  us: median {np.median(d):7.1f} p95 {np.percentile(d, 95):7.1f} max {d.max():7.1f}")
