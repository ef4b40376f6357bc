#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per orr:: kernel, mean counter values and
mean dispatch duration.  Usage: summarize_pmc.py <dir> [name-substring]"""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "orr::"
out = {}
for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if want in k:
            name = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for name, cs in agg.items():
        o = out.setdefault(name, {})
        o["avg_dispatch_ns"] = sum(dur[name]) / len(dur[name])
        for c, v in cs.items():
            o[c] = sum(v) / len(v)
            o[c + "_launches"] = len(v)
print(json.dumps(out, indent=1))
