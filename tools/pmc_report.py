#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean of each counter per dispatch."""
import csv, sys, glob, collections, re
def short(name):
    s = name.replace("(anonymous namespace)::", "").replace("cfx::", "").replace("void ", "")
    return s.split("(")[0][:70]
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("==", f)
        for k, cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
            n = max(len(v) for v in cs.values())
            print(f"{k:70s} n={n:4d} " + " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
