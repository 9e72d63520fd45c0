#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean of each counter per WORKLOAD-SIZED dispatch.
bench.py runs an 8^3 step first (library warm-up) and each kernel name therefore has a few tiny dispatches next to the
ones of the mesh under test; a mean over all of them dilutes the per-launch figures (round-3 verdict: classify_kernel
n = 9 of which 2 tiny).  A kernel that also runs once at set-up with a much larger grid (round-4 verdict: pattern_rows, whose
first-step full-hash dispatch was taken for the per-step one) has its steady-state dispatches in the grid-size class that
REPEATS most often: dispatches are grouped by grid size (classes a factor 1.5 apart) and the class with the most
dispatches is averaged (ties: the larger grid; the warm-up mesh runs fewer steps than the mesh under test).  `n=` is the
number used, `of=` the number seen."""
import csv, sys, glob, collections
def short(name):
    s = name.replace("(anonymous namespace)::", "").replace("cfx::", "").replace("void ", "")
    return s.split("(")[0][:70]
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = collections.defaultdict(list)          # kernel -> [(grid, counter, value)]
        for row in csv.DictReader(open(f)):
            grid = float(row.get("Grid_Size") or row.get("Grid_Size_X") or 0) or 1.0
            rows[short(row["Kernel_Name"])].append((grid, row["Counter_Name"], float(row["Counter_Value"])))
        acc, seen = {}, {}
        for k, rs in rows.items():
            gmax = max(g for g, _, _ in rs)
            import math
            cls = lambda g: int(math.log(max(g, 1.0)) / math.log(1.5))
            first = next(iter({c for _, c, _ in rs}))
            count = collections.Counter(cls(g) for g, c, _ in rs if c == first)
            best = max(count, key=lambda q: (count[q], q))
            cs = collections.defaultdict(list)
            for g, c, v in rs:
                if cls(g) == best:
                    cs[c].append(v)
            acc[k] = cs
            seen[k] = max(sum(1 for g, c2, _ in rs if c2 == c) for c in {c for _, c, _ in rs})
        print("==", f)
        for k, cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
            n = max(len(v) for v in cs.values())
            print(f"{k:70s} n={n:4d} of={seen[k]:4d} " + " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
