#!/usr/bin/env python3
"""usage: trace_gaps.py <dir with rocprofv3 kernel_trace.csv> [min_gap_us=8]: GPU idle time between consecutive kernels of
the last classify-to-classify step of a bench run, and the kernels that follow the longest gaps."""
import csv, glob, sys
d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("cfx::", "").replace("void ", "").split("(")[0][:60]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "classify_kernel" in r[2]]
# the big-mesh steps: classify launches whose duration is large
big = [i for i in starts if rows[i][1] - rows[i][0] > 500000]
if len(big) < 3:
    big = starts
a, b = big[-2], big[-1]
step = rows[a:b]
busy = sum(e - s for s, e, _ in step)
span = step[-1][1] - step[0][0]
print(f"kernels in the step: {len(step)}, busy {busy/1e6:.3f} ms, first start to last end {span/1e6:.3f} ms, next step starts {(rows[b][0]-step[-1][1])/1e3:.1f} us after the last kernel")
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(step[:-1], step[1:]):
    gaps.append(((s1 - e0) / 1e3, n0, n1))
print(f"sum of gaps {sum(g for g, _, _ in gaps)/1e3:.3f} ms, median gap {sorted(g for g, _, _ in gaps)[len(gaps)//2]:.1f} us")
for g, n0, n1 in sorted(gaps, reverse=True)[:25]:
    if g >= min_gap:
        print(f"  {g:8.1f} us  after {n0:45s} before {n1}")
