"""One-line summary of a bench.py JSON line: python tools/show_bench.py file.json [...]"""
import json
import sys

for f in sys.argv[1:]:
    t = open(f).read().strip()
    d = json.loads(t if t.startswith("{\n") else t.splitlines()[-1])   # bench_detail.json (indented) or a one-line record
    sm = d.get("step_mode") or {}
    ps = ((d.get("projected_scaling") or {}).get("by_world") or {}).get("8") or {}
    print(f, "ms/step", round(d["ms_per_step"], 4), "value", f"{d['value']:.4g}", "launches", sm.get("launches_per_step"),
          "read-backs", sm.get("read_backs_per_step"), "passes", sm.get("passes"), "published", sm.get("published"),
          "proj8", ps.get("projected_speedup"))
