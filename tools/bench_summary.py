"""Print the headline numbers and the slowest kernels of a bench.py JSON line."""
import json
import sys

t = open(sys.argv[1]).read().strip()
d = json.loads(t if t.startswith("{\n") else t.splitlines()[-1])   # bench_detail.json (indented) or a one-line record
print(f"{d['ms_per_step']:.3f} ms/step  {d['value']:.4e} {d['unit']}  phases {d.get('phases_ms')}")
r = d.get("roofline", {})
print("roofline", r.get("kernel"), r.get("achieved"), r.get("frac"), "traffic", r.get("traffic"))
ks = d.get("kernels", {})
for name, v in sorted(ks.items(), key=lambda kv: -kv[1].get("total_ms", 0))[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(f"  {name:28s} {v['total_ms']:8.3f} ms  x{int(v['launches'])}")
