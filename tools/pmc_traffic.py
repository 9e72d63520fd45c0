#!/usr/bin/env python3
"""profiles/<tag>_pmc_summary.txt -> profiles/<round>_traffic.json (usage: pmc_traffic.py <summary> <mesh> [round=r02]): HBM bytes per launch of every kernel
(FETCH_SIZE and WRITE_SIZE are reported in KiB... rocprofv3 derives them as requests x 64 B / 1024)."""
import json, re, sys
src, mesh = sys.argv[1], sys.argv[2]
rnd = sys.argv[3] if len(sys.argv) > 3 else "r04"
out = {}
for line in open(src):
    m = re.match(r"(\S.*?)\s+n=\s*\d+\s+(?:of=\s*\d+\s+)?(.*)", line)
    if not m:
        continue
    name = m.group(1).split("<")[0].replace("_kernel", "")
    vals = dict(kv.split("=") for kv in m.group(2).split())
    e = out.setdefault(name, {})
    if "FETCH_SIZE" in vals:
        e["fetch_bytes_raw"] = float(vals["FETCH_SIZE"]) * 1024.0
    if "WRITE_SIZE" in vals:
        e["write_bytes"] = float(vals["WRITE_SIZE"]) * 1024.0
    if "TCC_HIT_sum" in vals:
        e["l2_hit_rate"] = float(vals["TCC_HIT_sum"]) / (float(vals["TCC_HIT_sum"]) + float(vals["TCC_MISS_sum"]))
out = {k: v for k, v in out.items() if "fetch_bytes_raw" in v or "write_bytes" in v}
json.dump({"mesh": int(mesh), "source": src, "kernels": out}, open(f"profiles/{rnd}_traffic.json", "w"), indent=1, sort_keys=True)
print(len(out), "kernels")
