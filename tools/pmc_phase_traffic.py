#!/usr/bin/env python3
"""HBM traffic of one phase of a timing program from its rocprofv3 --pmc summary (tools/pmc_report.py output).

usage: pmc_phase_traffic.py <pmc_summary.txt> <nsteps> <out.json> <tag> <kernel-regex> [<kernel-regex> ...]

Every kernel whose name matches one of the regexes belongs to the phase; its FETCH_SIZE / WRITE_SIZE means per
dispatch (KiB, as rocprofv3 reports them) times its dispatches per step (dispatches / nsteps) are added up.  FETCH_SIZE
is quoted RAW: MI355X_MICROARCH.md's x2 correction applies to 16 B/lane coalesced streaming reads only, these are
gather kernels."""
import json
import re
import sys

src, nsteps, out, tag = sys.argv[1], float(sys.argv[2]), sys.argv[3], sys.argv[4]
pats = [re.compile(p) for p in sys.argv[5:]]
kern = {}
for line in open(src):
    m = re.match(r"(\S.*?)\s+n=\s*(\d+)\s+(?:of=\s*\d+\s+)?(.*)", line)
    if not m or not any(p.search(m.group(1)) for p in pats):
        continue
    vals = dict(kv.split("=") for kv in m.group(3).split())
    e = kern.setdefault(m.group(1).strip(), {"dispatches": int(m.group(2))})
    if "FETCH_SIZE" in vals:
        e["fetch_bytes_raw_per_dispatch"] = float(vals["FETCH_SIZE"]) * 1024.0
    if "WRITE_SIZE" in vals:
        e["write_bytes_per_dispatch"] = float(vals["WRITE_SIZE"]) * 1024.0
    if "TCC_HIT_sum" in vals:
        e["l2_hit_rate"] = float(vals["TCC_HIT_sum"]) / (float(vals["TCC_HIT_sum"]) + float(vals["TCC_MISS_sum"]))
    if "SQ_WAIT_ANY" in vals and "SQ_WAVE_CYCLES" in vals:
        e["wait_any_share"] = float(vals["SQ_WAIT_ANY"]) / float(vals["SQ_WAVE_CYCLES"])
        e["valu_wave_instr_per_dispatch"] = float(vals["SQ_INSTS_VALU"])
        e["waves_per_dispatch"] = float(vals["SQ_WAVES"])
fetch = sum(e.get("fetch_bytes_raw_per_dispatch", 0.0) * e["dispatches"] / nsteps for e in kern.values())
write = sum(e.get("write_bytes_per_dispatch", 0.0) * e["dispatches"] / nsteps for e in kern.values())
json.dump({"tag": tag, "source": src, "nsteps": nsteps, "phase_fetch_bytes_raw": fetch, "phase_write_bytes": write,
           "phase_traffic_bytes": fetch + write, "kernels": kern}, open(out, "w"), indent=1, sort_keys=True)
print(tag, "fetch %.3g B  write %.3g B per step over %d kernels" % (fetch, write, len(kern)))
