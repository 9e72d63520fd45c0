#!/usr/bin/env python3
"""Launches of one timed step by launch name (the library's own profile: cfx_profile_*), for a mesh size.
usage: python tools/launch_list.py N [steps]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
out = bench.measure(n, steps, 3, 4, 1, 0, torch.device("cuda:0"), profile=True)   # (order 4: the bench's rules)
ks = out.get("kernels") or {}
tot = 0.0
for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"{k:34s} launches {v['launches']:5.1f}  avg {v['avg_us']:9.1f} us  total {v['total_ms']:8.4f} ms")
    tot += v["total_ms"]
print(f"n={n} ms_per_step {out['ms_per_step']:.4f} kernels {tot:.4f} ms launches {sum(v['launches'] for v in ks.values()):.1f}")
