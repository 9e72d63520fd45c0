"""The launch sequence of one sync-free step at n^3 (CFX_LAUNCH_TRACE): python tools/launch_trace.py [n]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["CFX_LAUNCH_TRACE"] = "1"
import torch

import bench
import cutfemx_amd as cfx
from cutfemx_amd import poisson

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
mesh = cfx.Mesh.create_box(3, n)
V = cfx.FunctionSpace(mesh, 1)
phi = cfx.Function(V, bench.sphere_level_set(torch, n, dev))
values = torch.zeros(int(mesh.num_nodes) * 30, device=dev, dtype=torch.float64)
b = torch.zeros(mesh.num_nodes, device=dev, dtype=torch.float64)
for k in range(3):
    sys.stderr.write(f"==== step {k}\n")
    sys.stderr.flush()
    cfx.run_step(lambda: bench.hot_path_step(cfx, poisson, V, phi, values, b, 4, None, False), key="trace")
    torch.cuda.synchronize()
