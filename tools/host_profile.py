#!/usr/bin/env python3
"""Where the HOST time of a small step goes: cProfile over sync-free steps of the bench's hot path at N^3 (the 32^3 step is
bound by its host side, not by its kernels).  usage: python tools/host_profile.py [N] [steps]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import cutfemx_amd as cfx
from cutfemx_amd import poisson
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
mesh = cfx.Mesh.create_box(3, n)
V = cfx.FunctionSpace(mesh, 1)
phi = cfx.Function(V, bench.sphere_level_set(torch, n, dev))
nnz_cap = int(mesh.num_nodes) + 40 * int(0.2 * mesh.num_nodes + 100000)
vals = torch.zeros(nnz_cap, device=dev, dtype=torch.float64)
b = torch.zeros(mesh.num_nodes, device=dev, dtype=torch.float64)
def step():
    return cfx.run_step(lambda: bench.hot_path_step(cfx, poisson, V, phi, vals, b, 4, None, True), key="hp")
for _ in range(10): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"n={n}: {1e3 * dt:.4f} ms per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(steps): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
