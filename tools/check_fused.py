"""The fused count + write kernels against the three-launch form on a mesh with long chains:
python tools/check_fused.py [n]  (run once per setting of CFX_FUSED_TILES; prints digests to compare)"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
import cutfemx_amd as cfx
from cutfemx_amd import poisson

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
mesh = cfx.Mesh.create_box(3, n)
V = cfx.FunctionSpace(mesh, 1)
phi = cfx.Function(V, bench.sphere_level_set(torch, n, dev))
values = torch.zeros(int(mesh.num_nodes) * 30, device=dev, dtype=torch.float64)
b = torch.zeros(mesh.num_nodes, device=dev, dtype=torch.float64)
res = None
for k in range(3):   # the third step runs on the recorded sizes of the second
    res = cfx.run_step(lambda: bench.hot_path_step(cfx, poisson, V, phi, values, b, 4, None, False), key="check")
torch.cuda.synchronize()
A = res.A
ip, ix, va = A.torch_views(dev)
nnz = A.nnz
h = hashlib.sha256()
h.update(ip.cpu().numpy().tobytes())
h.update(ix[:nnz].cpu().numpy().tobytes())
print("fused tiles", os.environ.get("CFX_FUSED_TILES", "default"), "n", n, "nnz", nnz, "pattern", h.hexdigest()[:16],
      "values", hashlib.sha256(va[:nnz].cpu().numpy().tobytes()).hexdigest()[:16],
      "b", hashlib.sha256(b.cpu().numpy().tobytes()).hexdigest()[:16])
