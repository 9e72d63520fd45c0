"""Setup kernels of the first step at n^3 (mesh-static tables): per-kernel HIP-event times.
usage: python tools/time_setup.py [n]   (variants through CFX_ADJ_LDS / CFX_C2C, one process each)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
import cutfemx_amd as cfx
from cutfemx_amd import _lib, poisson

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
bench.library_warmup(torch, dev) if hasattr(bench, "library_warmup") else None
mesh = cfx.Mesh.create_box(3, n)
V = cfx.FunctionSpace(mesh, 1)
phi = cfx.Function(V, bench.sphere_level_set(torch, n, dev))
b = torch.zeros(mesh.num_nodes, device=dev, dtype=torch.float64)
_lib.check(_lib.lib().cfx_profile_enable(1))
_lib.check(_lib.lib().cfx_profile_reset())
torch.cuda.synchronize(); t0 = time.perf_counter()
cd = cfx.cut(phi)
system = poisson.build_forms(V, cd, order=4)
A = cfx.fem.create_matrix(system.a)
cfx.fem.assemble_matrix(system.a, A=A)
cfx.fem.assemble_vector(system.L, b)
torch.cuda.synchronize(); t1 = time.perf_counter()
rows = []
for i in range(_lib.lib().cfx_profile_count()):
    name, ms, cnt = C.c_char_p(), C.c_double(), C.c_int64()
    _lib.check(_lib.lib().cfx_profile_get(i, C.byref(name), C.byref(ms), C.byref(cnt)))
    if cnt.value:
        rows.append((ms.value, name.value.decode(), cnt.value))
setup = ("adj_", "stencil_", "cell_neighbours", "scan_reduce", "scan_write")
tot = sum(r[0] for r in rows if r[1].startswith(setup))
print(f"n={n} first step {1e3 * (t1 - t0):.1f} ms, setup kernels {tot:.1f} ms, nnz {A.nnz}  "
      f"[CFX_ADJ_LDS={os.environ.get('CFX_ADJ_LDS', '')} CFX_C2C={os.environ.get('CFX_C2C', '')}]")
for ms, name, cnt in sorted(rows, reverse=True)[:12]:
    print(f"  {name:28s} {ms:9.3f} ms  x{cnt}")
# a checksum of the tables' consumers: ghost facets and pattern must not depend on the variant
print("  check:", int(A.nnz), 0 if system.ghost_facets is None else system.ghost_facets.size,
      int(torch.as_tensor(A.indices[:1000000].astype('int64')).sum()))
