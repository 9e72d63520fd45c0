#!/usr/bin/env python3
"""Soak run of the moving-domain loop: many sync-free steps with the interface moving back and forth (P1 Poisson with
Nitsche + ghost penalty, and a degree-2 space every 10th step); reports the engine's HBM in use / cached / peak and the
repeated steps every `every` steps -- a leak or a size history that never settles shows here.
usage: python tools/soak.py [N] [steps] [every]"""
import ctypes as C, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import cutfemx_amd as cfx
from cutfemx_amd import _lib, poisson
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
every = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda:0")
mesh = cfx.Mesh.create_box(3, n)
V = cfx.FunctionSpace(mesh, 1)
V2 = cfx.FunctionSpace(mesh, 2)
ax = torch.arange(n + 1, device=dev, dtype=torch.float64) / n
phi = torch.empty((n + 1) ** 3, device=dev, dtype=torch.float64)
f = cfx.Function(V, phi)
values = torch.zeros(int(mesh.num_nodes) + 40 * int(0.3 * mesh.num_nodes + 100000), device=dev, dtype=torch.float64)
b = torch.zeros(mesh.num_nodes, device=dev, dtype=torch.float64)
state = {"cd": None}

def place(k):
    cx = 0.45 + 0.12 * math.sin(0.37 * k)
    R = 0.30 + 0.04 * math.sin(0.11 * k)
    d2 = (ax[:, None, None] - 0.41) ** 2 + (ax[None, :, None] - 0.43) ** 2 + (ax[None, None, :] - cx) ** 2
    phi.copy_((torch.sqrt(d2) - R).reshape(-1))

def body():
    if state["cd"] is None:
        state["cd"] = cfx.cut(f)
    else:
        cfx.update(state["cd"])
    system = poisson.build_forms(V, state["cd"], order=4)
    _lib.check(_lib.lib().cfx_device_memset(C.c_void_p(b.data_ptr()), 0, C.c_size_t(8 * b.numel())))
    A = cfx.fem.create_matrix(system.a, values=values)
    A.set_value(0.0)
    cfx.fem.assemble_matrix(system.a, A=A)
    cfx.fem.assemble_vector(system.L, b)
    dom = cfx.fem.deactivate_outside(A, b, cfx.fem.active_domain(system.a))
    return bench.StepResult(system, A, dom)

def p2_step():
    # the same problem on a degree-2 space over the same cut (hashed pattern rows, the row-reuse cache): outside run_step
    system = poisson.build_forms(V2, state["cd"], order=4)
    A = cfx.fem.create_matrix(system.a)
    A.set_value(0.0)
    cfx.fem.assemble_matrix(system.a, A=A)
    return A.nnz

redo = 0
t0 = time.perf_counter()
for k in range(steps):
    place(k)
    info = {}
    out = cfx.run_step(body, key="soak", info=info)
    redo += info.get("passes", 1) - 1
    nnz = out.A.nnz
    nnz2 = None
    if k % 10 == 9:
        nnz2 = p2_step()
    del out
    if k % every == every - 1:
        torch.cuda.synchronize()
        m = _lib.memory_stats()
        try:
            import psutil
            rss = psutil.Process().memory_info().rss / 2**20
        except ImportError:
            rss = float("nan")
        print(f"step {k + 1:5d}: host RSS {rss:8.1f} MiB in_use {m['in_use'] / 2**20:9.1f} MiB cached {m['cached'] / 2**20:9.1f} MiB peak {m['peak'] / 2**20:9.1f} MiB "
              f"repeated steps so far {redo} nnz {nnz} nnz(P2) {nnz2} {1e3 * (time.perf_counter() - t0) / (k + 1):.3f} ms/step", flush=True)
