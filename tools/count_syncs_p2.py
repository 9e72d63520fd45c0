"""Host read-backs of one configs[3]-style step (P2 space, gyroid) at n^3, outside and inside cutfemx_amd.run_step."""
import os; os.environ.setdefault("CFX_PATTERN_REUSE", "0")
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import cutfemx_amd as cfx
from cutfemx_amd import poisson, fem, _lib
from test_gpu_fullsize import level_set
dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
mesh = cfx.Mesh.create_box(3, n)
Vphi = cfx.FunctionSpace(mesh, 1)
phi = level_set('gyroid', n, 0, n, dev)
f = cfx.Function(Vphi, phi)
dm, nd = cfx.box_lagrange2_dofmap(mesh, n, dev)
V = cfx.FunctionSpace(mesh, 2, dofmap=dm, ndofs=nd)
b = torch.zeros(nd, device=dev, dtype=torch.float64)
def body():
    cd = cfx.cut(f)
    s = poisson.build_forms(V, cd, order=4)
    A = fem.create_matrix(s.a)
    fem.assemble_matrix(s.a, A=A)
    b.zero_()
    fem.assemble_vector(s.L, b)
    return fem.deactivate_outside(A, b, fem.active_domain(s.a))
for mode in ("plain", "step"):
    for it in range(4):
        torch.cuda.synchronize(); s0 = _lib.sync_count(); t0 = time.perf_counter()
        info = {}
        out = body() if mode == "plain" else cfx.run_step(body, key="p2-sync", info=info)
        torch.cuda.synchronize()
        print(mode, it, "ms", round(1e3 * (time.perf_counter() - t0), 2), "read-backs", _lib.sync_count() - s0, info, flush=True)
        del out
