import os; os.environ.setdefault("CFX_PATTERN_REUSE", "0")
"""create_matrix alone on BASELINE config 4 (256^3 gyroid, P2 scalar): per-kernel times (timing-variant libraries of the
sparsity kernels give wrong patterns: nothing is assembled here)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import ctypes as C
import torch
import cutfemx_amd as cfx
from cutfemx_amd import poisson, fem, _lib
from test_gpu_fullsize import level_set
dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mesh = cfx.Mesh.create_box(3, n)
Vphi = cfx.FunctionSpace(mesh, 1)
cd = cfx.cut(cfx.Function(Vphi, level_set('gyroid', n, 0, n, dev)))
dm, nd = cfx.box_lagrange2_dofmap(mesh, n, dev)
V = cfx.FunctionSpace(mesh, 2, dofmap=dm, ndofs=nd)
s = poisson.build_forms(V, cd, order=4)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    A = fem.create_matrix(s.a)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print('create_matrix ms', round(1e3 * (t1 - t0), 2), 'nnz', A.nnz, flush=True)
    del A
l = _lib.lib(); _lib.check(l.cfx_profile_enable(1)); _lib.check(l.cfx_profile_reset())
A = fem.create_matrix(s.a)
out = {}
for i in range(l.cfx_profile_count()):
    name, ms, cnt = C.c_char_p(), C.c_double(), C.c_int64()
    _lib.check(l.cfx_profile_get(i, C.byref(name), C.byref(ms), C.byref(cnt)))
    if cnt.value: out[name.value.decode()] = round(ms.value, 2)
print(' kernels', dict(sorted(out.items(), key=lambda kv: -kv[1])[:int(os.environ.get("TOPK", "8"))]), flush=True)
