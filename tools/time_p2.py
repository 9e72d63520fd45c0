import os; os.environ.setdefault("CFX_PATTERN_REUSE", "0")  # a full rebuild per step, as the bench line (the step re-cuts one level set)
"""Timing of BASELINE config 4 (256^3 gyroid, P2 scalar) and config 5's rank share (P2 vector elasticity)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import cutfemx_amd as cfx
from cutfemx_amd import poisson, fem, _lib
from test_gpu_fullsize import level_set
import ctypes as C
dev = torch.device('cuda', 0)
def T(name, fn, acc):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); acc[name] = round(1e3 * (time.perf_counter() - t0), 2); return r
def profile(fn):
    l = _lib.lib(); _lib.check(l.cfx_profile_enable(1)); _lib.check(l.cfx_profile_reset()); fn()
    out = {}
    for i in range(l.cfx_profile_count()):
        name, ms, cnt = C.c_char_p(), C.c_double(), C.c_int64()
        _lib.check(l.cfx_profile_get(i, C.byref(name), C.byref(ms), C.byref(cnt)))
        if cnt.value: out[name.value.decode()] = round(ms.value, 2)
    _lib.check(l.cfx_profile_enable(0))
    return dict(sorted(out.items(), key=lambda kv: -kv[1])[:int(__import__("os").environ.get("TOPK", "8"))])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mesh = cfx.Mesh.create_box(3, n)
Vphi = cfx.FunctionSpace(mesh, 1)
cd = cfx.cut(cfx.Function(Vphi, level_set('gyroid', n, 0, n, dev)))
dm, nd = cfx.box_lagrange2_dofmap(mesh, n, dev)
V = cfx.FunctionSpace(mesh, 2, dofmap=dm, ndofs=nd)
for it in range(2):
    acc = {}
    s = T('forms', lambda: poisson.build_forms(V, cd, order=4), acc)
    A = T('sparsity', lambda: fem.create_matrix(s.a), acc)
    T('assemble_matrix', lambda: fem.assemble_matrix(s.a, A=A), acc)
    b = torch.zeros(nd, device=dev, dtype=torch.float64)
    T('assemble_vector', lambda: fem.assemble_vector(s.L, b), acc)
    print('cfg4 P2 gyroid', n, acc, 'nnz', A.nnz, 'inside', s.inside_cells.size, flush=True)
    if it == 1:
        def fresh():   # A = 0 then assemble: the fused path the bench step takes (rows stored, not read-modified-written)
            A.set_value(0.0)
            fem.assemble_matrix(s.a, A=A)
        print(' matrix kernels', profile(fresh), flush=True)
        print(' sparsity kernels', profile(lambda: fem.create_matrix(s.a)), flush=True)
        print(' vector kernels', profile(lambda: fem.assemble_vector(s.L, b)), flush=True)
        print(' forms + plan + sparsity kernels', profile(lambda: fem.create_matrix(poisson.build_forms(V, cd, order=4).a)), flush=True)
    del A, s, b
