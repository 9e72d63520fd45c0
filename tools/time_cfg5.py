import os; os.environ.setdefault("CFX_PATTERN_REUSE", "0")  # a full rebuild per step, as the bench line (the step re-cuts one level set)
"""Timing of one rank's share of BASELINE config 5 (256^3, P2 vector elasticity, sphere): slab of 32 layers."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import cutfemx_amd as cfx
from cutfemx_amd import fem
from test_gpu_fullsize import level_set
dev = torch.device('cuda', 0)
n, z0, nz = 256, 89, 32
mesh = cfx.Mesh.create_slab(n, z0, nz)
Vphi = cfx.FunctionSpace(mesh, 1)
cd = cfx.cut(cfx.Function(Vphi, level_set('sphere', n, z0, nz, dev)))
dm, nd = cfx.box_lagrange2_dofmap(mesh, n, dev)
V = cfx.FunctionSpace(mesh, 2, dofmap=dm, ndofs=nd, bs=3)
inside = cfx.locate_entities_device(cd, "phi<0")
vol = cfx.runtime_quadrature(cd, "phi<0", 2)
ga = [fem.Integral(fem.ELASTICITY, cells=inside, rules=vol, params=(1.0e3, 0.3), qdegree=2)]
def T(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, round(1e3 * (time.perf_counter() - t0), 2)
for it in range(3):
    a = fem.form(ga, V)
    A, ts = T(lambda: fem.create_matrix(a))
    _, ta = T(lambda: fem.assemble_matrix(a, A=A))
    print('cfg5 share: inside', inside.size, 'nnz', A.nnz, 'sparsity ms', ts, 'assemble_matrix ms', ta, flush=True)
    if it == 2:
        import ctypes as C
        from cutfemx_amd import _lib
        l = _lib.lib(); _lib.check(l.cfx_profile_enable(1)); _lib.check(l.cfx_profile_reset())
        fem.assemble_matrix(a, A=fem.create_matrix(a))   # a fresh matrix: the set_value(0) + assemble path the timing loop takes
        out = {}
        for i in range(l.cfx_profile_count()):
            name, ms, cnt = C.c_char_p(), C.c_double(), C.c_int64()
            _lib.check(l.cfx_profile_get(i, C.byref(name), C.byref(ms), C.byref(cnt)))
            if cnt.value: out[name.value.decode()] = round(ms.value, 2)
        print(' kernels', {k: v for k, v in sorted(out.items(), key=lambda kv: -kv[1]) if k.startswith(('assemble', 'elasticity', 'fill', 'zero'))})
    del A, a
