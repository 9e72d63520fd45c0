"""Kernel times of the FIRST step (mesh-static table builds included) at the given mesh size."""
import sys, ctypes as C
sys.path.insert(0, '.')
import torch
import cutfemx_amd as cfx
from cutfemx_amd import poisson, fem, _lib
from bench import sphere_level_set
dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mesh = cfx.Mesh.create_box(3, n); V = cfx.FunctionSpace(mesh, 1)
phi = cfx.Function(V, sphere_level_set(torch, n, dev))
l = _lib.lib(); _lib.check(l.cfx_profile_enable(1)); _lib.check(l.cfx_profile_reset())
cd = cfx.cut(phi); s = poisson.build_forms(V, cd, order=4)
A = fem.create_matrix(s.a); fem.assemble_matrix(s.a, A=A)
b = torch.zeros(V.ndofs, device=dev, dtype=torch.float64); fem.assemble_vector(s.L, b)
fem.deactivate_outside(A, b, fem.active_domain(s.a))
out = {}
for i in range(l.cfx_profile_count()):
    name, ms, cnt = C.c_char_p(), C.c_double(), C.c_int64()
    _lib.check(l.cfx_profile_get(i, C.byref(name), C.byref(ms), C.byref(cnt)))
    if cnt.value: out[name.value.decode()] = (round(ms.value, 2), cnt.value)
for k, v in sorted(out.items(), key=lambda kv: -kv[1][0])[:16]: print(f"{k:28s} {v[0]:9.2f} ms x{v[1]}")
print(V.static_table_bytes())
