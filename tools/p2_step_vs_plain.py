#!/usr/bin/env python3
"""Per-launch-name kernel time of one configs[3]-style step (P2 space, gyroid) as the plain sequence and inside
cutfemx_amd.run_step: where the in-step form loses its ~4 ms.  usage: python tools/p2_step_vs_plain.py [n]"""
import os; os.environ.setdefault("CFX_PATTERN_REUSE", "0")
import ctypes as C, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import cutfemx_amd as cfx
from cutfemx_amd import poisson, fem, _lib
from test_gpu_fullsize import level_set
dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mesh = cfx.Mesh.create_box(3, n)
f = cfx.Function(cfx.FunctionSpace(mesh, 1), level_set('gyroid', n, 0, n, dev))
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
def profile(run, K=3):
    for _ in range(3): run()
    torch.cuda.synchronize()
    l = _lib.lib()
    _lib.check(l.cfx_profile_enable(1)); _lib.check(l.cfx_profile_reset())
    for _ in range(K): run()
    torch.cuda.synchronize()
    out = {}
    for i in range(l.cfx_profile_count()):
        nm, ms, cnt = C.c_char_p(), C.c_double(), C.c_int64()
        _lib.check(l.cfx_profile_get(i, C.byref(nm), C.byref(ms), C.byref(cnt)))
        if cnt.value: out[nm.value.decode()] = (ms.value / K, cnt.value / K)
    _lib.check(l.cfx_profile_enable(0))
    return out
plain = profile(body)
step = profile(lambda: cfx.run_step(body, key="p2-prof"))
names = sorted(set(plain) | set(step), key=lambda k: -(step.get(k, (0, 0))[0] - plain.get(k, (0, 0))[0]))
print(f"{'launch':28s} {'plain ms':>9s} {'x':>4s} {'step ms':>9s} {'x':>4s} {'diff':>8s}")
for k in names:
    p, s = plain.get(k, (0.0, 0)), step.get(k, (0.0, 0))
    if abs(s[0] - p[0]) > 0.02:
        print(f"{k:28s} {p[0]:9.3f} {p[1]:4.0f} {s[0]:9.3f} {s[1]:4.0f} {s[0] - p[0]:+8.3f}")
print("sum", round(sum(v[0] for v in plain.values()), 2), round(sum(v[0] for v in step.values()), 2))
