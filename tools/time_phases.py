import sys, time
sys.path.insert(0, '.')
import torch, cutfemx_amd as cfx
from cutfemx_amd import fem
from bench import sphere_level_set
dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
mesh = cfx.Mesh.create_box(3, n); V = cfx.FunctionSpace(mesh, 1)
phi = cfx.Function(V, sphere_level_set(torch, n, dev))
def T(name, fn, acc):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); acc[name] = acc.get(name, 0) + time.perf_counter() - t0; return r
for it in range(6):
    acc = {}
    cd = T('cut', lambda: cfx.cut(phi), acc)
    ins = T('locate', lambda: cfx.locate_entities_device(cd, "phi<0"), acc)
    vol = T('rq_vol', lambda: cfx.runtime_quadrature(cd, "phi<0", 4), acc)
    itf = T('rq_itf', lambda: cfx.runtime_quadrature(cd, "phi=0", 4), acc)
    nrm = T('normal', lambda: cfx.normal(cd, itf, device=True), acc)
    gh = T('ghost', lambda: cfx.ghost_penalty_facets(cd, "phi<0"), acc)
    a = T('form_a', lambda: fem.form([fem.Integral(fem.STIFFNESS, cells=ins, rules=vol, qdegree=0), fem.Integral(fem.NITSCHE, rules=itf, point_data=nrm, params=(40.,)), fem.Integral(fem.GHOST_GRADJUMP, facets=gh, params=(0.1,), qdegree=0)], V), acc)
    del a, gh, nrm, itf, vol, cd
print({k: round(1e3 * v, 3) for k, v in acc.items()})
