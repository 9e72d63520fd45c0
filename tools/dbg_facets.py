import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import cutfemx_amd as cfx
from oracle import pyoracle as O
tdim, n, degree = 2, 10, 1
om = O.mesh_box(tdim, n)
phi = om.x[:, 0] - 0.51 + 0.07 * om.x[:, 1]
mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
V = cfx.FunctionSpace(mesh, 1)
f = cfx.Function(V, phi)
oV = O.Space(om.conn, om.nnodes, 1)
orows = O.interior_facets_for_cells(om, np.arange(om.ncells, dtype=np.int32))
H = O.facet_hosts(om, orows, om.conn)
fdom = O.facet_classify(H, phi)
orun = O.facet_runtime_quadrature(om, H, phi, fdom, "phi<0", 2)
cdf = cfx.cut(f, orows, tdim - 1)
run = cfx.runtime_quadrature(cdf, "phi<0", 2)
for name in ("GHOST_GRADJUMP", "JUMP"):
    ok, gk = getattr(O, "K_" + name), getattr(cfx.fem, name)
    oa2 = [O.Integral(O.INTERIOR_FACET, ok, rules=orun, params=(0.3,), qdegree=2)]
    ga2 = [cfx.fem.Integral(gk, rules=run, params=(0.3,), qdegree=2)]
    ip, ix = O.create_sparsity(om, oV, oa2)
    a = cfx.fem.form(ga2, V)
    A2 = cfx.fem.assemble_matrix(a)
    want = O.assemble_matrix(om, oV, oa2, ip, ix)
    d = np.abs(A2.data - want)
    k = int(np.argmax(d))
    row = int(np.searchsorted(ip, k, side="right") - 1)
    print(name, "max diff", d.max(), "at", k, "row", row, "col", ix[k], "got", A2.data[k], "want", want[k], "scale", np.abs(want).max())
    for r in range(orun.parent_map.size):
        Ae = cfx.fem.tabulate_entity(a, 0, r, False)
        We = O.tabulate_entity(om, oV, oa2[0], r, False)
        if np.abs(Ae - We).max() > 1e-12 * max(np.abs(We).max(), 1e-30):
            print(" local tensor mismatch at rule", r, np.abs(Ae - We).max(), np.abs(We).max()); break
