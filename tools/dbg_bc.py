import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import cutfemx_amd as cfx
from oracle import pyoracle as O
from helpers import level_set_values
O.build()
om = O.mesh_box(3, 6); phi = level_set_values(om.x, 3)
dom = O.classify(om.conn, phi); inside = O.locate_entities(dom, "phi<0")
ovol = O.runtime_quadrature(om, om.conn, phi, dom, "phi<0", 2); oghost = O.ghost_penalty_facets(om, dom, "phi<0")
oV = O.Space(om.conn, om.nnodes, 1)
mesh = cfx.Mesh.from_arrays(3, om.x, om.conn); V = cfx.FunctionSpace(mesh, 1)
cd = cfx.cut(cfx.Function(V, phi)); vol = cfx.runtime_quadrature(cd, "phi<0", 2); ghost = cfx.ghost_penalty_facets(cd, "phi<0")
bc = np.zeros(om.nnodes, dtype=np.int8); touched = np.unique(om.conn[inside]); bc[touched[::7]] = 1
for which in ["cells", "facets", "both"]:
    oa, ga = [], []
    if which in ("cells", "both"):
        oa.append(O.Integral(O.CELL, O.K_STIFFNESS, entities=inside, rules=ovol, qdegree=0)); ga.append(cfx.fem.Integral(cfx.fem.STIFFNESS, cells=inside, rules=vol, qdegree=0))
    if which in ("facets", "both"):
        oa.append(O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=oghost, params=(0.1,), qdegree=0)); ga.append(cfx.fem.Integral(cfx.fem.GHOST_GRADJUMP, facets=ghost, params=(0.1,), qdegree=0))
    ip, ix = O.create_sparsity(om, oV, oa)
    for usebc in (False, True):
        want = O.assemble_matrix(om, oV, oa, ip, ix, bc if usebc else None, bc if usebc else None)
        A = cfx.fem.assemble_matrix(cfx.fem.form(ga, V), bcs=bc if usebc else None)
        d = np.abs(A.data - want); bad = np.flatnonzero(d > 1e-10 * np.abs(want).max())
        rows = np.searchsorted(ip, bad, side='right') - 1
        print(which, 'bc' if usebc else 'nobc', 'DET' if os.environ.get('CFX_DETERMINISTIC') else 'atomic-lds', 'max err', d.max() / np.abs(want).max(), 'nbad', bad.size, 'rows', rows[:6], 'cols', ix[bad][:6], 'got', A.data[bad][:3], 'want', want[bad][:3], 'bcrow', bc[rows[:6]] if bad.size else '', 'bccol', bc[ix[bad][:6]] if bad.size else '')
