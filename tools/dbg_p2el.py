import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import pyoracle as O
O.build()
import cutfemx_amd as cfx
from test_gpu_spaces import setup, elasticity_problem
tdim, n = int(sys.argv[1]), int(sys.argv[2])
s = setup(O, tdim, n, 2, tdim)
om, oV = s["om"], s["oV"]
inside, oa, ga = elasticity_problem(s, 2)
a = cfx.fem.form(ga[:1], s["V"])
oa = oa[:1]
ndofs = oV.ndofs * tdim
markers = np.zeros(ndofs, dtype=np.int8)
touched = np.unique(oV.dofmap[inside])
for k in range(tdim):
    markers[touched[::4] * tdim + k] = 1
ip, ix = O.create_sparsity(om, oV, oa)
for bc in (None, markers):
    want = O.assemble_matrix(om, oV, oa, ip, ix, bc, bc) if bc is not None else O.assemble_matrix(om, oV, oa, ip, ix)
    from helpers import profiled
    A, names = profiled(lambda: cfx.fem.assemble_matrix(a, bcs=bc) if bc is not None else cfx.fem.assemble_matrix(a))
    print(sorted(names))
    d = np.abs(A.data - want)
    if bc is None: A0 = A.data.copy()
    rows = np.repeat(np.arange(ndofs), np.diff(ip))
    bad = d > 1e-9 * np.abs(want).max()
    print('bc' if bc is not None else 'nobc', 'bad entries', bad.sum(), 'of', d.size, 'bad rows', np.unique(rows[bad]).size,
          'max', d.max(), 'scale', np.abs(want).max())
    if bad.any():
        br = np.unique(rows[bad])
        print(' bad rows: dof types vertex?', np.sum(br // tdim < om.nnodes), 'edge', np.sum(br // tdim >= om.nnodes), 'comp', np.bincount(br % tdim))
        k = np.nonzero(bad)[0][:8]
        for i in k: print('  row', rows[i], 'col', ix[i], 'got', A.data[i], 'want', want[i], 'bcrow', markers[rows[i]], 'bccol', markers[ix[i]], 'nobc value', A0[i])
