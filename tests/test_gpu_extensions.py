"""8f-3: cell aggregation and the extension-penalty stabilisation
(cpp/cutfemx/extensions/, python/tests/test_extensions_cell_aggregation.py)."""
import numpy as np
import pytest

from helpers import level_set_values, rel_err

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _setup(oracle, tdim, n, degree=1, kind="sphere"):
    import cutfemx_amd as cfx
    O = oracle
    om = O.mesh_box(tdim, n)
    phi = level_set_values(om.x, tdim, kind)
    dofmap, ndofs = cfx.lagrange_dofmap(tdim, om.conn, om.nnodes, degree)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, degree, dofmap=None if degree == 1 else dofmap, ndofs=ndofs)
    Vphi = V if degree == 1 else cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(Vphi, phi))
    return dict(O=O, om=om, phi=phi, oV=O.Space(dofmap, ndofs, degree), mesh=mesh, V=V, cd=cd,
                dom=O.classify(om.conn, phi))


@pytest.mark.parametrize("mode", ["rows", "atomic"])
@pytest.mark.parametrize("tdim,n,degree", [(2, 12, 1), (3, 6, 1), (2, 8, 2)])
def test_extension_penalty_pair_blocks_match_oracle(oracle, monkeypatch, mode, tdim, n, degree):
    # extension_penalty.cpp:191-369: the four (bad, root) blocks, symmetric, constants annihilated
    import cutfemx_amd as cfx
    if mode == "atomic":
        monkeypatch.setenv("CFX_ASSEMBLY", "atomic")
    s = _setup(oracle, tdim, n, degree)
    O, om = s["O"], s["om"]
    agg = O.cell_aggregation(om, om.conn, s["phi"], s["dom"], "phi<0", 0.6)
    pairs = O.extension_pairs(agg)
    assert pairs.shape[0] > 0
    q = 2 * degree
    beta_cell = 1.0 + 0.01 * np.arange(om.ncells)
    for pdata in (None, beta_cell[pairs[:, 0]]):
        oa = [O.Integral(O.INTERIOR_FACET, O.K_EXTENSION_L2, entities=pairs, params=(2.5,), qdegree=q, point_data=pdata)]
        ga = [cfx.fem.Integral(cfx.fem.EXTENSION_L2, facets=pairs, params=(2.5,), qdegree=q, point_data=pdata)]
        ip, ix = O.create_sparsity(om, s["oV"], oa)
        want = O.assemble_matrix(om, s["oV"], oa, ip, ix)
        A = cfx.fem.assemble_matrix(cfx.fem.form(ga, s["V"]))
        assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix)
        assert rel_err(A.data, want) < RTOL
        M = A.to_scipy()
        assert abs(M - M.T).max() < 1e-12 * abs(M).max()
        assert np.abs(M @ np.ones(M.shape[0])).max() < 1e-12 * abs(M).max()
    # together with the Poisson terms (pairs and ghost facets in one form: the folded and the
    # unfolded facet items must coexist)
    inside = O.locate_entities(s["dom"], "phi<0")
    ovol = O.runtime_quadrature(om, om.conn, s["phi"], s["dom"], "phi<0", 2)
    oghost = O.ghost_penalty_facets(om, s["dom"], "phi<0")
    vol = cfx.runtime_quadrature(s["cd"], "phi<0", 2)
    ghost = cfx.ghost_penalty_facets(s["cd"], "phi<0")
    qs = 2 * (degree - 1)
    oa = [O.Integral(O.CELL, O.K_STIFFNESS, entities=inside, rules=ovol, qdegree=qs),
          O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=oghost, params=(0.1,), qdegree=qs),
          O.Integral(O.INTERIOR_FACET, O.K_EXTENSION_L2, entities=pairs, params=(2.5,), qdegree=q)]
    ga = [cfx.fem.Integral(cfx.fem.STIFFNESS, cells=inside, rules=vol, qdegree=qs),
          cfx.fem.Integral(cfx.fem.GHOST_GRADJUMP, facets=ghost, params=(0.1,), qdegree=qs),
          cfx.fem.Integral(cfx.fem.EXTENSION_L2, facets=pairs, params=(2.5,), qdegree=q)]
    ip, ix = O.create_sparsity(om, s["oV"], oa)
    want = O.assemble_matrix(om, s["oV"], oa, ip, ix)
    A = cfx.fem.assemble_matrix(cfx.fem.form(ga, s["V"]))
    assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix)
    assert rel_err(A.data, want) < RTOL
