"""8f-3: cell aggregation and the extension-penalty stabilisation
(cpp/cutfemx/extensions/, python/tests/test_extensions_cell_aggregation.py)."""
import numpy as np
import pytest

from helpers import level_set_values, rel_err, scrambled_mesh

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _setup(oracle, tdim, n, degree=1, kind="sphere"):
    import cutfemx_amd as cfx
    O = oracle
    om = scrambled_mesh(O, tdim, n) if kind.endswith("-scrambled") else O.mesh_box(tdim, n)
    phi = level_set_values(om.x, tdim, kind.replace("-scrambled", ""))
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


def _compare_aggregation(a, o):
    for name in ("active_cells", "cut_cells", "interior_cells", "well_posed_cells", "ill_posed_cells", "rootless_cells",
                 "root_cell", "aggregate_id", "propagation_depth"):
        assert np.array_equal(getattr(a, name), o[name]), name
    assert rel_err(a.cut_volume_fraction, o["cut_volume_fraction"]) < RTOL


@pytest.mark.parametrize("tdim,n,kind", [(2, 16, "sphere"), (3, 8, "sphere"), (3, 10, "gyroid"), (2, 24, "gyroid"),
                                         (3, 8, "sphere-scrambled"), (2, 20, "gyroid-scrambled")])
@pytest.mark.parametrize("threshold,policy", [(0.3, "interior_or_well_cut"), (1.0, "interior_or_well_cut"),
                                              (0.0, "interior_only")])
def test_cell_aggregation_matches_sequential_oracle(oracle, tdim, n, kind, threshold, policy):
    # cell_aggregation.cpp:143-270: the parallel relaxation must reproduce the sequential sweeps
    import cutfemx_amd as cfx
    s = _setup(oracle, tdim, n, 1, kind)
    O, om = s["O"], s["om"]
    for selector in ("phi<0", "phi > 0"):
        want = O.cell_aggregation(om, om.conn, s["phi"], s["dom"], selector, threshold, root_policy=policy,
                                  allow_rootless=True)
        got = cfx.extensions.create_cell_aggregation(s["cd"], selector, threshold, root_policy=policy,
                                                     allow_rootless=True)
        _compare_aggregation(got, want)
        assert np.array_equal(got.pairs, O.extension_pairs(want))
        # test_extensions_cell_aggregation.py:30-71
        assert set(got.well_posed_cells.tolist()) <= set(got.active_cells.tolist())
        assert set(got.ill_posed_cells.tolist()) <= set(got.cut_cells.tolist())
        rooted = got.ill_posed_cells[got.root_cell[got.ill_posed_cells] >= 0]
        assert np.all(got.propagation_depth[rooted] > 0)
        assert set(got.root_cell[rooted].tolist()) <= set(got.well_posed_cells.tolist())


def test_cell_aggregation_iteration_limit_and_rootless(oracle):
    # test_extensions_cell_aggregation.py:101-120
    import cutfemx_amd as cfx
    s = _setup(oracle, 2, 16)
    O, om = s["O"], s["om"]
    for limit in (0, 1, 2):
        want = O.cell_aggregation(om, om.conn, s["phi"], s["dom"], "phi<0", 1.0, max_iterations=limit,
                                  allow_rootless=True)
        got = cfx.extensions.create_cell_aggregation(s["cd"], "phi<0", 1.0, max_iterations=limit, allow_rootless=True)
        _compare_aggregation(got, want)
    got = cfx.extensions.create_cell_aggregation(s["cd"], "phi<0", 1.0, max_iterations=0, allow_rootless=True)
    assert got.rootless_cells.size == got.ill_posed_cells.size > 0
    with pytest.raises(RuntimeError, match="without an admissible root"):
        cfx.extensions.create_cell_aggregation(s["cd"], "phi<0", 1.0, max_iterations=0)
    with pytest.raises(ValueError):
        cfx.extensions.create_cell_aggregation(s["cd"], "phi<=0", 0.5)
    with pytest.raises(ValueError):
        cfx.extensions.create_cell_aggregation(s["cd"], "phi<0", 1.5)
    with pytest.raises(ValueError, match="Unknown root policy"):
        cfx.extensions.create_cell_aggregation(s["cd"], "phi<0", 0.5, root_policy="nearest")


def test_opposite_volume_fractions_sum_to_one(oracle):
    # test_extensions_cell_aggregation.py:74-98
    import cutfemx_amd as cfx
    s = _setup(oracle, 3, 8)
    neg = cfx.extensions.create_cell_aggregation(s["cd"], "phi<0", 0.5, allow_rootless=True)
    pos = cfx.extensions.create_cell_aggregation(s["cd"], "phi>0", 0.5, allow_rootless=True)
    assert np.array_equal(neg.cut_cells, pos.cut_cells)
    total = neg.cut_volume_fraction[neg.cut_cells] + pos.cut_volume_fraction[pos.cut_cells]
    np.testing.assert_allclose(total, 1.0, atol=1e-12)


def test_extension_penalty_api(oracle):
    # test_extensions_cell_aggregation.py:123-146 (symmetric, annihilates constants), :178-209 (DG0 beta),
    # :212-235 (term API), :297-327 (a runtime matrix accepts extension terms)
    import cutfemx_amd as cfx
    ext = cfx.extensions
    s = _setup(oracle, 2, 16)
    V, cd, O, om = s["V"], s["cd"], s["O"], s["om"]
    agg = ext.create_cell_aggregation(cd, "phi<0", 1.0)
    assert agg.num_pairs == agg.ill_posed_cells.size > 0
    A = ext.extension_penalty_matrix(V, cd, agg, 2.5, 2)
    dense = A.to_dense()
    np.testing.assert_allclose(dense, dense.T, atol=1e-12)
    np.testing.assert_allclose(dense @ np.ones(dense.shape[1]), 0.0, atol=1e-12)
    np.testing.assert_allclose(dense @ om.x[:, 0], 0.0, atol=1e-12)     # P1 extension of a linear function is exact
    assert np.linalg.norm(dense) > 0.0
    # against the oracle restatement
    oagg = O.cell_aggregation(om, om.conn, s["phi"], s["dom"], "phi<0", 1.0)
    oint = [O.Integral(O.INTERIOR_FACET, O.K_EXTENSION_L2, entities=O.extension_pairs(oagg), params=(2.5,), qdegree=2)]
    ip, ix = O.create_sparsity(om, s["oV"], oint)
    assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix)
    assert rel_err(A.data, O.assemble_matrix(om, s["oV"], oint, ip, ix)) < RTOL
    # cellwise beta == scalar beta
    A_dg0 = ext.extension_penalty_matrix(V, cd, agg, np.full(om.ncells, 2.5), 2)
    np.testing.assert_allclose(A_dg0.to_dense(), dense, atol=1e-12)
    # term API == direct call
    term = ext.ExtensionPenaltyTerm(V, 2.5, 2).with_domain(cd, agg)
    np.testing.assert_allclose(ext.extension_penalty_matrix(term, cd, agg).to_dense(), dense, atol=1e-12)
    with pytest.raises(ValueError):
        ext.extension_penalty_matrix(term, cd, agg, beta=1.0)
    with pytest.raises(NotImplementedError):
        ext.ExtensionPenaltyTerm(V, 1.0, 2, product="H1")
    # create + assemble into an existing matrix; a form that carries the term next to the PDE terms
    A2 = ext.assemble_extension_penalty(ext.create_extension_penalty_matrix(V, cd, agg), V, cd, agg, 2.5, 2)
    np.testing.assert_allclose(A2.to_dense(), dense, atol=1e-12)
    inside = cfx.locate_entities(cd, "phi<0")
    vol = cfx.runtime_quadrature(cd, "phi<0", 2)
    pde = [cfx.fem.Integral(cfx.fem.STIFFNESS, cells=inside, rules=vol, qdegree=0)]
    A_ref = cfx.fem.assemble_matrix(cfx.fem.form(pde, V))
    A_all = cfx.fem.assemble_matrix(cfx.fem.form(pde + [ext.extension_penalty_integral(agg, 2.5, 2)], V))
    np.testing.assert_allclose(A_all.to_dense(), A_ref.to_dense() + dense, atol=1e-12)
