"""Empty / degenerate inputs through the C ABI (GPU): no cut cells, everything
inside, a single cube, level-set values exactly zero at vertices."""
import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu


def build(oracle, tdim, n, phi_fn):
    import cutfemx_amd as cfx
    om = oracle.mesh_box(tdim, n)
    phi = phi_fn(om.x)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    return om, phi, mesh, V, cfx.cut(cfx.Function(V, phi))


@pytest.mark.parametrize("tdim", [2, 3])
def test_level_set_positive_everywhere(oracle, tdim):
    import cutfemx_amd as cfx
    om, phi, mesh, V, cd = build(oracle, tdim, 4, lambda x: np.ones(x.shape[0]))
    assert np.all(cd.domain() == 1)
    for sel in ("phi<0", "phi=0", "phi<=0"):
        assert cfx.locate_entities(cd, sel).size == 0
    assert cfx.locate_entities(cd, "phi>0").size == om.ncells
    for sel in ("phi<0", "phi=0", "phi>0"):
        r = cfx.runtime_quadrature(cd, sel, 2)
        assert r.total_points == 0 and r.num_rules == 0 and np.array_equal(r.offsets, [0])
        assert r.points.shape == (0, tdim) and r.parent_map.size == 0
    assert cfx.ghost_penalty_facets(cd, "phi<0").size == 0
    vol = cfx.runtime_quadrature(cd, "phi<0", 2)
    a = cfx.fem.form([cfx.fem.Integral(cfx.fem.STIFFNESS, cells=np.zeros(0, dtype=np.int32), rules=vol, qdegree=0)], V)
    A = cfx.fem.assemble_matrix(a)
    # only the all-rows diagonal of assembler.h:538-560 is present, all zeros
    assert np.array_equal(A.indptr, np.arange(om.nnodes + 1)) and np.array_equal(A.indices, np.arange(om.nnodes))
    assert np.all(A.data == 0.0)
    L = cfx.fem.form([cfx.fem.Integral(cfx.fem.SOURCE, cells=np.zeros(0, dtype=np.int32), rules=vol,
                                       params=(cfx.fem.F_ONE, 1.0), qdegree=1)], V)
    assert np.all(cfx.fem.assemble_vector(L) == 0.0)
    with pytest.raises(ValueError):      # deactivate.h:155-160: "found no active background cells"
        cfx.fem.active_domain(a)


@pytest.mark.parametrize("tdim", [2, 3])
def test_level_set_negative_everywhere(oracle, tdim):
    import cutfemx_amd as cfx
    om, phi, mesh, V, cd = build(oracle, tdim, 4, lambda x: -np.ones(x.shape[0]))
    inside = cfx.locate_entities(cd, "phi<0")
    assert np.array_equal(inside, np.arange(om.ncells))
    vol = cfx.runtime_quadrature(cd, "phi<0", 2)
    assert vol.num_rules == 0
    a = cfx.fem.form([cfx.fem.Integral(cfx.fem.MASS, cells=inside, rules=vol, qdegree=2)], V)
    A = cfx.fem.assemble_matrix(a)
    assert abs(A.data.sum() - 1.0) < 1e-13     # the whole unit box
    dom = cfx.fem.active_domain(a)
    assert dom.inactive_dofs.size == 0 and np.array_equal(dom.active_cells, np.arange(om.ncells))
    O = oracle
    oV = O.Space(om.conn, om.nnodes, 1)
    oa = [O.Integral(O.CELL, O.K_MASS, entities=inside, qdegree=2)]
    ip, ix = O.create_sparsity(om, oV, oa)
    assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix)
    assert rel_err(A.data, O.assemble_matrix(om, oV, oa, ip, ix)) < 1e-12


def test_single_cube_mesh(oracle):
    import cutfemx_amd as cfx
    from cutfemx_amd import poisson
    from helpers import oracle_poisson
    om, phi, mesh, V, cd = build(oracle, 3, 1, lambda x: np.linalg.norm(x - 0.1, axis=1) - 0.6)
    ref = oracle_poisson(oracle, om, phi)
    assert np.array_equal(cd.domain(), ref["domain"])
    s = poisson.build_forms(V, cd)
    A = cfx.fem.assemble_matrix(s.a)
    assert np.array_equal(A.indptr, ref["indptr"]) and np.array_equal(A.indices, ref["indices"])
    assert rel_err(A.data, ref["values"]) < 1e-12
    assert rel_err(cfx.fem.assemble_vector(s.L), ref["b"]) < 1e-12


@pytest.mark.parametrize("tdim,n", [(2, 8), (3, 4)])
def test_zero_values_at_vertices(oracle, tdim, n):
    # test_cut_api.py:191-208: a zero at a vertex makes the cell "intersected"; the
    # degenerate sub-simplices must have finite, non-negative weights and tile the cells
    import cutfemx_amd as cfx
    om, phi, mesh, V, cd = build(oracle, tdim, n, lambda x: x[:, 0] - 0.5)
    O = oracle
    dom = O.classify(om.conn, phi)
    assert np.array_equal(cd.domain(), dom) and (dom == 0).sum() > 0
    total = 0.0
    for sel in ("phi<0", "phi>0", "phi=0"):
        got = cfx.runtime_quadrature(cd, sel, 2)
        want = O.runtime_quadrature(om, om.conn, phi, dom, sel, 2)
        assert np.array_equal(got.offsets, want.offsets) and np.array_equal(got.parent_map, want.parent_map)
        assert np.all(np.isfinite(got.weights)) and np.all(got.weights >= 0)
        assert np.allclose(got.weights, want.weights, rtol=0, atol=1e-15)
        if sel != "phi=0":
            total += got.weights.sum()
    cut = O.locate_entities(dom, "phi=0")
    assert abs(total - O.full_cell_rules(om, cut, 1).weights.sum()) < 1e-13
    # the interface x = 0.5 is counted once: its measure is 1
    itf = cfx.runtime_quadrature(cd, "phi=0", 2)
    assert abs(itf.weights.sum() - 1.0) < 1e-13


@pytest.mark.parametrize("tdim,n", [(3, 14), (2, 40), (3, 9)])
def test_block_culled_classification_equals_the_cell_loop(oracle, tdim, n, monkeypatch):
    """Classification decides whole blocks of 1024 cells from the sign codes of their distinct vertices (mesh-static
    vertex runs) and goes cell by cell only where the interface passes: the domain array, the located lists and their
    counts must equal the oracle's and the cell-by-cell kernel's (CFX_CLASSIFY_CULL=0) -- on the generated numbering, on a
    mesh whose cells and vertices are shuffled (no runs: every block falls back), with zeros at vertices, with one sign."""
    import cutfemx_amd as cfx
    O = oracle
    om = O.mesh_box(tdim, n)
    rng = np.random.default_rng(12)
    levels = {
        "sphere": np.linalg.norm(om.x[:, :tdim] - 0.47, axis=1) - 0.31,
        "zeros": np.round(8.0 * (om.x[:, 0] - 0.5)) / 8.0 + 0.0 * om.x[:, 1],        # exact zeros on a vertex plane
        "random": rng.standard_normal(om.nnodes),
        "negative": -np.ones(om.nnodes), "positive": np.ones(om.nnodes),
    }
    # a second mesh: the same cells in random order over randomly renumbered vertices
    vperm = rng.permutation(om.nnodes)
    inv = np.empty_like(vperm); inv[vperm] = np.arange(om.nnodes)
    cperm = rng.permutation(om.ncells)
    x2 = om.x[vperm]
    conn2 = np.ascontiguousarray(inv[om.conn[cperm]].astype(np.int32))
    for shuffled in (False, True):
        x, conn = (x2, conn2) if shuffled else (om.x, om.conn)
        for name, phi0 in levels.items():
            phi = phi0[vperm] if shuffled else phi0
            want = O.classify(conn, phi)
            got = {}
            for mode in ("1", "0"):
                monkeypatch.setenv("CFX_CLASSIFY_CULL", mode)
                mesh = cfx.Mesh.from_arrays(tdim, x, conn)
                V = cfx.FunctionSpace(mesh, 1)
                cd = cfx.cut(cfx.Function(V, phi))
                got[mode] = (cd.domain(), cfx.locate_entities(cd, "phi<0"), cfx.locate_entities(cd, "phi=0"))
                # update() with a moved level set goes through the same (already built) summary
                phi2 = phi + 0.03
                cd2 = cfx.cut(cfx.Function(V, phi2))
                assert np.array_equal(cd2.domain(), O.classify(conn, phi2)), (name, shuffled, mode)
            for mode in ("1", "0"):
                assert np.array_equal(got[mode][0], want), (name, shuffled, mode)
                assert np.array_equal(got[mode][1], O.locate_entities(want, "phi<0")), (name, shuffled, mode)
                assert np.array_equal(got[mode][2], O.locate_entities(want, "phi=0")), (name, shuffled, mode)
    monkeypatch.delenv("CFX_CLASSIFY_CULL")
