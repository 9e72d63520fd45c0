"""Randomised edge cases against the oracle: level sets with many exact zeros and tiny values at the vertices
(a value == 0 makes the cell intersected, cut.cpp / docs/user-guide/level-sets.md:84-88; sub-simplices of zero
measure must give zero weights, never NaN), on unstructured meshes.  Integers bit-exact, FP64 to 1e-12."""
import numpy as np
import pytest

from helpers import oracle_poisson, rel_err, scrambled_mesh

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _nasty_level_set(rng, x, tdim):
    base = np.linalg.norm(x[:, :tdim] - 0.5, axis=1) - rng.uniform(0.2, 0.45)
    kind = rng.integers(0, 4, size=base.size)
    phi = base.copy()
    phi[kind == 0] = 0.0                                         # exact zeros on a quarter of the vertices
    phi[kind == 1] *= 1e-13                                      # nearly on the interface
    phi[kind == 2] = np.sign(base[kind == 2]) * rng.uniform(1e-300, 1e-200, size=int((kind == 2).sum()))
    return phi


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("tdim,n", [(2, 7), (3, 4)])
def test_degenerate_level_sets_match_oracle(oracle, tdim, n, seed):
    import cutfemx_amd as cfx
    from cutfemx_amd import poisson
    O = oracle
    rng = np.random.default_rng(100 * tdim + seed)
    om = scrambled_mesh(O, tdim, n, seed=seed) if seed % 2 else O.mesh_box(tdim, n)
    phi = _nasty_level_set(rng, om.x, tdim)
    ref = oracle_poisson(O, om, phi, order=3)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(V, phi))
    assert np.array_equal(cd.domain(), ref["domain"])
    for sel in ("phi<0", "phi=0", "phi>0", "phi<=0", "phi>=0"):
        assert np.array_equal(cfx.locate_entities(cd, sel), O.locate_entities(ref["domain"], sel))
    for sel, want in (("phi<0", ref["vol"]), ("phi=0", ref["itf"]),
                      ("phi>0", O.runtime_quadrature(om, om.conn, phi, ref["domain"], "phi>0", 3))):
        R = cfx.runtime_quadrature(cd, sel, 3)
        assert np.array_equal(R.offsets, want.offsets) and np.array_equal(R.parent_map, want.parent_map)
        assert np.all(np.isfinite(R.weights)) and np.all(R.weights >= 0.0) and np.all(np.isfinite(R.points))
        assert np.max(np.abs(R.weights - want.weights), initial=0.0) <= RTOL * max(np.max(want.weights, initial=0.0), 1e-300)
        assert np.max(np.abs(R.points - want.points), initial=0.0) < 1e-13
    s = poisson.build_forms(V, cd, order=3)
    assert np.array_equal(s.ghost_facets.rows.reshape(-1, 4), ref["ghost"].reshape(-1, 4))
    A = cfx.fem.assemble_matrix(s.a)
    b = cfx.fem.assemble_vector(s.L)
    assert np.array_equal(A.indptr, ref["indptr"]) and np.array_equal(A.indices, ref["indices"])
    assert np.all(np.isfinite(A.data)) and np.all(np.isfinite(b))
    assert rel_err(A.data, ref["values"]) < RTOL and rel_err(b, ref["b"]) < RTOL
    dom = cfx.fem.active_domain(s.a)
    assert np.array_equal(dom.active_cells, ref["active"]) and np.array_equal(dom.inactive_dofs, ref["inactive"])
    # the facets as hosts of the same level set
    orows = O.interior_facets_for_cells(om, np.arange(om.ncells, dtype=np.int32))
    H = O.facet_hosts(om, orows, om.conn)
    fdom = O.facet_classify(H, phi)
    fcd = cfx.cut(cfx.Function(V, phi), orows, tdim - 1)
    assert np.array_equal(fcd.domain(), fdom)
    for sel in ("phi<0", "phi>0", "phi=0"):
        R = cfx.runtime_quadrature(fcd, sel, 2)
        want = O.facet_runtime_quadrature(om, H, phi, fdom, sel, 2)
        assert np.array_equal(R.offsets, want.offsets) and np.array_equal(R.parent_map, want.parent_map)
        assert np.all(np.isfinite(R.weights)) and np.all(R.weights >= 0.0)
        assert np.max(np.abs(R.weights - want.weights), initial=0.0) <= RTOL * max(np.max(want.weights, initial=0.0), 1e-300)
        assert np.max(np.abs(R.points - want.points), initial=0.0) < 1e-13
