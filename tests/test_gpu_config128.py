"""BASELINE configs[1] at its real size: 3-D Poisson, sphere level set on the 128^3
background mesh (12.6 M tets), P1, one MI355X -- the WHOLE domain against the FULL
oracle (no slab): every stage of the path, index results bit-exact, points 1e-14
absolute, weights / normals / CSR values / RHS 1e-12 relative (north_star).
The oracle needs ~10 s and ~3 GB of host memory at this size."""
import numpy as np
import pytest

from helpers import level_set_values, oracle_poisson, profiled, rel_err

pytestmark = pytest.mark.gpu
RTOL = 1e-12
N = 128


@pytest.fixture(scope="module")
def cfg(oracle):
    import cutfemx_amd as cfx
    from cutfemx_amd import poisson
    O = oracle
    om = O.mesh_box(3, N)
    phi = level_set_values(om.x, 3, "sphere")
    ref = oracle_poisson(O, om, phi, order=4)
    mesh = cfx.Mesh.create_box(3, N)
    V = cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(V, phi))
    sysm = poisson.build_forms(V, cd, order=4)
    yield dict(O=O, om=om, phi=phi, ref=ref, mesh=mesh, V=V, cd=cd, sys=sysm)


def test_cfg128_mesh_and_classification(cfg):
    assert cfg["mesh"].num_cells == 6 * N ** 3 == cfg["om"].ncells
    assert np.array_equal(cfg["mesh"].conn, cfg["om"].conn)
    dom = cfg["cd"].domain()
    assert dom.dtype == np.int8 and np.array_equal(dom, cfg["ref"]["domain"])


@pytest.mark.parametrize("sel", ["phi<0", "phi=0", "phi>0", "phi<=0"])
def test_cfg128_located_lists(cfg, sel):
    import cutfemx_amd as cfx
    got = cfx.locate_entities(cfg["cd"], sel)
    assert got.dtype == np.int32
    assert np.array_equal(got, cfg["O"].locate_entities(cfg["ref"]["domain"], sel))


@pytest.mark.parametrize("which", ["vol", "itf"])
def test_cfg128_rules(cfg, which):
    got = cfg["sys"].volume_rules if which == "vol" else cfg["sys"].interface_rules
    want = cfg["ref"][which]
    assert np.array_equal(got.offsets, want.offsets) and got.offsets.dtype == np.int32
    assert np.array_equal(got.parent_map, want.parent_map) and got.parent_map.dtype == np.int32
    assert np.abs(got.points - want.points).max() <= 1e-14
    assert rel_err(got.weights, want.weights) < RTOL
    assert got.weights.min() > 0.0


def test_cfg128_normals(cfg):
    import cutfemx_amd as cfx
    nrm = cfx.normal(cfg["cd"], cfg["sys"].interface_rules)
    assert rel_err(nrm, cfg["ref"]["normals"]) < RTOL


def test_cfg128_ghost_rows(cfg):
    got = cfg["sys"].ghost_facets.rows
    assert got.dtype == np.int32 and np.array_equal(got, cfg["ref"]["ghost"])


def test_cfg128_csr_and_rhs(cfg):
    import cutfemx_amd as cfx
    ref = cfg["ref"]
    A = cfx.fem.create_matrix(cfg["sys"].a)
    assert A.indptr.dtype == np.int64 and np.array_equal(A.indptr, ref["indptr"])
    assert A.indices.dtype == np.int32 and np.array_equal(A.indices, ref["indices"])
    cfx.fem.assemble_matrix(cfg["sys"].a, A=A)
    assert rel_err(A.data, ref["values"]) < RTOL
    # entry-wise, not only against the largest entry: every entry within 1e-12 of its row's scale
    rowmax = np.maximum.reduceat(np.abs(ref["values"]), ref["indptr"][:-1])
    rows = np.repeat(np.arange(A.nrows), np.diff(ref["indptr"]))
    assert np.all(np.abs(A.data - ref["values"]) <= RTOL * rowmax[rows])
    b = cfx.fem.assemble_vector(cfg["sys"].L)
    assert rel_err(b, ref["b"]) < RTOL
    dom = cfx.fem.active_domain(cfg["sys"].a)
    assert np.array_equal(dom.active_cells, ref["active"])
    assert np.array_equal(dom.inactive_dofs, ref["inactive"])
    cfx.fem.deactivate_outside(A, b, dom)
    O = cfg["O"]
    vals, bb = ref["values"].copy(), ref["b"].copy()
    O.deactivate(ref["inactive"], ref["indptr"], ref["indices"], vals, bb)
    assert rel_err(A.data, vals) < RTOL and rel_err(b, bb) < RTOL
    assert np.all(b[ref["inactive"]] == 0.0)


def test_cfg128_takes_the_specialised_kernels(cfg):
    """The kernels the 512^3 numbers are measured on must be the ones that run here: row tiles for the plain rows,
    facet records, the series source kernel, mask-based sparsity (a silent fall back to the generic paths would keep
    every parity test green)."""
    import cutfemx_amd as cfx

    def run():
        A = cfx.fem.create_matrix(cfg["sys"].a)
        cfx.fem.assemble_matrix(cfg["sys"].a, A=A)
        return A, cfx.fem.assemble_vector(cfg["sys"].L)
    (A, b), names = profiled(run)
    assert rel_err(A.data, cfg["ref"]["values"]) < RTOL and rel_err(b, cfg["ref"]["b"]) < RTOL
    import os
    # the series source term on P1 keeps the row-ordered staging (CFX_VEC_BLOCKS=2: by cell block like every other form)
    vec = ("vec_blocks_std", "vec_blocks_rows") if os.environ.get("CFX_VEC_BLOCKS") == "2" else ("vec_tensors_std", "assemble_vec_plain")
    for k in ("assemble_tiles_plain", "assemble_rows_p1", "assemble_rows_cut", "assemble_facets",
              "pattern_plain_write", "pattern_rows") + vec:   # (the plan and its masks are cached)
        assert k in names, (k, sorted(names))
    assert "assemble_rows_plain" not in names and "assemble_rows" not in names    # per-row fallback / unsplit path
