"""The specialised kernels must actually run where they are meant to (a silent fall back to a generic path would
keep the parity tests green and hide a performance regression): each case assembles a problem, checks it against the
oracle and reads the engine's per-kernel profile for the kernel names."""
import numpy as np
import pytest

from helpers import level_set_values, oracle_poisson, profiled, rel_err

pytestmark = pytest.mark.gpu


def test_p2_poisson_takes_the_list_and_slot_kernels(oracle, monkeypatch):
    """configs[3] at 12^3 (CFX_ROWS_SPLIT=1: the row split does not wait for the interface rows to be a minority): plain rows copy their static list, the copied rows take the slot-record kernel, the hashed
    rows split by list length, gradient-jump facets as records per quadrature point."""
    import cutfemx_amd as cfx
    from cutfemx_amd import fem, poisson
    monkeypatch.setenv("CFX_ROWS_SPLIT", "1")
    O, n = oracle, 12
    om = O.mesh_box(3, n)
    phi = level_set_values(om.x, 3)
    dofmap, ndofs = cfx.lagrange_dofmap(3, om.conn, om.nnodes, 2)
    ref = oracle_poisson(O, om, phi, degree=2, dofmap=dofmap, ndofs=ndofs)
    mesh = cfx.Mesh.from_arrays(3, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 2, dofmap=dofmap, ndofs=ndofs)
    cd = cfx.cut(cfx.Function(cfx.FunctionSpace(mesh, 1), phi))
    s = poisson.build_forms(V, cd, order=4)

    def run():
        A = fem.create_matrix(s.a)
        fem.assemble_matrix(s.a, A=A)
        return A
    A, names = profiled(run)
    assert np.array_equal(A.indptr, ref["indptr"]) and np.array_equal(A.indices, ref["indices"])
    assert rel_err(A.data, ref["values"]) < 1e-12
    for k in ("pattern_plain_full", "pattern_plain_write", "pattern_rows_short", "pattern_rows_wide",
              "assemble_rows_p2_plain", "assemble_rows_cut", "assemble_facets"):
        assert k in names, (k, sorted(names))
    assert "assemble_cells_std" not in names           # no staged uncut tensors: closed-form stiffness rows
    # a second assembly into the same matrix accumulates (the copied rows are only stored on a fresh zero)
    fem.assemble_matrix(s.a, A=A)
    assert rel_err(A.data, 2.0 * ref["values"]) < 1e-12
    # Dirichlet markers: the slot-record kernel steps aside, the result is the oracle's
    rng = np.random.default_rng(3)
    bc = (rng.random(ndofs) < 0.05).astype(np.int8)
    want = O.assemble_matrix(om, ref["V"], ref["a"], ref["indptr"], ref["indices"], bc0=bc, bc1=bc)
    B, names_bc = profiled(lambda: fem.assemble_matrix(s.a, bcs=bc))
    assert rel_err(B.data, want) < 1e-12
    assert "assemble_rows_p2_plain" not in names_bc


def test_vector_p2_elasticity_takes_the_closed_form_mfma_and_record_kernels(oracle, monkeypatch):
    """configs[4] at 12^3.  Default: closed-form block rows for the uncut cells (no staged 30 x 30 tensors:
    `elasticity_tensors_mfma` must NOT run), 30 x 30 tensors of the CUT cells on the FP64 matrix cores, vector ghost
    penalty from the facet records.  CFX_P2_CLOSED=0: the staged path (MFMA tensors for every cell + the block gather)."""
    import cutfemx_amd as cfx
    from cutfemx_amd import fem
    from test_gpu_spaces import elasticity_problem, setup
    s = setup(oracle, 3, 12, 2, 3)
    O, om, oV = s["O"], s["om"], s["oV"]
    inside, oa, ga = elasticity_problem(s, 2)
    ip, ix = O.create_sparsity(om, oV, oa)
    want = O.assemble_matrix(om, oV, oa, ip, ix)
    a = fem.form(ga, s["V"])
    A, names = profiled(lambda: fem.assemble_matrix(a))
    assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix)
    assert rel_err(A.data, want) < 1e-12
    for k in ("assemble_rows_block_p2", "elasticity_tensors_mfma_cut", "assemble_rows_block", "assemble_facets",
              "pattern_plain_full"):
        assert k in names, (k, sorted(names))
    assert "elasticity_tensors_mfma" not in names and "assemble_rows_block_plain" not in names, sorted(names)
    # accumulate on top of what is there (not a fresh matrix): twice the values
    fem.assemble_matrix(a, A=A)
    assert rel_err(A.data, 2.0 * want) < 1e-12
    monkeypatch.setenv("CFX_P2_CLOSED", "0")
    a2 = fem.form(ga, s["V"])
    A2, names2 = profiled(lambda: fem.assemble_matrix(a2))
    assert rel_err(A2.data, want) < 1e-12
    for k in ("elasticity_tensors_mfma", "assemble_rows_block_plain"):
        assert k in names2, (k, sorted(names2))


def test_static_table_bytes_and_form_prepare(oracle):
    """cfx_space_static_bytes reports what the first assembly built (the bench's `setup` record), cfx_form_prepare
    builds a form's derived tables ahead of the first assembly call and changes no result."""
    import cutfemx_amd as cfx
    from cutfemx_amd import fem, poisson
    O, n = oracle, 12
    om = O.mesh_box(3, n)
    phi = level_set_values(om.x, 3)
    ref = oracle_poisson(O, om, phi)
    mesh = cfx.Mesh.create_box(3, n)
    V = cfx.FunctionSpace(mesh, 1)
    before = V.static_table_bytes()
    assert set(before) == {"dof_cells", "row_stencil", "row_tiles", "cell_neighbours"} and before["row_stencil"] == 0
    cd = cfx.cut(cfx.Function(V, phi))
    s = poisson.build_forms(V, cd, order=4)
    s.a.prepare()
    s.L.prepare()
    after = V.static_table_bytes()
    assert after["dof_cells"] > 0 and after["row_stencil"] > 0 and after["row_tiles"] > 0 and after["cell_neighbours"] > 0
    A = fem.assemble_matrix(s.a)
    b = fem.assemble_vector(s.L)
    assert np.array_equal(A.indices, ref["indices"]) and rel_err(A.data, ref["values"]) < 1e-12 and rel_err(b, ref["b"]) < 1e-12
    assert V.static_table_bytes() == after          # nothing mesh-static is built twice
