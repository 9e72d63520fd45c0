"""Bulk rows (round 5): where a form's uncut entities are the located list of a cut whose level set lives on the space's
own dofmap (P1 on the geometry dofmap), the row plan takes the cell marks from the classification bytes and the row marks
from the vertices' sign codes + the cut's touch bytes instead of walking the 10^8-entry list
(cfx_rowasm.hip: row_class_kernel, cellmark_from_domain_kernel, mix_rowmark_kernel).  The reference marks cell by cell
(cpp/cutfemx/fem/deactivate.h:103-183, cpp/dolfinx_custom_data/fem/assembler.h:442-560): every result must be the one the
list walk gives (CFX_BULK_ROWS=0) and the oracle's."""
import numpy as np
import pytest

from helpers import level_set_values, oracle_poisson, profiled, rel_err, scrambled_mesh

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def bulk_expected() -> bool:
    """The diagnostic modes that switch the row stencil (or the bulk rows themselves) off take the list walk: same results."""
    import os
    return not (os.environ.get("CFX_STENCIL") == "0" or os.environ.get("CFX_BULK_ROWS") == "0")


def poisson_system(cfx, V, cd, order=4):
    from cutfemx_amd import poisson
    s = poisson.build_forms(V, cd, order=order)
    A = cfx.fem.create_matrix(s.a)
    cfx.fem.assemble_matrix(s.a, A=A)
    b = cfx.fem.assemble_vector(s.L)
    dom = cfx.fem.deactivate_outside(A, b, cfx.fem.active_domain(s.a))
    return s, A, b, dom


def against_oracle(O, om, phi, A, b, dom):
    ref = oracle_poisson(O, om, phi)
    vals, bb = ref["values"].copy(), ref["b"].copy()
    O.deactivate(ref["inactive"], ref["indptr"], ref["indices"], vals, bb)
    assert np.array_equal(A.indptr, ref["indptr"]) and np.array_equal(A.indices, ref["indices"])
    assert rel_err(A.data, vals) < RTOL and rel_err(b, bb) < RTOL
    assert np.array_equal(dom.inactive_dofs, ref["inactive"]) and np.array_equal(dom.active_cells, ref["active"])


@pytest.mark.parametrize("tdim,n", [(3, 14), (2, 40)])
def test_the_bulk_path_is_taken_and_equals_the_list_walk_and_the_oracle(oracle, tdim, n, monkeypatch):
    import cutfemx_amd as cfx
    om = oracle.mesh_box(tdim, n)
    phi = level_set_values(om.x, tdim)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    monkeypatch.setenv("CFX_DETERMINISTIC", "1")
    cd = cfx.cut(cfx.Function(V, phi))
    (s, A, b, dom), names = profiled(lambda: poisson_system(cfx, V, cd))
    import os
    switched_off = not bulk_expected()
    if not switched_off:
        for k in ("plan_bulk_init", "plan_mix_rows"):
            assert k in names, (k, sorted(names))
        assert "plan_mark_entities" not in names
    against_oracle(oracle, om, phi, A, b, dom)
    # the list walk: same pattern, same values bit for bit (deterministic mode: fixed summation orders)
    monkeypatch.setenv("CFX_BULK_ROWS", "0")
    cd2 = cfx.cut(cfx.Function(V, phi))
    (s2, A2, b2, dom2), names2 = profiled(lambda: poisson_system(cfx, V, cd2))
    assert "plan_bulk_init" not in names2 and ("plan_mark_entities" in names2 or switched_off)
    assert np.array_equal(A.indptr, A2.indptr) and np.array_equal(A.indices, A2.indices)
    if os.environ.get("CFX_ASSEMBLY") == "atomic":      # (FP64 atomics: the order of the sums is the schedule's)
        assert rel_err(A.data, A2.data) < 1e-13 and rel_err(b, b2) < 1e-13
    else:
        assert np.array_equal(A.data, A2.data) and np.array_equal(b, b2)
    assert np.array_equal(dom.inactive_dofs, dom2.inactive_dofs)


def test_a_host_copy_of_the_list_has_no_provenance_and_takes_the_list_walk(oracle):
    import cutfemx_amd as cfx
    from cutfemx_amd import fem
    om = oracle.mesh_box(3, 10)
    phi = level_set_values(om.x, 3)
    mesh = cfx.Mesh.from_arrays(3, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(V, phi))
    inside_dev = cfx.locate_entities_device(cd, "phi<0")
    inside_host = cfx.locate_entities(cd, "phi<0")              # a numpy copy: the engine cannot know where it came from
    vol = cfx.runtime_quadrature(cd, "phi<0", 2)
    out = {}
    for tag, cells in (("dev", inside_dev), ("host", inside_host), ("prefix", inside_host[: inside_host.size // 2])):
        a = fem.form([fem.Integral(fem.STIFFNESS, cells=cells, rules=vol, qdegree=0)], V)
        A, names = profiled(lambda: fem.assemble_matrix(a))
        out[tag] = (A, names)
    assert ("plan_bulk_init" in out["dev"][1] or not bulk_expected()) and "plan_bulk_init" not in out["host"][1]
    assert "plan_bulk_init" not in out["prefix"][1]
    assert np.array_equal(out["dev"][0].indices, out["host"][0].indices)
    assert rel_err(out["dev"][0].data, out["host"][0].data) < 1e-14
    # half of the list is another form: fewer entries
    assert out["prefix"][0].nnz < out["host"][0].nnz


@pytest.mark.parametrize("tdim,n", [(3, 9), (2, 21)])
def test_zeros_at_vertices_only_inside_cells_and_scrambled_numbering(oracle, tdim, n):
    """(i) level-set values that vanish exactly at vertices (every cell around such a vertex is cut); (ii) a form of the
    inside cells alone: no rule integral marks the rows of the cut cells, the mark gather decides them; (iii) a mesh
    without any locality in its numbering (the culled classification falls back to the cell loop everywhere)."""
    import cutfemx_amd as cfx
    from cutfemx_amd import fem
    O = oracle
    om = O.mesh_box(tdim, n)
    x = om.x
    phi_plane = np.round((x[:, 0] - 1.0 / 3.0) * n) / n          # exact zeros on the vertex plane x = 1/3
    assert np.count_nonzero(phi_plane == 0.0) > n
    meshes = [(om, phi_plane, "zeros"), (om, level_set_values(om.x, tdim), "inside-only")]
    sm = scrambled_mesh(O, tdim, min(n, 8))
    meshes.append((sm, level_set_values(sm.x, tdim), "scrambled"))
    for m, phi, tag in meshes:
        mesh = cfx.Mesh.from_arrays(tdim, m.x, m.conn)
        V = cfx.FunctionSpace(mesh, 1)
        cd = cfx.cut(cfx.Function(V, phi))
        d = O.classify(m.conn, phi)
        assert np.array_equal(cd.domain(), d), tag
        if tag == "inside-only":
            inside = cfx.locate_entities_device(cd, "phi<0")
            a = fem.form([fem.Integral(fem.MASS, cells=inside, qdegree=2)], V)
            (A, dom), names = profiled(lambda: (fem.assemble_matrix(a), fem.active_domain(a)))
            assert "plan_bulk_init" in names or not bulk_expected()
            oV = O.Space(m.conn, m.nnodes, 1)
            oi = [O.Integral(O.CELL, O.K_MASS, entities=O.locate_entities(d, "phi<0"), qdegree=2)]
            ip, ix = O.create_sparsity(m, oV, oi)
            assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix)
            assert rel_err(A.data, O.assemble_matrix(m, oV, oi, ip, ix)) < RTOL
            assert np.array_equal(dom.inactive_dofs, O.inactive_dofs(oV, O.active_cells(oi, m.ncells)))
        else:
            s, A, b, dom = poisson_system(cfx, V, cd)
            against_oracle(O, m, phi, A, b, dom)


def test_facets_from_another_source_mark_rows_inside_the_bulk(oracle):
    """A facet integral over facets that have nothing to do with the cut (the interior facets of a few cells deep inside
    the domain): their rows are 'special' although every cell around them is an uncut entity."""
    import cutfemx_amd as cfx
    from cutfemx_amd import fem
    O = oracle
    tdim, n = 3, 10
    om = O.mesh_box(tdim, n)
    phi = level_set_values(om.x, tdim)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(V, phi))
    d = O.classify(om.conn, phi)
    inside = O.locate_entities(d, "phi<0")
    xc = om.x[om.conn].mean(axis=1)
    deep = inside[np.linalg.norm(xc[inside] - np.array([0.47, 0.43, 0.41]), axis=1) < 0.12].astype(np.int32)
    assert deep.size > 20
    facets = cfx.interior_facets_for_cells(mesh, deep)
    inside_dev = cfx.locate_entities_device(cd, "phi<0")
    vol = cfx.runtime_quadrature(cd, "phi<0", 2)
    a = fem.form([fem.Integral(fem.STIFFNESS, cells=inside_dev, rules=vol, qdegree=0),
                  fem.Integral(fem.GHOST_GRADJUMP, facets=facets, params=(0.1,), qdegree=0)], V)
    A, names = profiled(lambda: fem.assemble_matrix(a))
    assert "plan_bulk_init" in names or not bulk_expected()
    oV = O.Space(om.conn, om.nnodes, 1)
    ovol = O.runtime_quadrature(om, om.conn, phi, d, "phi<0", 2)
    oi = [O.Integral(O.CELL, O.K_STIFFNESS, entities=inside, rules=ovol, qdegree=0),
          O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=facets.rows, params=(0.1,), qdegree=0)]
    ip, ix = O.create_sparsity(om, oV, oi)
    assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix)
    assert rel_err(A.data, O.assemble_matrix(om, oV, oi, ip, ix)) < RTOL


@pytest.mark.parametrize("tdim,n,bs", [(3, 7, 1), (2, 14, 1), (3, 5, 3), (2, 10, 2)])
def test_bulk_rows_of_degree_two_spaces(oracle, tdim, n, bs, monkeypatch):
    """Degree 2 (scalar and vector): a dof sits between two mesh vertices (cfx_space_s::dof_verts); with an end vertex on
    the entities' side that no cut cell touches, every cell around the dof is an entity.  Same pattern, values and
    inactive dofs as the list walk and as the oracle; the forms carry a ghost penalty so that rows next to the interface
    are special, and a second form has the inside cells alone (the mark gather decides the rows of the cut cells)."""
    import cutfemx_amd as cfx
    from cutfemx_amd import fem
    O = oracle
    om = O.mesh_box(tdim, n)
    phi = level_set_values(om.x, tdim)
    dofmap, ndofs = cfx.lagrange_dofmap(tdim, om.conn, om.nnodes, 2)
    oV = O.Space(dofmap, ndofs, 2, bs)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 2, dofmap=dofmap, ndofs=ndofs, bs=bs)
    Vphi = cfx.FunctionSpace(mesh, 1)
    kern_g, kern_o, params = (fem.ELASTICITY, O.K_ELASTICITY, (1.0, 0.3)) if bs > 1 else (fem.STIFFNESS, O.K_STIFFNESS, ())
    d = O.classify(om.conn, phi)
    o_in = O.locate_entities(d, "phi<0")
    o_vol = O.runtime_quadrature(om, om.conn, phi, d, "phi<0", 2)
    o_ghost = O.ghost_penalty_facets(om, d, "phi<0")

    def build(with_rules):
        cd = cfx.cut(cfx.Function(Vphi, phi))
        inside = cfx.locate_entities_device(cd, "phi<0")
        ints = [fem.Integral(kern_g, cells=inside, params=params, qdegree=2,
                             **(dict(rules=cfx.runtime_quadrature(cd, "phi<0", 2)) if with_rules else {}))]
        if with_rules:
            ints.append(fem.Integral(fem.GHOST_GRADJUMP, facets=cfx.ghost_penalty_facets(cd, "phi<0"), params=(0.1,), qdegree=2))
        a = fem.form(ints, V)
        (A, dom), names = profiled(lambda: (fem.assemble_matrix(a), fem.active_domain(a)))
        return A, dom, names, cd
    for with_rules in (True, False):
        o_ints = [O.Integral(O.CELL, kern_o, entities=o_in, params=params, qdegree=2, **(dict(rules=o_vol) if with_rules else {}))]
        if with_rules:
            o_ints.append(O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=o_ghost, params=(0.1,), qdegree=2))
        ip, ix = O.create_sparsity(om, oV, o_ints)
        want = O.assemble_matrix(om, oV, o_ints, ip, ix)
        ina = O.inactive_dofs(oV, O.active_cells(o_ints, om.ncells))
        A, dom, names, _keep = build(with_rules)
        if bulk_expected():
            assert "plan_bulk_init" in names, sorted(names)     # (plan_mark_cells still marks the rule cells)
        assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix), with_rules
        assert rel_err(A.data, want) < RTOL and np.array_equal(dom.inactive_dofs, ina)
        monkeypatch.setenv("CFX_BULK_ROWS", "0")
        A2, dom2, names2, _keep2 = build(with_rules)
        monkeypatch.delenv("CFX_BULK_ROWS")
        assert "plan_bulk_init" not in names2
        assert np.array_equal(A2.indptr, ip) and np.array_equal(A2.indices, ix) and rel_err(A2.data, want) < RTOL
        assert np.array_equal(dom2.inactive_dofs, ina)
