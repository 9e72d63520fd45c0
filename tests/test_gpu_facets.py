"""GPU parity for facet hosts: cut(level_set, facets, tdim - 1) -- classification, selector
answers in the caller's facet numbering, runtime quadrature on the facets' reference simplices, physical
points, the cell views of the rules (ds terms as dx-type integrals), dS integrals over [facets, rules], and the
functionals of python/tests/test_cut_api.py:499-605.  Integers bit-exact, FP64 to 1e-12."""
import numpy as np
import pytest

from helpers import level_set_values, rel_err, scrambled_mesh

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _setup(oracle, tdim, n, kind="plane", scrambled=False):
    import cutfemx_amd as cfx
    O = oracle
    om = scrambled_mesh(O, tdim, n) if scrambled else O.mesh_box(tdim, n)
    phi = level_set_values(om.x, tdim, kind) if kind != "oblique" else om.x[:, 0] - 0.51 + 0.07 * om.x[:, 1]
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    f = cfx.Function(V, phi)
    return O, om, phi, mesh, V, f


def _rules_equal(R, oR, tdim_pts):
    assert R.tdim == oR.tdim == tdim_pts
    assert np.array_equal(R.offsets, oR.offsets) and np.array_equal(R.parent_map, oR.parent_map)
    assert rel_err(R.weights, oR.weights) < RTOL
    assert np.max(np.abs(R.points - oR.points), initial=0.0) < 1e-13


@pytest.mark.parametrize("tdim,n,kind,scr", [(2, 9, "plane", False), (2, 16, "sphere", False), (3, 5, "oblique", False),
                                             (3, 6, "sphere", True), (2, 11, "oblique", True)])
@pytest.mark.parametrize("which", ["exterior", "interior"])
def test_facet_hosts_cut_parity(oracle, tdim, n, kind, scr, which):
    import cutfemx_amd as cfx
    O, om, phi, mesh, V, f = _setup(oracle, tdim, n, kind, scr)
    if which == "exterior":
        rows_g = cfx.exterior_facets(mesh)
        orows = O.exterior_facets(om)
    else:
        rows_g = cfx.interior_facets_for_cells(mesh, np.arange(om.ncells, dtype=np.int32))
        orows = O.interior_facets_for_cells(om, np.arange(om.ncells, dtype=np.int32))
    assert np.array_equal(rows_g.rows, orows)
    ids = (7 + 3 * np.arange(orows.shape[0])).astype(np.int32)
    H = O.facet_hosts(om, orows, om.conn, facet_ids=ids)
    dom = O.facet_classify(H, phi)
    cd = cfx.cut(f, rows_g, tdim - 1, facet_ids=ids)
    assert cd.tdim == tdim - 1 and cd.num_local_cells == orows.shape[0] and cd.entity_dim == tdim - 1
    assert np.array_equal(cd.domain(), dom)
    for sel in ("phi<0", "phi=0", "phi>0", "phi<=0"):
        assert np.array_equal(cfx.locate_entities(cd, sel), O.facet_locate_entities(H, dom, sel))
    for sel, order in (("phi<0", 2), ("phi>0", 4), ("phi<0", 0), ("phi=0", 3)):
        R = cfx.runtime_quadrature(cd, sel, order)
        oR = O.facet_runtime_quadrature(om, H, phi, dom, sel, order)
        _rules_equal(R, oR, tdim - 1)
        assert R.host_width == orows.shape[1] and np.array_equal(R.host_rows, oR.host_rows)
        assert np.max(np.abs(R.physical_points.T - O.facet_physical_points(om, oR)), initial=0.0) < 1e-13
        for side in range(orows.shape[1] // 2):
            Rc, oRc = R.to_cells(side), O.facet_rules_to_cells(om, oR, side)
            _rules_equal(Rc, oRc, tdim)
            assert Rc.host_width == 0
    for sel in ("phi<0", None):
        W = cfx.full_facet_rules(cd, sel, 3)
        oW = O.facet_runtime_quadrature(om, H, phi, dom, sel, 3, whole=True)
        _rules_equal(W, oW, tdim - 1)
    with pytest.raises(ValueError):
        cfx.runtime_quadrature(cd, "psi<0", 2)       # unknown level-set name
    with pytest.raises(ValueError):
        cfx.ghost_penalty_facets(cd, "phi<0")        # needs cell hosts
    # the level set moves: update() re-classifies the hosts (cut.cpp:845-868)
    moved = phi + 0.013
    f.values[:] = moved
    cfx.update(cd)
    assert np.array_equal(cd.domain(), O.facet_classify(H, moved))


@pytest.mark.parametrize("tdim,n", [(2, 8), (3, 4)])
def test_facet_hosts_entity_geometry_and_errors(oracle, tdim, n):
    import cutfemx_amd as cfx
    O, om, phi, mesh, V, f = _setup(oracle, tdim, n, "oblique")
    orows = O.exterior_facets(om)
    H0 = O.facet_hosts(om, orows, om.conn)
    geom = H0.verts[:, ::-1].copy()                   # the caller's vertex order (entities_to_geometry)
    H = O.facet_hosts(om, orows, om.conn, entity_geometry=geom)
    dom = O.facet_classify(H, phi)
    cd = cfx.cut(f, orows, tdim - 1, entity_geometry=geom)
    R = cfx.runtime_quadrature(cd, "phi<0", 3)
    _rules_equal(R, O.facet_runtime_quadrature(om, H, phi, dom, "phi<0", 3), tdim - 1)
    bad = geom.copy()
    bad[0, 0] = om.conn[orows[0, 0], orows[0, 1]]     # the vertex opposite the facet is not on it
    with pytest.raises(ValueError):
        cfx.cut(f, orows, tdim - 1, entity_geometry=bad)
    far = orows.copy()
    far[1, 0] = om.ncells
    with pytest.raises(IndexError):
        cfx.cut(f, far, tdim - 1)
    irows = O.interior_facets_for_cells(om, np.arange(om.ncells, dtype=np.int32)).copy()
    irows[0, 3] = (irows[0, 3] + 1) % (tdim + 1)       # cell1's facet no longer matches cell0's
    with pytest.raises(ValueError):
        cfx.cut(f, irows, tdim - 1)
    with pytest.raises(ValueError):
        cfx.cut(f, orows, 0)


def test_facet_functionals_line_3x3(oracle):
    # test_cut_api.py:499-605: int 1 ds / dS over phi<0, runtime-only and mixed [standard facets, rules]
    import cutfemx_amd as cfx
    O, om, phi, mesh, V, f = _setup(oracle, 2, 3, "plane")
    one = (cfx.fem.F_ONE, 1.0)
    ext = cfx.exterior_facets(mesh)
    cd = cfx.cut(f, ext, 1)
    run = cfx.runtime_quadrature(cd, "phi<0", 2)
    std = cfx.full_facet_rules(cd, "phi<0", 2)
    v_run = cfx.fem.assemble_scalar(cfx.fem.form([cfx.fem.Integral(cfx.fem.SOURCE, rules=run.to_cells(), params=one)], V))
    v_std = cfx.fem.assemble_scalar(cfx.fem.form([cfx.fem.Integral(cfx.fem.SOURCE, rules=std.to_cells(), params=one)], V))
    v_mix = cfx.fem.assemble_scalar(cfx.fem.form([cfx.fem.Integral(cfx.fem.SOURCE, rules=std.to_cells(), params=one),
                                                  cfx.fem.Integral(cfx.fem.SOURCE, rules=run.to_cells(), params=one)], V))
    assert abs(v_run - 2 * (0.51 - 1 / 3)) < 1e-14 and abs(v_std - (1 + 2 / 3)) < 1e-14
    assert abs(v_mix - (v_std + v_run)) < 1e-14
    inter = cfx.interior_facets_for_cells(mesh, np.arange(om.ncells, dtype=np.int32))
    cd = cfx.cut(f, inter, 1)
    run = cfx.runtime_quadrature(cd, "phi<0", 2)
    std = cfx.full_facet_rules(cd, "phi<0", 2)
    vals = [cfx.fem.assemble_scalar(cfx.fem.form([cfx.fem.Integral(cfx.fem.SOURCE, rules=r.to_cells(side), params=one)], V))
            for r in (run, std) for side in (0, 1)]
    assert abs(vals[0] - vals[1]) < 1e-14 and abs(vals[2] - vals[3]) < 1e-14
    # the Kuhn split has the other diagonal than create_rectangle: same closed form
    assert abs(vals[0] + vals[2] - (1 + 2 * 0.51 + 3 * 0.51 * np.sqrt(2.0))) < 1e-13


@pytest.mark.parametrize("tdim,n,degree,scr", [(2, 10, 1, False), (3, 5, 1, False), (2, 8, 2, False), (3, 4, 2, False),
                                               (3, 5, 1, True)])
def test_exterior_facet_terms_as_cell_rules(oracle, tdim, n, degree, scr):
    """Robin / Neumann boundary terms on the wet part of the boundary: int_{ds, phi<0} u v and f v, mixed measure
    [uncut boundary facets, rules of the cut ones], added to a cut stiffness form so that sparsity, row-gather
    assembly and deactivation all see cells with several rules."""
    import cutfemx_amd as cfx
    O, om, phi, mesh, V1, f = _setup(oracle, tdim, n, "oblique", scr)
    dofmap, ndofs = cfx.lagrange_dofmap(tdim, om.conn, om.nnodes, degree)
    oV = O.Space(dofmap, ndofs, degree)
    V = V1 if degree == 1 else cfx.FunctionSpace(mesh, degree, dofmap=dofmap, ndofs=ndofs)
    # volume part
    dom = O.classify(om.conn, phi)
    inside = O.locate_entities(dom, "phi<0")
    ovol = O.runtime_quadrature(om, om.conn, phi, dom, "phi<0", 2 * degree)
    cdc = cfx.cut(f)
    vol = cfx.runtime_quadrature(cdc, "phi<0", 2 * degree)
    # boundary part
    orows = O.exterior_facets(om)
    H = O.facet_hosts(om, orows, om.conn)
    fdom = O.facet_classify(H, phi)
    ostd = O.facet_rules_to_cells(om, O.facet_runtime_quadrature(om, H, phi, fdom, "phi<0", 2 * degree, whole=True))
    orun = O.facet_rules_to_cells(om, O.facet_runtime_quadrature(om, H, phi, fdom, "phi<0", 2 * degree))
    cdf = cfx.cut(f, cfx.exterior_facets(mesh), tdim - 1)
    std = cfx.full_facet_rules(cdf, "phi<0", 2 * degree).to_cells()
    run = cfx.runtime_quadrature(cdf, "phi<0", 2 * degree).to_cells()
    oa = [O.Integral(O.CELL, O.K_STIFFNESS, entities=inside, rules=ovol, qdegree=2 * (degree - 1)),
          O.Integral(O.CELL, O.K_MASS, rules=ostd, params=()), O.Integral(O.CELL, O.K_MASS, rules=orun)]
    ga = [cfx.fem.Integral(cfx.fem.STIFFNESS, cells=inside, rules=vol, qdegree=2 * (degree - 1)),
          cfx.fem.Integral(cfx.fem.MASS, rules=std), cfx.fem.Integral(cfx.fem.MASS, rules=run)]
    ip, ix = O.create_sparsity(om, oV, oa)
    want = O.assemble_matrix(om, oV, oa, ip, ix)
    a = cfx.fem.form(ga, V)
    A = cfx.fem.assemble_matrix(a)
    assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix)
    assert rel_err(A.data, want) < RTOL
    # Neumann datum g = 2 (analytic id) and g = a Function of the space (sin-products vanish on the box boundary)
    w = np.random.default_rng(3).uniform(0.5, 1.5, ndofs)
    for params, co in (((O.F_ONE, 2.0), None), ((O.F_COEFFICIENT, 0.5), w)):
        oL = [O.Integral(O.CELL, O.L_SOURCE, rules=ostd, params=params, coefficient=co),
              O.Integral(O.CELL, O.L_SOURCE, rules=orun, params=params, coefficient=co)]
        gL = [cfx.fem.Integral(cfx.fem.SOURCE, rules=std, params=params, coefficient=co),
              cfx.fem.Integral(cfx.fem.SOURCE, rules=run, params=params, coefficient=co)]
        b = cfx.fem.assemble_vector(cfx.fem.form(gL, V))
        assert rel_err(b, O.assemble_vector(om, oV, oL)) < RTOL


@pytest.mark.parametrize("tdim,n,degree,scr", [(2, 10, 1, False), (3, 5, 1, False), (2, 7, 2, False), (3, 4, 2, False),
                                               (2, 9, 1, True), (3, 4, 1, True)])
@pytest.mark.parametrize("kernel", ["GHOST_GRADJUMP", "JUMP"])
def test_interior_facet_integrals_over_facet_rules(oracle, tdim, n, degree, scr, kernel):
    """dS(subdomain_data=[standard facets, rules]): skeleton penalties on the wet part of every interior facet,
    next to a cut stiffness form (sparsity + assembly + deactivation parity, local tensors of cut facets)."""
    import cutfemx_amd as cfx
    O, om, phi, mesh, V1, f = _setup(oracle, tdim, n, "oblique", scr)
    dofmap, ndofs = cfx.lagrange_dofmap(tdim, om.conn, om.nnodes, degree)
    if kernel == "JUMP":
        # the value jump of a continuous space vanishes: the DG space of the same degree (every cell owns its dofs)
        ndofs = om.ncells * dofmap.shape[1]
        dofmap = np.arange(ndofs, dtype=np.int32).reshape(om.ncells, -1)
    oV = O.Space(dofmap, ndofs, degree)
    V = V1 if (degree == 1 and kernel != "JUMP") else cfx.FunctionSpace(mesh, degree, dofmap=dofmap, ndofs=ndofs)
    dom = O.classify(om.conn, phi)
    inside = O.locate_entities(dom, "phi<0")
    ovol = O.runtime_quadrature(om, om.conn, phi, dom, "phi<0", 2 * degree)
    cdc = cfx.cut(f)
    vol = cfx.runtime_quadrature(cdc, "phi<0", 2 * degree)
    orows = O.interior_facets_for_cells(om, np.arange(om.ncells, dtype=np.int32))
    H = O.facet_hosts(om, orows, om.conn)
    fdom = O.facet_classify(H, phi)
    ostd_rows = orows[O.facet_locate_entities(H, fdom, "phi<0")]
    orun = O.facet_runtime_quadrature(om, H, phi, fdom, "phi<0", 2 * degree)
    cdf = cfx.cut(f, orows, tdim - 1)
    std_rows = orows[cfx.locate_entities(cdf, "phi<0")]
    assert np.array_equal(std_rows, ostd_rows)
    run = cfx.runtime_quadrature(cdf, "phi<0", 2 * degree)
    ok, gk = getattr(O, "K_" + kernel), getattr(cfx.fem, kernel)
    oa = [O.Integral(O.CELL, O.K_STIFFNESS, entities=inside, rules=ovol, qdegree=2 * (degree - 1)),
          O.Integral(O.INTERIOR_FACET, ok, entities=ostd_rows, rules=orun, params=(0.3,), qdegree=2 * degree)]
    ga = [cfx.fem.Integral(cfx.fem.STIFFNESS, cells=inside, rules=vol, qdegree=2 * (degree - 1)),
          cfx.fem.Integral(gk, facets=std_rows, rules=run, params=(0.3,), qdegree=2 * degree)]
    ip, ix = O.create_sparsity(om, oV, oa)
    want = O.assemble_matrix(om, oV, oa, ip, ix)
    a = cfx.fem.form(ga, V)
    A = cfx.fem.assemble_matrix(a)
    assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix)
    assert rel_err(A.data, want) < RTOL
    # local tensors of a few cut facets (entity index = n_std + rule)
    n_std = ostd_rows.shape[0]
    for r in range(0, orun.parent_map.size, max(1, orun.parent_map.size // 5)):
        Ae = cfx.fem.tabulate_entity(a, 1, n_std + r, False)
        assert rel_err(Ae, O.tabulate_entity(om, oV, oa[1], n_std + r, False)) < RTOL
    # deactivation sees the facet cells (deactivate.h:387-418)
    dom_act = cfx.fem.active_domain(a)
    act = O.active_cells(oa, om.ncells)
    assert np.array_equal(dom_act.inactive_dofs, O.inactive_dofs(oV, act))
    # runtime rules only (dS(subdomain_data=rules))
    oa2 = [O.Integral(O.INTERIOR_FACET, ok, rules=orun, params=(0.3,), qdegree=2 * degree)]
    ga2 = [cfx.fem.Integral(gk, rules=run, params=(0.3,), qdegree=2 * degree)]
    ip, ix = O.create_sparsity(om, oV, oa2)
    a2 = cfx.fem.form(ga2, V)
    A2 = cfx.fem.assemble_matrix(a2)
    assert np.array_equal(A2.indptr, ip) and np.array_equal(A2.indices, ix)
    assert rel_err(A2.data, O.assemble_matrix(om, oV, oa2, ip, ix)) < RTOL
    # a form without cell integrals: its active cells are the facets' cells alone
    act2 = cfx.fem.active_domain(a2)
    want2 = O.active_cells(oa2, om.ncells)
    assert np.array_equal(act2.active_cells, want2) and np.array_equal(act2.inactive_dofs, O.inactive_dofs(oV, want2))


@pytest.mark.parametrize("tdim,n,degree", [(2, 12, 1), (3, 5, 1), (2, 8, 2), (3, 4, 2)])
def test_dg_poisson_system_parity(oracle, tdim, n, degree):
    """The whole cut DG Poisson system (demo_dg_poisson.py) against the oracle: facet lists, rules, sparsity,
    matrix, vector, deactivation."""
    import cutfemx_amd as cfx
    from helpers import oracle_dg_poisson
    O = oracle
    om = O.mesh_box(tdim, n)
    phi = level_set_values(om.x, tdim)
    o = oracle_dg_poisson(O, om, phi, degree=degree)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    f = cfx.Function(cfx.FunctionSpace(mesh, 1), phi)
    from cutfemx_amd import poisson
    g = poisson.build_dg_forms(f, degree)
    a, L = g.a, g.L
    assert np.array_equal(g.skeleton.rows, o["skeleton"]) and np.array_equal(g.omega_facets, o["omega_facets"])
    assert np.array_equal(g.facet_rules.host_rows, o["facet_rules"].host_rows)
    assert rel_err(g.facet_rules.weights, o["facet_rules"].weights) < RTOL
    ip, ix = O.create_sparsity(om, o["V"], o["a"])
    want = O.assemble_matrix(om, o["V"], o["a"], ip, ix)
    bw = O.assemble_vector(om, o["V"], o["L"])
    A = cfx.fem.assemble_matrix(a)
    b = cfx.fem.assemble_vector(L)
    assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix)
    assert rel_err(A.data, want) < RTOL and rel_err(b, bw) < RTOL
    act = cfx.fem.active_domain(a)
    ina = O.inactive_dofs(o["V"], O.active_cells(o["a"], om.ncells))
    assert np.array_equal(act.inactive_dofs, ina)


def test_dg_poisson_solution_converges():
    """The engine's own DG system solved (scipy on the host): second-order L2 convergence to
    sin(pi x) sin(pi y) on the disk, as python/demo/demo_dg_poisson.py reports."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import cutfemx_amd as cfx
    from cutfemx_amd import poisson
    errs = []
    for n in (8, 16, 32):
        x, conn = cfx.box_mesh_arrays(2, n)
        mesh = cfx.Mesh.from_arrays(2, x, conn)
        phi = level_set_values(x, 2)
        f = cfx.Function(cfx.FunctionSpace(mesh, 1), phi)
        ndofs = conn.shape[0] * 3
        g = poisson.build_dg_forms(f, 1)
        V, a = g.function_space, g.a
        A = cfx.fem.assemble_matrix(a)
        b = cfx.fem.assemble_vector(g.L)
        cfx.fem.deactivate_outside(A, b, cfx.fem.active_domain(a))
        As = sp.csr_matrix((A.data, A.indices, A.indptr), shape=(ndofs, ndofs))
        u = spla.spsolve(As.tocsc(), np.asarray(b))
        xd = x[conn.ravel()][:, :2]
        e = u - np.sin(np.pi * xd[:, 0]) * np.sin(np.pi * xd[:, 1])
        M = cfx.fem.assemble_matrix(cfx.fem.form([cfx.fem.Integral(
            cfx.fem.MASS, cells=cfx.locate_entities(g.cell_cut, "phi<0"), rules=g.volume_rules, qdegree=2)], V))
        Ms = sp.csr_matrix((M.data, M.indices, M.indptr), shape=(ndofs, ndofs))
        errs.append(float(np.sqrt(e @ (Ms @ e))))
    rates = np.log2(np.array(errs[:-1]) / np.array(errs[1:]))
    assert errs[-1] < 2e-3 and np.all(rates > 1.7), (errs, rates)
