"""GPU parity beyond scalar P1: P2 spaces over a P1 level set (BASELINE configs
4/5 as parity cases), vector-valued elasticity, Dirichlet markers, runtime-vs-
standard assembly on whole-cell rules, and the reference's error classes."""
import numpy as np
import pytest

from helpers import level_set_values, oracle_poisson, rel_err

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def setup(oracle, tdim, n, degree, bs, kind="sphere"):
    import cutfemx_amd as cfx
    O = oracle
    om = O.mesh_box(tdim, n)
    phi = level_set_values(om.x, tdim, kind)
    dofmap, ndofs = cfx.lagrange_dofmap(tdim, om.conn, om.nnodes, degree)
    oV = O.Space(dofmap, ndofs, degree, bs)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, degree, dofmap=None if degree == 1 else dofmap, ndofs=ndofs, bs=bs)
    Vphi = V if (degree == 1 and bs == 1) else cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(Vphi, phi))
    dom = O.classify(om.conn, phi)
    return dict(O=O, om=om, phi=phi, oV=oV, mesh=mesh, V=V, cd=cd, dom=dom)


def compare_forms(s, o_integrals, g_integrals, rank=2):
    import cutfemx_amd as cfx
    O, om, oV = s["O"], s["om"], s["oV"]
    a = cfx.fem.form(g_integrals, s["V"])
    if rank == 2:
        ip, ix = O.create_sparsity(om, oV, o_integrals)
        want = O.assemble_matrix(om, oV, o_integrals, ip, ix)
        A = cfx.fem.assemble_matrix(a)
        assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix)
        assert rel_err(A.data, want) < RTOL
        return A
    want = O.assemble_vector(om, oV, o_integrals)
    b = cfx.fem.assemble_vector(a)
    assert rel_err(b, want) < RTOL
    return b


@pytest.mark.parametrize("tdim,n", [(2, 12), (3, 6)])
def test_p2_scalar_poisson_with_ghost_penalty(oracle, tdim, n):
    import cutfemx_amd as cfx
    s = setup(oracle, tdim, n, 2, 1)
    O, om, dom, cd = s["O"], s["om"], s["dom"], s["cd"]
    inside = O.locate_entities(dom, "phi<0")
    ovol = O.runtime_quadrature(om, om.conn, s["phi"], dom, "phi<0", 4)
    oitf = O.runtime_quadrature(om, om.conn, s["phi"], dom, "phi=0", 4)
    onrm = O.evaluate_normals(om, om.conn, s["phi"], oitf)
    oghost = O.ghost_penalty_facets(om, dom, "phi<0")
    vol = cfx.runtime_quadrature(cd, "phi<0", 4)
    itf = cfx.runtime_quadrature(cd, "phi=0", 4)
    nrm = cfx.normal(cd, itf)
    ghost = cfx.ghost_penalty_facets(cd, "phi<0")
    oa = [O.Integral(O.CELL, O.K_STIFFNESS, entities=inside, rules=ovol, qdegree=2),
          O.Integral(O.CELL, O.K_MASS, entities=inside, rules=ovol, qdegree=4),
          O.Integral(O.CELL, O.K_NITSCHE, rules=oitf, point_data=onrm, params=(40.0,)),
          O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=oghost, params=(0.1,), qdegree=2)]
    ga = [cfx.fem.Integral(cfx.fem.STIFFNESS, cells=inside, rules=vol, qdegree=2),
          cfx.fem.Integral(cfx.fem.MASS, cells=inside, rules=vol, qdegree=4),
          cfx.fem.Integral(cfx.fem.NITSCHE, rules=itf, point_data=nrm, params=(40.0,)),
          cfx.fem.Integral(cfx.fem.GHOST_GRADJUMP, facets=ghost, params=(0.1,), qdegree=2)]
    A = compare_forms(s, oa, ga)
    M = A.to_scipy()
    assert abs(M - M.T).max() < 1e-10 * abs(M).max()
    oL = [O.Integral(O.CELL, O.L_SOURCE, entities=inside, rules=ovol, params=(O.F_POISSON_RHS, 1.0), qdegree=4),
          O.Integral(O.CELL, O.L_NITSCHE_RHS, rules=oitf, point_data=onrm, params=(40.0, O.F_SINPROD, 1.0))]
    gL = [cfx.fem.Integral(cfx.fem.SOURCE, cells=inside, rules=vol, params=(cfx.fem.F_POISSON_RHS, 1.0), qdegree=4),
          cfx.fem.Integral(cfx.fem.NITSCHE_RHS, rules=itf, point_data=nrm, params=(40.0, cfx.fem.F_SINPROD, 1.0))]
    compare_forms(s, oL, gL, rank=1)
    # local tensors of a cut P2 cell and of a facet
    a = cfx.fem.form(ga, s["V"])
    for integral, idx, use_rule in [(0, 0, True), (1, ovol.parent_map.size // 2, True), (0, len(inside) // 2, False),
                                    (3, len(oghost) // 2, False)]:
        got = cfx.fem.tabulate_entity(a, integral, idx, use_rule)
        want = O.tabulate_entity(om, s["oV"], oa[integral], idx, use_rule)
        assert rel_err(got, want) < RTOL


@pytest.mark.parametrize("tdim,n,degree", [(2, 10, 1), (3, 6, 1), (3, 4, 2)])
def test_vector_elasticity(oracle, tdim, n, degree):
    # python/demo/demo_elasticity.py:167-238 (E = 1e3, nu = 0.3), vector space bs = gdim
    import cutfemx_amd as cfx
    s = setup(oracle, tdim, n, degree, tdim)
    O, om, dom, cd = s["O"], s["om"], s["dom"], s["cd"]
    inside = O.locate_entities(dom, "phi<0")
    ovol = O.runtime_quadrature(om, om.conn, s["phi"], dom, "phi<0", 2)
    vol = cfx.runtime_quadrature(cd, "phi<0", 2)
    q = 2 * (degree - 1)
    oa = [O.Integral(O.CELL, O.K_ELASTICITY, entities=inside, rules=ovol, params=(1.0e3, 0.3), qdegree=q),
          O.Integral(O.CELL, O.K_MASS, entities=inside, rules=ovol, qdegree=2 * degree)]
    ga = [cfx.fem.Integral(cfx.fem.ELASTICITY, cells=inside, rules=vol, params=(1.0e3, 0.3), qdegree=q),
          cfx.fem.Integral(cfx.fem.MASS, cells=inside, rules=vol, qdegree=2 * degree)]
    A = compare_forms(s, oa, ga)
    M = A.to_scipy()
    assert abs(M - M.T).max() < 1e-10 * abs(M).max()
    # rigid translations lie in the null space of the elasticity block alone
    a_el = cfx.fem.form(ga[:1], s["V"])
    K = cfx.fem.assemble_matrix(a_el).to_scipy()
    t = np.zeros(K.shape[0]); t[0::tdim] = 1.0
    assert np.abs(K @ t).max() < 1e-9 * abs(K).max()
    dom_a = cfx.fem.active_domain(a_el)
    active = O.active_cells(oa[:1], om.ncells)
    assert np.array_equal(dom_a.active_cells, active)
    assert np.array_equal(dom_a.inactive_dofs, O.inactive_dofs(s["oV"], active))


def elasticity_problem(s, degree, order=2, E=1.0e3, nu=0.3, gamma_ghost=0.05):
    """a of python/demo/demo_elasticity.py:214-235 on the cut domain: sigma(u):eps(v) over [solid cells, rules]
    plus gamma (2 mu + lambda) h_avg [grad u . n].[grad v . n] over the ghost-penalty facets, for the oracle and
    for the engine."""
    import cutfemx_amd as cfx
    O, om, dom, cd = s["O"], s["om"], s["dom"], s["cd"]
    mu, lmbda = E / (2.0 * (1.0 + nu)), E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu))
    gpar = gamma_ghost * (2.0 * mu + lmbda)
    q = 2 * (degree - 1)
    inside = O.locate_entities(dom, "phi<0")
    ovol = O.runtime_quadrature(om, om.conn, s["phi"], dom, "phi<0", order)
    oghost = O.ghost_penalty_facets(om, dom, "phi<0")
    vol = cfx.runtime_quadrature(cd, "phi<0", order)
    ghost = cfx.ghost_penalty_facets(cd, "phi<0")
    assert np.array_equal(ghost.rows, oghost) and len(oghost) > 0
    oa = [O.Integral(O.CELL, O.K_ELASTICITY, entities=inside, rules=ovol, params=(E, nu), qdegree=q),
          O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=oghost, params=(gpar,), qdegree=q)]
    ga = [cfx.fem.Integral(cfx.fem.ELASTICITY, cells=inside, rules=vol, params=(E, nu), qdegree=q),
          cfx.fem.Integral(cfx.fem.GHOST_GRADJUMP, facets=ghost, params=(gpar,), qdegree=q)]
    return inside, oa, ga


@pytest.mark.parametrize("mode", ["rows", "atomic", "deterministic"])
@pytest.mark.parametrize("tdim,n,degree", [(2, 10, 1), (3, 6, 1), (2, 8, 2), (3, 4, 2)])
def test_vector_elasticity_with_ghost_penalty_and_lifting(oracle, tdim, n, degree, mode, monkeypatch):
    """BASELINE configs[4] as a parity case: the ghost penalty on a VECTOR space (bs = gdim, P1 and P2) next to
    the elasticity term, strong Dirichlet data lifted through both (demo_elasticity.py:224-235, :67-93;
    assemble_matrix_impl.h:409-607 interior-facet loop with bs > 1, assemble_vector_impl.h:383-436)."""
    import cutfemx_amd as cfx
    if mode == "atomic":
        monkeypatch.setenv("CFX_ASSEMBLY", "atomic")
    if mode == "deterministic":
        monkeypatch.setenv("CFX_DETERMINISTIC", "1")
    s = setup(oracle, tdim, n, degree, tdim)
    O, om, oV = s["O"], s["om"], s["oV"]
    inside, oa, ga = elasticity_problem(s, degree)
    A = compare_forms(s, oa, ga)                    # sparsity bit-exact (facet coupling of all components), values 1e-12
    M = A.to_scipy()
    assert abs(M - M.T).max() < 1e-10 * abs(M).max()
    # the ghost-penalty block alone: componentwise, annihilates affine displacement fields (normal-gradient jumps vanish)
    G = cfx.fem.assemble_matrix(cfx.fem.form(ga[1:], s["V"])).to_scipy()
    xdof = np.zeros((oV.ndofs, 3))
    xdof[:om.nnodes] = om.x
    if degree == 2:                                 # edge dofs sit at the edge midpoints (Basix edge order)
        edges = [(1, 2), (0, 2), (0, 1)] if tdim == 2 else [(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)]
        for k, (p, q) in enumerate(edges):
            xdof[oV.dofmap[:, tdim + 1 + k]] = 0.5 * (om.x[om.conn[:, p]] + om.x[om.conn[:, q]])
    for comp in range(tdim):
        u = np.zeros(oV.ndofs * tdim)
        u[comp::tdim] = 1.0 + 2.0 * xdof[:, 0] - 0.5 * xdof[:, 1]
        assert np.abs(G @ u).max() < 1e-9 * abs(G).max()
    # local tensor of one facet: 2 cells x nd dofs x bs components, block layout [[00, 01], [10, 11]]
    a = cfx.fem.form(ga, s["V"])
    idx = len(oa[1].entities) // 2
    got = cfx.fem.tabulate_entity(a, 1, idx, False)
    want = O.tabulate_entity(om, oV, oa[1], idx, False)
    assert got.shape == want.shape and rel_err(got, want) < RTOL
    # strong Dirichlet data on every component of some dofs touched by the facets and the cells
    ndofs = oV.ndofs * tdim
    rng = np.random.default_rng(9)
    markers = np.zeros(ndofs, dtype=np.int8)
    touched = np.unique(oV.dofmap[inside])
    for k in range(tdim):
        markers[touched[::4] * tdim + k] = 1
    g, x0, b0 = rng.standard_normal(ndofs), rng.standard_normal(ndofs), rng.standard_normal(ndofs)
    for alpha, x in [(1.0, None), (0.6, x0)]:
        want = O.apply_lifting(om, oV, oa, markers, g, b0.copy(), x0=x, alpha=alpha)
        got = cfx.fem.apply_lifting(b0.copy(), a, markers, g, x0=x, alpha=alpha)
        assert rel_err(got, want) < RTOL
        d = np.where(markers == 1, alpha * (g - (0.0 if x is None else x)), 0.0)
        assert rel_err(got, b0 - M @ d) < 1e-11
    ip, ix = O.create_sparsity(om, oV, oa)
    want = O.assemble_matrix(om, oV, oa, ip, ix, markers, markers)
    Abc = cfx.fem.assemble_matrix(a, bcs=markers)
    assert rel_err(Abc.data, want) < RTOL
    # active domain / deactivation of the vector system
    dom_a = cfx.fem.active_domain(a)
    active = O.active_cells(oa, om.ncells)
    assert np.array_equal(dom_a.active_cells, active)
    assert np.array_equal(dom_a.inactive_dofs, O.inactive_dofs(oV, active))


@pytest.mark.parametrize("mode", ["rows", "atomic"])
def test_dirichlet_markers_zero_rows_and_columns(oracle, mode, monkeypatch):
    # assemble_matrix_impl.h:151-185
    import cutfemx_amd as cfx
    if mode == "atomic":
        monkeypatch.setenv("CFX_ASSEMBLY", "atomic")
    s = setup(oracle, 3, 6, 1, 1)
    O, om, dom, cd = s["O"], s["om"], s["dom"], s["cd"]
    inside = O.locate_entities(dom, "phi<0")
    ovol = O.runtime_quadrature(om, om.conn, s["phi"], dom, "phi<0", 2)
    oghost = O.ghost_penalty_facets(om, dom, "phi<0")
    vol = cfx.runtime_quadrature(cd, "phi<0", 2)
    ghost = cfx.ghost_penalty_facets(cd, "phi<0")
    oa = [O.Integral(O.CELL, O.K_STIFFNESS, entities=inside, rules=ovol, qdegree=0),
          O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=oghost, params=(0.1,), qdegree=0)]
    ga = [cfx.fem.Integral(cfx.fem.STIFFNESS, cells=inside, rules=vol, qdegree=0),
          cfx.fem.Integral(cfx.fem.GHOST_GRADJUMP, facets=ghost, params=(0.1,), qdegree=0)]
    bc = np.zeros(om.nnodes, dtype=np.int8)
    touched = np.unique(om.conn[inside])
    bc[touched[::7]] = 1
    ip, ix = O.create_sparsity(om, s["oV"], oa)
    want = O.assemble_matrix(om, s["oV"], oa, ip, ix, bc, bc)
    A = cfx.fem.assemble_matrix(cfx.fem.form(ga, s["V"]), bcs=bc)
    assert rel_err(A.data, want) < RTOL
    M = A.to_scipy().toarray()
    assert np.all(M[bc == 1, :] == 0) and np.all(M[:, bc == 1] == 0)


@pytest.mark.parametrize("tdim,n,degree,bs", [(3, 6, 1, 1), (2, 10, 2, 1), (3, 4, 2, 3)])
def test_apply_lifting_and_set_bc(oracle, tdim, n, degree, bs):
    # assemble_vector_impl.h:383-436 (lifting_fn), python/demo/demo_elasticity.py:67-93
    import cutfemx_amd as cfx
    s = setup(oracle, tdim, n, degree, bs)
    O, om, dom, cd, oV = s["O"], s["om"], s["dom"], s["cd"], s["oV"]
    inside = O.locate_entities(dom, "phi<0")
    ovol = O.runtime_quadrature(om, om.conn, s["phi"], dom, "phi<0", 2)
    vol = cfx.runtime_quadrature(cd, "phi<0", 2)
    q = 2 * (degree - 1)
    if bs == 1:
        oghost = O.ghost_penalty_facets(om, dom, "phi<0")
        ghost = cfx.ghost_penalty_facets(cd, "phi<0")
        oa = [O.Integral(O.CELL, O.K_STIFFNESS, entities=inside, rules=ovol, qdegree=q),
              O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=oghost, params=(0.1,), qdegree=q)]
        ga = [cfx.fem.Integral(cfx.fem.STIFFNESS, cells=inside, rules=vol, qdegree=q),
              cfx.fem.Integral(cfx.fem.GHOST_GRADJUMP, facets=ghost, params=(0.1,), qdegree=q)]
    else:
        oa = [O.Integral(O.CELL, O.K_ELASTICITY, entities=inside, rules=ovol, params=(1.0e3, 0.3), qdegree=q)]
        ga = [cfx.fem.Integral(cfx.fem.ELASTICITY, cells=inside, rules=vol, params=(1.0e3, 0.3), qdegree=q)]
    ndofs = oV.ndofs * bs
    rng = np.random.default_rng(5)
    markers = np.zeros(ndofs, dtype=np.int8)
    touched = np.unique(oV.dofmap[inside])
    chosen = touched[::5]
    for k in range(bs):
        markers[chosen * bs + k] = 1
    g = rng.standard_normal(ndofs)
    x0 = rng.standard_normal(ndofs)
    b0 = rng.standard_normal(ndofs)
    a = cfx.fem.form(ga, s["V"])
    for alpha, x in [(1.0, None), (0.7, x0)]:
        want = O.apply_lifting(om, oV, oa, markers, g, b0.copy(), x0=x, alpha=alpha)
        got = cfx.fem.apply_lifting(b0.copy(), a, markers, g, x0=x, alpha=alpha)
        assert rel_err(got, want) < RTOL
        # the same thing through the unconstrained matrix: b - alpha A (g - x0) restricted to marked columns
        A = cfx.fem.assemble_matrix(a).to_scipy()
        d = np.where(markers == 1, alpha * (g - (0.0 if x is None else x)), 0.0)
        assert rel_err(got, b0 - A @ d) < 1e-11
        fixed = cfx.fem.set_bc(got.copy(), markers, g, x0=x, alpha=alpha)
        assert np.array_equal(fixed[markers == 0], got[markers == 0])
        assert np.allclose(fixed[markers == 1], d[markers == 1], rtol=0, atol=0)
    # device vectors
    import torch
    bt = torch.tensor(b0, device="cuda")
    cfx.fem.apply_lifting(bt, a, torch.tensor(markers, device="cuda"), torch.tensor(g, device="cuda"))
    assert rel_err(bt.cpu().numpy(), O.apply_lifting(om, oV, oa, markers, g, b0.copy())) < RTOL


@pytest.mark.parametrize("kernel,order,tol", [("stiffness", 2, 1e-12), ("mass", 2, 1e-12), ("elasticity", 2, 1e-9)])
def test_runtime_vs_standard_matrix_on_gpu(oracle, kernel, order, tol):
    # test_assembly_poisson.py:18-59, test_assembly_elasticity.py:18-68: whole-cell runtime
    # rules (reference points, weights*|detJ|) reproduce the standard assembly
    import cutfemx_amd as cfx
    tdim, n = 2, 4
    om = oracle.mesh_box(tdim, n)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    bs = 2 if kernel == "elasticity" else 1
    V = cfx.FunctionSpace(mesh, 1, bs=bs)
    cells = np.arange(om.ncells, dtype=np.int32)
    rules = cfx.full_cell_rules(mesh, cells, order)
    k, params = {"stiffness": (cfx.fem.STIFFNESS, ()), "mass": (cfx.fem.MASS, ()),
                 "elasticity": (cfx.fem.ELASTICITY, (1.0e3, 0.3))}[kernel]
    A = cfx.fem.assemble_matrix(cfx.fem.form([cfx.fem.Integral(k, cells=cells, params=params, qdegree=order)], V))
    B = cfx.fem.assemble_matrix(cfx.fem.form([cfx.fem.Integral(k, rules=rules, params=params)], V))
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
    assert np.linalg.norm(A.data - B.data) < tol
    # unsorted caller-supplied rules go through the entity-parallel path and agree too
    perm = np.random.default_rng(20260630).permutation(om.ncells)
    r2 = cfx.RuntimeQuadratureRules.from_arrays(
        mesh, rules.points.reshape(om.ncells, -1, tdim)[perm].reshape(-1, tdim),
        rules.weights.reshape(om.ncells, -1)[perm].ravel(), rules.offsets, rules.parent_map[perm])
    Cm = cfx.fem.assemble_matrix(cfx.fem.form([cfx.fem.Integral(k, rules=r2, params=params)], V))
    assert rel_err(Cm.data, A.data) < 1e-12


def test_reference_error_classes(oracle):
    # test_cut_api.py:214-233,252,1321-1333: ValueError / RuntimeError / IndexError surface
    import cutfemx_amd as cfx
    om = oracle.mesh_box(2, 4)
    mesh = cfx.Mesh.from_arrays(2, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    f = cfx.Function(V, level_set_values(om.x, 2))
    cd = cfx.cut(f)
    with pytest.raises(ValueError):
        cfx.locate_entities(cd, "psi<0")
    with pytest.raises(ValueError):
        cfx.locate_entities(cd, "phi<")
    with pytest.raises(ValueError):
        cfx.locate_entities(cd, "phi1<0")          # only one level set was cut
    with pytest.raises(ValueError):
        cfx.runtime_quadrature(cd, "phi<0", 4, backend="algoim")
    with pytest.raises(ValueError):
        cfx.runtime_quadrature(cd, "phi<0", -1)
    with pytest.raises(ValueError):
        cfx.cut([])
    with pytest.raises(TypeError):
        cfx.cut("phi")
    with pytest.raises(NotImplementedError):
        cfx.ghost_penalty_facets(cd, "phi<0", depth=2)
    L = cfx.fem.form([cfx.fem.Integral(cfx.fem.SOURCE, cells=np.arange(4, dtype=np.int32), params=(0, 1.0))], V)
    with pytest.raises(RuntimeError):
        cfx.fem.create_matrix(L)                   # "Form is not a bilinear" (assembler.h:570-574)
    with pytest.raises(ValueError):
        cfx.fem.active_domain(L)                   # rank-2 required (deactivate.h:80-85)
    a = cfx.fem.form([cfx.fem.Integral(cfx.fem.MASS, cells=np.arange(4, dtype=np.int32))], V)
    with pytest.raises(IndexError):
        cfx.fem.tabulate_entity(a, 0, 99, False)
    # two level sets: selector algebra on the device
    f1 = cfx.Function(V, om.x[:, 0] - 0.5)
    cd2 = cfx.cut([f, f1])
    dom = np.stack([cd2.domain(0), cd2.domain(1)])
    assert np.array_equal(cfx.locate_entities(cd2, "phi<0 and phi1>0"),
                          oracle.locate_entities(dom, "phi<0 and phi1>0"))


def test_update_reclassifies(oracle):
    # cutfemx.update(): cut.cpp:845-868, python/demo/demo_moving_poisson.py:53-67
    import cutfemx_amd as cfx
    om = oracle.mesh_box(3, 8)
    mesh = cfx.Mesh.from_arrays(3, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    f = cfx.Function(V, level_set_values(om.x, 3))
    cd = cfx.cut(f)
    before = cfx.locate_entities(cd, "phi=0")
    f.values = np.linalg.norm(om.x - np.array([0.55, 0.5, 0.45]), axis=1) - 0.27
    cfx.update(cd)
    dom = oracle.classify(om.conn, f.values)
    assert np.array_equal(cd.domain(), dom)
    after = cfx.locate_entities(cd, "phi=0")
    assert np.array_equal(after, oracle.locate_entities(dom, "phi=0")) and not np.array_equal(before, after)
    r = cfx.runtime_quadrature(cd, "phi<0", 2)
    want = oracle.runtime_quadrature(om, om.conn, f.values, dom, "phi<0", 2)
    assert np.array_equal(r.parent_map, want.parent_map) and rel_err(r.weights, want.weights) < RTOL


@pytest.mark.parametrize("mode", ["rows", "atomic"])
@pytest.mark.parametrize("tdim,n,degree", [(3, 6, 1), (2, 10, 2)])
def test_source_term_from_a_coefficient_function(oracle, monkeypatch, mode, tdim, n, degree):
    # a10 (pack_form.h:32-170): f given as a Function of the space instead of an expression
    import cutfemx_amd as cfx
    if mode == "atomic":
        monkeypatch.setenv("CFX_ASSEMBLY", "atomic")
    s = setup(oracle, tdim, n, degree, 1)
    O, om, dom, cd, oV = s["O"], s["om"], s["dom"], s["cd"], s["oV"]
    inside = O.locate_entities(dom, "phi<0")
    ovol = O.runtime_quadrature(om, om.conn, s["phi"], dom, "phi<0", 4)
    vol = cfx.runtime_quadrature(cd, "phi<0", 4)
    w = np.random.default_rng(3).standard_normal(oV.ndofs)
    q = 2 * degree
    oL = [O.Integral(O.CELL, O.L_SOURCE, entities=inside, rules=ovol, params=(O.F_COEFFICIENT, 1.5), qdegree=q,
                     coefficient=w)]
    gL = [cfx.fem.Integral(cfx.fem.SOURCE, cells=inside, rules=vol, params=(cfx.fem.F_COEFFICIENT, 1.5), qdegree=q,
                           coefficient=cfx.Function(s["V"], w))]
    b = compare_forms(s, oL, gL, rank=1)
    # (f, v) with f in the space is the mass matrix applied to its dof values
    M = cfx.fem.assemble_matrix(cfx.fem.form([cfx.fem.Integral(cfx.fem.MASS, cells=inside, rules=vol, qdegree=q)],
                                             s["V"])).to_scipy()
    assert rel_err(b, 1.5 * (M @ w)) < 1e-11
    with pytest.raises(ValueError):      # the field id and the array go together
        cfx.fem.form([cfx.fem.Integral(cfx.fem.SOURCE, cells=inside, params=(cfx.fem.F_COEFFICIENT, 1.0))], s["V"])
    with pytest.raises(ValueError):
        cfx.fem.form([cfx.fem.Integral(cfx.fem.SOURCE, cells=inside, params=(cfx.fem.F_ONE, 1.0), coefficient=w)], s["V"])


def test_named_level_sets_in_selectors(oracle):
    # python/tests/test_cut_api.py:713-773: real Function names are the selector names, frozen at cut()
    import cutfemx_amd as cfx
    om = oracle.mesh_box(2, 12)
    mesh = cfx.Mesh.from_arrays(2, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    a_vals = level_set_values(om.x, 2)
    b_vals = om.x[:, 1] - 0.51
    fluid, second = cfx.Function(V, a_vals, name="fluid"), cfx.Function(V, b_vals)
    cd = cfx.cut([fluid, second])
    assert cd.level_set_names == ("fluid", "phi1")
    ref = cfx.cut([cfx.Function(V, a_vals), cfx.Function(V, b_vals)])
    assert ref.level_set_names == ("phi", "phi1")
    for named, plain in [("fluid<0", "phi<0"), ("fluid=0 or phi1=0", "phi=0 or phi1=0"),
                         ("fluid<0 and phi1>0", "phi<0 and phi1>0")]:
        assert np.array_equal(cfx.locate_entities(cd, named), cfx.locate_entities(ref, plain))
    second.name = "renamed_after_cut"
    cd.update()
    assert cd.level_set_names == ("fluid", "phi1")
    assert cfx.locate_entities(cd, "fluid=0 or phi1=0").size > 0
    with pytest.raises(ValueError):
        cfx.locate_entities(cd, "phi<0")                       # no level set of that name here
    with pytest.raises(ValueError, match="Duplicate level-set function name"):
        cfx.cut([cfx.Function(V, a_vals, name="fluid"), cfx.Function(V, b_vals, name="fluid")])
    single = cfx.cut(cfx.Function(V, a_vals, name="fluid"))
    r = cfx.runtime_quadrature(single, "fluid<0", 2)
    want = cfx.runtime_quadrature(cfx.cut(cfx.Function(V, a_vals)), "phi<0", 2)
    assert np.array_equal(r.parent_map, want.parent_map) and np.array_equal(r.weights, want.weights)


def test_interior_facets_scalar_functional_and_zero_rows(oracle):
    import cutfemx_amd as cfx
    # interior_facets_for_cells stays inside the cell set (python/tests/test_cut_api.py:1176-1196)
    for tdim, n in ((2, 6), (3, 4)):
        om = oracle.mesh_box(tdim, n)
        mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
        sel = np.arange(0, om.ncells, 2, dtype=np.int32)
        for cells in (sel, sel[::-1].copy(), np.arange(om.ncells, dtype=np.int32)):
            rows = cfx.interior_facets_for_cells(mesh, cells).rows
            assert np.array_equal(rows, oracle.interior_facets_for_cells(om, np.sort(cells)))
            assert np.all(np.isin(rows[:, 0], cells)) and np.all(np.isin(rows[:, 2], cells))
    # functionals: area and perimeter of the circle (test_cut_api.py:796-811, :1268-1300)
    om = oracle.mesh_box(2, 21)
    mesh = cfx.Mesh.from_arrays(2, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(V, level_set_values(om.x, 2)))
    inside = cfx.locate_entities(cd, "phi<0")
    vol = cfx.runtime_quadrature(cd, "phi<0", 4)
    itf = cfx.runtime_quadrature(cd, "phi=0", 4)
    one = (cfx.fem.F_ONE, 1.0)
    area = cfx.fem.assemble_scalar(cfx.fem.form([cfx.fem.Integral(cfx.fem.SOURCE, cells=inside, rules=vol, params=one,
                                                                   qdegree=1)], V))
    perimeter = cfx.fem.assemble_scalar(cfx.fem.form([cfx.fem.Integral(cfx.fem.SOURCE, rules=itf, params=one)], V))
    assert abs(area - np.pi * 0.31 ** 2) < 1e-2 and abs(perimeter - 2 * np.pi * 0.31) < 1e-2
    assert abs(area - (vol.weights.sum() + inside.size * 0.5 / 21 ** 2)) < 1e-13
    # zero_rows: exactly the rows that no entity touches (fem.py:777-782)
    A = cfx.fem.assemble_matrix(cfx.fem.form([cfx.fem.Integral(cfx.fem.STIFFNESS, cells=inside, rules=vol, qdegree=0)], V))
    dom = cfx.fem.active_domain(cfx.fem.form([cfx.fem.Integral(cfx.fem.STIFFNESS, cells=inside, rules=vol, qdegree=0)], V))
    assert np.array_equal(cfx.fem.zero_rows(A), dom.inactive_dofs)
    assert cfx.fem.zero_rows(A, tol=1e300).size == A.nrows


def test_cut_with_cell_subset_as_host(oracle):
    # python/tests/test_cut_api.py:160-168, :211-222, python/tests/test_locate_entities.py:40-71
    import cutfemx_amd as cfx
    om = oracle.mesh_box(2, 8)
    mesh = cfx.Mesh.from_arrays(2, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    phi = om.x[:, 0] - 0.51
    f = cfx.Function(V, phi)
    full = cfx.cut(f)
    part = np.arange(0, om.ncells, 3, dtype=np.int32)
    sub = cfx.cut(f, part, mesh.tdim)
    assert sub.entity_dim == 2 and np.array_equal(sub.entities, part)
    for sel in ("phi=0", "phi<0", "phi>0", "phi<=0"):
        assert np.array_equal(cfx.locate_entities(sub, sel), np.intersect1d(cfx.locate_entities(full, sel), part))
    dom = oracle.classify(om.conn, phi)
    want = np.where(np.isin(np.arange(om.ncells), part), dom, 2)
    assert np.array_equal(sub.domain(), want)
    # rules only on the candidate cut cells, identical to the full cut's rules there
    r_sub, r_full = cfx.runtime_quadrature(sub, "phi<0", 2), cfx.runtime_quadrature(full, "phi<0", 2)
    keep = np.isin(r_full.parent_map, part)
    assert np.array_equal(r_sub.parent_map, r_full.parent_map[keep])
    assert np.array_equal(r_sub.weights, np.concatenate(
        [r_full.weights[r_full.offsets[k]:r_full.offsets[k + 1]] for k in np.flatnonzero(keep)]))
    # the subset survives update()
    f.values = om.x[:, 0] - 0.37
    sub.update()
    full.update()
    assert np.array_equal(cfx.locate_entities(sub, "phi=0"), np.intersect1d(cfx.locate_entities(full, "phi=0"), part))
    with pytest.raises(ValueError, match="entity_dim must be supplied"):
        cfx.cut(f, entities=part)
    with pytest.raises(ValueError, match="entity_dim is only valid"):
        cfx.cut(f, entity_dim=0)
    with pytest.raises(ValueError, match="integration rows"):
        cfx.cut(f, part, 1)                       # facet hosts are (cell, local facet) rows: tests/test_gpu_facets.py
    with pytest.raises(IndexError):
        cfx.cut(f, np.array([om.ncells], dtype=np.int32), 2)


def test_moving_domain_loop(oracle):
    # python/demo/demo_moving_poisson.py:53-90: one CutData, the level set moves in place on the
    # device, update() + rules + forms + sparsity + assembly every step; each step equals the oracle
    import torch

    import cutfemx_amd as cfx
    from cutfemx_amd import poisson
    n = 12
    om = oracle.mesh_box(3, n)
    mesh = cfx.Mesh.from_arrays(3, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    xt = torch.tensor(om.x, device="cuda")
    phi = torch.empty(om.nnodes, device="cuda", dtype=torch.float64)
    f = cfx.Function(V, phi)
    cd = None
    seen = []
    for step in range(4):
        centre = torch.tensor([0.40 + 0.05 * step, 0.45, 0.5 - 0.03 * step], device="cuda", dtype=torch.float64)
        phi.copy_(torch.linalg.norm(xt - centre, dim=1) - 0.27)        # in place: the engine aliases this array
        if cd is None:
            cd = cfx.cut(f)
        else:
            cfx.update(cd)
        system = poisson.build_forms(V, cd, order=4)
        A = cfx.fem.assemble_matrix(system.a)
        b = cfx.fem.assemble_vector(system.L)
        dom = cfx.fem.deactivate_outside(A, b, cfx.fem.active_domain(system.a))
        ref = oracle_poisson(oracle, om, phi.cpu().numpy())
        vals, bb = ref["values"].copy(), ref["b"].copy()
        oracle.deactivate(ref["inactive"], ref["indptr"], ref["indices"], vals, bb)
        assert np.array_equal(cd.domain(), ref["domain"])
        assert np.array_equal(A.indptr, ref["indptr"]) and np.array_equal(A.indices, ref["indices"])
        assert rel_err(A.data, vals) < RTOL and rel_err(b, bb) < RTOL
        assert np.array_equal(dom.inactive_dofs, ref["inactive"])
        seen.append(int(A.nnz))
    assert len(set(seen)) > 1   # the pattern really changed between steps


def test_deterministic_mode_is_bitwise_reproducible(oracle, monkeypatch):
    # CFX_DETERMINISTIC=1: lane-ordered LDS reduction + sorted facet lists; two assemblies of
    # the same system must agree bit for bit (the default LDS-atomic reduction only to round-off)
    import os

    import cutfemx_amd as cfx
    from cutfemx_amd import poisson
    if os.environ.get("CFX_ASSEMBLY") == "atomic":
        pytest.skip("global FP64 atomics are order dependent by construction")
    monkeypatch.setenv("CFX_DETERMINISTIC", "1")
    om = oracle.mesh_box(3, 14)
    mesh = cfx.Mesh.from_arrays(3, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    phi = level_set_values(om.x, 3)
    runs = []
    for _ in range(3):
        cd = cfx.cut(cfx.Function(V, phi))
        s = poisson.build_forms(V, cd)
        A = cfx.fem.assemble_matrix(s.a)
        b = cfx.fem.assemble_vector(s.L)
        runs.append((A.data.copy(), b.copy()))
    for vals, b in runs[1:]:
        assert np.array_equal(vals, runs[0][0]) and np.array_equal(b, runs[0][1])


@pytest.mark.parametrize("mode", ["rows", "atomic"])
@pytest.mark.parametrize("tdim,n,degree,bs", [(2, 10, 1, 1), (3, 6, 1, 1), (2, 8, 2, 1), (3, 4, 2, 1), (3, 5, 1, 3), (2, 6, 2, 2)])
def test_coefficients_in_bilinear_forms(oracle, tdim, n, degree, bs, mode, monkeypatch):
    """a10, pack_form.h:69-158: a scalar coefficient Function inside bilinear forms -- kappa grad u . grad v,
    rho u v, and the density-weighted elasticity of python/demo/demo_compliance_optimization.py -- over
    [inside cells, cut-cell rules]; constants travel in `params`."""
    import cutfemx_amd as cfx
    if mode == "atomic":
        monkeypatch.setenv("CFX_ASSEMBLY", "atomic")
    s = setup(oracle, tdim, n, degree, bs)
    O, om, dom, cd, oV = s["O"], s["om"], s["dom"], s["cd"], s["oV"]
    inside = O.locate_entities(dom, "phi<0")
    ovol = O.runtime_quadrature(om, om.conn, s["phi"], dom, "phi<0", 4)
    vol = cfx.runtime_quadrature(cd, "phi<0", 4)
    rng = np.random.default_rng(3)
    kappa = 1.0 + rng.uniform(0.0, 2.0, oV.ndofs)            # dof values of a scalar Function of the form's element
    q = 2 * degree
    terms = [(O.K_STIFFNESS, cfx.fem.STIFFNESS, ()), (O.K_MASS, cfx.fem.MASS, ())]
    if bs > 1:
        terms.append((O.K_ELASTICITY, cfx.fem.ELASTICITY, (1.0e3, 0.3)))
    for ok, gk, params in terms:
        oa = [O.Integral(O.CELL, ok, entities=inside, rules=ovol, params=params, qdegree=q, coefficient=kappa)]
        ga = [cfx.fem.Integral(gk, cells=inside, rules=vol, params=params, qdegree=q, coefficient=kappa)]
        A = compare_forms(s, oa, ga)
        # a constant coefficient is a constant factor
        c = np.full(oV.ndofs, 2.5)
        Ac = cfx.fem.assemble_matrix(cfx.fem.form(
            [cfx.fem.Integral(gk, cells=inside, rules=vol, params=params, qdegree=q, coefficient=c)], s["V"]))
        A1 = cfx.fem.assemble_matrix(cfx.fem.form(
            [cfx.fem.Integral(gk, cells=inside, rules=vol, params=params, qdegree=q)], s["V"]))
        assert rel_err(Ac.data, 2.5 * A1.data) < 1e-12
        assert rel_err(A.data, A1.data) > 1e-3               # the varying coefficient does change the matrix
    with pytest.raises(ValueError):                          # only these three terms take a coefficient
        cfx.fem.form([cfx.fem.Integral(cfx.fem.GHOST_GRADJUMP, facets=np.zeros((0, 4), np.int32), params=(0.1,),
                                       coefficient=kappa)], s["V"])


@pytest.mark.parametrize("tdim,n,degree", [(2, 10, 1), (3, 5, 1), (2, 6, 2), (3, 4, 2)])
def test_vector_valued_source_coefficient(oracle, tdim, n, degree):
    """a10: L = f . v with f a vector-valued Function of the (vector) space: be[(i, a)] = int f_a N_i."""
    import cutfemx_amd as cfx
    s = setup(oracle, tdim, n, degree, tdim)
    O, om, dom, cd, oV = s["O"], s["om"], s["dom"], s["cd"], s["oV"]
    inside = O.locate_entities(dom, "phi<0")
    ovol = O.runtime_quadrature(om, om.conn, s["phi"], dom, "phi<0", 4)
    vol = cfx.runtime_quadrature(cd, "phi<0", 4)
    rng = np.random.default_rng(4)
    f = rng.standard_normal(oV.ndofs * tdim)
    q = 2 * degree
    oL = [O.Integral(O.CELL, O.L_SOURCE, entities=inside, rules=ovol, params=(O.F_COEFFICIENT, 1.5), qdegree=q, coefficient=f)]
    gL = [cfx.fem.Integral(cfx.fem.SOURCE, cells=inside, rules=vol, params=(cfx.fem.F_COEFFICIENT, 1.5), qdegree=q,
                           coefficient=f)]
    b = compare_forms(s, oL, gL, rank=1)
    # the same through the vector mass matrix: b = 1.5 M f
    M = cfx.fem.assemble_matrix(cfx.fem.form(
        [cfx.fem.Integral(cfx.fem.MASS, cells=inside, rules=vol, qdegree=q)], s["V"])).to_scipy()
    assert rel_err(b, 1.5 * (M @ f)) < 1e-11
    with pytest.raises(ValueError):                          # an analytic scalar field cannot source a vector space
        cfx.fem.form([cfx.fem.Integral(cfx.fem.SOURCE, cells=inside, params=(cfx.fem.F_ONE, 1.0))], s["V"])


def test_form_alive_across_update_is_refused_and_new_forms_get_new_plans(oracle):
    """A form points at the located lists / rules of its cut.  cfx_cut_update drops them (their HBM blocks go back to
    the block cache and are handed out again at the same address): the old form must raise instead of assembling
    from recycled memory, and a form built afterwards must not adopt the old form's row plan (plans are keyed on
    block / handle serial numbers, cfx_rowasm.hip row_plan)."""
    import torch

    import cutfemx_amd as cfx
    from cutfemx_amd import poisson
    n = 10
    om = oracle.mesh_box(3, n)
    mesh = cfx.Mesh.from_arrays(3, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    xt = torch.tensor(om.x, device="cuda")
    phi = torch.empty(om.nnodes, device="cuda", dtype=torch.float64)
    f = cfx.Function(V, phi)

    def move(c):
        centre = torch.tensor(c, device="cuda", dtype=torch.float64)
        phi.copy_(torch.linalg.norm(xt - centre, dim=1) - 0.27)
    move([0.40, 0.45, 0.50])
    cd = cfx.cut(f)
    old = poisson.build_forms(V, cd, order=4)
    A0 = cfx.fem.assemble_matrix(old.a)          # builds and caches the plan of `old`
    move([0.55, 0.50, 0.42])
    cfx.update(cd)                               # the lists `old` points at are gone
    new = poisson.build_forms(V, cd, order=4)    # same sizes class, recycled addresses
    A1 = cfx.fem.assemble_matrix(new.a)
    b1 = cfx.fem.assemble_vector(new.L)
    ref = oracle_poisson(oracle, om, phi.cpu().numpy())
    assert np.array_equal(A1.indptr, ref["indptr"]) and np.array_equal(A1.indices, ref["indices"])
    assert rel_err(A1.data, ref["values"]) < RTOL and rel_err(b1, ref["b"]) < RTOL
    assert A0.nnz != A1.nnz or not np.array_equal(A0.indices, A1.indices)
    for use in (lambda: cfx.fem.assemble_matrix(old.a), lambda: cfx.fem.create_matrix(old.a),
                lambda: cfx.fem.assemble_vector(old.L), lambda: cfx.fem.active_domain(old.a)):
        with pytest.raises(RuntimeError, match="stale form"):
            use()


@pytest.mark.parametrize("degree,bs", [(1, 1), (2, 1), (2, 3)])
def test_overlap_without_prepare(oracle, degree, bs):
    """Two lanes with NOTHING pre-built (no prepare(), fresh spaces: incidence lists, neighbour lists, slot records and
    the shared row plan are built lazily inside the section by whichever lane needs them first and published to the
    other, cfx::publish_across_lanes): same matrix and vector as the oracle, degree 1 / 2 and vector degree 2."""
    import torch

    import cutfemx_amd as cfx
    fem = cfx.fem
    for rep in range(2):     # second repetition: fresh space again, block cache warm (recycled addresses)
        s = setup(oracle, 3, 6, degree, bs)
        O, om, dom, cd = s["O"], s["om"], s["dom"], s["cd"]
        inside = O.locate_entities(dom, "phi<0")
        ovol = O.runtime_quadrature(om, om.conn, s["phi"], dom, "phi<0", 2)
        oghost = O.ghost_penalty_facets(om, dom, "phi<0")
        vol = cfx.runtime_quadrature(cd, "phi<0", 2)
        ghost = cfx.ghost_penalty_facets(cd, "phi<0")
        if bs == 1:
            oa = [O.Integral(O.CELL, O.K_STIFFNESS, entities=inside, rules=ovol, qdegree=2),
                  O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=oghost, params=(0.1,), qdegree=2)]
            ga = [fem.Integral(fem.STIFFNESS, cells=inside, rules=vol, qdegree=2),
                  fem.Integral(fem.GHOST_GRADJUMP, facets=ghost, params=(0.1,), qdegree=2)]
            oL = [O.Integral(O.CELL, O.L_SOURCE, entities=inside, rules=ovol, params=(O.F_SINPROD, 1.0), qdegree=4)]
            gL = [fem.Integral(fem.SOURCE, cells=inside, rules=vol, params=(fem.F_SINPROD, 1.0), qdegree=4)]
        else:
            oa = [O.Integral(O.CELL, O.K_ELASTICITY, entities=inside, rules=ovol, params=(10.0, 0.3), qdegree=2),
                  O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=oghost, params=(0.1,), qdegree=2)]
            ga = [fem.Integral(fem.ELASTICITY, cells=inside, rules=vol, params=(10.0, 0.3), qdegree=2),
                  fem.Integral(fem.GHOST_GRADJUMP, facets=ghost, params=(0.1,), qdegree=2)]
            w = np.random.default_rng(5).standard_normal(s["oV"].ndofs * bs)
            oL = [O.Integral(O.CELL, O.L_SOURCE, entities=inside, rules=ovol, params=(O.F_COEFFICIENT, 1.0), qdegree=4,
                             coefficient=w)]
            gL = [fem.Integral(fem.SOURCE, cells=inside, rules=vol, params=(fem.F_COEFFICIENT, 1.0), qdegree=4,
                               coefficient=w)]
        a, L = fem.form(ga, s["V"]), fem.form(gL, s["V"], rank=1)
        b = torch.zeros(s["oV"].ndofs * bs, device="cuda", dtype=torch.float64)
        with fem.overlap() as lanes:
            lanes.side(lambda: fem.assemble_vector(L, b))
            A = fem.create_matrix(a)
            fem.assemble_matrix(a, A=A)
        ip, ix = O.create_sparsity(om, s["oV"], oa)
        assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix)
        assert rel_err(A.data, O.assemble_matrix(om, s["oV"], oa, ip, ix)) < RTOL
        assert rel_err(b.cpu().numpy(), O.assemble_vector(om, s["oV"], oL)) < RTOL


def test_block_deactivation_uses_per_row_active_domains(oracle):
    # python/tests/test_cut_api.py:880-952: a 2 x 2 block system, block row i supported on one side of the interface;
    # zero_block_rows lists a row when it is zero in every block of its block row, deactivate_outside_blocks touches the
    # diagonal blocks and the right-hand sides only
    import cutfemx_amd as cfx
    from cutfemx_amd import fem
    om = oracle.mesh_box(2, 12)
    phi = level_set_values(om.x, 2)
    mesh = cfx.Mesh.from_arrays(2, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(V, phi))
    inside, outside = cfx.locate_entities(cd, "phi<0"), cfx.locate_entities(cd, "phi>0")
    r_in, r_out = cfx.runtime_quadrature(cd, "phi<0", 2), cfx.runtime_quadrature(cd, "phi>0", 2)

    def mass(cells, rules, scale=1.0):
        a = fem.form([fem.Integral(fem.MASS, cells=cells, rules=rules, qdegree=2)], V)
        A = fem.assemble_matrix(a)
        if scale != 1.0:
            import ctypes as C
            vals = A.data * scale
            cfx._lib.check(cfx._lib.lib().cfx_copy(C.c_void_p(A._vptr), vals.ctypes.data_as(C.c_void_p), C.c_size_t(vals.nbytes)))
        return a, A
    a00, A00 = mass(inside, r_in)
    a11, A11 = mass(outside, r_out)
    _, A01 = mass(inside, r_in, 0.25)        # u2 v1 dx_inside: rows of block row 0 live on the inside
    _, A10 = mass(outside, r_out, 0.5)
    A_blocks = [[A00, A01], [A10, A11]]

    def rhs(cells, rules, scale):
        L = fem.form([fem.Integral(fem.SOURCE, cells=cells, rules=rules, params=(fem.F_ONE, scale), qdegree=2)], V)
        return fem.assemble_vector(L)
    b_blocks = [rhs(inside, r_in, 1.0), rhs(outside, r_out, 2.0)]
    domains = [fem.active_domain(a00), fem.active_domain(a11)]
    zero_before = fem.zero_block_rows(A_blocks)
    for i in range(2):
        assert set(domains[i].inactive_dofs.tolist()).issubset(set(zero_before[i].tolist()))
        assert domains[i].inactive_dofs.size > 0
        b_blocks[i][domains[i].inactive_dofs] = 3.0
    off_before = [A01.data.copy(), A10.data.copy()]
    returned = fem.deactivate_outside_blocks(A_blocks, domains, b_blocks)
    assert returned == domains
    assert all(rows.size == 0 for rows in fem.zero_block_rows(A_blocks))
    for i, dom in enumerate(domains):
        np.testing.assert_allclose(b_blocks[i][dom.inactive_dofs], 0.0)
        np.testing.assert_allclose(A_blocks[i][i].to_scipy().diagonal()[dom.inactive_dofs], 1.0)
    assert np.array_equal(A01.data, off_before[0]) and np.array_equal(A10.data, off_before[1])   # off-diagonal blocks untouched
    # the reference's input checks (deactivate.h:352-385, 323-337)
    with pytest.raises(RuntimeError, match="one ActiveDomain per block row"):
        fem.deactivate_outside_blocks(A_blocks, domains[:1])
    with pytest.raises(RuntimeError, match="square block matrix"):
        fem.deactivate_outside_blocks([[A00, A01], [A11]], domains)
    with pytest.raises(RuntimeError, match="every diagonal matrix block"):
        fem.zero_block_rows([[None, A01], [A10, A11]])
    with pytest.raises(RuntimeError, match="one RHS vector per block row"):
        fem.deactivate_outside_blocks(A_blocks, domains, b_blocks[:1])


@pytest.mark.parametrize("tdim,n,degree,bs", [(2, 14, 2, 1), (3, 6, 2, 1), (3, 5, 2, 3), (2, 12, 1, 2)])
def test_pattern_rows_are_reused_between_the_steps_of_a_moving_domain(oracle, tdim, n, degree, bs, monkeypatch):
    # cut.cpp:845-868 + assembler.h:567-592 rebuild the pattern every step; the engine copies the rows whose incident
    # cells kept their marks and facet sides from the previous pattern of the space.  After K moving steps the pattern
    # must equal a from-scratch build (CFX_PATTERN_REUSE=0) and the oracle's, bit for bit, and some rows must have
    # been copied, some hashed.
    import cutfemx_amd as cfx
    from cutfemx_amd import fem
    O = oracle
    om = O.mesh_box(tdim, n)
    dofmap, ndofs = cfx.lagrange_dofmap(tdim, om.conn, om.nnodes, degree)
    oV = O.Space(dofmap, ndofs, degree, bs)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, degree, dofmap=None if degree == 1 else dofmap, ndofs=ndofs, bs=bs)
    Vphi = cfx.FunctionSpace(mesh, 1)
    kern_g, kern_o = (fem.ELASTICITY, O.K_ELASTICITY) if bs > 1 else (fem.STIFFNESS, O.K_STIFFNESS)
    params = (1.0, 0.3) if bs > 1 else ()
    reused_total = hashed_total = 0
    for k in range(4):
        c = np.array([0.40 + 0.013 * k, 0.45, 0.5][:tdim])
        phi = np.linalg.norm(om.x[:, :tdim] - c, axis=1) - 0.27
        cd = cfx.cut(cfx.Function(Vphi, phi))
        dom = O.classify(om.conn, phi)
        inside = cfx.locate_entities(cd, "phi<0")
        vol = cfx.runtime_quadrature(cd, "phi<0", 2)
        ghost = cfx.ghost_penalty_facets(cd, "phi<0")

        def gform():
            ints = [fem.Integral(kern_g, cells=inside, rules=vol, params=params, qdegree=2)]
            if ghost.size > 0:
                ints.append(fem.Integral(fem.GHOST_GRADJUMP, facets=ghost, params=(0.1,), qdegree=2))
            return fem.form(ints, V)
        A = fem.create_matrix(gform())
        hashed, reused = A.reuse_stats
        hashed_total += hashed
        reused_total += reused
        if k == 0:
            assert reused == 0
        monkeypatch.setenv("CFX_PATTERN_REUSE", "0")
        A0 = fem.create_matrix(gform())
        monkeypatch.delenv("CFX_PATTERN_REUSE")
        assert A0.reuse_stats[1] == 0
        o_in = O.locate_entities(dom, "phi<0")
        o_vol = O.runtime_quadrature(om, om.conn, phi, dom, "phi<0", 2)
        o_ghost = O.ghost_penalty_facets(om, dom, "phi<0")
        o_ints = [O.Integral(O.CELL, kern_o, entities=o_in, rules=o_vol, params=params, qdegree=2)]
        if len(o_ghost):
            o_ints.append(O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=o_ghost, params=(0.1,), qdegree=2))
        ip, ix = O.create_sparsity(om, oV, o_ints)
        assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix)
        assert np.array_equal(A0.indptr, ip) and np.array_equal(A0.indices, ix)
        # the values assembled into the reused pattern equal the oracle's too
        want = O.assemble_matrix(om, oV, o_ints, ip, ix)
        assert rel_err(fem.assemble_matrix(gform(), A=A).data, want) < RTOL
        del A0
    assert 0 < reused_total < hashed_total, (reused_total, hashed_total)
