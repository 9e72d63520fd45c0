"""Shared problem builders for the parity tests: the same synthetic inputs
 fed to the CPU oracle and to the HIP engine."""
from __future__ import annotations

import numpy as np

CIRCLE = (np.array([0.47, 0.43]), 0.31)           # python/tests/test_cut_api.py:36-38
SPHERE = (np.array([0.47, 0.43, 0.41]), 0.31)     # python/tests/test_cut_api.py:50-52


def level_set_values(x: np.ndarray, tdim: int, kind: str = "sphere") -> np.ndarray:
    if kind == "sphere":
        c, r = CIRCLE if tdim == 2 else SPHERE
        return np.linalg.norm(x[:, :tdim] - c, axis=1) - r
    if kind == "gyroid":
        k = 2.0 * np.pi * 4.0 if tdim == 3 else 2.0 * np.pi * 2.0
        X, Y = x[:, 0], x[:, 1]
        Z = x[:, 2] if tdim == 3 else 0.3 * np.ones_like(X)
        return (np.sin(k * X) * np.cos(k * Y) + np.sin(k * Y) * np.cos(k * Z)
                + np.sin(k * Z) * np.cos(k * X)) + 0.0137
    if kind == "plane":
        return x[:, 0] - 0.51
    raise ValueError(kind)


def oracle_poisson(O, mesh, phi, order=4, gamma=40.0, gamma_g=0.1, degree=1, dofmap=None, ndofs=None):
    """Oracle restatement of the demo_poisson.py system on `mesh`."""
    dom = O.classify(mesh.conn, phi)
    inside = O.locate_entities(dom, "phi<0")
    vol = O.runtime_quadrature(mesh, mesh.conn, phi, dom, "phi<0", order)
    itf = O.runtime_quadrature(mesh, mesh.conn, phi, dom, "phi=0", order)
    normals = O.evaluate_normals(mesh, mesh.conn, phi, itf)
    ghost = O.ghost_penalty_facets(mesh, dom, "phi<0")
    V = O.Space(mesh.conn if dofmap is None else dofmap, mesh.nnodes if ndofs is None else ndofs, degree)
    a = [O.Integral(O.CELL, O.K_STIFFNESS, entities=inside, rules=vol, qdegree=2 * (degree - 1)),
         O.Integral(O.CELL, O.K_NITSCHE, rules=itf, point_data=normals, params=(gamma,))]
    if len(ghost):
        a.append(O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=ghost, params=(gamma_g,),
                            qdegree=2 * (degree - 1)))
    L = [O.Integral(O.CELL, O.L_SOURCE, entities=inside, rules=vol, params=(O.F_POISSON_RHS, 1.0), qdegree=4),
         O.Integral(O.CELL, O.L_NITSCHE_RHS, rules=itf, point_data=normals, params=(gamma, O.F_SINPROD, 1.0))]
    indptr, indices = O.create_sparsity(mesh, V, a)
    values = O.assemble_matrix(mesh, V, a, indptr, indices)
    b = O.assemble_vector(mesh, V, L)
    active = O.active_cells(a, mesh.ncells)
    inactive = O.inactive_dofs(V, active)
    return dict(domain=dom, inside=inside, vol=vol, itf=itf, normals=normals, ghost=ghost, V=V, a=a, L=L,
                indptr=indptr, indices=indices, values=values, b=b, active=active, inactive=inactive)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.max(np.abs(b)) if b.size else 0.0, 1e-300)
    return float(np.max(np.abs(a - b)) / scale) if a.size else 0.0


def scrambled_mesh(O, tdim: int, n: int, seed: int = 11):
    """The n^tdim Kuhn box mesh made unstructured: interior vertices moved by up to 0.2 h,
    vertices and cells renumbered at random, local vertex order of every cell shuffled (both
    orientations occur).  Nothing of the generator's numbering survives."""
    rng = np.random.default_rng(seed)
    om = O.mesh_box(tdim, n)
    x = om.x.copy()
    interior = np.all((x[:, :tdim] > 1e-12) & (x[:, :tdim] < 1.0 - 1e-12), axis=1)
    x[interior, :tdim] += (0.2 / n) * rng.uniform(-1.0, 1.0, size=(int(interior.sum()), tdim))
    p = rng.permutation(om.nnodes)                 # new id of old vertex v: p[v]
    xn = np.empty_like(x)
    xn[p] = x
    conn = p[om.conn]
    conn = conn[rng.permutation(om.ncells)]
    conn = rng.permuted(conn, axis=1)
    return O.Mesh(tdim, xn, conn.astype(np.int32))


def oracle_dg_poisson(O, mesh, phi, degree=1, order=4, sigma=10.0, sigma_gamma=20.0, gamma_g=0.1):
    """Oracle restatement of python/demo/demo_dg_poisson.py:205-277 on `mesh`: cut DG Poisson problem on
    {phi < 0}: volume terms on [inside cells, cut-cell rules], symmetric interior penalty on the skeleton of the
    active mesh restricted to the domain ([facets inside, rules of the cut facets]), Nitsche on the interface,
    ghost penalty on the cut band."""
    nd = {1: mesh.tdim + 1, 2: (mesh.tdim + 1) * (mesh.tdim + 2) // 2}[degree]
    ndofs = mesh.ncells * nd
    dofmap = np.arange(ndofs, dtype=np.int32).reshape(mesh.ncells, nd)
    V = O.Space(dofmap, ndofs, degree)
    dom = O.classify(mesh.conn, phi)
    inside = O.locate_entities(dom, "phi<0")
    active = O.locate_entities(dom, "phi<=0")
    vol = O.runtime_quadrature(mesh, mesh.conn, phi, dom, "phi<0", order)
    itf = O.runtime_quadrature(mesh, mesh.conn, phi, dom, "phi=0", order)
    normals = O.evaluate_normals(mesh, mesh.conn, phi, itf)
    skeleton = O.interior_facets_for_cells(mesh, active)
    H = O.facet_hosts(mesh, skeleton, mesh.conn)
    fdom = O.facet_classify(H, phi)
    omega_facets = skeleton[O.facet_locate_entities(H, fdom, "phi<0")]
    facet_rules = O.facet_runtime_quadrature(mesh, H, phi, fdom, "phi<0", order)
    ghost = O.ghost_penalty_facets(mesh, dom, "phi<0")
    s2 = degree * degree
    a = [O.Integral(O.CELL, O.K_STIFFNESS, entities=inside, rules=vol, qdegree=2 * (degree - 1)),
         O.Integral(O.INTERIOR_FACET, O.K_SIP, entities=omega_facets, rules=facet_rules, params=(sigma * s2,),
                    qdegree=2 * degree),
         O.Integral(O.CELL, O.K_NITSCHE, rules=itf, point_data=normals, params=(sigma_gamma * s2,)),
         O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=ghost, params=(gamma_g,), qdegree=2 * (degree - 1))]
    L = [O.Integral(O.CELL, O.L_SOURCE, entities=inside, rules=vol, params=(O.F_POISSON_RHS, 1.0), qdegree=4),
         O.Integral(O.CELL, O.L_NITSCHE_RHS, rules=itf, point_data=normals, params=(sigma_gamma * s2, O.F_SINPROD, 1.0))]
    return dict(V=V, dofmap=dofmap, ndofs=ndofs, domain=dom, inside=inside, active=active, vol=vol, itf=itf,
                normals=normals, skeleton=skeleton, hosts=H, fdom=fdom, omega_facets=omega_facets,
                facet_rules=facet_rules, ghost=ghost, a=a, L=L)


def profiled(fn):
    """fn() with the engine's per-kernel profile on: (result, {kernel name: launches})."""
    import ctypes as C
    from cutfemx_amd import _lib
    l = _lib.lib()
    _lib.check(l.cfx_profile_enable(1))
    _lib.check(l.cfx_profile_reset())
    try:
        out = fn()
        names = {}
        for i in range(l.cfx_profile_count()):
            name, ms, cnt = C.c_char_p(), C.c_double(), C.c_int64()
            _lib.check(l.cfx_profile_get(i, C.byref(name), C.byref(ms), C.byref(cnt)))
            if cnt.value:
                names[name.value.decode()] = cnt.value
    finally:
        _lib.check(l.cfx_profile_enable(0))
    return out, names


