"""complex128 at the boundary (the <std::complex<double>, double> rows of python/cutfemx/wrappers/fem.cpp:490-500).
Invariants of python/tests/test_complex_assembly.py:24-95: assemble_scalar(kappa dx_runtime) = kappa |Omega|, and
runtime == standard assembly to 1e-12 for the scalar, the vector inner(kappa, v) dx and the matrix
kappa inner(grad u, grad v) dx with kappa = 2 + 3j.  Geometry and integrands are real, so the complex oracle is the
real oracle times the constants: A = sum_k s_k A_k (+ i times the form of the imaginary parts of a complex
coefficient Function)."""
import numpy as np
import pytest

from helpers import level_set_values

KAPPA = 2.0 + 3.0j


def rel_err(a, b):
    """max |a - b| / max |b| on complex arrays (helpers.rel_err is for real ones)."""
    a, b = np.asarray(a, dtype=np.complex128), np.asarray(b, dtype=np.complex128)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def test_oracle_complex_invariants(oracle):
    # test_complex_assembly.py:24-50 and :53-95 restated on the oracle: the real runtime-vs-standard identities
    # (pinned in tests/test_oracle_kat.py) carry over to every complex constant
    O = oracle
    om = O.mesh_box(2, 4)
    V = O.Space(om.conn, om.nnodes, 1, 1)
    cells = np.arange(om.ncells, dtype=np.int32)
    full = O.full_cell_rules(om, cells, 2)
    none = np.zeros(0, dtype=np.int32)
    one_run = [O.Integral(O.CELL, O.L_SOURCE, entities=none, rules=full, params=(O.F_ONE, 1.0), qdegree=2)]
    one_std = [O.Integral(O.CELL, O.L_SOURCE, entities=cells, params=(O.F_ONE, 1.0), qdegree=2)]
    v_run, v_std = KAPPA * O.assemble_vector(om, V, one_run), KAPPA * O.assemble_vector(om, V, one_std)
    assert abs(v_run.sum() - KAPPA) < 1e-12 and abs(v_std.sum() - KAPPA) < 1e-12     # assemble_scalar(kappa dx) = kappa |Omega|
    assert np.linalg.norm(v_run - v_std) < 1e-12
    a_run = [O.Integral(O.CELL, O.K_STIFFNESS, entities=none, rules=full, qdegree=0)]
    a_std = [O.Integral(O.CELL, O.K_STIFFNESS, entities=cells, qdegree=0)]
    ip, ix = O.create_sparsity(om, V, a_std)
    assert np.linalg.norm(KAPPA * O.assemble_matrix(om, V, a_run, ip, ix) - KAPPA * O.assemble_matrix(om, V, a_std, ip, ix)) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("tdim,n", [(2, 10), (3, 6)])
def test_gpu_complex_poisson(oracle, tdim, n):
    """The cut Poisson forms with complex constants per integral (kappa on the stiffness term, a second constant on
    Nitsche, a real one on the ghost penalty) and a complex source Function: values / vector / scalar against the
    oracle combination at 1e-12; deactivation with a complex diagonal; lifting and set_bc with complex data."""
    import cutfemx_amd as cfx
    from cutfemx_amd import fem
    O = oracle
    om = O.mesh_box(tdim, n)
    phi = level_set_values(om.x, tdim)
    dom = O.classify(om.conn, phi)
    inside = O.locate_entities(dom, "phi<0")
    ovol = O.runtime_quadrature(om, om.conn, phi, dom, "phi<0", 4)
    oitf = O.runtime_quadrature(om, om.conn, phi, dom, "phi=0", 4)
    onrm = O.evaluate_normals(om, om.conn, phi, oitf)
    oghost = O.ghost_penalty_facets(om, dom, "phi<0")
    oV = O.Space(om.conn, om.nnodes, 1, 1)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(V, phi))
    vol, itf = cfx.runtime_quadrature(cd, "phi<0", 4), cfx.runtime_quadrature(cd, "phi=0", 4)
    nrm, ghost = cfx.normal(cd, itf), cfx.ghost_penalty_facets(cd, "phi<0")
    s = [KAPPA, 0.5 - 1.5j, 1.0]
    o_int = [O.Integral(O.CELL, O.K_STIFFNESS, entities=inside, rules=ovol, qdegree=0),
             O.Integral(O.CELL, O.K_NITSCHE, rules=oitf, point_data=onrm, params=(40.0,)),
             O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=oghost, params=(0.1,), qdegree=0)]
    g_int = [fem.Integral(fem.STIFFNESS, cells=inside, rules=vol, qdegree=0, scale=s[0]),
             fem.Integral(fem.NITSCHE, rules=itf, point_data=nrm, params=(40.0,), scale=s[1]),
             fem.Integral(fem.GHOST_GRADJUMP, facets=ghost, params=(0.1,), qdegree=0, scale=s[2])]
    ip, ix = O.create_sparsity(om, oV, o_int)
    want = sum(sk * O.assemble_matrix(om, oV, [Ik], ip, ix) for sk, Ik in zip(s, o_int))
    a = fem.form(g_int, V)
    assert a.dtype == np.dtype(np.complex128)
    A = fem.assemble_matrix(a)
    assert A.dtype == np.dtype(np.complex128) and A.data.dtype == np.complex128
    assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix)
    assert rel_err(A.data, want) < 1e-12
    fem.assemble_matrix(a, A=A)                                   # accumulates
    assert rel_err(A.data, 2.0 * want) < 1e-12
    with pytest.raises(TypeError):
        fem.assemble_matrix(a, A=fem.create_matrix(a, dtype=np.float64))
    # markers
    rng = np.random.default_rng(8)
    bc = (rng.random(om.nnodes) < 0.1).astype(np.int8)
    wantb = sum(sk * O.assemble_matrix(om, oV, [Ik], ip, ix, bc, bc) for sk, Ik in zip(s, o_int))
    assert rel_err(fem.assemble_matrix(a, bcs=bc).data, wantb) < 1e-12
    # linear form: complex constant on an analytic source, a complex source FUNCTION, Nitsche datum
    w = rng.standard_normal(om.nnodes) + 1j * rng.standard_normal(om.nnodes)
    oL = [O.Integral(O.CELL, O.L_SOURCE, entities=inside, rules=ovol, params=(O.F_SINPROD, 1.0), qdegree=4),
          O.Integral(O.CELL, O.L_NITSCHE_RHS, rules=oitf, point_data=onrm, params=(40.0, O.F_SINPROD, 1.0))]
    gL = [fem.Integral(fem.SOURCE, cells=inside, rules=vol, params=(fem.F_SINPROD, 1.0), qdegree=4, scale=KAPPA),
          fem.Integral(fem.NITSCHE_RHS, rules=itf, point_data=nrm, params=(40.0, fem.F_SINPROD, 1.0), scale=1j),
          fem.Integral(fem.SOURCE, cells=inside, rules=vol, params=(fem.F_COEFFICIENT, 1.0), qdegree=4, coefficient=w, scale=s[1])]

    def src(values):
        return O.assemble_vector(om, oV, [O.Integral(O.CELL, O.L_SOURCE, entities=inside, rules=ovol,
                                                     params=(O.F_COEFFICIENT, 1.0), qdegree=4, coefficient=values)])
    wantv = KAPPA * O.assemble_vector(om, oV, oL[:1]) + 1j * O.assemble_vector(om, oV, oL[1:]) \
        + s[1] * (src(np.ascontiguousarray(w.real)) + 1j * src(np.ascontiguousarray(w.imag)))
    L = fem.form(gL, V)
    b = fem.assemble_vector(L)
    assert b.dtype == np.complex128 and rel_err(b, wantv) < 1e-12
    # assemble_scalar(kappa dx) = kappa |Omega_h|
    M = fem.form([fem.Integral(fem.SOURCE, cells=inside, rules=vol, params=(fem.F_ONE, 1.0), qdegree=0, scale=KAPPA)], V, rank=1)
    vol_h = O.assemble_vector(om, oV, [O.Integral(O.CELL, O.L_SOURCE, entities=inside, rules=ovol, params=(O.F_ONE, 1.0), qdegree=0)]).sum()
    assert abs(fem.assemble_scalar(M) - KAPPA * vol_h) < 1e-12
    # lifting b -= alpha A (g - x0) and set_bc with complex data
    import scipy.sparse as sp
    Asp = sp.csr_matrix((want, ix, ip), shape=(om.nnodes, om.nnodes))
    g = rng.standard_normal(om.nnodes) + 1j * rng.standard_normal(om.nnodes)
    x0 = rng.standard_normal(om.nnodes) + 1j * rng.standard_normal(om.nnodes)
    b0 = rng.standard_normal(om.nnodes) + 1j * rng.standard_normal(om.nnodes)
    alpha = 0.6 - 0.2j
    got = fem.apply_lifting(b0.copy(), a, bc, g, x0=x0, alpha=alpha)
    assert rel_err(got, b0 - Asp @ np.where(bc == 1, alpha * (g - x0), 0.0)) < 1e-11
    got = fem.set_bc(b0.copy(), bc, g, x0=x0, alpha=alpha)
    assert np.allclose(got, np.where(bc == 1, alpha * (g - x0), b0), rtol=0, atol=1e-15)
    # deactivation: complex diagonal / rhs on the dofs outside the active domain
    A1 = fem.assemble_matrix(a)
    b1 = fem.assemble_vector(L)
    domn = fem.deactivate_outside(A1, b1, fem.active_domain(a), diagonal=1.0 + 0.5j, rhs_value=0.25j)
    ina = O.inactive_dofs(oV, O.active_cells(o_int, om.ncells))
    assert np.array_equal(domn.inactive_dofs, ina)
    vals, bb = want.copy(), wantv.copy()
    diag_pos = [ip[r] + int(np.searchsorted(ix[ip[r]:ip[r + 1]], r)) for r in ina]
    vals[diag_pos] = 1.0 + 0.5j
    bb[ina] = 0.25j
    assert rel_err(A1.data, vals) < 1e-12 and rel_err(b1, bb) < 1e-12


@pytest.mark.gpu
def test_gpu_complex_form_of_real_constants(oracle):
    """dtype=complex128 with real constants: the float64 numbers in the real parts, zero imaginary parts
    (test_complex_assembly.py:24-50: `form.dtype == scalar_dtype`)."""
    import cutfemx_amd as cfx
    from cutfemx_amd import fem
    O = oracle
    om = O.mesh_box(2, 6)
    mesh = cfx.Mesh.from_arrays(2, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    cells = np.arange(om.ncells, dtype=np.int32)
    ar = fem.form([fem.Integral(fem.MASS, cells=cells, qdegree=2)], V)
    ac = fem.form([fem.Integral(fem.MASS, cells=cells, qdegree=2)], V, dtype=np.complex128)
    assert ar.dtype == np.dtype(np.float64) and ac.dtype == np.dtype(np.complex128)
    Ar, Ac = fem.assemble_matrix(ar), fem.assemble_matrix(ac)
    import os
    if os.environ.get("CFX_ASSEMBLY") == "atomic":   # FP64 atomics: the order of the additions differs from run to run
        assert np.allclose(Ac.data.real, Ar.data, rtol=1e-14, atol=0.0) and not Ac.data.imag.any()
    else:
        assert np.array_equal(Ac.data.real, Ar.data) and not Ac.data.imag.any()
    with pytest.raises(TypeError):
        fem.form([fem.Integral(fem.MASS, cells=cells, qdegree=2, scale=2j)], V, dtype=np.float64)


@pytest.mark.gpu
@pytest.mark.parametrize("tdim,n", [(2, 10), (3, 6)])
def test_gpu_complex64_containers(oracle, tdim, n):
    """complex64 (the <std::complex<float>, float> rows of wrappers/fem.cpp:490-500): interleaved float32 containers, the
    complex128 arithmetic rounded once -- every entry within a few float32 ulps of the complex128 result; accumulate,
    markers, lifting, set_bc and deactivation keep their meaning."""
    import cutfemx_amd as cfx
    from cutfemx_amd import fem
    O = oracle
    om = O.mesh_box(tdim, n)
    phi = level_set_values(om.x, tdim)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(V, phi))
    inside = cfx.locate_entities(cd, "phi<0")
    vol, itf = cfx.runtime_quadrature(cd, "phi<0", 4), cfx.runtime_quadrature(cd, "phi=0", 4)
    nrm, ghost = cfx.normal(cd, itf), cfx.ghost_penalty_facets(cd, "phi<0")
    s = [KAPPA, 0.5 - 1.5j, 1.0]

    def forms(dtype):
        a = fem.form([fem.Integral(fem.STIFFNESS, cells=inside, rules=vol, qdegree=0, scale=s[0]),
                      fem.Integral(fem.NITSCHE, rules=itf, point_data=nrm, params=(40.0,), scale=s[1]),
                      fem.Integral(fem.GHOST_GRADJUMP, facets=ghost, params=(0.1,), qdegree=0, scale=s[2])], V, dtype=dtype)
        L = fem.form([fem.Integral(fem.SOURCE, cells=inside, rules=vol, params=(fem.F_SINPROD, 1.0), qdegree=4, scale=KAPPA),
                      fem.Integral(fem.NITSCHE_RHS, rules=itf, point_data=nrm, params=(40.0, fem.F_SINPROD, 1.0), scale=1j)],
                     V, dtype=dtype)
        return a, L
    a64, L64 = forms(np.complex64)
    a128, L128 = forms(np.complex128)
    assert a64.dtype == np.dtype(np.complex64)

    def close(x, y, ulps=4):          # componentwise: float32 rounding of the complex128 value (+ accumulation slack)
        y = np.asarray(y)
        tol = ulps * np.finfo(np.float32).eps * np.maximum(np.abs(y), np.abs(y).max() * 1e-3)
        return bool(np.all(np.abs(x.real - y.real) <= tol) and np.all(np.abs(x.imag - y.imag) <= tol))
    A64, A128 = fem.assemble_matrix(a64), fem.assemble_matrix(a128)
    assert A64.dtype == np.dtype(np.complex64) and A64.data.dtype == np.complex64
    assert np.array_equal(A64.indptr, A128.indptr) and np.array_equal(A64.indices, A128.indices)
    assert close(A64.data, A128.data, 2)
    fem.assemble_matrix(a64, A=A64)                               # accumulates: widen, add, round
    assert close(A64.data, 2.0 * A128.data, 4)
    rng = np.random.default_rng(5)
    bc = (rng.random(om.nnodes) < 0.1).astype(np.int8)
    assert close(fem.assemble_matrix(a64, bcs=bc).data, fem.assemble_matrix(a128, bcs=bc).data, 2)
    b64, b128 = fem.assemble_vector(L64), fem.assemble_vector(L128)
    assert b64.dtype == np.complex64 and close(b64, b128, 2)
    with pytest.raises(TypeError):
        fem.assemble_vector(L64, np.zeros(om.nnodes, dtype=np.float32))
    # lifting and set_bc with complex64 data
    g = (rng.standard_normal(om.nnodes) + 1j * rng.standard_normal(om.nnodes)).astype(np.complex64)
    x0 = (rng.standard_normal(om.nnodes) + 1j * rng.standard_normal(om.nnodes)).astype(np.complex64)
    b0 = (rng.standard_normal(om.nnodes) + 1j * rng.standard_normal(om.nnodes)).astype(np.complex64)
    alpha = 0.6 - 0.2j
    got = fem.apply_lifting(b0.copy(), a64, bc, g, x0=x0, alpha=alpha)
    ref = fem.apply_lifting(b0.astype(np.complex128), a128, bc, g.astype(np.complex128), x0=x0.astype(np.complex128), alpha=alpha)
    assert got.dtype == np.complex64 and close(got, ref, 4)
    got = fem.set_bc(b0.copy(), bc, g, x0=x0, alpha=alpha)
    want = np.where(bc == 1, alpha * (g.astype(np.complex128) - x0.astype(np.complex128)), b0.astype(np.complex128))
    assert close(got, want, 2) and np.array_equal(got[bc == 0], b0[bc == 0])
    # deactivation with a complex diagonal
    A1, b1 = fem.assemble_matrix(a64), fem.assemble_vector(L64)
    fem.deactivate_outside(A1, b1, fem.active_domain(a64), diagonal=1.0 + 0.5j, rhs_value=0.25j)
    A2, b2 = fem.assemble_matrix(a128), fem.assemble_vector(L128)
    fem.deactivate_outside(A2, b2, fem.active_domain(a128), diagonal=1.0 + 0.5j, rhs_value=0.25j)
    assert close(A1.data, A2.data, 2) and close(b1, b2, 2)
