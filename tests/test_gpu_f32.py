"""float32 instantiation of the boundary (python/cutfemx/wrappers/fem.cpp:490-500, wrappers/cut.cpp:403-407):
float32 meshes, level sets, rules, matrices and vectors through the *_f32 entry points.  The engine widens the
inputs exactly and rounds the fp64 results once, so the oracle (fp64) is fed the widened float32 inputs and the
outputs must agree to float32 rounding: integer outputs bit-exact, real outputs within 4 float32 ulps of the
largest entry."""
import ctypes as C

import numpy as np
import pytest

from helpers import level_set_values, oracle_poisson, rel_err

pytestmark = pytest.mark.gpu
F32_TOL = 4 * np.finfo(np.float32).eps   # 4.8e-7, relative to the largest entry


def f32_problem(oracle, tdim, n):
    import cutfemx_amd as cfx
    O = oracle
    om64 = O.mesh_box(tdim, n)
    x32 = om64.x.astype(np.float32)
    phi32 = level_set_values(om64.x, tdim).astype(np.float32)
    om = O.Mesh(tdim, x32.astype(np.float64), om64.conn)          # what the engine computes on
    phi = phi32.astype(np.float64)
    mesh = cfx.Mesh.from_arrays(tdim, x32, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(V, phi32))
    return O, om, phi, phi32, mesh, V, cd


@pytest.mark.parametrize("tdim,n", [(2, 16), (3, 8)])
def test_f32_cut_rules_normals(oracle, tdim, n):
    import cutfemx_amd as cfx
    O, om, phi, phi32, mesh, V, cd = f32_problem(oracle, tdim, n)
    assert mesh.dtype == np.float32 and cd.dtype == np.float32
    dom = O.classify(om.conn, phi)
    assert np.array_equal(cd.domain(0), dom)
    for sel in ("phi<0", "phi=0"):
        want = O.runtime_quadrature(om, om.conn, phi, dom, sel, 4)
        got = cfx.runtime_quadrature(cd, sel, 4)
        assert got.dtype == np.float32 and got.points.dtype == np.float32 and got.weights.dtype == np.float32
        assert np.array_equal(got.offsets, want.offsets) and np.array_equal(got.parent_map, want.parent_map)
        assert np.abs(got.points - want.points.reshape(got.points.shape)).max() < F32_TOL
        assert rel_err(got.weights, want.weights) < F32_TOL
        assert got.physical_points.dtype == np.float32
        assert np.abs(got.physical_points.T - O.physical_points(om, want)).max() < F32_TOL
    itf = cfx.runtime_quadrature(cd, "phi=0", 4)
    oitf = O.runtime_quadrature(om, om.conn, phi, dom, "phi=0", 4)
    nrm = cfx.normal(cd, itf)
    assert nrm.dtype == np.float32
    assert np.abs(nrm - O.evaluate_normals(om, om.conn, phi, oitf).reshape(nrm.shape)).max() < F32_TOL
    val = cfx.level_set_value(cd, itf)
    assert val.dtype == np.float32 and np.abs(val).max() < 1e-6      # phi = 0 on the interface


def test_f32_level_set_update(oracle):
    import cutfemx_amd as cfx
    O, om, phi, phi32, mesh, V, cd = f32_problem(oracle, 3, 8)
    f = cd.level_sets[0]
    f.values = (phi32 + np.float32(0.05)).astype(np.float32)
    cd.update()
    assert np.array_equal(cd.domain(0), O.classify(om.conn, f.values.astype(np.float64)))


@pytest.mark.parametrize("tdim,n", [(2, 16), (3, 8)])
def test_f32_poisson_system(oracle, tdim, n):
    """demo_poisson.py with float32 containers end to end: matrix, vector, lifting, set_bc, zero_rows, deactivation."""
    import cutfemx_amd as cfx
    from cutfemx_amd import fem, poisson
    O, om, phi, phi32, mesh, V, cd = f32_problem(oracle, tdim, n)
    ref = oracle_poisson(O, om, phi)
    s = poisson.build_forms(V, cd, order=4)          # float32 rules and normals inside
    assert s.volume_rules.dtype == np.float32 and s.normals.dtype == np.float32
    A = fem.create_matrix(s.a, dtype=np.float32)
    assert A.dtype == np.float32
    fem.assemble_matrix(s.a, A=A)
    assert np.array_equal(A.indptr, ref["indptr"]) and np.array_equal(A.indices, ref["indices"])
    assert A.data.dtype == np.float32 and rel_err(A.data, ref["values"]) < F32_TOL
    # accumulate semantics: a second assembly without set_value(0) doubles the entries (one rounding each)
    fem.assemble_matrix(s.a, A=A)
    assert rel_err(A.data, 2.0 * ref["values"]) < 2 * F32_TOL
    A.set_value(0.0)
    fem.assemble_matrix(s.a, A=A)
    assert rel_err(A.data, ref["values"]) < F32_TOL
    b = fem.assemble_vector(s.L, dtype=np.float32)
    assert b.dtype == np.float32 and rel_err(b, ref["b"]) < F32_TOL
    fem.assemble_vector(s.L, b)
    assert rel_err(b, 2.0 * ref["b"]) < 2 * F32_TOL
    # zero_rows before deactivation = the inactive rows (their lone diagonal entry is 0)
    assert np.array_equal(fem.zero_rows(A), ref["inactive"])
    # Dirichlet data in float32
    rng = np.random.default_rng(5)
    markers = (rng.random(om.nnodes) < 0.1).astype(np.int8)
    g = rng.standard_normal(om.nnodes).astype(np.float32)
    x0 = rng.standard_normal(om.nnodes).astype(np.float32)
    b2 = np.zeros(om.nnodes, dtype=np.float32)
    fem.apply_lifting(b2, s.a, markers, g, x0, alpha=0.5)
    want = O.apply_lifting(om, ref["V"], ref["a"], markers, g.astype(np.float64), np.zeros(om.nnodes),
                           x0=x0.astype(np.float64), alpha=0.5)
    assert rel_err(b2, want) < F32_TOL
    fem.set_bc(b2, markers, g, x0, alpha=0.5)
    m = markers.astype(bool)
    assert np.allclose(b2[m], 0.5 * (g[m].astype(np.float64) - x0[m]), rtol=1e-6, atol=1e-7)
    # deactivation
    dom = fem.active_domain(s.a)
    b3 = fem.assemble_vector(s.L, dtype=np.float32)
    fem.deactivate_outside(A, b3, dom)
    vals, bref = ref["values"].copy(), ref["b"].copy()
    O.deactivate(ref["inactive"], ref["indptr"], ref["indices"], vals, bref)
    assert rel_err(A.data, vals) < F32_TOL and rel_err(b3, bref) < F32_TOL
    with pytest.raises(TypeError):
        fem.deactivate_outside(A, np.zeros(om.nnodes), dom)       # float32 matrix with a float64 vector


def test_f32_device_buffers_and_widen(oracle):
    """Caller-owned float32 HBM buffers (torch tensors) and the cfx_widen_f32 helper of the C ABI."""
    import torch

    import cutfemx_amd as cfx
    from cutfemx_amd import _lib, fem, poisson
    O, om, phi, phi32, mesh, V, cd = f32_problem(oracle, 3, 8)
    ref = oracle_poisson(O, om, phi)
    s = poisson.build_forms(V, cd, order=4)
    vals = torch.zeros(ref["values"].size + 7, device="cuda", dtype=torch.float32)
    A = fem.create_matrix(s.a, values=vals)
    assert A.dtype == np.float32
    A.set_value(0.0)
    fem.assemble_matrix(s.a, A=A)
    assert rel_err(vals[:A.nnz].cpu().numpy(), ref["values"]) < F32_TOL and float(vals[A.nnz:].abs().max()) == 0.0
    b = torch.zeros(om.nnodes, device="cuda", dtype=torch.float32)
    fem.assemble_vector(s.L, b)
    assert rel_err(b.cpu().numpy(), ref["b"]) < F32_TOL
    with pytest.raises(TypeError):
        fem.create_matrix(s.a, values=vals, dtype=np.float64)
    src = np.linspace(-3.0, 7.0, 1001, dtype=np.float32)
    p = C.c_void_p()
    _lib.check(_lib.lib().cfx_widen_f32(src.ctypes.data_as(C.c_void_p), C.c_int64(src.size), C.byref(p)))
    assert np.array_equal(_lib.download(p.value, src.size, np.float64), src.astype(np.float64))
    _lib.check(_lib.lib().cfx_device_free(p))


def test_f32_rules_from_arrays(oracle):
    """runintgen-style per-entity rules handed over as float32 arrays (RuntimeQuadrature<float>)."""
    import cutfemx_amd as cfx
    from cutfemx_amd import fem
    O, om, phi, phi32, mesh, V, cd = f32_problem(oracle, 2, 12)
    dom = O.classify(om.conn, phi)
    want = O.runtime_quadrature(om, om.conn, phi, dom, "phi<0", 3)
    p32, w32 = want.points.astype(np.float32), want.weights.astype(np.float32)
    r = cfx.RuntimeQuadratureRules.from_arrays(mesh, p32, w32, want.offsets, want.parent_map)
    assert r.dtype == np.float32 and np.array_equal(r.points.ravel(), p32.ravel()) and np.array_equal(r.weights, w32)
    # mass matrix over those rules = the oracle's on the widened rule arrays
    orules = O.Rules(2, p32.astype(np.float64), w32.astype(np.float64), want.offsets, want.parent_map)
    oV = O.Space(om.conn, om.nnodes, 1)
    oa = [O.Integral(O.CELL, O.K_MASS, rules=orules)]
    ip, ix = O.create_sparsity(om, oV, oa)
    ref = O.assemble_matrix(om, oV, oa, ip, ix)
    A = fem.assemble_matrix(fem.form([fem.Integral(fem.MASS, rules=r)], V), A=None)
    assert rel_err(A.data, ref) < 1e-12           # float64 matrix from float32 rules: <double, float> mixing
