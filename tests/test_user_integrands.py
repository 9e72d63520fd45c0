"""Integrands supplied by the caller and compiled at run time with hipRTC (cfx_integrand_register): the GPU counterpart of
the reference's runtime-generated tabulate_tensor kernels (cpp/dolfinx_custom_data/fem/Form.h:59-75,
python/cutfemx/_runintgen_adapter.py:181-217).  Compilation targets gfx950 explicitly and needs no GPU; the parity
tests (the registered source of the stiffness / Nitsche / source integrands against the built-in ids) run on the GPU."""
import numpy as np
import pytest

from helpers import level_set_values, rel_err

STIFFNESS_SRC = r"""
__device__ void user_stiffness(double* A, const double* w, const double* c, const double* coordinate_dofs, int nq,
                               const double* points, const double* weights, const double* point_data)
{
  double K[CFX_TDIM][CFX_TDIM];
  (void)cfx_inverse_jacobian(coordinate_dofs, K);
  for (int q = 0; q < nq; ++q)
  {
    double N[CFX_ND], dN[CFX_ND][CFX_TDIM], g[CFX_ND][CFX_TDIM];
    cfx_tabulate(points + q * CFX_TDIM, N, dN);
    for (int i = 0; i < CFX_ND; ++i)
      for (int d = 0; d < CFX_TDIM; ++d)
      {
        double v = 0.0;
        for (int t = 0; t < CFX_TDIM; ++t) v += dN[i][t] * K[t][d];
        g[i][d] = v;
      }
    for (int i = 0; i < CFX_ND; ++i)
      for (int j = 0; j < CFX_ND; ++j)
      {
        double v = 0.0;
        for (int d = 0; d < CFX_TDIM; ++d) v += g[i][d] * g[j][d];
        A[i * CFX_ND + j] += weights[q] * v;
      }
  }
}
"""

# -dn(u) v - dn(v) u + gamma / h u v on the interface; point_data = unit normals (the built-in CFX_K_NITSCHE)
NITSCHE_SRC = r"""
__device__ void user_nitsche(double* A, const double* w, const double* c, const double* coordinate_dofs, int nq,
                             const double* points, const double* weights, const double* point_data)
{
  double K[CFX_TDIM][CFX_TDIM];
  (void)cfx_inverse_jacobian(coordinate_dofs, K);
  const double h = cfx_cell_diameter(coordinate_dofs), gamma = c[0];
  for (int q = 0; q < nq; ++q)
  {
    double N[CFX_ND], dN[CFX_ND][CFX_TDIM], dn[CFX_ND];
    cfx_tabulate(points + q * CFX_TDIM, N, dN);
    const double* nrm = point_data + q * CFX_TDIM;
    for (int i = 0; i < CFX_ND; ++i)
    {
      double v = 0.0;
      for (int d = 0; d < CFX_TDIM; ++d)
      {
        double gd = 0.0;
        for (int t = 0; t < CFX_TDIM; ++t) gd += dN[i][t] * K[t][d];
        v += gd * nrm[d];
      }
      dn[i] = v;
    }
    for (int i = 0; i < CFX_ND; ++i)
      for (int j = 0; j < CFX_ND; ++j)
        A[i * CFX_ND + j] += weights[q] * (-dn[j] * N[i] - dn[i] * N[j] + gamma / h * N[i] * N[j]);
  }
}
"""

# (f, v) with f = 1 scaled by c[1]: the built-in CFX_L_SOURCE with field id CFX_F_ONE
SOURCE_SRC = r"""
__device__ void user_source(double* b, const double* w, const double* c, const double* coordinate_dofs, int nq,
                            const double* points, const double* weights, const double* point_data)
{
  for (int q = 0; q < nq; ++q)
  {
    double N[CFX_ND], dN[CFX_ND][CFX_TDIM];
    cfx_tabulate(points + q * CFX_TDIM, N, dN);
    for (int i = 0; i < CFX_ND; ++i) b[i] += c[1] * weights[q] * N[i];
  }
}
"""


def test_integrand_sources_compile_for_gfx950_without_a_gpu():
    from cutfemx_amd import fem
    ids = [fem.register_integrand("user_stiffness", STIFFNESS_SRC, rank=2),
           fem.register_integrand("user_nitsche", NITSCHE_SRC, rank=2),
           fem.register_integrand("user_source", SOURCE_SRC, rank=1)]
    assert all(i >= 1000 for i in ids) and len(set(ids)) == 3
    for kid in ids:                                  # the other (tdim, dofs per cell) variants
        for tdim, nd in ((2, 3), (2, 6), (3, 10)):
            fem.compile_integrand(kid, tdim, nd)


def test_a_source_that_does_not_compile_is_refused_with_the_compiler_log():
    from cutfemx_amd import fem
    with pytest.raises(ValueError, match="does not compile"):
        fem.register_integrand("broken", "__device__ void broken(double* A) { this is not C++; }", rank=2)
    with pytest.raises(ValueError, match="C identifier"):
        fem.register_integrand("2bad", "", rank=2)


@pytest.mark.gpu
@pytest.mark.parametrize("tdim,n,degree", [(3, 8, 1), (2, 16, 1), (2, 10, 2), (3, 5, 2)])
def test_registered_sources_reproduce_the_builtin_integrands(oracle, tdim, n, degree, monkeypatch):
    import cutfemx_amd as cfx
    from cutfemx_amd import fem
    om = oracle.mesh_box(tdim, n)
    phi = level_set_values(om.x, tdim)
    dofmap, ndofs = cfx.lagrange_dofmap(tdim, om.conn, om.nnodes, degree)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, degree, dofmap=None if degree == 1 else dofmap, ndofs=ndofs)
    Vphi = V if degree == 1 else cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(Vphi, phi))
    inside = cfx.locate_entities(cd, "phi<0")
    vol, itf = cfx.runtime_quadrature(cd, "phi<0", 4), cfx.runtime_quadrature(cd, "phi=0", 4)
    nrm = cfx.normal(cd, itf)
    k_stiff = fem.register_integrand("user_stiffness", STIFFNESS_SRC, rank=2)
    k_nit = fem.register_integrand("user_nitsche", NITSCHE_SRC, rank=2)
    k_src = fem.register_integrand("user_source", SOURCE_SRC, rank=1)
    qd = 2 * (degree - 1)

    def forms(ks, kn, kl):
        a = fem.form([fem.Integral(ks, cells=inside, rules=vol, qdegree=qd),
                      fem.Integral(kn, rules=itf, point_data=nrm, params=(40.0,))], V, rank=2)
        L = fem.form([fem.Integral(kl, cells=inside, rules=vol, params=(fem.F_ONE, 2.5), qdegree=2)], V, rank=1)
        return a, L
    a_ref, L_ref = forms(fem.STIFFNESS, fem.NITSCHE, fem.SOURCE)
    a_usr, L_usr = forms(k_stiff, k_nit, k_src)
    A_ref, A_usr = fem.assemble_matrix(a_ref), fem.assemble_matrix(a_usr)
    assert np.array_equal(A_ref.indptr, A_usr.indptr) and np.array_equal(A_ref.indices, A_usr.indices)
    assert rel_err(A_usr.data, A_ref.data) < 1e-13
    assert rel_err(fem.assemble_vector(L_usr), fem.assemble_vector(L_ref)) < 1e-13
    # local tensors of single entities (tabulate_entity), standard and runtime
    for use_rule in (False, True):
        assert rel_err(fem.tabulate_entity(a_usr, 0, 3, use_rule), fem.tabulate_entity(a_ref, 0, 3, use_rule)) < 1e-13
    assert rel_err(fem.tabulate_entity(a_usr, 1, 2, True), fem.tabulate_entity(a_ref, 1, 2, True)) < 1e-13
    # the entity-parallel scatter serves them too
    monkeypatch.setenv("CFX_ASSEMBLY", "atomic")
    a2, L2 = forms(k_stiff, k_nit, k_src)
    assert rel_err(fem.assemble_matrix(a2).data, A_ref.data) < 1e-13
    assert rel_err(fem.assemble_vector(L2), fem.assemble_vector(L_ref)) < 1e-13
