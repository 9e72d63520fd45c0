// Drives the hot path through the C++ facade (include/cutfemx_amd.hpp), the way
// a C++ user of cutfemx::cut / runtime_quadrature / assemble_matrix would, and
// dumps the results for tests/test_gpu_cpp_facade.py to compare with the oracle.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>

#include "cutfemx_amd.hpp"

namespace cfx = cutfemx_amd;

template <typename T>
static void dump(std::ofstream& f, const std::vector<T>& v)
{
  const std::int64_t n = static_cast<std::int64_t>(v.size());
  f.write(reinterpret_cast<const char*>(&n), sizeof(n));
  f.write(reinterpret_cast<const char*>(v.data()), sizeof(T) * v.size());
}

int main(int argc, char** argv)
{
  const int tdim = argc > 1 ? std::atoi(argv[1]) : 3;
  const int n = argc > 2 ? std::atoi(argv[2]) : 8;
  const char* out_path = argc > 3 ? argv[3] : "facade.bin";
  try
  {
    if (cfx_init(0) != CFX_OK) throw std::runtime_error(cfx_last_error());
    cfx::Mesh mesh = cfx::Mesh::create_box(tdim, n);
    const std::vector<double> x = mesh.x();
    const std::vector<std::int32_t> conn = mesh.connectivity();
    // phi = |x - c| - R, c = (0.47, 0.43, 0.41), R = 0.31 (python/tests/test_cut_api.py:36-52)
    const double c[3] = {0.47, 0.43, 0.41};
    std::vector<double> phi(static_cast<std::size_t>(mesh.num_nodes));
    for (std::int64_t v = 0; v < mesh.num_nodes; ++v)
    {
      double r2 = 0.0;
      for (int d = 0; d < tdim; ++d) r2 += (x[3 * v + d] - c[d]) * (x[3 * v + d] - c[d]);
      phi[v] = std::sqrt(r2) - 0.31;
    }
    const double* values[1] = {phi.data()};
    cfx::CutData cd = cfx::cut(mesh, conn, tdim + 1, mesh.num_nodes, values);

    // error classes of the reference (cut.cpp:97-107,166-167)
    bool threw = false;
    try { (void)cfx::locate_entities(cd, "psi<0"); } catch (const std::invalid_argument&) { threw = true; }
    if (!threw) throw std::runtime_error("expected std::invalid_argument for an unknown level-set name");
    threw = false;
    try { (void)cfx::runtime_quadrature(cd, "phi<0", 4, "algoim"); } catch (const std::invalid_argument&) { threw = true; }
    if (!threw) throw std::runtime_error("expected std::invalid_argument for an unsupported backend");

    const std::vector<std::int32_t> inside = cfx::locate_entities(cd, "phi<0");
    cfx::RuntimeQuadrature vol = cfx::runtime_quadrature(cd, "phi<0", 4);
    cfx::RuntimeQuadrature itf = cfx::runtime_quadrature(cd, "phi=0", 4);
    const std::vector<double> normals = cfx::level_set::evaluate_normals(cd, 0, itf);
    const std::vector<std::int32_t> ghost = cfx::ghost_penalty_facets(cd, "phi<0");

    cfx::fem::FunctionSpace V = cfx::fem::FunctionSpace::create(mesh, 1, 1, mesh.num_nodes, conn, tdim + 1);
    std::vector<cfx::fem::Integral> ai(3), Li(2);
    ai[0] = {CFX_CELL, CFX_K_STIFFNESS, inside, &vol, {}, 0, {}, 0};
    ai[1] = {CFX_CELL, CFX_K_NITSCHE, {}, &itf, normals, tdim, {40.0}, 0};
    ai[2] = {CFX_INTERIOR_FACET, CFX_K_GHOST_GRADJUMP, ghost, nullptr, {}, 0, {0.1}, 0};
    Li[0] = {CFX_CELL, CFX_L_SOURCE, inside, &vol, {}, 0, {double(CFX_F_POISSON_RHS), 1.0}, 4};
    Li[1] = {CFX_CELL, CFX_L_NITSCHE_RHS, {}, &itf, normals, tdim, {40.0, double(CFX_F_SINPROD), 1.0}, 0};
    cfx::fem::Form a = cfx::fem::Form::create(V, 2, ai);
    cfx::fem::Form L = cfx::fem::Form::create(V, 1, Li);

    if (argc > 99)
    {
      // (type-checked, not run: the complex64 overloads, the block merge and a run-time integrand)
      std::vector<std::complex<float>> Ac, bc;
      cfx::fem::SparsityPattern spc = cfx::fem::create_sparsity_pattern(a);
      cfx::fem::assemble_matrix(std::span<std::complex<float>>(Ac), a, spc);
      cfx::fem::assemble_vector(std::span<std::complex<float>>(bc), L);
      std::vector<double> Ad;
      const cfx::fem::MergedCSR merged = cfx::fem::merge_blocks({{cfx::fem::MatrixBlock{std::span<double>(Ad), &spc}}});
      (void)merged;
      (void)cfx::fem::register_integrand("user_k", "__device__ void user_k(double*, const double*, const double*, const double*, int, const double*, const double*, const double*) {}", 2);
    }
    cfx::fem::SparsityPattern sp = cfx::fem::create_sparsity_pattern(a);
    std::vector<double> A(static_cast<std::size_t>(sp.num_nonzeros()), 0.0), b(static_cast<std::size_t>(mesh.num_nodes), 0.0);
    cfx::fem::assemble_matrix(A, a, sp);
    cfx::fem::assemble_vector(b, L);
    cfx::fem::ActiveDomain dom = cfx::fem::active_domain(a);
    cfx::fem::deactivate_outside(A, sp, b, dom);

    std::ofstream f(out_path, std::ios::binary);
    dump(f, cd.domain());
    dump(f, inside);
    dump(f, vol.offsets());
    dump(f, vol.parent_map());
    dump(f, vol.weights());
    dump(f, itf.weights());
    dump(f, ghost);
    dump(f, sp.row_ptr());
    dump(f, sp.cols());
    dump(f, A);
    dump(f, b);
    dump(f, dom.inactive_dofs);

    // facets as hosts (cut.cpp:788-830): the boundary facets against the plane x = 0.51
    std::vector<double> plane(static_cast<std::size_t>(mesh.num_nodes));
    for (std::int64_t v = 0; v < mesh.num_nodes; ++v) plane[v] = x[3 * v] - 0.51;
    const double* pvalues[1] = {plane.data()};
    const cfx::FacetRows ext = cfx::exterior_facets(mesh);
    cfx::CutData fcd = cfx::cut(mesh, conn, tdim + 1, mesh.num_nodes, pvalues, ext);
    if (fcd.tdim != tdim - 1 || fcd.num_local_cells != (std::int32_t)ext.size())
      throw std::runtime_error("facet-hosted CutData reports the wrong host dimension / count");
    const std::vector<std::int32_t> cut_facets = cfx::locate_entities(fcd, "phi=0");
    cfx::RuntimeQuadrature frules = cfx::runtime_quadrature(fcd, "phi<0", 2);
    cfx::RuntimeQuadrature fcell = cfx::facet_rules_to_cells(frules, 0);
    cfx::RuntimeQuadrature fstd = cfx::full_facet_rules(fcd, "phi<0", 2);
    dump(f, ext.rows);
    dump(f, cut_facets);
    dump(f, frules.parent_map());
    dump(f, frules.weights());
    dump(f, frules.physical_points());
    dump(f, fcell.parent_map());
    dump(f, fcell.points());
    dump(f, fstd.weights());
    std::printf("facade ok: %lld cells, %zu inside, %zu cut rules, nnz %lld\n", (long long)mesh.num_cells,
                inside.size(), vol.num_rules(), (long long)sp.num_nonzeros());
    return 0;
  }
  catch (const std::exception& e)
  {
    std::cerr << "facade FAILED: " << e.what() << "\n";
    return 1;
  }
}
