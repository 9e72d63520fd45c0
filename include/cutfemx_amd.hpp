// cutfemx_amd.hpp -- header-only C++ facade over the C ABI (cutfemx_amd.h).
//
// Mirrors the C++ surface of the reference for the hot path with the same
// names, argument meaning and exception classes:
//   cutfemx::cut / update / locate_entities / runtime_quadrature
//                                   cpp/cutfemx/cut/cut.h:104-181
//   cutfemx::RuntimeQuadrature      cpp/cutfemx/cut/runtime_quadrature.h:43-232
//   dolfinx_custom_data::fem::create_sparsity_pattern / assemble_matrix /
//   assemble_vector                 cpp/dolfinx_custom_data/fem/assembler.h:252-262,567-592,690-703
//   cutfemx::fem::active_domain / deactivate_outside
//                                   cpp/cutfemx/fem/deactivate.h:387-418
// DOLFINx objects are replaced by the flat arrays the reference reads from them
// (geometry.x stride 3, int32 dofmaps, dof values); JIT kernel pointers by
// integrand ids.  Status codes become the exceptions the reference throws:
// std::invalid_argument, std::runtime_error, std::out_of_range.
#pragma once

#include <complex>
#include <algorithm>
#include <cstdint>
#include <iterator>
#include <memory>
#include <span>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "cutfemx_amd.h"

namespace cutfemx_amd
{

inline void check(int rc)
{
  if (rc == CFX_OK) return;
  const std::string msg = cfx_last_error();
  if (rc == CFX_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
  if (rc == CFX_ERR_OUT_OF_RANGE) throw std::out_of_range(msg);
  throw std::runtime_error(msg);
}

template <typename T>
inline std::vector<T> download(const T* dev, std::int64_t n)
{
  std::vector<T> v(static_cast<std::size_t>(n));
  if (n > 0) check(cfx_copy(v.data(), dev, sizeof(T) * static_cast<std::size_t>(n)));
  return v;
}

namespace detail
{
template <typename H, int (*Destroy)(H)>
struct Handle
{
  H h = nullptr;
  Handle() = default;
  explicit Handle(H handle) : h(handle) {}
  Handle(const Handle&) = delete;
  Handle& operator=(const Handle&) = delete;
  Handle(Handle&& o) noexcept : h(std::exchange(o.h, nullptr)) {}
  Handle& operator=(Handle&& o) noexcept
  {
    if (this != &o) { reset(); h = std::exchange(o.h, nullptr); }
    return *this;
  }
  ~Handle() { reset(); }
  void reset() { if (h) { Destroy(h); h = nullptr; } }
};
} // namespace detail

/// Background simplex mesh: the cutcells::MeshView of cut.cpp:500-538.
struct Mesh
{
  detail::Handle<cfx_mesh_t, cfx_mesh_destroy> handle;
  int tdim = 0, gdim = 0;
  std::int64_t num_nodes = 0, num_cells = 0;

  /// x: geometry.x (stride 3); connectivity: geometry dofmap [ncells x cell_stride]
  static Mesh create(int tdim, std::span<const double> x, std::span<const std::int32_t> connectivity,
                     int cell_stride)
  {
    Mesh m;
    cfx_mesh_t h = nullptr;
    const std::int64_t ncells = static_cast<std::int64_t>(connectivity.size()) / cell_stride;
    check(cfx_mesh_create(tdim, tdim, static_cast<std::int64_t>(x.size()) / 3, x.data(), ncells,
                          connectivity.data(), cell_stride, &h));
    m.handle = detail::Handle<cfx_mesh_t, cfx_mesh_destroy>(h);
    check(cfx_mesh_info(h, &m.tdim, &m.gdim, &m.num_nodes, &m.num_cells, nullptr, nullptr));
    return m;
  }
  static Mesh create_box(int tdim, int n)
  {
    Mesh m;
    cfx_mesh_t h = nullptr;
    check(cfx_mesh_create_box(tdim, n, &h));
    m.handle = detail::Handle<cfx_mesh_t, cfx_mesh_destroy>(h);
    check(cfx_mesh_info(h, &m.tdim, &m.gdim, &m.num_nodes, &m.num_cells, nullptr, nullptr));
    return m;
  }
  std::vector<double> x() const
  {
    const double* p;
    check(cfx_mesh_info(handle.h, nullptr, nullptr, nullptr, nullptr, &p, nullptr));
    return download(p, num_nodes * 3);
  }
  std::vector<std::int32_t> connectivity() const
  {
    const std::int32_t* p;
    check(cfx_mesh_info(handle.h, nullptr, nullptr, nullptr, nullptr, nullptr, &p));
    return download(p, num_cells * (tdim + 1));
  }
};

/// cutcells::CutOptions as filled by make_cut_options (wrappers/cut.cpp:117-140)
struct CutOptions
{
  int cut_approximation_order = 1;
  int max_refinement_iterations = 8;
  int edge_max_depth = 20;
};

/// cutfemx::CutData<T> (cut.h:31-102): move-only, owns the classification.
struct CutData
{
  detail::Handle<cfx_cut_t, cfx_cut_destroy> handle;
  int gdim = 0, tdim = 0;
  std::int32_t num_local_cells = 0;
  std::vector<std::string> level_set_names;

  /// ParentCellClassification::domain(level_set, cell) for every cell
  std::vector<std::int8_t> domain(int level_set = 0) const
  {
    const std::int8_t* p;
    check(cfx_cut_domain(handle.h, level_set, &p));
    return download(p, num_local_cells);
  }
};

/// cut(): cut.cpp:743-786.  One P1/P2 Lagrange dofmap shared by the level sets;
/// level_set_values[k] are the dof values of level set k (host or device pointers).
inline CutData cut(const Mesh& mesh, std::span<const std::int32_t> ls_dofmap, int ls_ndofs_cell,
                   std::int64_t ls_ndofs, std::span<const double* const> level_set_values,
                   const CutOptions& options = CutOptions{})
{
  if (level_set_values.empty())
    throw std::invalid_argument("cutfemx.cut requires at least one level-set function"); // cut.cpp:97-107
  cfx_cut_options opt{options.cut_approximation_order, options.max_refinement_iterations, options.edge_max_depth, 0};
  cfx_cut_t h = nullptr;
  check(cfx_cut_create(mesh.handle.h, static_cast<int>(level_set_values.size()), ls_dofmap.data(), ls_ndofs_cell,
                       ls_ndofs, level_set_values.data(), &opt, &h));
  CutData cd;
  cd.handle = detail::Handle<cfx_cut_t, cfx_cut_destroy>(h);
  std::int64_t nc = 0;
  int nls = 0;
  check(cfx_cut_info(h, &cd.tdim, &cd.gdim, &nc, &nls));
  cd.num_local_cells = static_cast<std::int32_t>(nc);
  for (int k = 0; k < nls; ++k) cd.level_set_names.push_back(k == 0 ? "phi" : "phi" + std::to_string(k));
  return cd;
}

/// cut(level_set, entities, entity_dim): cut.cpp:788-830 with entity_dim == tdim -- only the listed
/// background cells are candidates (facets as hosts: the FacetRows overload below)
inline CutData cut(const Mesh& mesh, std::span<const std::int32_t> ls_dofmap, int ls_ndofs_cell,
                   std::int64_t ls_ndofs, std::span<const double* const> level_set_values,
                   std::span<const std::int32_t> entities, int entity_dim, const CutOptions& options = CutOptions{})
{
  CutData cd = cut(mesh, ls_dofmap, ls_ndofs_cell, ls_ndofs, level_set_values, options);
  if (entity_dim != cd.tdim) throw std::invalid_argument("cut: a cell subset has entity_dim == tdim (facets are passed as FacetRows)");
  check(cfx_cut_restrict(cd.handle.h, entities.data(), static_cast<std::int64_t>(entities.size())));
  return cd;
}

/// Facets as integration rows: (cell, local facet) for exterior facets (width 2) or
/// (cell0, local facet0, cell1, local facet1) for interior facets (width 4) --
/// facet_integration_rows(), python/cutfemx/wrappers/cut.cpp:54-115.
struct FacetRows
{
  std::vector<std::int32_t> rows;
  int width = 4;
  std::size_t size() const { return rows.size() / static_cast<std::size_t>(width); }
};

/// exterior_facet_indices() + facet_integration_rows(): the boundary facets, ascending
inline FacetRows exterior_facets(const Mesh& mesh)
{
  std::int32_t* p = nullptr;
  std::int64_t n = 0;
  check(cfx_exterior_facets(mesh.handle.h, &p, &n));
  FacetRows out{download(p, 2 * n), 2};
  check(cfx_device_free(p));
  return out;
}

/// cut(level_set, facets, tdim - 1): cut.cpp:540-591, 788-830 -- the facets host the cut.
/// facet_ids (empty: positions) are the numbers locate_entities / parent_map answer with;
/// entity_geometry (empty: cell0's vertices other than the opposite one, ascending) is the
/// host vertex order of entities_to_geometry() (cut.cpp:567-569).
inline CutData cut(const Mesh& mesh, std::span<const std::int32_t> ls_dofmap, int ls_ndofs_cell,
                   std::int64_t ls_ndofs, std::span<const double* const> level_set_values, const FacetRows& facets,
                   std::span<const std::int32_t> facet_ids = {}, std::span<const std::int32_t> entity_geometry = {},
                   const CutOptions& options = CutOptions{})
{
  if (level_set_values.empty())
    throw std::invalid_argument("cutfemx.cut requires at least one level-set function");
  cfx_cut_options opt{options.cut_approximation_order, options.max_refinement_iterations, options.edge_max_depth, 0};
  cfx_cut_t h = nullptr;
  check(cfx_cut_create_facets(mesh.handle.h, static_cast<std::int64_t>(facets.size()),
                              facet_ids.empty() ? nullptr : facet_ids.data(), facets.rows.data(), facets.width,
                              entity_geometry.empty() ? nullptr : entity_geometry.data(),
                              static_cast<int>(level_set_values.size()), ls_dofmap.data(), ls_ndofs_cell, ls_ndofs,
                              level_set_values.data(), &opt, &h));
  CutData cd;
  cd.handle = detail::Handle<cfx_cut_t, cfx_cut_destroy>(h);
  std::int64_t nc = 0;
  int nls = 0;
  check(cfx_cut_info(h, &cd.tdim, &cd.gdim, &nc, &nls)); // tdim = mesh tdim - 1, nc = number of hosts
  cd.num_local_cells = static_cast<std::int32_t>(nc);
  for (int k = 0; k < nls; ++k) cd.level_set_names.push_back(k == 0 ? "phi" : "phi" + std::to_string(k));
  return cd;
}

/// update(): cut.cpp:845-868 -- re-classify from the current values (same pointers when empty)
inline void update(CutData& cut_data, std::span<const double* const> level_set_values = {})
{
  check(cfx_cut_update(cut_data.handle.h, level_set_values.empty() ? nullptr : level_set_values.data()));
}

/// locate_entities(): cut.cpp:877-924 -- ascending background cell ids
inline std::vector<std::int32_t> locate_entities(const CutData& cut_data, std::string_view ls_part)
{
  const std::int32_t* p;
  std::int64_t n;
  check(cfx_locate_entities(cut_data.handle.h, std::string(ls_part).c_str(), &p, &n));
  return download(p, n);
}

/// cutfemx::RuntimeQuadrature<T> (runtime_quadrature.h:43-232).  Arrays stay in HBM;
/// the accessors download them.
struct RuntimeQuadrature
{
  detail::Handle<cfx_rules_t, cfx_rules_destroy> handle;
  cfx_rules_view view{};

  int tdim() const { return view.tdim; }
  int gdim() const { return view.gdim; }
  std::size_t num_points() const { return static_cast<std::size_t>(view.nq); }
  std::size_t num_rules() const { return static_cast<std::size_t>(view.nr); }
  std::vector<double> points() const { return download(view.points, view.nq * view.tdim); }
  std::vector<double> weights() const { return download(view.weights, view.nq); }
  std::vector<std::int32_t> offsets() const { return download(view.offsets, view.nr + 1); }
  std::vector<std::int32_t> parent_map() const { return download(view.parent_map, view.nr); }
  /// physical_points(): runtime_quadrature.h:102-221, row-major (nq, gdim)
  std::vector<double> physical_points() const
  {
    std::vector<double> out(static_cast<std::size_t>(view.nq * view.gdim));
    check(cfx_rules_physical_points(handle.h, out.data()));
    return out;
  }
};

/// runtime_quadrature(): cut.cpp:1311-1335
inline RuntimeQuadrature runtime_quadrature(const CutData& cut_data, std::string_view ls_part, int order,
                                            std::string_view backend = "straight")
{
  cfx_rules_t h = nullptr;
  check(cfx_runtime_quadrature(cut_data.handle.h, std::string(ls_part).c_str(), order, std::string(backend).c_str(),
                               &h));
  RuntimeQuadrature r;
  r.handle = detail::Handle<cfx_rules_t, cfx_rules_destroy>(h);
  check(cfx_rules_view_get(h, &r.view));
  return r;
}

/// whole-facet rules over the hosts of a facet-hosted cut matching ls_part (empty: all): the standard facets
/// of a mixed [facets, rules] measure (python/tests/test_cut_api.py:527-560)
inline RuntimeQuadrature full_facet_rules(const CutData& facet_cut, std::string_view ls_part, int order)
{
  cfx_rules_t h = nullptr;
  const std::string sel(ls_part);
  check(cfx_full_facet_rules(facet_cut.handle.h, sel.empty() ? nullptr : sel.c_str(), order, &h));
  RuntimeQuadrature r;
  r.handle = detail::Handle<cfx_rules_t, cfx_rules_destroy>(h);
  check(cfx_rules_view_get(h, &r.view));
  return r;
}

/// facet_runtime_quadrature_payload / interior_facet_runtime_quadrature_payload
/// (python/cutfemx/_runintgen_adapter.py:605-680): facet-hosted rules seen from cell `side` of their facets
inline RuntimeQuadrature facet_rules_to_cells(const RuntimeQuadrature& facet_rules, int side = 0)
{
  cfx_rules_t h = nullptr;
  check(cfx_facet_rules_to_cells(facet_rules.handle.h, side, &h));
  RuntimeQuadrature r;
  r.handle = detail::Handle<cfx_rules_t, cfx_rules_destroy>(h);
  check(cfx_rules_view_get(h, &r.view));
  return r;
}

/// runtime_quadratures(): cut.cpp:1357-1406
inline std::vector<std::pair<std::string, RuntimeQuadrature>>
runtime_quadratures(const CutData& cut_data, std::span<const std::string> ls_parts, int order,
                    std::string_view backend = "straight")
{
  std::vector<std::pair<std::string, RuntimeQuadrature>> out;
  for (const std::string& p : ls_parts) out.emplace_back(p, runtime_quadrature(cut_data, p, order, backend));
  return out;
}

/// ghost_penalty_facets() + facet_integration_rows(): python/cutfemx/cut.py:340-380,
/// wrappers/cut.cpp:54-115 -- rows (cell0, lf0, cell1, lf1), cell0 < cell1
inline std::vector<std::int32_t> ghost_penalty_facets(const CutData& cut_data, std::string_view selector)
{
  const std::int32_t* p;
  std::int64_t n;
  check(cfx_ghost_penalty_facets(cut_data.handle.h, std::string(selector).c_str(), &p, &n));
  return download(p, 4 * n);
}

namespace level_set
{
/// evaluate_normals(): level_set/normal.h:39-187 -- row-major (nq, gdim), always double
inline std::vector<double> evaluate_normals(const CutData& cut_data, int level_set, const RuntimeQuadrature& rules,
                                            double sign = 1.0)
{
  std::vector<double> out(rules.num_points() * static_cast<std::size_t>(rules.gdim()));
  check(cfx_evaluate_normals(cut_data.handle.h, level_set, rules.handle.h, sign, out.data()));
  return out;
}
/// evaluate_values(): level_set/value.h:34-119
inline std::vector<double> evaluate_values(const CutData& cut_data, int level_set, const RuntimeQuadrature& rules)
{
  std::vector<double> out(rules.num_points());
  check(cfx_evaluate_values(cut_data.handle.h, level_set, rules.handle.h, out.data()));
  return out;
}
} // namespace level_set

namespace fem
{

struct FunctionSpace
{
  detail::Handle<cfx_space_t, cfx_space_destroy> handle;
  int degree = 1, bs = 1, ndofs_cell = 0;
  std::int64_t ndofs = 0;
  static FunctionSpace create(const Mesh& mesh, int degree, int bs, std::int64_t ndofs,
                              std::span<const std::int32_t> dofmap, int ndofs_cell)
  {
    cfx_space_t h = nullptr;
    check(cfx_space_create(mesh.handle.h, degree, bs, ndofs, dofmap.data(), ndofs_cell, &h));
    FunctionSpace V;
    V.handle = detail::Handle<cfx_space_t, cfx_space_destroy>(h);
    V.degree = degree; V.bs = bs; V.ndofs = ndofs; V.ndofs_cell = ndofs_cell;
    return V;
  }
};

/// One integral: the integral_data of Form.h:46-89 with the kernel chosen by id.
struct Integral
{
  int type = CFX_CELL;
  int kernel = CFX_K_STIFFNESS;
  std::span<const std::int32_t> entities{};    // cells, or (c0,lf0,c1,lf1) rows
  const RuntimeQuadrature* rules = nullptr;    // runtime entities of the same measure
  std::span<const double> point_data{};        // per-point coefficients, row-major
  int point_stride = 0;
  std::vector<double> params{};
  int quadrature_degree = 2;
  std::span<const double> coefficient{};       // dof values of a CFX_F_COEFFICIENT field
};

struct Form
{
  detail::Handle<cfx_form_t, cfx_form_destroy> handle;
  int rank = 2;
  static Form create(const FunctionSpace& V, int rank, std::span<const Integral> integrals)
  {
    return create_impl(V, nullptr, rank, integrals);
  }
  /// bilinear form with different test and trial spaces (Form::function_spaces() = {V_test, V_trial}, Form.h:119-178;
  /// assemble_matrix_impl.h:68-189): the off-diagonal blocks of a Stokes system (CFX_K_DIV_TEST / CFX_K_DIV_TRIAL),
  /// mass / stiffness between spaces of different degree
  static Form create(const FunctionSpace& V_test, const FunctionSpace& V_trial, std::span<const Integral> integrals)
  {
    return create_impl(V_test, &V_trial, 2, integrals);
  }

private:
  static Form create_impl(const FunctionSpace& V, const FunctionSpace* V_trial, int rank, std::span<const Integral> integrals)
  {
    std::vector<cfx_integral> raw(integrals.size());
    for (std::size_t i = 0; i < integrals.size(); ++i)
    {
      const Integral& in = integrals[i];
      cfx_integral& r = raw[i];
      r = cfx_integral{};
      r.type = in.type; r.kernel = in.kernel; r.qdegree = in.quadrature_degree; r.point_stride = in.point_stride;
      r.entities = in.entities.data();
      r.n_entities = static_cast<std::int64_t>(in.entities.size()) / (in.type == CFX_INTERIOR_FACET ? 4 : 1);
      r.rules = in.rules ? in.rules->handle.h : nullptr;
      r.point_data = in.point_data.empty() ? nullptr : in.point_data.data();
      r.coefficient = in.coefficient.empty() ? nullptr : in.coefficient.data();
      for (std::size_t k = 0; k < in.params.size() && k < 8; ++k) r.params[k] = in.params[k];
    }
    cfx_form_t h = nullptr;
    if (V_trial) check(cfx_form_create2(V.handle.h, V_trial->handle.h, static_cast<int>(raw.size()), raw.data(), &h));
    else check(cfx_form_create(V.handle.h, rank, static_cast<int>(raw.size()), raw.data(), &h));
    Form a;
    a.handle = detail::Handle<cfx_form_t, cfx_form_destroy>(h);
    a.rank = rank;
    return a;
  }
};

/// la::SparsityPattern after finalize(): row_ptr (int64) / cols (int32)
struct SparsityPattern
{
  detail::Handle<cfx_pattern_t, cfx_pattern_destroy> handle;
  cfx_pattern_view view{};
  std::int64_t num_rows() const { return view.nrows; }
  std::int64_t num_cols() const { return view.ncols; }
  std::int64_t num_nonzeros() const { return view.nnz; }
  std::vector<std::int64_t> row_ptr() const { return download(view.indptr, view.nrows + 1); }
  std::vector<std::int32_t> cols() const { return download(view.indices, view.nnz); }
};

/// create_sparsity_pattern(): assembler.h:567-592 (incl. the all-rows diagonal :538-560)
inline SparsityPattern create_sparsity_pattern(const Form& a)
{
  cfx_pattern_t h = nullptr;
  check(cfx_create_sparsity(a.handle.h, &h));
  SparsityPattern p;
  p.handle = detail::Handle<cfx_pattern_t, cfx_pattern_destroy>(h);
  check(cfx_pattern_view_get(h, &p.view));
  return p;
}

/// assemble_matrix(): assembler.h:690-703 -- accumulates into `values` (host or device,
/// length nnz); bc0/bc1 are the int8 dof markers of assembler.h:643-683 (may be empty)
inline void assemble_matrix(std::span<double> values, const Form& a, const SparsityPattern& pattern,
                            std::span<const std::int8_t> bc0 = {}, std::span<const std::int8_t> bc1 = {})
{
  check(cfx_assemble_matrix(a.handle.h, pattern.handle.h, bc0.empty() ? nullptr : bc0.data(),
                            bc1.empty() ? nullptr : bc1.data(), values.data()));
}

/// A.set_value(0) followed by assemble_matrix(A.mat_add_values(), a, bcs) as one call (how a matrix is assembled
/// once per step, python/demo/demo_poisson.py:40-60): `values` is overwritten
inline void assemble_matrix_zeroed(std::span<double> values, const Form& a, const SparsityPattern& pattern,
                                   std::span<const std::int8_t> bc0 = {}, std::span<const std::int8_t> bc1 = {})
{
  check(cfx_assemble_matrix_zeroed(a.handle.h, pattern.handle.h, bc0.empty() ? nullptr : bc0.data(),
                                   bc1.empty() ? nullptr : bc1.data(), values.data()));
}

/// assemble_vector(): assembler.h:252-262
inline void assemble_vector(std::span<double> b, const Form& L) { check(cfx_assemble_vector(L.handle.h, b.data())); }

/// complex128 instantiation (T = std::complex<double>, wrappers/fem.cpp:490-500; test_complex_assembly.py:24-95):
/// `scales` = the complex constant of every integral of the form (empty: all 1); accumulates into `values` / `b`
inline void assemble_matrix(std::span<std::complex<double>> values, const Form& a, const SparsityPattern& pattern,
                            std::span<const std::complex<double>> scales = {}, std::span<const std::int8_t> bc0 = {},
                            std::span<const std::int8_t> bc1 = {})
{
  check(cfx_assemble_matrix_c128(a.handle.h, pattern.handle.h, bc0.empty() ? nullptr : bc0.data(),
                                 bc1.empty() ? nullptr : bc1.data(),
                                 scales.empty() ? nullptr : reinterpret_cast<const double*>(scales.data()), 0,
                                 reinterpret_cast<double*>(values.data())));
}
inline void assemble_vector(std::span<std::complex<double>> b, const Form& L, std::span<const std::complex<double>> scales = {})
{
  check(cfx_assemble_vector_c128(L.handle.h, scales.empty() ? nullptr : reinterpret_cast<const double*>(scales.data()),
                                 reinterpret_cast<double*>(b.data())));
}

/// complex64 instantiation (T = std::complex<float>): interleaved float32 containers, the complex128 arithmetic, one
/// rounding per entry; `scales` stay double
inline void assemble_matrix(std::span<std::complex<float>> values, const Form& a, const SparsityPattern& pattern,
                            std::span<const std::complex<double>> scales = {}, std::span<const std::int8_t> bc0 = {},
                            std::span<const std::int8_t> bc1 = {})
{
  check(cfx_assemble_matrix_c64(a.handle.h, pattern.handle.h, bc0.empty() ? nullptr : bc0.data(),
                                bc1.empty() ? nullptr : bc1.data(),
                                scales.empty() ? nullptr : reinterpret_cast<const double*>(scales.data()), 0,
                                reinterpret_cast<float*>(values.data())));
}
inline void assemble_vector(std::span<std::complex<float>> b, const Form& L, std::span<const std::complex<double>> scales = {})
{
  check(cfx_assemble_vector_c64(L.handle.h, scales.empty() ? nullptr : reinterpret_cast<const double*>(scales.data()),
                                reinterpret_cast<float*>(b.data())));
}

/// A user integrand compiled for gfx950 at run time (the generated tabulate_tensor of a form, Form.h:59-75): `source`
/// defines `__device__ void name(double* A, const double* w, const double* c, const double* coordinate_dofs, int nq,
/// const double* points, const double* weights, const double* point_data)`; the id returned goes into Integral::kernel.
inline int register_integrand(const std::string& name, const std::string& source, int rank)
{
  int id = 0;
  check(cfx_integrand_register(name.c_str(), source.c_str(), rank, &id));
  return id;
}

/// (Sync-free steps -- cfx_step_begin / cfx_step_end -- are not wrapped here: this facade returns host vectors sized by the
/// counts of the moment, and inside a step those are capacities.  Callers that pass device handles on use the C ABI's
/// step functions directly, as cutfemx_amd/step.py does.)

/// apply_lifting(): b <- b - alpha A (g - x0) over the Dirichlet columns
/// (cpp/dolfinx_custom_data/fem/assemble_vector_impl.h:383-436); one form, markers/values per dof
inline void apply_lifting(std::span<double> b, const Form& a, std::span<const std::int8_t> bc_markers,
                          std::span<const double> bc_values, std::span<const double> x0 = {}, double alpha = 1.0)
{
  if (bc_markers.size() != b.size() || bc_values.size() != b.size() || (!x0.empty() && x0.size() != b.size()))
    throw std::invalid_argument("apply_lifting: marker / value arrays must have one entry per dof");
  check(cfx_apply_lifting(a.handle.h, bc_markers.data(), bc_values.data(), x0.empty() ? nullptr : x0.data(), alpha,
                          b.data()));
}

/// DirichletBC::set: b[dofs] = alpha (g - x0)
inline void set_bc(std::span<double> b, std::span<const std::int8_t> bc_markers, std::span<const double> bc_values,
                   std::span<const double> x0 = {}, double alpha = 1.0)
{
  if (bc_markers.size() != b.size() || bc_values.size() != b.size() || (!x0.empty() && x0.size() != b.size()))
    throw std::invalid_argument("set_bc: marker / value arrays must have one entry per dof");
  check(cfx_set_bc((std::int64_t)b.size(), bc_markers.data(), bc_values.data(), x0.empty() ? nullptr : x0.data(), alpha,
                   b.data()));
}

/// cutfemx::fem::ActiveDomain (deactivate.h:387-400)
struct ActiveDomain
{
  detail::Handle<cfx_active_t, cfx_active_destroy> handle;
  std::vector<std::int32_t> active_cells, inactive_dofs;
};

inline ActiveDomain active_domain(const Form& a)
{
  cfx_active_t h = nullptr;
  check(cfx_active_domain(a.handle.h, &h));
  ActiveDomain d;
  d.handle = detail::Handle<cfx_active_t, cfx_active_destroy>(h);
  const std::int32_t *ac, *id;
  std::int64_t na, ni;
  check(cfx_active_view(h, &ac, &na, &id, &ni));
  d.active_cells = download(ac, na);
  d.inactive_dofs = download(id, ni);
  return d;
}

/// deactivate_outside(): deactivate.h:402-418 -- diag = 1, rhs = 0 on the inactive dofs
inline ActiveDomain& deactivate_outside(std::span<double> values, const SparsityPattern& pattern, std::span<double> b,
                                        ActiveDomain& domain, double diagonal = 1.0, double rhs_value = 0.0)
{
  check(cfx_deactivate_outside(domain.handle.h, pattern.handle.h, values.empty() ? nullptr : values.data(),
                               b.empty() ? nullptr : b.data(), diagonal, rhs_value));
  return domain;
}

/// zero_rows(A, tol): rows whose assembled entries are all <= tol in magnitude (python/cutfemx/fem.py:777-782)
inline std::vector<std::int32_t> zero_rows(std::span<const double> values, const SparsityPattern& pattern, double tol = 0.0)
{
  std::int32_t* rows = nullptr;
  std::int64_t n = 0;
  check(cfx_zero_rows(pattern.handle.h, values.data(), tol, &rows, &n));
  std::vector<std::int32_t> out = download(rows, n);
  check(cfx_device_free(rows));
  return out;
}

/// One block of a MatrixCSR block system: the values of A[i][j] with their pattern (nullptr: an absent block)
struct MatrixBlock
{
  std::span<double> values;
  const SparsityPattern* pattern = nullptr;
};

/// zero_block_rows(): deactivate.h:279-320 -- row r of block row i is listed when it is zero in every block A[i][j]
inline std::vector<std::vector<std::int32_t>> zero_block_rows(const std::vector<std::vector<MatrixBlock>>& A_blocks,
                                                              double tol = 0.0)
{
  if (A_blocks.empty()) throw std::runtime_error("Zero-row scan requires at least one block row");
  const std::size_t nb = A_blocks.size();
  std::vector<std::vector<std::int32_t>> rows(nb);
  for (std::size_t i = 0; i < nb; ++i)
  {
    if (A_blocks[i].size() != nb) throw std::runtime_error("Zero-row scan requires a square block matrix");
    if (A_blocks[i][i].pattern == nullptr) throw std::runtime_error("Zero-row scan requires every diagonal matrix block");
    const std::int64_t nrows = A_blocks[i][i].pattern->num_rows();
    bool first = true;
    for (std::size_t j = 0; j < nb; ++j)
    {
      const MatrixBlock& B = A_blocks[i][j];
      if (B.pattern == nullptr) continue;
      if (B.pattern->num_rows() != nrows) throw std::runtime_error("Zero-row scan found incompatible row maps in a block row");
      std::vector<std::int32_t> z = zero_rows(B.values, *B.pattern, tol);
      if (first) { rows[i] = std::move(z); first = false; continue; }
      std::vector<std::int32_t> both;
      std::set_intersection(rows[i].begin(), rows[i].end(), z.begin(), z.end(), std::back_inserter(both));
      rows[i] = std::move(both);
    }
  }
  return rows;
}

/// deactivate_outside_blocks(): deactivate.h:420-457 -- the inactive rows of block row i come from active_domains[i];
/// only the diagonal block A[i][i] and the optional right-hand side b[i] are modified
inline std::vector<ActiveDomain*> deactivate_outside_blocks(const std::vector<std::vector<MatrixBlock>>& A_blocks,
                                                            const std::vector<ActiveDomain*>& active_domains,
                                                            const std::vector<std::span<double>>& b_blocks = {},
                                                            double diagonal = 1.0, double rhs_value = 0.0)
{
  if (A_blocks.empty()) throw std::runtime_error("Block deactivation requires at least one block row");
  if (A_blocks.size() != active_domains.size())
    throw std::runtime_error("Block deactivation requires one ActiveDomain per block row");
  const std::size_t nb = A_blocks.size();
  for (std::size_t i = 0; i < nb; ++i)
  {
    if (A_blocks[i].size() != nb) throw std::runtime_error("Block deactivation requires a square block matrix");
    if (active_domains[i] == nullptr) throw std::runtime_error("Block deactivation received a null ActiveDomain");
    if (A_blocks[i][i].pattern == nullptr) throw std::runtime_error("Block deactivation requires every diagonal matrix block");
  }
  if (!b_blocks.empty() && b_blocks.size() != nb)
    throw std::runtime_error("Block deactivation requires one RHS vector per block row");
  for (std::size_t i = 0; i < nb; ++i)
    deactivate_outside(A_blocks[i][i].values, *A_blocks[i][i].pattern, b_blocks.empty() ? std::span<double>{} : b_blocks[i],
                       *active_domains[i], diagonal, rhs_value);
  return active_domains;
}

/// The monolithic matrix of a block system (host copies): what the reference assembles as ONE matrix on a mixed element
/// (python/tests/test_assembly_stokes.py:34-95), with block-ordered dofs -- see cfx_csr_block_merge
struct MergedCSR
{
  std::vector<std::int64_t> row_ptr;
  std::vector<std::int32_t> cols;
  std::vector<double> values;
  std::int64_t num_rows = 0, num_cols = 0;
};

/// merge_blocks(): A_blocks[i][j] with a null pattern is an empty block; every block row and column needs one block
inline MergedCSR merge_blocks(const std::vector<std::vector<MatrixBlock>>& A_blocks)
{
  if (A_blocks.empty() || A_blocks[0].empty()) throw std::runtime_error("merge_blocks requires at least one block");
  const std::size_t nbr = A_blocks.size(), nbc = A_blocks[0].size();
  std::vector<const std::int64_t*> ip(nbr * nbc, nullptr);
  std::vector<const std::int32_t*> ix(nbr * nbc, nullptr);
  std::vector<const double*> va(nbr * nbc, nullptr);
  std::vector<std::int64_t> nrows(nbr, -1), ncols(nbc, -1);
  for (std::size_t i = 0; i < nbr; ++i)
  {
    if (A_blocks[i].size() != nbc) throw std::runtime_error("merge_blocks requires the same number of blocks in every block row");
    for (std::size_t j = 0; j < nbc; ++j)
    {
      const MatrixBlock& B = A_blocks[i][j];
      if (B.pattern == nullptr) continue;
      if ((nrows[i] >= 0 && nrows[i] != B.pattern->num_rows()) || (ncols[j] >= 0 && ncols[j] != B.pattern->num_cols()))
        throw std::runtime_error("merge_blocks found incompatible block sizes");
      nrows[i] = B.pattern->num_rows(); ncols[j] = B.pattern->num_cols();
      ip[i * nbc + j] = B.pattern->view.indptr; ix[i * nbc + j] = B.pattern->view.indices; va[i * nbc + j] = B.values.data();
    }
  }
  for (std::int64_t n : nrows) if (n < 0) throw std::runtime_error("merge_blocks requires a block in every block row and every block column");
  for (std::int64_t n : ncols) if (n < 0) throw std::runtime_error("merge_blocks requires a block in every block row and every block column");
  std::int64_t* o_ip = nullptr; std::int32_t* o_ix = nullptr; double* o_va = nullptr; std::int64_t nnz = 0;
  check(cfx_csr_block_merge((int)nbr, (int)nbc, ip.data(), ix.data(), va.data(), nrows.data(), ncols.data(), &o_ip, &o_ix, &o_va, &nnz));
  MergedCSR M;
  for (std::int64_t n : nrows) M.num_rows += n;
  for (std::int64_t n : ncols) M.num_cols += n;
  M.row_ptr = download(o_ip, M.num_rows + 1);
  M.cols = download(o_ix, nnz);
  M.values = download(o_va, nnz);
  check(cfx_device_free(o_ip)); check(cfx_device_free(o_ix)); check(cfx_device_free(o_va));
  return M;
}

} // namespace fem

namespace extensions
{
/// cutfemx::extensions::RootPolicy / CellAggregation (cpp/cutfemx/extensions/cell_aggregation.h:24-47)
enum class RootPolicy { interior_only = 0, interior_or_well_cut = 1 };

inline RootPolicy root_policy_from_string(std::string_view policy)
{
  if (policy == "interior_only") return RootPolicy::interior_only;
  if (policy == "interior_or_well_cut") return RootPolicy::interior_or_well_cut;
  throw std::invalid_argument("Unknown root policy. Expected 'interior_only' or 'interior_or_well_cut'.");
}

struct CellAggregation
{
  detail::Handle<cfx_aggregation_t, cfx_cell_aggregation_destroy> handle;
  std::vector<std::int32_t> active_cells, cut_cells, interior_cells, well_posed_cells, ill_posed_cells, root_cell,
      aggregate_id, propagation_depth, rootless_cells;
  std::vector<double> cut_volume_fraction;
  /// (bad, 0, root, 0) rows: extension_pairs(), the entities of a CFX_K_EXTENSION_L2 integral
  std::vector<std::int32_t> pairs;
};

/// create_cell_aggregation(): cell_aggregation.cpp:143-270
inline CellAggregation create_cell_aggregation(const CutData& cut_data, std::string_view selector,
                                               double volume_fraction_threshold,
                                               RootPolicy root_policy = RootPolicy::interior_or_well_cut,
                                               int max_iterations = -1, bool allow_rootless = false)
{
  cfx_aggregation_t h = nullptr;
  check(cfx_cell_aggregation_create(cut_data.handle.h, std::string(selector).c_str(), volume_fraction_threshold,
                                    static_cast<int>(root_policy), max_iterations, allow_rootless ? 1 : 0, &h));
  CellAggregation a;
  a.handle = detail::Handle<cfx_aggregation_t, cfx_cell_aggregation_destroy>(h);
  cfx_aggregation_view v;
  check(cfx_cell_aggregation_view_get(h, &v));
  a.active_cells = download(v.active_cells, v.n_active);
  a.cut_cells = download(v.cut_cells, v.n_cut);
  a.interior_cells = download(v.interior_cells, v.n_interior);
  a.well_posed_cells = download(v.well_posed_cells, v.n_well_posed);
  a.ill_posed_cells = download(v.ill_posed_cells, v.n_ill_posed);
  a.rootless_cells = download(v.rootless_cells, v.n_rootless);
  a.root_cell = download(v.root_cell, v.ncells);
  a.aggregate_id = download(v.aggregate_id, v.ncells);
  a.propagation_depth = download(v.propagation_depth, v.ncells);
  a.cut_volume_fraction = download(v.cut_volume_fraction, v.ncells);
  a.pairs = download(v.pairs, 4 * v.n_pairs);
  return a;
}
} // namespace extensions
} // namespace cutfemx_amd
