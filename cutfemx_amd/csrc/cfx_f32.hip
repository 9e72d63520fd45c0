// float32 instantiation of the boundary: the <float, float> rows of
// python/cutfemx/wrappers/fem.cpp:490-500 (declare_runtime_fem<T, U>) and wrappers/cut.cpp:403-407
// (declare_cut_api<T>).
//
// What is f32 here is what the reference's template parameters name: the CONTAINERS that cross the boundary --
// mesh coordinates (U), level-set dof values, rule points / weights, normals, CSR values, vectors and Dirichlet data
// (T).  Inside, every kernel keeps its fp64 registers: inputs are widened once on upload (exact), outputs are
// rounded once on the way out.  That is a deliberate MI355X choice, not a shortcut: the hot kernels of this path
// are bound by gather latency, index decoding and HBM streams of int32 connectivity, the FP64 vector
// rate is half the FP32 rate and none of them is near either; a second set of kernels with f32 registers would
// double the code to test and make the cut geometry (edge intersections of nearly parallel level sets) less
// accurate than the reference's own f32 path.  The f32 results are therefore the fp64 results rounded to nearest
// -- at least as close to the exact integrals as the reference's <float, float> instantiation.
#include "cfx_common.h"
#include "cfx_device.h"
#include "cfx_elem.h"

using namespace cfx;

namespace
{

__global__ void __launch_bounds__(kBlock) widen_kernel(int64_t n, const float* __restrict__ in, double* __restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (double)in[i];
}

// ADD: out += in in fp64, one rounding (the accumulate semantics of assemble_matrix / assemble_vector)
template <bool ADD>
__global__ void __launch_bounds__(kBlock) narrow_kernel(int64_t n, const double* __restrict__ in, float* __restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ADD ? (float)((double)out[i] + in[i]) : (float)in[i];
}

__global__ void __launch_bounds__(kBlock) set_bc_f32_kernel(int64_t n, const int8_t* __restrict__ markers,
                                                            const float* __restrict__ g, const float* __restrict__ x0,
                                                            double alpha, float* __restrict__ b)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && markers[i]) b[i] = (float)(alpha * ((double)g[i] - (x0 ? (double)x0[i] : 0.0)));
}

__global__ void __launch_bounds__(kBlock) deactivate_f32_kernel(int64_t n, const int32_t* __restrict__ rows,
                                                                const int64_t* __restrict__ indptr,
                                                                const int32_t* __restrict__ indices, float* values, float* b,
                                                                float diagonal, float rhs_value, int* error)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t r = rows[i];
  if (values)
  {
    const int64_t rb = indptr[r], re = indptr[r + 1];
    const int64_t pos = (re - rb == 1 && indices[rb] == r) ? rb : csr_find(indices, rb, re, r);
    if (pos < 0) *error = 1; else values[pos] = diagonal;
  }
  if (b) b[r] = rhs_value;
}

struct RowAllZero32
{
  const int64_t* indptr;
  const float* values;
  float tol;
  __device__ bool operator()(int64_t r) const
  {
    for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k)
      if (fabsf(values[k]) > tol) return false;
    return true;
  }
};

// library-owned fp64 copy of a caller's f32 array (host or device)
DevArray<double> widen(const float* src, int64_t n)
{
  DevArray<double> out(n);
  if (n > 0)
  {
    DevArray<float> in = to_device(src, n);
    launch("widen_f32", widen_kernel, grid_for(n), dim3(kBlock), 0, n, in.p, out.p);
    if (in.owned) CFX_HIP(hipStreamSynchronize(ctx().stream)); // the staging copy dies with `in`
  }
  return out;
}

template <bool ADD>
void narrow(const double* src, float* user, int64_t n)
{
  OutArray<float> out(user, n, ADD);
  if (n > 0) launch("narrow_f32", narrow_kernel<ADD>, grid_for(n), dim3(kBlock), 0, n, src, out.dev);
  out.finish();
}

} // namespace

extern "C" {

int cfx_mesh_create_f32(int tdim, int gdim, int64_t nnodes, const float* x, int64_t ncells, const int32_t* conn,
                        int cell_stride, cfx_mesh_t* out)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(out && x && nnodes > 0, CFX_ERR_INVALID_ARGUMENT, "cfx_mesh_create_f32: null argument");
  DevArray<double> dx = widen(x, nnodes * 3);
  const int rc = cfx_mesh_create(tdim, gdim, nnodes, dx.p, ncells, conn, cell_stride, out);
  if (rc != CFX_OK) return rc;
  (*out)->x = std::move(dx); // cfx_mesh_create aliased the device pointer: the mesh now owns the widened copy
  CFX_API_END
}

int cfx_cut_create_f32(cfx_mesh_t mesh, int nls, const int32_t* ls_dofmap, int ls_ndofs_cell, int64_t ls_ndofs,
                       const float* const* ls_values, const cfx_cut_options* opt, cfx_cut_t* out)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(out != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_create_f32: null output");
  require(nls >= 1 && ls_values, CFX_ERR_INVALID_ARGUMENT, "cutfemx.cut requires at least one level-set function");
  require(nls <= 8 && ls_ndofs > 0, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_create_f32: invalid level-set sizes");
  std::vector<DevArray<double>> wide;
  std::vector<const double*> ptrs;
  for (int k = 0; k < nls; ++k)
  {
    require(ls_values[k] != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_create_f32: null level-set values");
    wide.push_back(widen(ls_values[k], ls_ndofs));
    ptrs.push_back(wide.back().p);
  }
  const int rc = cfx_cut_create(mesh, nls, ls_dofmap, ls_ndofs_cell, ls_ndofs, ptrs.data(), opt, out);
  if (rc != CFX_OK) return rc;
  for (int k = 0; k < nls; ++k) (*out)->ls_values[k] = std::move(wide[k]); // owned by the cut from here on
  CFX_API_END
}

int cfx_cut_update_f32(cfx_cut_t cut, const float* const* ls_values)
{
  CFX_API_BEGIN
  require(cut != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_update_f32: null handle");
  // the cut holds widened COPIES of float32 level sets (not aliases of the caller's arrays, as the f64 path does):
  // "reuse what you have" would silently re-classify stale values, so the arrays must be passed again
  require(ls_values != nullptr, CFX_ERR_INVALID_ARGUMENT,
          "cfx_cut_update_f32: pass the float32 level-set arrays again (the cut keeps widened copies, not aliases)");
  for (int k = 0; k < cut->nls; ++k)
  {
    require(ls_values[k] != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_update_f32: null level-set values");
    cut->ls_values[k] = widen(ls_values[k], cut->ls_ndofs);
  }
  return cfx_cut_update(cut, nullptr);
  CFX_API_END
}

int cfx_rules_create_f32(cfx_mesh_t mesh, int tdim, int64_t nq, int64_t nr, const float* points, const float* weights,
                         const int32_t* offsets, const int32_t* parent_map, cfx_rules_t* out)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(mesh && out && nq >= 0 && nr >= 0 && (nq == 0 || (points && weights)), CFX_ERR_INVALID_ARGUMENT,
          "cfx_rules_create_f32: null argument");
  DevArray<double> dp = widen(points, nq * tdim), dw = widen(weights, nq);
  const int rc = cfx_rules_create(mesh, tdim, nq, nr, dp.p, dw.p, offsets, parent_map, out);
  if (rc != CFX_OK) return rc;
  (*out)->points = std::move(dp);
  (*out)->weights = std::move(dw);
  CFX_API_END
}

int cfx_rules_view_get_f32(cfx_rules_t r, cfx_rules_view_f32* v)
{
  CFX_API_BEGIN
  require(r && v, CFX_ERR_INVALID_ARGUMENT, "cfx_rules_view_get_f32: null argument");
  if (r->points_f32.n != r->points.n || r->points_f32.p == nullptr)
  {
    r->points_f32.alloc(r->points.n);
    r->weights_f32.alloc(r->weights.n);
    narrow<false>(r->points.p, r->points_f32.p, r->points.n);
    narrow<false>(r->weights.p, r->weights_f32.p, r->weights.n);
  }
  v->tdim = r->tdim; v->gdim = r->gdim; v->nq = r->nq.value(); v->nr = r->nr.value(); // (the float32 boundary works with exact lengths)
  v->points = r->points_f32.p; v->weights = r->weights_f32.p;
  v->offsets = r->offsets.p; v->parent_map = r->parent_map.p;
  v->host_width = r->host_width; v->reserved = 0;
  v->host_rows = r->host_width ? r->host_rows.p : nullptr;
  v->host_verts = r->host_width ? r->host_verts.p : nullptr;
  CFX_API_END
}

int cfx_rules_physical_points_f32(cfx_rules_t r, float* out)
{
  CFX_API_BEGIN
  require(r && out, CFX_ERR_INVALID_ARGUMENT, "cfx_rules_physical_points_f32: null argument");
  DevArray<double> tmp(r->nq.value() * r->gdim);
  const int rc = cfx_rules_physical_points(r, tmp.p);
  if (rc != CFX_OK) return rc;
  narrow<false>(tmp.p, out, tmp.n);
  CFX_API_END
}

int cfx_evaluate_normals_f32(cfx_cut_t cut, int level_set, cfx_rules_t rules, float sign, float* out)
{
  CFX_API_BEGIN
  require(cut && rules && out, CFX_ERR_INVALID_ARGUMENT, "cfx_evaluate_normals_f32: null argument");
  DevArray<double> tmp(rules->nq.value() * rules->gdim);
  const int rc = cfx_evaluate_normals(cut, level_set, rules, (double)sign, tmp.p);
  if (rc != CFX_OK) return rc;
  narrow<false>(tmp.p, out, tmp.n);
  CFX_API_END
}

int cfx_evaluate_values_f32(cfx_cut_t cut, int level_set, cfx_rules_t rules, float* out)
{
  CFX_API_BEGIN
  require(cut && rules && out, CFX_ERR_INVALID_ARGUMENT, "cfx_evaluate_values_f32: null argument");
  DevArray<double> tmp(rules->nq.value());
  const int rc = cfx_evaluate_values(cut, level_set, rules, tmp.p);
  if (rc != CFX_OK) return rc;
  narrow<false>(tmp.p, out, tmp.n);
  CFX_API_END
}

int cfx_widen_f32(const float* src, int64_t n, double** out)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(out && n >= 0 && (src || n == 0), CFX_ERR_INVALID_ARGUMENT, "cfx_widen_f32: null argument");
  DevArray<double> w = widen(src, n);
  *out = w.p;
  w.p = nullptr; w.owned = false; // released by the caller with cfx_device_free
  CFX_API_END
}

static int assemble_matrix_f32(cfx_form_t a, cfx_pattern_t P, const int8_t* bc0, const int8_t* bc1, float* values, bool add)
{
  CFX_API_BEGIN
  require(a && P && values, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_matrix_f32: null argument");
  const int64_t nnz = P->nnz.value();
  DevArray<double> tmp(nnz);
  const int rc = cfx_assemble_matrix_zeroed(a, P, bc0, bc1, tmp.p);
  if (rc != CFX_OK) return rc;
  if (add) narrow<true>(tmp.p, values, nnz); else narrow<false>(tmp.p, values, nnz);
  CFX_API_END
}

int cfx_assemble_matrix_f32(cfx_form_t a, cfx_pattern_t P, const int8_t* bc0, const int8_t* bc1, float* values)
{
  return assemble_matrix_f32(a, P, bc0, bc1, values, true);
}

int cfx_assemble_matrix_zeroed_f32(cfx_form_t a, cfx_pattern_t P, const int8_t* bc0, const int8_t* bc1, float* values)
{
  return assemble_matrix_f32(a, P, bc0, bc1, values, false);
}

int cfx_assemble_vector_f32(cfx_form_t L, float* b)
{
  CFX_API_BEGIN
  require(L && b, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_vector_f32: null argument");
  const int64_t n = L->V->ndofs * L->V->bs;
  DevArray<double> tmp(n);
  tmp.zero();
  const int rc = cfx_assemble_vector(L, tmp.p);
  if (rc != CFX_OK) return rc;
  narrow<true>(tmp.p, b, n);
  CFX_API_END
}

int cfx_apply_lifting_f32(cfx_form_t a, const int8_t* bc_markers, const float* bc_values, const float* x0, float alpha,
                          float* b)
{
  CFX_API_BEGIN
  require(a && bc_markers && bc_values && b, CFX_ERR_INVALID_ARGUMENT, "cfx_apply_lifting_f32: null argument");
  const int64_t n = a->V->ndofs * a->V->bs;
  DevArray<double> g = widen(bc_values, n), w0, tmp(n);
  if (x0) w0 = widen(x0, n);
  tmp.zero();
  const int rc = cfx_apply_lifting(a, bc_markers, g.p, x0 ? w0.p : nullptr, (double)alpha, tmp.p);
  if (rc != CFX_OK) return rc;
  narrow<true>(tmp.p, b, n);
  CFX_API_END
}

int cfx_set_bc_f32(int64_t n, const int8_t* bc_markers, const float* bc_values, const float* x0, float alpha, float* b)
{
  CFX_API_BEGIN
  require(n >= 0 && bc_markers && bc_values && b, CFX_ERR_INVALID_ARGUMENT, "cfx_set_bc_f32: null argument");
  ctx().ensure();
  DevArray<int8_t> dm = to_device(bc_markers, n);
  DevArray<float> dv = to_device(bc_values, n), dx0 = to_device(x0, x0 ? n : 0);
  OutArray<float> out(b, n, true);
  launch("set_bc", set_bc_f32_kernel, grid_for(n), dim3(kBlock), 0, n, dm.p, dv.p, x0 ? dx0.p : nullptr, (double)alpha,
         out.dev);
  out.finish();
  CFX_API_END
}

int cfx_zero_rows_f32(cfx_pattern_t P, const float* values, float tol, int32_t** rows, int64_t* n_rows)
{
  CFX_API_BEGIN
  require(P && values && rows && n_rows, CFX_ERR_INVALID_ARGUMENT, "cfx_zero_rows_f32: null argument");
  DevArray<float> dv = to_device(values, P->nnz.value());
  DevArray<int32_t> list;
  const int64_t n = compact("zero_rows", P->nrows, RowAllZero32{P->indptr.p, dv.p, tol}, list);
  int32_t* out = static_cast<int32_t*>(dev_alloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1)));
  if (n > 0)
    CFX_HIP(hipMemcpyAsync(out, list.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, ctx().stream));
  CFX_HIP(hipStreamSynchronize(ctx().stream));
  *rows = out;
  *n_rows = n;
  CFX_API_END
}

int cfx_deactivate_outside_f32(cfx_active_t d, cfx_pattern_t P, float* values, float* b, float diagonal, float rhs_value)
{
  CFX_API_BEGIN
  require(d && (values == nullptr || P), CFX_ERR_INVALID_ARGUMENT, "cfx_deactivate_outside_f32: null argument");
  const int64_t nrows = d->V->ndofs * d->V->bs;
  std::unique_ptr<OutArray<float>> ov, ob;
  if (values) ov = std::make_unique<OutArray<float>>(values, P->nnz.value(), true);
  if (b) ob = std::make_unique<OutArray<float>>(b, nrows, true);
  ZeroFlag err;
  cfx::active_lists(d);
  const int64_t n_inactive = d->n_inactive.value();
  if (n_inactive > 0)
    launch("deactivate", deactivate_f32_kernel, grid_for(n_inactive), dim3(kBlock), 0, n_inactive,
           d->inactive_dofs.p, P ? P->indptr.p : nullptr, P ? P->indices.p : nullptr, values ? ov->dev : nullptr,
           b ? ob->dev : nullptr, diagonal, rhs_value, err.p);
  require(!read_scalar(err.p), CFX_ERR_RUNTIME, "Deactivated matrix row has no diagonal entry.");
  if (ov) ov->finish();
  if (ob) ob->finish();
  CFX_API_END
}

} // extern "C"
