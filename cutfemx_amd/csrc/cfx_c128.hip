// complex128 instantiation of the boundary: the <std::complex<double>, double> rows of
// python/cutfemx/wrappers/fem.cpp:490-500 (declare_runtime_fem<T, U>); invariants
// python/tests/test_complex_assembly.py:24-95.
//
// What is complex here is what the reference's scalar type T names: CSR values, vectors, Dirichlet data and the
// constants / coefficients that multiply an integrand.  Geometry, quadrature rules, basis functions and every
// integrand of this path are real, so a complex form is a complex combination of real ones:
//     A = sum_k s_k A_k,   s_k = the complex constant of integral k (kappa in `kappa inner(grad u, grad v) dx`),
// and a complex coefficient FUNCTION enters linearly (its real part with scale s, its imaginary part with i s: the
// binding above passes two real forms).  The entry points below take the per-integral constants as an array
// `scales` (interleaved re, im; NULL = 1), assemble each group of integrals that shares a constant with the
// float64 kernels into a temporary and add s x it to the interleaved complex array -- no second set of kernels
// with complex registers for a path whose kernels are bound by gather latency and int32 connectivity streams,
// and results that are the float64 results times the constant (one complex multiply-add per entry).
// inner(u, v) conjugates the test function in the reference; every basis function here is real.
#include <map>

#include "cfx_common.h"
#include "cfx_device.h"

using namespace cfx;

namespace
{

// out (complex, interleaved) += (sr + i si) * in (real)
__global__ void __launch_bounds__(kBlock) axpy_c128_kernel(int64_t n, const double* __restrict__ in, double sr, double si,
                                                           double2* __restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = in[i];
  double2 o = out[i];
  o.x += sr * v; o.y += si * v;
  out[i] = o;
}

// out (complex) += (sr + i si) * (re + i im)
__global__ void __launch_bounds__(kBlock) axpy2_c128_kernel(int64_t n, const double* __restrict__ re, const double* __restrict__ im,
                                                            double sr, double si, double2* __restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = re[i], b = im[i];
  double2 o = out[i];
  o.x += sr * a - si * b; o.y += sr * b + si * a;
  out[i] = o;
}

// part: 0 real parts, 1 imaginary parts of an interleaved complex array
__global__ void __launch_bounds__(kBlock) split_c128_kernel(int64_t n, const double2* __restrict__ in, int part, double* __restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = part ? in[i].y : in[i].x;
}

__global__ void __launch_bounds__(kBlock) set_bc_c128_kernel(int64_t n, const int8_t* __restrict__ markers,
                                                             const double2* __restrict__ g, const double2* __restrict__ x0,
                                                             double ar, double ai, double2* __restrict__ b)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !markers[i]) return;
  const double dr = g[i].x - (x0 ? x0[i].x : 0.0), di = g[i].y - (x0 ? x0[i].y : 0.0);
  b[i] = make_double2(ar * dr - ai * di, ar * di + ai * dr);
}

template <typename C2> // double2 (complex128) or float2 (complex64)
__global__ void __launch_bounds__(kBlock) deactivate_c128_kernel(int64_t n, const int32_t* __restrict__ rows,
                                                                 const int64_t* __restrict__ indptr,
                                                                 const int32_t* __restrict__ indices, C2* values, C2* b,
                                                                 C2 diagonal, C2 rhs_value, int* error)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t r = rows[i];
  if (values)
  {
    const int64_t rb = indptr[r], re = indptr[r + 1];
    int64_t pos = -1;
    if (re - rb == 1 && indices[rb] == r) pos = rb;
    else
    {
      int64_t lo = rb, hi = re;
      while (lo < hi)
      {
        const int64_t mid = (lo + hi) >> 1;
        if (indices[mid] < r) lo = mid + 1; else hi = mid;
      }
      pos = (lo < re && indices[lo] == r) ? lo : -1;
    }
    if (pos < 0) *error = 1; else values[pos] = diagonal; // set, not add (set_diagonal)
  }
  if (b) b[r] = rhs_value;
}

// complex64 containers: widened once on the way in, rounded once on the way out (the policy of the float32 boundary)
__global__ void __launch_bounds__(kBlock) widen_c64_kernel(int64_t n, const float* __restrict__ in, double* __restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (double)in[i];
}
template <bool ADD>
__global__ void __launch_bounds__(kBlock) narrow_c64_kernel(int64_t n, const double* __restrict__ in, float* __restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ADD ? (float)((double)out[i] + in[i]) : (float)in[i];
}
DevArray<double> widen_c64(const float* src, int64_t n)
{
  DevArray<double> out(n);
  if (n > 0)
  {
    DevArray<float> in = to_device(src, n);
    launch("widen_c64", widen_c64_kernel, grid_for(n), dim3(kBlock), 0, n, (const float*)in.p, out.p);
    if (in.owned) CFX_HIP(hipStreamSynchronize(ctx().stream)); // the staging copy dies with `in`
  }
  return out;
}
template <bool ADD>
void narrow_c64(const double* src, float* user, int64_t n)
{
  OutArray<float> out(user, n, ADD);
  if (n > 0) launch("narrow_c64", narrow_c64_kernel<ADD>, grid_for(n), dim3(kBlock), 0, n, src, out.dev);
  out.finish();
}

template <typename T>
DevArray<T> alias(const DevArray<T>& a)
{
  DevArray<T> o;
  o.p = a.p; o.n = a.n; o.owned = false;
  return o;
}

// the integrals `which` of form a as a form of their own (arrays aliased: it lives in a's cache and dies with a, so
// that its row plan -- and the tables keyed on it -- serve every later assembly of the same group)
cfx_form_s* sub_form(cfx_form_s* a, const std::vector<int>& which)
{
  auto hit = a->sub_forms.find(which);
  if (hit != a->sub_forms.end()) return hit->second.get();
  auto f = std::make_unique<cfx_form_s>();
  f->V = a->V; f->V1 = a->V1; f->rank = a->rank;
  for (int i : which)
  {
    const cfx_integral_dev& I = a->integrals[i];
    cfx_integral_dev J;
    J.type = I.type; J.kernel = I.kernel; J.qdegree = I.qdegree; J.point_stride = I.point_stride;
    J.entities = alias(I.entities); J.n_entities = I.n_entities; J.rules = I.rules;
    J.entities_serial = I.entities_serial; J.rules_serial = I.rules_serial; J.n_std = I.n_std;
    J.point_data = alias(I.point_data); J.coefficient = alias(I.coefficient);
    for (int k = 0; k < 8; ++k) J.params[k] = I.params[k];
    f->integrals.push_back(std::move(J));
  }
  cfx_form_s* out = f.get();
  a->sub_forms[which] = std::move(f);
  return out;
}

// integrals grouped by their complex constant; a zero constant drops its integrals
std::map<std::pair<double, double>, std::vector<int>> groups(const cfx_form_s* a, const double* scales)
{
  std::map<std::pair<double, double>, std::vector<int>> g;
  for (int i = 0; i < (int)a->integrals.size(); ++i)
  {
    const double sr = scales ? scales[2 * i] : 1.0, si = scales ? scales[2 * i + 1] : 0.0;
    if (sr == 0.0 && si == 0.0) continue;
    g[{sr, si}].push_back(i);
  }
  return g;
}

} // namespace

extern "C" {

int cfx_assemble_matrix_c128(cfx_form_t a, cfx_pattern_t P, const int8_t* bc0, const int8_t* bc1, const double* scales,
                             int zero_first, double* values)
{
  CFX_API_BEGIN
  require(a && P && values, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_matrix_c128: null argument");
  require(a->rank == 2, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_matrix_c128: form is not bilinear");
  const int64_t nnz = P->nnz.value(); // (the complex boundary works with exact lengths)
  OutArray<double> out(values, 2 * nnz, !zero_first);
  if (zero_first) dev_fill(out.dev, 0, sizeof(double) * 2 * (size_t)nnz);
  DevArray<double> tmp(nnz);
  for (const auto& kv : groups(a, scales))
  {
    const bool whole = kv.second.size() == a->integrals.size();
    cfx_form_s* sub = whole ? nullptr : sub_form(a, kv.second);
    const int rc = cfx_assemble_matrix_zeroed(whole ? a : sub, P, bc0, bc1, tmp.p);
    if (rc != CFX_OK) return rc;
    launch("axpy_c128", axpy_c128_kernel, grid_for(nnz), dim3(kBlock), 0, nnz, (const double*)tmp.p, kv.first.first,
           kv.first.second, reinterpret_cast<double2*>(out.dev));
  }
  out.finish();
  CFX_API_END
}

int cfx_assemble_vector_c128(cfx_form_t L, const double* scales, double* b)
{
  CFX_API_BEGIN
  require(L && b, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_vector_c128: null argument");
  require(L->rank == 1, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_vector_c128: form is not linear");
  const int64_t n = L->V->ndofs * L->V->bs;
  OutArray<double> out(b, 2 * n, true);
  DevArray<double> tmp(n);
  for (const auto& kv : groups(L, scales))
  {
    const bool whole = kv.second.size() == L->integrals.size();
    cfx_form_s* sub = whole ? nullptr : sub_form(L, kv.second);
    tmp.zero();
    const int rc = cfx_assemble_vector(whole ? L : sub, tmp.p);
    if (rc != CFX_OK) return rc;
    launch("axpy_c128", axpy_c128_kernel, grid_for(n), dim3(kBlock), 0, n, (const double*)tmp.p, kv.first.first, kv.first.second,
           reinterpret_cast<double2*>(out.dev));
  }
  out.finish();
  CFX_API_END
}

int cfx_apply_lifting_c128(cfx_form_t a, const int8_t* bc_markers, const double* bc_values, const double* x0, double alpha_re,
                           double alpha_im, const double* scales, double* b)
{
  CFX_API_BEGIN
  require(a && bc_markers && bc_values && b, CFX_ERR_INVALID_ARGUMENT, "cfx_apply_lifting_c128: null argument");
  require(a->rank == 2, CFX_ERR_INVALID_ARGUMENT, "cfx_apply_lifting_c128: form is not bilinear");
  const int64_t n = a->V->ndofs * a->V->bs, n1 = a->V1->ndofs * a->V1->bs;
  // b -= sum_k s_k alpha A_k (g - x0): the real lifting kernels on the real and the imaginary part of (g - x0)
  DevArray<double> g = to_device(bc_values, 2 * n1), x = to_device(x0, x0 ? 2 * n1 : 0);
  DevArray<double> part[2] = {DevArray<double>(n1), DevArray<double>(n1)}, xpart[2];
  for (int p = 0; p < 2; ++p)
  {
    launch("split_c128", split_c128_kernel, grid_for(n1), dim3(kBlock), 0, n1, reinterpret_cast<const double2*>(g.p), p, part[p].p);
    if (x0)
    {
      xpart[p].alloc(n1);
      launch("split_c128", split_c128_kernel, grid_for(n1), dim3(kBlock), 0, n1, reinterpret_cast<const double2*>(x.p), p, xpart[p].p);
    }
  }
  DevArray<int8_t> dm = to_device(bc_markers, n1);
  OutArray<double> out(b, 2 * n, true);
  DevArray<double> t[2] = {DevArray<double>(n), DevArray<double>(n)};
  for (const auto& kv : groups(a, scales))
  {
    const bool whole = kv.second.size() == a->integrals.size();
    cfx_form_s* sub = whole ? nullptr : sub_form(a, kv.second);
    for (int p = 0; p < 2; ++p)
    {
      t[p].zero();
      const int rc = cfx_apply_lifting(whole ? a : sub, dm.p, part[p].p, x0 ? xpart[p].p : nullptr, 1.0, t[p].p);
      if (rc != CFX_OK) return rc;
    }
    // t = -A_k (g - x0) (real, imaginary part); b += s_k alpha t
    const double sr = kv.first.first * alpha_re - kv.first.second * alpha_im, si = kv.first.first * alpha_im + kv.first.second * alpha_re;
    launch("axpy_c128", axpy2_c128_kernel, grid_for(n), dim3(kBlock), 0, n, (const double*)t[0].p, (const double*)t[1].p, sr, si,
           reinterpret_cast<double2*>(out.dev));
  }
  out.finish();
  CFX_API_END
}

int cfx_set_bc_c128(int64_t n, const int8_t* bc_markers, const double* bc_values, const double* x0, double alpha_re,
                    double alpha_im, double* b)
{
  CFX_API_BEGIN
  require(n >= 0 && bc_markers && bc_values && b, CFX_ERR_INVALID_ARGUMENT, "cfx_set_bc_c128: null argument");
  ctx().ensure();
  DevArray<int8_t> dm = to_device(bc_markers, n);
  DevArray<double> g = to_device(bc_values, 2 * n), x = to_device(x0, x0 ? 2 * n : 0);
  OutArray<double> out(b, 2 * n, true);
  launch("set_bc", set_bc_c128_kernel, grid_for(n), dim3(kBlock), 0, n, (const int8_t*)dm.p, reinterpret_cast<const double2*>(g.p),
         x0 ? reinterpret_cast<const double2*>(x.p) : (const double2*)nullptr, alpha_re, alpha_im, reinterpret_cast<double2*>(out.dev));
  out.finish();
  CFX_API_END
}

int cfx_deactivate_outside_c128(cfx_active_t d, cfx_pattern_t P, double* values, double* b, double diag_re, double diag_im,
                                double rhs_re, double rhs_im)
{
  CFX_API_BEGIN
  require(d && (values == nullptr || P), CFX_ERR_INVALID_ARGUMENT, "cfx_deactivate_outside_c128: null argument");
  const int64_t nrows = d->V->ndofs * d->V->bs;
  std::unique_ptr<OutArray<double>> ov, ob;
  if (values) ov = std::make_unique<OutArray<double>>(values, 2 * P->nnz.value(), true);
  if (b) ob = std::make_unique<OutArray<double>>(b, 2 * nrows, true);
  ZeroFlag err;
  cfx::active_lists(d);
  const int64_t n_inactive = d->n_inactive.value();
  if (n_inactive > 0)
    launch("deactivate", deactivate_c128_kernel<double2>, grid_for(n_inactive), dim3(kBlock), 0, n_inactive,
           (const int32_t*)d->inactive_dofs.p, P ? (const int64_t*)P->indptr.p : (const int64_t*)nullptr,
           P ? (const int32_t*)P->indices.p : (const int32_t*)nullptr, values ? reinterpret_cast<double2*>(ov->dev) : (double2*)nullptr,
           b ? reinterpret_cast<double2*>(ob->dev) : (double2*)nullptr, make_double2(diag_re, diag_im), make_double2(rhs_re, rhs_im), err.p);
  require(!read_scalar(err.p), CFX_ERR_RUNTIME, "Deactivated matrix row has no diagonal entry.");
  if (ov) ov->finish();
  if (ob) ob->finish();
  CFX_API_END
}

// ---- complex64: the <std::complex<float>, float> rows of python/cutfemx/wrappers/fem.cpp:490-500.  Containers (CSR values,
// vectors, Dirichlet data) are interleaved float32; constants stay double; every sum is formed in fp64 by the complex128
// path above and rounded once.
int cfx_assemble_matrix_c64(cfx_form_t a, cfx_pattern_t P, const int8_t* bc0, const int8_t* bc1, const double* scales,
                            int zero_first, float* values)
{
  CFX_API_BEGIN
  require(a && P && values, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_matrix_c64: null argument");
  const int64_t nnz = P->nnz.value();
  DevArray<double> wide(2 * nnz);
  const int rc = cfx_assemble_matrix_c128(a, P, bc0, bc1, scales, 1, wide.p);
  if (rc != CFX_OK) return rc;
  if (zero_first) narrow_c64<false>(wide.p, values, 2 * nnz); else narrow_c64<true>(wide.p, values, 2 * nnz);
  CFX_API_END
}

int cfx_assemble_vector_c64(cfx_form_t L, const double* scales, float* b)
{
  CFX_API_BEGIN
  require(L && b, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_vector_c64: null argument");
  require(L->rank == 1, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_vector_c64: form is not linear");
  const int64_t n = L->V->ndofs * L->V->bs;
  DevArray<double> wide(2 * n);
  wide.zero();
  const int rc = cfx_assemble_vector_c128(L, scales, wide.p);
  if (rc != CFX_OK) return rc;
  narrow_c64<true>(wide.p, b, 2 * n);
  CFX_API_END
}

int cfx_apply_lifting_c64(cfx_form_t a, const int8_t* bc_markers, const float* bc_values, const float* x0, double alpha_re,
                          double alpha_im, const double* scales, float* b)
{
  CFX_API_BEGIN
  require(a && bc_markers && bc_values && b, CFX_ERR_INVALID_ARGUMENT, "cfx_apply_lifting_c64: null argument");
  require(a->rank == 2, CFX_ERR_INVALID_ARGUMENT, "cfx_apply_lifting_c64: form is not bilinear");
  const int64_t n = a->V->ndofs * a->V->bs, n1 = a->V1->ndofs * a->V1->bs;
  DevArray<double> g = widen_c64(bc_values, 2 * n1), x = widen_c64(x0, x0 ? 2 * n1 : 0), wide(2 * n);
  wide.zero();
  const int rc = cfx_apply_lifting_c128(a, bc_markers, g.p, x0 ? x.p : nullptr, alpha_re, alpha_im, scales, wide.p);
  if (rc != CFX_OK) return rc;
  narrow_c64<true>(wide.p, b, 2 * n);
  CFX_API_END
}

int cfx_set_bc_c64(int64_t n, const int8_t* bc_markers, const float* bc_values, const float* x0, double alpha_re,
                   double alpha_im, float* b)
{
  CFX_API_BEGIN
  require(n >= 0 && bc_markers && bc_values && b, CFX_ERR_INVALID_ARGUMENT, "cfx_set_bc_c64: null argument");
  ctx().ensure();
  // (unmarked entries make the round trip float -> double -> float unchanged)
  DevArray<double> g = widen_c64(bc_values, 2 * n), x = widen_c64(x0, x0 ? 2 * n : 0), wide = widen_c64(b, 2 * n);
  const int rc = cfx_set_bc_c128(n, bc_markers, g.p, x0 ? x.p : nullptr, alpha_re, alpha_im, wide.p);
  if (rc != CFX_OK) return rc;
  narrow_c64<false>(wide.p, b, 2 * n);
  CFX_API_END
}

int cfx_deactivate_outside_c64(cfx_active_t d, cfx_pattern_t P, float* values, float* b, double diag_re, double diag_im,
                               double rhs_re, double rhs_im)
{
  CFX_API_BEGIN
  require(d && (values == nullptr || P), CFX_ERR_INVALID_ARGUMENT, "cfx_deactivate_outside_c64: null argument");
  const int64_t nrows = d->V->ndofs * d->V->bs;
  std::unique_ptr<OutArray<float>> ov, ob;
  if (values) ov = std::make_unique<OutArray<float>>(values, 2 * P->nnz.value(), true);
  if (b) ob = std::make_unique<OutArray<float>>(b, 2 * nrows, true);
  ZeroFlag err;
  cfx::active_lists(d);
  const int64_t n_inactive = d->n_inactive.value();
  if (n_inactive > 0)
    launch("deactivate", deactivate_c128_kernel<float2>, grid_for(n_inactive), dim3(kBlock), 0, n_inactive,
           (const int32_t*)d->inactive_dofs.p, P ? (const int64_t*)P->indptr.p : (const int64_t*)nullptr,
           P ? (const int32_t*)P->indices.p : (const int32_t*)nullptr, values ? reinterpret_cast<float2*>(ov->dev) : (float2*)nullptr,
           b ? reinterpret_cast<float2*>(ob->dev) : (float2*)nullptr, make_float2((float)diag_re, (float)diag_im),
           make_float2((float)rhs_re, (float)rhs_im), err.p);
  require(!read_scalar(err.p), CFX_ERR_RUNTIME, "Deactivated matrix row has no diagonal entry.");
  if (ov) ov->finish();
  if (ob) ob->finish();
  CFX_API_END
}

} // extern "C"
