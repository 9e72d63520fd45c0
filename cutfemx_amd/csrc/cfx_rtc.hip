// cutfemx_amd: user-supplied integrands, compiled at run time with hipRTC.
//
// The reference compiles every form to its own tabulate_tensor kernel (runintgen / FFCx: Form.h:59-75,
// python/cutfemx/_runintgen_adapter.py:181-217) and calls that CPU function pointer per entity.  The GPU counterpart:
// the caller registers the SOURCE of a device function with the argument list below; the engine compiles it for
// gfx950 together with a stage-1 wrapper (one thread per entity: geometry, packed coefficient, rule slice -> the
// function -> local tensor staged in HBM) and the existing row gather -- or the entity-parallel atomic scatter --
// consumes the staged tensors exactly as it does for the built-in integrands that are not formed in line.
//
//   __device__ void NAME(double* A,                    // local tensor, zero on entry: [ND][ND] row-major (rank 2) or [ND]
//                        const double* w,              // packed coefficient: the ND cell dofs of `coefficient`, or NULL
//                        const double* c,              // constants: cfx_integral.params[8]
//                        const double* coordinate_dofs,// vertex coordinates of the cell, [(TDIM+1)][3] (UFCx layout)
//                        int nq, const double* points, // rule of this entity: [nq][TDIM] parent-reference coordinates
//                        const double* weights,        // [nq] PHYSICAL-measure weights (standard entities: reference
//                                                      // weights x |det J|; runtime rules: as the rules carry them)
//                        const double* point_data);    // [nq][point_stride] per-point data of the rules, or NULL
//
// CFX_TDIM, CFX_ND (scalar dofs per cell of the form's Lagrange space), CFX_BS (block size of the space) and CFX_NDB =
// CFX_ND * CFX_BS (local dimension: entry (i, a) = dof i, component a sits at i * CFX_BS + a) are defined when the source
// is compiled; the prelude below offers cfx_tabulate_p1 / cfx_tabulate_p2 (values and reference gradients) and
// cfx_inverse_jacobian.
//
// Interior-facet integrands (cfx_integrand_register_facet; the reference's call with entity_local_index = {lf0, lf1},
// assemble_matrix_impl.h:528-542):
//
//   __device__ void NAME(double* A,                     // macro tensor, zero on entry: [2 NDB][2 NDB] row-major, rows and
//                                                       // columns = [cell 0 dofs, cell 1 dofs]: [[00, 01], [10, 11]]
//                        const double* w,               // packed coefficient of both cells [2][ND], or NULL
//                        const double* c,               // constants: cfx_integral.params[8]
//                        const double* coordinate_dofs, // [2][(TDIM+1)][3]: cell 0, then cell 1
//                        const int* entity_local_index, // {lf0, lf1}: the local facet in either cell
//                        int nq,
//                        const double* points0,         // [nq][TDIM] the facet's points in the reference cell of cell 0
//                        const double* points1,         // ... and the SAME physical points in the reference cell of cell 1
//                        const double* weights);        // [nq] physical (facet-measure) weights
#include <dlfcn.h>

#include <mutex>

#include <hip/hiprtc.h>

#include "cfx_device.h"

namespace cfx
{
int quad_npoints(int dim, int degree);
const double* quad_points_host(int dim, int degree);  // cfx_quadhost.cpp
const double* quad_weights_host(int dim, int degree);
} // namespace cfx

using namespace cfx;

namespace
{
// what the compiled wrapper takes (the same struct is spelled out in the wrapper's source below)
struct RtcArgs
{
  const double* x;
  const int32_t* conn;
  const int32_t* dofmap;
  int64_t n_cap;
  const int64_t* n_dev;
  const int32_t* entities;
  const int32_t* offsets;
  const int32_t* parent_map;
  const double* points;
  const double* weights;
  const double* point_data;
  int point_stride, runtime, nref, rank;
  const double* ref_points;
  const double* ref_weights;
  double params[8];
  const double* coeff;
  double* out;
  int64_t out_stride; // out_mode 1 / 2: entry i of the local vector lives at out[i * out_stride + index]
  int out_mode;       // 0: [entity][NT]; 1: [i][stride] indexed by the entity; 2: [i][stride] indexed by the cell
  int coeff_bs;       // components per dof of `coeff` (rank 1 on a vector space: the space's block size; else 1)
};

const char* kPrelude = R"RTC(
typedef long long cfx_i64;
typedef int cfx_i32;
struct RtcArgs
{
  const double* x; const cfx_i32* conn; const cfx_i32* dofmap;
  cfx_i64 n_cap; const cfx_i64* n_dev;
  const cfx_i32* entities; const cfx_i32* offsets; const cfx_i32* parent_map;
  const double* points; const double* weights; const double* point_data;
  int point_stride, runtime, nref, rank;
  const double* ref_points; const double* ref_weights;
  double params[8];
  const double* coeff;
  double* out;
  cfx_i64 out_stride;
  int out_mode, coeff_bs;
};
// Lagrange bases on the reference simplex, vertex order of the mesh connectivity; degree 2: vertices, then the edge
// midpoints in the order of the engine's dofmaps (cutfemx_amd.lagrange_dofmap)
__device__ inline void cfx_tabulate_p1(const double* X, double* N, double (*dN)[CFX_TDIM])
{
  double l0 = 1.0;
  for (int t = 0; t < CFX_TDIM; ++t) l0 -= X[t];
  N[0] = l0;
  for (int t = 0; t < CFX_TDIM; ++t) { N[t + 1] = X[t]; dN[0][t] = -1.0; }
  for (int i = 0; i < CFX_TDIM; ++i)
    for (int t = 0; t < CFX_TDIM; ++t) dN[i + 1][t] = (i == t) ? 1.0 : 0.0;
}
// degree 2: vertices, then edges -- tri: (1,2),(0,2),(0,1); tet: (2,3),(1,3),(1,2),(0,3),(0,2),(0,1) (the Basix order)
__device__ inline void cfx_tabulate_p2(const double* X, double* N, double (*dN)[CFX_TDIM])
{
  double lam[CFX_TDIM + 1], g[CFX_TDIM + 1][CFX_TDIM];
  lam[0] = 1.0;
  for (int t = 0; t < CFX_TDIM; ++t) { lam[0] -= X[t]; lam[t + 1] = X[t]; }
  for (int i = 0; i <= CFX_TDIM; ++i)
    for (int t = 0; t < CFX_TDIM; ++t) g[i][t] = (i == 0) ? -1.0 : ((i - 1 == t) ? 1.0 : 0.0);
  for (int i = 0; i <= CFX_TDIM; ++i)
  {
    N[i] = lam[i] * (2.0 * lam[i] - 1.0);
    for (int t = 0; t < CFX_TDIM; ++t) dN[i][t] = (4.0 * lam[i] - 1.0) * g[i][t];
  }
#if CFX_TDIM == 2
  const int ne = 3, ea[3] = {1, 0, 0}, eb[3] = {2, 2, 1};
#else
  const int ne = 6, ea[6] = {2, 1, 1, 0, 0, 0}, eb[6] = {3, 3, 2, 3, 2, 1};
#endif
  for (int e = 0; e < ne; ++e)
  {
    N[CFX_TDIM + 1 + e] = 4.0 * lam[ea[e]] * lam[eb[e]];
    for (int t = 0; t < CFX_TDIM; ++t) dN[CFX_TDIM + 1 + e][t] = 4.0 * (lam[ea[e]] * g[eb[e]][t] + g[ea[e]][t] * lam[eb[e]]);
  }
}
// the basis of the form's space: CFX_ND dofs per cell
__device__ inline void cfx_tabulate(const double* X, double* N, double (*dN)[CFX_TDIM])
{
#if CFX_ND == CFX_TDIM + 1
  cfx_tabulate_p1(X, N, dN);
#else
  cfx_tabulate_p2(X, N, dN);
#endif
}
// K[t][d] = d xi_t / d x_d and det J of the affine cell with vertices coordinate_dofs[(CFX_TDIM + 1)][3]
__device__ inline double cfx_inverse_jacobian(const double* xc, double (*K)[CFX_TDIM])
{
  double J[CFX_TDIM][CFX_TDIM];
  for (int d = 0; d < CFX_TDIM; ++d)
    for (int t = 0; t < CFX_TDIM; ++t) J[d][t] = xc[3 * (t + 1) + d] - xc[d];
#if CFX_TDIM == 2
  const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
  K[0][0] = J[1][1] / det;  K[0][1] = -J[0][1] / det;
  K[1][0] = -J[1][0] / det; K[1][1] = J[0][0] / det;
#else
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
  K[0][0] = c00 / det;
  K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
  K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
  K[1][0] = c01 / det;
  K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
  K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
  K[2][0] = c02 / det;
  K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
  K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
#endif
  return det;
}
// UFL CellDiameter: the largest vertex-to-vertex distance
__device__ inline double cfx_cell_diameter(const double* xc)
{
  double h2 = 0.0;
  for (int i = 0; i <= CFX_TDIM; ++i)
    for (int j = i + 1; j <= CFX_TDIM; ++j)
    {
      double d2 = 0.0;
      for (int d = 0; d < CFX_TDIM; ++d) d2 += (xc[3 * i + d] - xc[3 * j + d]) * (xc[3 * i + d] - xc[3 * j + d]);
      h2 = d2 > h2 ? d2 : h2;
    }
  return sqrt(h2);
}
)RTC";

// the wrapper: CFX_USER_FN is the registered function
const char* kWrapper = R"RTC(
#define CFX_MAXQ 64
extern "C" __global__ void __launch_bounds__(256) cfx_user_stage1(RtcArgs A)
{
  const cfx_i64 e = (cfx_i64)blockIdx.x * 256 + threadIdx.x;
  cfx_i64 n = A.n_cap;
  if (A.n_dev) { const cfx_i64 v = *A.n_dev; n = v < n ? v : n; }
  if (e >= n) return;
  const cfx_i64 cell = A.runtime ? A.parent_map[e] : A.entities[e];
  double xc[(CFX_TDIM + 1) * 3];
  for (int v = 0; v <= CFX_TDIM; ++v)
  {
    const cfx_i64 node = A.conn[cell * (CFX_TDIM + 1) + v];
    for (int d = 0; d < 3; ++d) xc[3 * v + d] = A.x[3 * node + d];
  }
  // packed coefficient: [ND][coeff_bs] (a scalar Function for bilinear forms, a Function of the form's space -- CFX_BS
  // components per dof -- for linear forms on vector spaces)
  double w[CFX_NDB];
  if (A.coeff)
    for (int j = 0; j < CFX_ND; ++j)
      for (int b = 0; b < A.coeff_bs; ++b) w[j * A.coeff_bs + b] = A.coeff[(cfx_i64)A.dofmap[cell * CFX_ND + j] * A.coeff_bs + b];
  const int NT = A.rank == 2 ? CFX_NDB * CFX_NDB : CFX_NDB;
  double T[CFX_NDB * CFX_NDB];
  for (int i = 0; i < CFX_NDB * CFX_NDB; ++i) T[i] = 0.0;
  if (A.runtime)
  {
    const int q0 = A.offsets[e], nq = A.offsets[e + 1] - q0;
    CFX_USER_FN(T, A.coeff ? w : (const double*)0, A.params, xc, nq, A.points + (cfx_i64)q0 * CFX_TDIM, A.weights + q0,
                A.point_data ? A.point_data + (cfx_i64)q0 * A.point_stride : (const double*)0);
  }
  else
  {
    double K[CFX_TDIM][CFX_TDIM];
    const double det = fabs(cfx_inverse_jacobian(xc, K));
    double wq[CFX_MAXQ];
    const int nq = A.nref < CFX_MAXQ ? A.nref : CFX_MAXQ;
    for (int q = 0; q < nq; ++q) wq[q] = A.ref_weights[q] * det;
    CFX_USER_FN(T, A.coeff ? w : (const double*)0, A.params, xc, nq, A.ref_points, wq, (const double*)0);
  }
  if (A.out_mode == 0)
    for (int i = 0; i < NT; ++i) A.out[e * NT + i] = T[i];
  else
  {
    const cfx_i64 at = A.out_mode == 2 ? cell : e;
    for (int i = 0; i < NT; ++i) A.out[(cfx_i64)i * A.out_stride + at] = T[i];
  }
}
)RTC";

// the facet wrapper: one thread per (c0, lf0, c1, lf1) row.  The points: the reference facet rule pushed to physical
// space from cell 0's facet lf0 (its vertices in ascending local order) and pulled back to both reference cells -- what
// the built-in facet kernels do (cfx_elem.h: facet_local_row) -- with the facet's measure in the weights.
struct RtcFacetArgs
{
  const double* x;
  const int32_t* conn;
  const int32_t* dofmap;
  int64_t n_cap;
  const int64_t* n_dev;
  const int32_t* rows;
  int nref, pad;
  const double* ref_points;
  const double* ref_weights;
  double params[8];
  const double* coeff;
  double* out;
};
const char* kFacetWrapper = R"RTC(
struct RtcFacetArgs
{
  const double* x; const cfx_i32* conn; const cfx_i32* dofmap;
  cfx_i64 n_cap; const cfx_i64* n_dev;
  const cfx_i32* rows;
  int nref, pad;
  const double* ref_points; const double* ref_weights;
  double params[8];
  const double* coeff;
  double* out;
};
#define CFX_MAXQF 32
extern "C" __global__ void __launch_bounds__(256) cfx_user_stage1(RtcFacetArgs A)
{
  const cfx_i64 f = (cfx_i64)blockIdx.x * 256 + threadIdx.x;
  cfx_i64 n = A.n_cap;
  if (A.n_dev) { const cfx_i64 v = *A.n_dev; n = v < n ? v : n; }
  if (f >= n) return;
  const cfx_i64 c0 = A.rows[4 * f], c1 = A.rows[4 * f + 2];
  const int eli[2] = {A.rows[4 * f + 1], A.rows[4 * f + 3]};
  double xc[2 * (CFX_TDIM + 1) * 3];
  for (int s = 0; s < 2; ++s)
    for (int v = 0; v <= CFX_TDIM; ++v)
    {
      const cfx_i64 node = A.conn[(s ? c1 : c0) * (CFX_TDIM + 1) + v];
      for (int d = 0; d < 3; ++d) xc[(s * (CFX_TDIM + 1) + v) * 3 + d] = A.x[3 * node + d];
    }
  double w[2 * CFX_ND];
  if (A.coeff)
    for (int s = 0; s < 2; ++s)
      for (int j = 0; j < CFX_ND; ++j) w[s * CFX_ND + j] = A.coeff[A.dofmap[(s ? c1 : c0) * CFX_ND + j]];
  double K0[CFX_TDIM][CFX_TDIM], K1[CFX_TDIM][CFX_TDIM];
  (void)cfx_inverse_jacobian(xc, K0);
  (void)cfx_inverse_jacobian(xc + (CFX_TDIM + 1) * 3, K1);
  // the facet's vertices: those of cell 0 except lf0, ascending local index
  double xf[CFX_TDIM][3];
  {
    int k = 0;
    for (int i = 0; i <= CFX_TDIM; ++i)
    {
      if (i == eli[0]) continue;
      for (int d = 0; d < 3; ++d) xf[k][d] = xc[3 * i + d];
      ++k;
    }
  }
  double measure;
#if CFX_TDIM == 2
  {
    const double dx = xf[1][0] - xf[0][0], dy = xf[1][1] - xf[0][1];
    measure = sqrt(dx * dx + dy * dy);
  }
#else
  {
    double a[3], b[3];
    for (int d = 0; d < 3; ++d) { a[d] = xf[1][d] - xf[0][d]; b[d] = xf[2][d] - xf[0][d]; }
    const double cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
    measure = sqrt(cx * cx + cy * cy + cz * cz);
  }
#endif
  const int nq = A.nref < CFX_MAXQF ? A.nref : CFX_MAXQF;
  double P0[CFX_MAXQF * CFX_TDIM], P1[CFX_MAXQF * CFX_TDIM], wq[CFX_MAXQF];
  for (int q = 0; q < nq; ++q)
  {
    double l0 = 1.0, xq[CFX_TDIM];
    for (int t = 0; t < CFX_TDIM - 1; ++t) l0 -= A.ref_points[q * (CFX_TDIM - 1) + t];
    for (int d = 0; d < CFX_TDIM; ++d)
    {
      double v = l0 * xf[0][d];
      for (int t = 0; t < CFX_TDIM - 1; ++t) v += A.ref_points[q * (CFX_TDIM - 1) + t] * xf[t + 1][d];
      xq[d] = v;
    }
    for (int t = 0; t < CFX_TDIM; ++t)
    {
      double a = 0.0, b = 0.0;
      for (int d = 0; d < CFX_TDIM; ++d)
      {
        a += K0[t][d] * (xq[d] - xc[d]);
        b += K1[t][d] * (xq[d] - xc[(CFX_TDIM + 1) * 3 + d]);
      }
      P0[q * CFX_TDIM + t] = a;
      P1[q * CFX_TDIM + t] = b;
    }
    wq[q] = A.ref_weights[q] * measure;
  }
  double T[4 * CFX_NDB * CFX_NDB];
  for (int i = 0; i < 4 * CFX_NDB * CFX_NDB; ++i) T[i] = 0.0;
  CFX_USER_FN(T, A.coeff ? w : (const double*)0, A.params, xc, eli, nq, P0, P1, wq);
  for (int i = 0; i < 4 * CFX_NDB * CFX_NDB; ++i) A.out[f * (4 * CFX_NDB * CFX_NDB) + i] = T[i];
}
)RTC";

// hipRTC through dlopen: the engine does not link it (a process that never registers an integrand never loads it,
// and one that has PyTorch's copy mapped gets that copy)
struct Rtc
{
  void* lib = nullptr;
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetCodeSize) code_size = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
};

Rtc& rtc()
{
  static Rtc r;
  if (r.lib) return r;
  for (const char* name : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"})
  {
    r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.lib) break;
  }
  if (!r.lib) throw Error(CFX_ERR_RUNTIME, std::string("cfx_integrand_register: cannot load hipRTC (") + dlerror() + ")");
  auto sym = [&](const char* s) {
    void* p = dlsym(r.lib, s);
    if (!p) throw Error(CFX_ERR_RUNTIME, std::string("hipRTC: missing symbol ") + s);
    return p;
  };
  r.create = reinterpret_cast<decltype(r.create)>(sym("hiprtcCreateProgram"));
  r.compile = reinterpret_cast<decltype(r.compile)>(sym("hiprtcCompileProgram"));
  r.log_size = reinterpret_cast<decltype(r.log_size)>(sym("hiprtcGetProgramLogSize"));
  r.log = reinterpret_cast<decltype(r.log)>(sym("hiprtcGetProgramLog"));
  r.code_size = reinterpret_cast<decltype(r.code_size)>(sym("hiprtcGetCodeSize"));
  r.code = reinterpret_cast<decltype(r.code)>(sym("hiprtcGetCode"));
  r.destroy = reinterpret_cast<decltype(r.destroy)>(sym("hiprtcDestroyProgram"));
  return r;
}

struct UserIntegrand
{
  std::string name, source;
  int rank = 2;
  int kind = 0;                              // 0: cell integrand, 1: interior-facet integrand
  std::map<int, std::vector<char>> code;     // (tdim * 100 + nd) * 10 + bs -> code object for gfx950
  std::map<int, hipModule_t> module;
  std::map<int, hipFunction_t> function;
};

std::vector<UserIntegrand>& integrands()
{
  static std::vector<UserIntegrand> v;
  return v;
}

// compile for (tdim, nd); no GPU needed (the code object is loaded on first launch)
// (registry and per-variant maps are shared state: one lock around every use)
std::mutex& rtc_mutex()
{
  static std::mutex m;
  return m;
}

const std::vector<char>& compiled(UserIntegrand& u, int tdim, int nd, int bs = 1)
{
  const int key = (tdim * 100 + nd) * 10 + bs;
  auto it = u.code.find(key);
  if (it != u.code.end()) return it->second;
  Rtc& r = rtc();
  const std::string src = std::string("#define CFX_TDIM ") + std::to_string(tdim) + "\n#define CFX_ND " + std::to_string(nd)
                          + "\n#define CFX_BS " + std::to_string(bs) + "\n#define CFX_NDB (CFX_ND * CFX_BS)"
                          + "\n#define CFX_USER_FN " + u.name + "\n" + kPrelude + "\n" + u.source + "\n"
                          + (u.kind == 1 ? kFacetWrapper : kWrapper);
  hiprtcProgram prog = nullptr;
  if (r.create(&prog, src.c_str(), (u.name + ".hip").c_str(), 0, nullptr, nullptr) != HIPRTC_SUCCESS)
    throw Error(CFX_ERR_RUNTIME, "hiprtcCreateProgram failed");
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast"};
  const hiprtcResult rc = r.compile(prog, 4, opts);
  size_t ls = 0;
  r.log_size(prog, &ls);
  std::string log(ls, '\0');
  if (ls > 1) r.log(prog, log.data());
  if (rc != HIPRTC_SUCCESS)
  {
    r.destroy(&prog);
    throw Error(CFX_ERR_INVALID_ARGUMENT, "cfx_integrand_register: the integrand '" + u.name + "' does not compile:\n" + log);
  }
  size_t cs = 0;
  r.code_size(prog, &cs);
  std::vector<char> code(cs);
  r.code(prog, code.data());
  r.destroy(&prog);
  return u.code.emplace(key, std::move(code)).first->second;
}

hipFunction_t function_of(UserIntegrand& u, int tdim, int nd, int bs = 1)
{
  const int key = (tdim * 100 + nd) * 10 + bs;
  auto it = u.function.find(key);
  if (it != u.function.end()) return it->second;
  const std::vector<char>& code = compiled(u, tdim, nd, bs);
  hipModule_t mod = nullptr;
  CFX_HIP(hipModuleLoadData(&mod, code.data()));
  hipFunction_t fn = nullptr;
  CFX_HIP(hipModuleGetFunction(&fn, mod, "cfx_user_stage1"));
  u.module[key] = mod;
  u.function[key] = fn;
  return fn;
}

// reference rule of (dim, degree) in HBM for the wrapper (the engine's own kernels read the tables from their module)
struct RuleCopy { DevArray<double> points, weights; int n = 0; };
RuleCopy& reference_rule(int dim, int degree)
{
  static std::map<int, RuleCopy> cache;
  RuleCopy& rc = cache[dim * 100 + degree];
  if (rc.n == 0)
  {
    rc.n = quad_npoints(dim, degree);
    rc.points = to_device(quad_points_host(dim, degree), (int64_t)rc.n * dim);
    rc.weights = to_device(quad_weights_host(dim, degree), (int64_t)rc.n);
  }
  return rc;
}

bool user_integrand_known_locked(int kernel) { return kernel >= CFX_K_USER_BASE && kernel - CFX_K_USER_BASE < (int)integrands().size(); }

// one thread per entity, 256 per block, on the library stream -- with what cfx::launch does for the engine's own
// kernels: the launch trace, the 2^32 work-item check, HIP-event bracketing when profiling
void module_launch(hipFunction_t fn, int64_t n, void* args)
{
  void* kargs[] = {args};
  const int64_t blocks = (n + 255) / 256;
  if (blocks * 256 > 0xffffffffll) throw Error(CFX_ERR_RUNTIME, "user_integrand: launch exceeds 2^32 threads");
  const unsigned grid = (unsigned)blocks;
  Context& c = ctx();
  c.last_launch = "user_integrand";
  static const bool trace = getenv("CFX_LAUNCH_TRACE") != nullptr;
  if (trace) fprintf(stderr, "cutfemx_amd: launch user_integrand grid %u\n", grid);
  if (c.profile)
  {
    hipEvent_t e0 = c.get_event(), e1 = c.get_event();
    CFX_HIP(hipEventRecord(e0, c.stream));
    CFX_HIP(hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, c.stream, kargs, nullptr));
    CFX_HIP(hipEventRecord(e1, c.stream));
    c.pending.push_back({c.entry("user_integrand"), e0, e1});
  }
  else
    CFX_HIP(hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, c.stream, kargs, nullptr));
}
} // namespace

namespace cfx
{
bool user_integrand_known(int kernel)
{
  std::lock_guard<std::mutex> lock(rtc_mutex());
  return user_integrand_known_locked(kernel);
}
int user_integrand_rank(int kernel)
{
  std::lock_guard<std::mutex> lock(rtc_mutex());
  require(user_integrand_known_locked(kernel), CFX_ERR_INVALID_ARGUMENT, "unknown user integrand id");
  return integrands()[kernel - CFX_K_USER_BASE].rank;
}
int user_integrand_kind(int kernel)
{
  std::lock_guard<std::mutex> lock(rtc_mutex());
  require(user_integrand_known_locked(kernel), CFX_ERR_INVALID_ARGUMENT, "unknown user integrand id");
  return integrands()[kernel - CFX_K_USER_BASE].kind;
}

// stage 1 of a user integrand over the standard entities (runtime = false) or the runtime rules of integral I of form a:
// local tensors into `out` (out_mode / out_stride: see RtcArgs); `first` / `count` >= 0 restrict the launch to one entity
void user_stage1(const cfx_form_s* a, const cfx_integral_dev& I, bool runtime, double* out, int out_mode, int64_t out_stride,
                 int64_t only_index)
{
  std::lock_guard<std::mutex> lock(rtc_mutex());
  const cfx_space_s* V = a->V;
  require(user_integrand_known_locked(I.kernel), CFX_ERR_INVALID_ARGUMENT, "unknown user integrand id");
  require((V->degree == 1 || V->degree == 2) && I.type == CFX_CELL && !a->rectangular(), CFX_ERR_INVALID_ARGUMENT,
          "user integrands serve cell integrals of Lagrange spaces of degree 1 or 2");
  UserIntegrand& u = integrands()[I.kernel - CFX_K_USER_BASE];
  require(u.kind == 0, CFX_ERR_INVALID_ARGUMENT, "this user integrand was registered for interior-facet integrals");
  require(V->ndofs_cell * V->bs <= 30, CFX_ERR_INVALID_ARGUMENT, "user integrands: at most 30 local dofs");
  const int tdim = V->mesh->tdim, nd = V->ndofs_cell;
  RtcArgs A{};
  A.x = V->mesh->x.p; A.conn = V->mesh->conn.p; A.dofmap = V->dofmap.p;
  A.rank = a->rank; A.runtime = runtime ? 1 : 0;
  for (int k = 0; k < 8; ++k) A.params[k] = I.params[k];
  A.coeff = I.coefficient.n > 0 ? I.coefficient.p : nullptr;
  A.coeff_bs = a->rank == 1 ? V->bs : 1; // (as the built-in kernels pack it: cfx_fem.hip, assemble_cells_kernel)
  A.out = out; A.out_mode = out_mode; A.out_stride = out_stride;
  DevN n;
  if (runtime)
  {
    require(I.rules != nullptr, CFX_ERR_INVALID_ARGUMENT, "user integrand: no runtime rules");
    const int64_t o = only_index >= 0 ? only_index : 0;
    A.offsets = I.rules->offsets.p + o; A.parent_map = I.rules->parent_map.p + o;
    A.points = I.rules->points.p; A.weights = I.rules->weights.p;
    // (offsets are absolute: the slices of points / weights / point_data start at offsets[e])
    A.point_data = I.point_data.n > 0 ? I.point_data.p : nullptr; A.point_stride = I.point_stride;
    n = only_index >= 0 ? DevN(1) : I.rules->nr.devn();
  }
  else
  {
    RuleCopy& rc = reference_rule(tdim, I.qdegree);
    require(rc.n <= 64, CFX_ERR_INVALID_ARGUMENT, "user integrand: the standard rule has more than 64 points");
    A.ref_points = rc.points.p; A.ref_weights = rc.weights.p; A.nref = rc.n;
    A.entities = I.entities.p + (only_index >= 0 ? only_index : 0);
    n = only_index >= 0 ? DevN(1) : I.n_entities.devn();
  }
  A.n_cap = n.cap; A.n_dev = n.dev;
  if (n.cap == 0) return;
  module_launch(function_of(u, tdim, nd, V->bs), n.cap, &A);
}

// stage 1 of a user interior-facet integrand over the (c0, lf0, c1, lf1) rows of integral I: macro tensors
// [facet][2 NDB][2 NDB] into `out` (the layout the row gather and the scatter read); only_index >= 0: one facet
void user_stage1_facets(const cfx_form_s* a, const cfx_integral_dev& I, double* out, int64_t only_index)
{
  std::lock_guard<std::mutex> lock(rtc_mutex());
  const cfx_space_s* V = a->V;
  require(user_integrand_known_locked(I.kernel), CFX_ERR_INVALID_ARGUMENT, "unknown user integrand id");
  UserIntegrand& u = integrands()[I.kernel - CFX_K_USER_BASE];
  require(u.kind == 1 && I.type == CFX_INTERIOR_FACET, CFX_ERR_INVALID_ARGUMENT,
          "this user integrand was registered for cell integrals (cfx_integrand_register_facet registers facet integrands)");
  require((V->degree == 1 || V->degree == 2) && !a->rectangular() && a->rank == 2 && I.rules == nullptr, CFX_ERR_INVALID_ARGUMENT,
          "user facet integrands serve bilinear forms on Lagrange spaces of degree 1 or 2 over standard facets");
  const int tdim = V->mesh->tdim, nd = V->ndofs_cell;
  require(2 * nd * V->bs <= 24, CFX_ERR_INVALID_ARGUMENT,
          "user facet integrands: at most 24 macro dofs (scalar spaces of degree 1 or 2, vector spaces of degree 1)");
  RtcFacetArgs A{};
  A.x = V->mesh->x.p; A.conn = V->mesh->conn.p; A.dofmap = V->dofmap.p;
  for (int k = 0; k < 8; ++k) A.params[k] = I.params[k];
  A.coeff = I.coefficient.n > 0 ? I.coefficient.p : nullptr;
  RuleCopy& rc = reference_rule(tdim - 1, I.qdegree);
  require(rc.n <= 32, CFX_ERR_INVALID_ARGUMENT, "user facet integrand: the facet rule has more than 32 points");
  A.ref_points = rc.points.p; A.ref_weights = rc.weights.p; A.nref = rc.n;
  A.rows = I.entities.p + 4 * (only_index >= 0 ? only_index : 0);
  A.out = out;
  const DevN n = only_index >= 0 ? DevN(1) : I.n_entities.devn();
  A.n_cap = n.cap; A.n_dev = n.dev;
  if (n.cap == 0) return;
  module_launch(function_of(u, tdim, nd, V->bs), n.cap, &A);
}
} // namespace cfx

extern "C" {

static int register_integrand(const char* name, const char* source, int rank, int kind, int tdim, int nd, int bs, int* kernel_id)
{
  CFX_API_BEGIN
  require(name && source && kernel_id, CFX_ERR_INVALID_ARGUMENT, "cfx_integrand_register: null argument");
  require(rank == 1 || rank == 2, CFX_ERR_INVALID_ARGUMENT, "cfx_integrand_register: rank must be 1 or 2");
  require(kind == 0 || rank == 2, CFX_ERR_INVALID_ARGUMENT, "cfx_integrand_register_facet: interior-facet integrands are bilinear");
  for (const char* p = name; *p; ++p)
    require((*p >= 'a' && *p <= 'z') || (*p >= 'A' && *p <= 'Z') || *p == '_' || (p != name && *p >= '0' && *p <= '9'),
            CFX_ERR_INVALID_ARGUMENT, "cfx_integrand_register: the name must be a C identifier");
  require((tdim == 2 || tdim == 3) && nd >= tdim + 1 && nd <= 10 && bs >= 1 && bs <= 3, CFX_ERR_INVALID_ARGUMENT,
          "cfx_integrand_register: variant to validate: tdim 2 or 3, dofs per cell of a degree-1 or degree-2 space, block size 1..3");
  std::lock_guard<std::mutex> lock(rtc_mutex());
  UserIntegrand u;
  u.name = name; u.source = source; u.rank = rank; u.kind = kind;
  // compiled here for ONE variant so that a source that does not compile is refused at registration (no GPU needed:
  // hipRTC targets gfx950 explicitly); the other (tdim, dofs per cell, block size) variants are compiled on first use
  (void)compiled(u, tdim, nd, bs);
  integrands().push_back(std::move(u));
  *kernel_id = CFX_K_USER_BASE + (int)integrands().size() - 1;
  CFX_API_END
}

int cfx_integrand_register(const char* name, const char* source, int rank, int* kernel_id)
{
  return register_integrand(name, source, rank, 0, 3, 4, 1, kernel_id);
}

int cfx_integrand_register_facet(const char* name, const char* source, int* kernel_id)
{
  return register_integrand(name, source, 2, 1, 3, 4, 1, kernel_id);
}

int cfx_integrand_register_variant(const char* name, const char* source, int rank, int facet, int tdim, int ndofs_cell, int bs,
                                   int* kernel_id)
{
  return register_integrand(name, source, rank, facet ? 1 : 0, tdim, ndofs_cell, bs, kernel_id);
}

int cfx_integrand_compile(int kernel_id, int tdim, int ndofs_cell)
{
  return cfx_integrand_compile_bs(kernel_id, tdim, ndofs_cell, 1);
}

int cfx_integrand_compile_bs(int kernel_id, int tdim, int ndofs_cell, int bs)
{
  CFX_API_BEGIN
  std::lock_guard<std::mutex> lock(rtc_mutex());
  require(user_integrand_known_locked(kernel_id), CFX_ERR_INVALID_ARGUMENT, "cfx_integrand_compile: unknown id");
  require((tdim == 2 || tdim == 3) && ndofs_cell >= tdim + 1 && ndofs_cell <= 10 && bs >= 1 && bs <= 3, CFX_ERR_INVALID_ARGUMENT,
          "cfx_integrand_compile: tdim 2 or 3, dofs per cell of a degree-1 or degree-2 space, block size 1..3");
  (void)compiled(integrands()[kernel_id - CFX_K_USER_BASE], tdim, ndofs_cell, bs);
  CFX_API_END
}

} // extern "C"
