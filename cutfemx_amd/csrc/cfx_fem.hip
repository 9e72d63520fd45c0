// cutfemx_amd: function spaces, forms, CSR sparsity, matrix/vector assembly and
// deactivation -- HIP kernels for gfx950 and their C ABI.
//
// Replaces (paths relative to the CutFEMx tree):
//   cpp/dolfinx_custom_data/fem/assembler.h:567-592 (+:442-560) sparsity      (a9)
//   cpp/dolfinx_custom_data/fem/assemble_matrix_impl.h:68-189   cell loop     (a5)
//   cpp/dolfinx_custom_data/fem/assemble_matrix_impl.h:409-607  facet loop    (a7)
//   cpp/dolfinx_custom_data/fem/assemble_vector_impl.h:62-122   vector loop   (a8)
//   runintgen/FFCx generated tabulate_tensor kernels (third party)            (a6)
//   cpp/cutfemx/fem/deactivate.h:387-418                                      (a11)
#include "cfx_device.h"

#define CFX_QUAD_TABLE_QUALIFIER static __device__ const
#include "cfx_quadrature_tables.h"
#undef CFX_QUAD_TABLE_QUALIFIER

using namespace cfx;

namespace cfx
{
int quad_npoints(int dim, int degree);
}

namespace
{

constexpr double kPi = 3.14159265358979323846;

template <int TDIM, int DEG>
struct Elem
{
  static constexpr int ND = DEG == 1 ? TDIM + 1 : (TDIM == 2 ? 6 : 10);
};

// Lagrange tabulation; dof order = Basix (vertices, then edges
// tri: (1,2),(0,2),(0,1); tet: (2,3),(1,3),(1,2),(0,3),(0,2),(0,1))
template <int TDIM, int DEG>
__device__ __forceinline__ void tabulate(const double* X, double* N, double (*dN)[TDIM])
{
  double lam[TDIM + 1];
  lam[0] = 1.0;
#pragma unroll
  for (int t = 0; t < TDIM; ++t) { lam[0] -= X[t]; lam[t + 1] = X[t]; }
  if constexpr (DEG == 1)
  {
#pragma unroll
    for (int i = 0; i <= TDIM; ++i)
    {
      N[i] = lam[i];
#pragma unroll
      for (int t = 0; t < TDIM; ++t) dN[i][t] = (i == 0) ? -1.0 : ((i - 1 == t) ? 1.0 : 0.0);
    }
  }
  else
  {
#pragma unroll
    for (int i = 0; i <= TDIM; ++i)
    {
      N[i] = lam[i] * (2.0 * lam[i] - 1.0);
#pragma unroll
      for (int t = 0; t < TDIM; ++t)
        dN[i][t] = (4.0 * lam[i] - 1.0) * ((i == 0) ? -1.0 : ((i - 1 == t) ? 1.0 : 0.0));
    }
    constexpr int NE = TDIM == 2 ? 3 : 6;
    constexpr int ea2[3] = {1, 0, 0}, eb2[3] = {2, 2, 1};
    constexpr int ea3[6] = {2, 1, 1, 0, 0, 0}, eb3[6] = {3, 3, 2, 3, 2, 1};
#pragma unroll
    for (int e = 0; e < NE; ++e)
    {
      const int a = TDIM == 2 ? ea2[e % 3] : ea3[e], b = TDIM == 2 ? eb2[e % 3] : eb3[e];
      N[TDIM + 1 + e] = 4.0 * lam[a] * lam[b];
#pragma unroll
      for (int t = 0; t < TDIM; ++t)
      {
        const double da = (a == 0) ? -1.0 : ((a - 1 == t) ? 1.0 : 0.0);
        const double db = (b == 0) ? -1.0 : ((b - 1 == t) ? 1.0 : 0.0);
        dN[TDIM + 1 + e][t] = 4.0 * (lam[a] * db + da * lam[b]);
      }
    }
  }
}

__device__ __forceinline__ const double* ref_rule(int dim, int degree, int& n, const double*& w)
{
  if (dim == 1)
  {
    n = cfx_quad_offset_1d[degree + 1] - cfx_quad_offset_1d[degree];
    w = cfx_quad_weights_1d + cfx_quad_offset_1d[degree];
    return cfx_quad_points_1d + cfx_quad_offset_1d[degree];
  }
  if (dim == 2)
  {
    n = cfx_quad_offset_2d[degree + 1] - cfx_quad_offset_2d[degree];
    w = cfx_quad_weights_2d + cfx_quad_offset_2d[degree];
    return cfx_quad_points_2d + 2 * cfx_quad_offset_2d[degree];
  }
  n = cfx_quad_offset_3d[degree + 1] - cfx_quad_offset_3d[degree];
  w = cfx_quad_weights_3d + cfx_quad_offset_3d[degree];
  return cfx_quad_points_3d + 3 * cfx_quad_offset_3d[degree];
}

template <int GDIM>
__device__ __forceinline__ double field_eval(int id, const double* x)
{
  if (id == CFX_F_ONE) return 1.0;
  double p = 1.0;
#pragma unroll
  for (int d = 0; d < GDIM; ++d) p *= sin(kPi * x[d]);
  if (id == CFX_F_SINPROD) return p;
  return (double)GDIM * kPi * kPi * p;
}

// arguments shared by the assembly kernels
struct AsmArgs
{
  const double* x;
  const int32_t* conn;
  const int32_t* dofmap;
  // entity description
  int64_t n;                 // entities in this launch
  const int32_t* entities;   // standard: cell ids / facet rows
  const int32_t* offsets;    // runtime rules
  const int32_t* parent_map;
  const double* points;
  const double* weights;
  const double* point_data;
  int point_stride;
  int kernel;
  int qdegree;
  double params[8];
  // sinks
  const int8_t* bc0;
  const int8_t* bc1;
  const int64_t* indptr;
  const int32_t* indices;
  double* values; // rank 2: CSR values; rank 1: vector
  double* dump;   // if set: write the local tensor here instead of scattering
  int* error;
};

// position of column `col` in CSR row [b,e); -1 if absent
__device__ __forceinline__ int64_t csr_find(const int32_t* __restrict__ indices, int64_t b, int64_t e, int32_t col)
{
  int64_t lo = b, hi = e;
  while (lo < hi)
  {
    const int64_t mid = (lo + hi) >> 1;
    if (indices[mid] < col) lo = mid + 1; else hi = mid;
  }
  return (lo < e && indices[lo] == col) ? lo : -1;
}

// ---------------------------------------------------------------------------
// a5/a6/a8 cell integrals.  One thread per (entity, local row): the thread
// forms row i of the element tensor in registers (geometry is recomputed per
// row, it is a handful of flops against the memory traffic) and adds it into
// its CSR row with one search per column and FP64 atomics.
// RUNTIME: the entity integrates over its runtime rule slice (weights are
// physical); otherwise the reference rule of degree qdegree times |detJ|.
// ---------------------------------------------------------------------------
template <int TDIM, int DEG, int BS, int RANK, bool RUNTIME>
__global__ void __launch_bounds__(kBlock) assemble_cells_kernel(AsmArgs A)
{
  constexpr int ND = Elem<TDIM, DEG>::ND;
  constexpr int NLOC = ND * BS;
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t e = tid / NLOC;
  if (e >= A.n) return;
  const int i = (int)(tid - e * NLOC);
  const int ia = i / BS, ik = i - ia * BS;
  const int64_t cell = RUNTIME ? A.parent_map[e] : A.entities[e];

  Geo<TDIM> g;
  load_cell<TDIM>(A.x, A.conn, cell, g);
  jacobian<TDIM>(g);
  const double h = cell_diameter<TDIM>(g);

  int npts;
  const double* pts;
  const double* wts;
  const double* pdata = nullptr;
  double wscale = 1.0;
  if constexpr (RUNTIME)
  {
    const int32_t q0 = A.offsets[e], q1 = A.offsets[e + 1];
    npts = q1 - q0;
    pts = A.points + (int64_t)q0 * TDIM;
    wts = A.weights + q0;
    if (A.point_data) pdata = A.point_data + (int64_t)q0 * A.point_stride;
  }
  else
  {
    pts = ref_rule(TDIM, A.qdegree, npts, wts);
    wscale = fabs(g.detJ);
  }

  double acc[RANK == 2 ? NLOC : 1];
#pragma unroll
  for (int j = 0; j < (RANK == 2 ? NLOC : 1); ++j) acc[j] = 0.0;

  for (int q = 0; q < npts; ++q)
  {
    double X[TDIM];
#pragma unroll
    for (int t = 0; t < TDIM; ++t) X[t] = pts[(int64_t)q * TDIM + t];
    const double w = wts[q] * wscale;
    double N[ND], dN[ND][TDIM], G[ND][TDIM];
    tabulate<TDIM, DEG>(X, N, dN);
#pragma unroll
    for (int j = 0; j < ND; ++j)
#pragma unroll
      for (int d = 0; d < TDIM; ++d)
      {
        double v = 0.0;
#pragma unroll
        for (int t = 0; t < TDIM; ++t) v += g.K[t][d] * dN[j][t];
        G[j][d] = v;
      }
    // row basis function (static indexing through a select chain)
    double Ni = 0.0, Gi[TDIM];
#pragma unroll
    for (int d = 0; d < TDIM; ++d) Gi[d] = 0.0;
#pragma unroll
    for (int j = 0; j < ND; ++j)
      if (j == ia)
      {
        Ni = N[j];
#pragma unroll
        for (int d = 0; d < TDIM; ++d) Gi[d] = G[j][d];
      }

    if constexpr (RANK == 2)
    {
      switch (A.kernel)
      {
      case CFX_K_MASS:
#pragma unroll
        for (int j = 0; j < ND; ++j) acc[j * BS + ik] += w * Ni * N[j];
        break;
      case CFX_K_STIFFNESS:
#pragma unroll
        for (int j = 0; j < ND; ++j)
        {
          double s = 0.0;
#pragma unroll
          for (int d = 0; d < TDIM; ++d) s += Gi[d] * G[j][d];
          acc[j * BS + ik] += w * s;
        }
        break;
      case CFX_K_NITSCHE:
        if constexpr (BS == 1)
        {
          const double* nrm = pdata + (int64_t)q * A.point_stride;
          const double gam = A.params[0] / h;
          double dni = 0.0;
#pragma unroll
          for (int d = 0; d < TDIM; ++d) dni += Gi[d] * nrm[d];
#pragma unroll
          for (int j = 0; j < ND; ++j)
          {
            double dnj = 0.0;
#pragma unroll
            for (int d = 0; d < TDIM; ++d) dnj += G[j][d] * nrm[d];
            acc[j] += w * (-dnj * Ni - dni * N[j] + gam * N[j] * Ni);
          }
        }
        break;
      case CFX_K_ELASTICITY:
        if constexpr (BS == TDIM)
        {
          const double E = A.params[0], nu = A.params[1];
          const double mu = E / (2.0 * (1.0 + nu));
          const double lmbda = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu));
          double Gia = 0.0;
#pragma unroll
          for (int d = 0; d < TDIM; ++d) Gia = (d == ik) ? Gi[d] : Gia;
#pragma unroll
          for (int j = 0; j < ND; ++j)
          {
            double gg = 0.0, Gja = 0.0;
#pragma unroll
            for (int d = 0; d < TDIM; ++d) { gg += Gi[d] * G[j][d]; Gja = (d == ik) ? G[j][d] : Gja; }
#pragma unroll
            for (int b = 0; b < BS; ++b)
              acc[j * BS + b] += w * (mu * ((b == ik ? gg : 0.0) + Gi[b] * Gja) + lmbda * Gia * G[j][b]);
          }
        }
        break;
      default: break;
      }
    }
    else
    {
      double xq[TDIM], l0 = 1.0;
#pragma unroll
      for (int t = 0; t < TDIM; ++t) l0 -= X[t];
#pragma unroll
      for (int d = 0; d < TDIM; ++d)
      {
        double v = l0 * g.x[0][d];
#pragma unroll
        for (int t = 0; t < TDIM; ++t) v += X[t] * g.x[t + 1][d];
        xq[d] = v;
      }
      if (A.kernel == CFX_L_SOURCE)
      {
        const double f = A.params[1] * field_eval<TDIM>((int)A.params[0], xq);
        acc[0] += w * f * Ni;
      }
      else if (A.kernel == CFX_L_NITSCHE_RHS)
      {
        const double* nrm = pdata + (int64_t)q * A.point_stride;
        const double gam = A.params[0] / h;
        const double gv = A.params[2] * field_eval<TDIM>((int)A.params[1], xq);
        double dni = 0.0;
#pragma unroll
        for (int d = 0; d < TDIM; ++d) dni += Gi[d] * nrm[d];
        acc[0] += w * (-dni * gv + gam * gv * Ni);
      }
    }
  }

  if (A.dump)
  {
    if constexpr (RANK == 2)
    {
#pragma unroll
      for (int j = 0; j < NLOC; ++j) A.dump[(e * NLOC + i) * NLOC + j] = acc[j];
    }
    else
      A.dump[e * NLOC + i] = acc[0];
    return;
  }

  const int32_t* cd = A.dofmap + cell * ND;
  const int32_t row = BS * cd[ia] + ik;
  if constexpr (RANK == 1)
  {
    atomicAdd(A.values + row, acc[0]);
  }
  else
  {
    // zero BC rows / columns: assemble_matrix_impl.h:151-185
    const bool row_bc = A.bc0 && A.bc0[row];
    const int64_t rb = A.indptr[row], re = A.indptr[row + 1];
#pragma unroll
    for (int j = 0; j < ND; ++j)
    {
      const int32_t col0 = BS * cd[j];
      const int64_t pos = csr_find(A.indices, rb, re, col0);
      if (pos < 0) { *A.error = 1; continue; }
#pragma unroll
      for (int b = 0; b < BS; ++b)
      {
        double v = acc[j * BS + b];
        if (row_bc || (A.bc1 && A.bc1[col0 + b])) v = 0.0;
        atomicAdd(A.values + pos + b, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// a7 interior-facet integrals (ghost penalty).  One thread per (facet, macro
// row).  Macro element = [cell0 dofs, cell1 dofs]; Ae block layout
// [[00,01],[10,11]] (assemble_matrix_impl.h:537-542).  The facet quadrature
// points are pushed to physical space from cell0's facet and pulled back to
// both reference cells.
// ---------------------------------------------------------------------------
template <int TDIM, int DEG, int BS>
__global__ void __launch_bounds__(kBlock) assemble_facets_kernel(AsmArgs A)
{
  constexpr int ND = Elem<TDIM, DEG>::ND;
  constexpr int NLOC = 2 * ND * BS;
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t f = tid / NLOC;
  if (f >= A.n) return;
  const int I = (int)(tid - f * NLOC);
  const int ia = I / BS, ik = I - ia * BS; // macro basis index, component
  const int4 row4 = *reinterpret_cast<const int4*>(A.entities + 4 * f);
  const int64_t c0 = row4.x, c1 = row4.z;
  const int lf0 = row4.y;

  Geo<TDIM> g0, g1;
  load_cell<TDIM>(A.x, A.conn, c0, g0);
  load_cell<TDIM>(A.x, A.conn, c1, g1);
  jacobian<TDIM>(g0);
  jacobian<TDIM>(g1);
  const double havg = 0.5 * (cell_diameter<TDIM>(g0) + cell_diameter<TDIM>(g1));

  // outward unit normal of cell0 on facet lf0: -grad(lambda_lf0)/|.|
  double nrm[TDIM];
  {
    double nn = 0.0;
#pragma unroll
    for (int d = 0; d < TDIM; ++d)
    {
      double v = 0.0;
#pragma unroll
      for (int t = 0; t < TDIM; ++t)
      {
        const double dl = (lf0 == 0) ? -1.0 : ((lf0 - 1 == t) ? 1.0 : 0.0);
        v -= g0.K[t][d] * dl;
      }
      nrm[d] = v;
      nn += v * v;
    }
    nn = sqrt(nn);
#pragma unroll
    for (int d = 0; d < TDIM; ++d) nrm[d] /= nn;
  }
  // facet vertices (cell0 vertices except lf0, ascending local index)
  double xf[TDIM][TDIM];
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i <= TDIM; ++i)
    {
      if (i == lf0) continue;
#pragma unroll
      for (int j = 0; j < TDIM; ++j)
        if (j == k)
        {
#pragma unroll
          for (int d = 0; d < TDIM; ++d) xf[j][d] = g0.x[i][d];
        }
      ++k;
    }
  }
  double scale;
  if constexpr (TDIM == 2)
  {
    const double dx = xf[1][0] - xf[0][0], dy = xf[1][1] - xf[0][1];
    scale = sqrt(dx * dx + dy * dy);
  }
  else
  {
    double a[3], b[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { a[d] = xf[1][d] - xf[0][d]; b[d] = xf[2][d] - xf[0][d]; }
    const double cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
    scale = sqrt(cx * cx + cy * cy + cz * cz);
  }

  double acc[NLOC];
#pragma unroll
  for (int j = 0; j < NLOC; ++j) acc[j] = 0.0;

  int nref;
  const double* wref;
  const double* pref = ref_rule(TDIM - 1, A.qdegree, nref, wref);
  for (int q = 0; q < nref; ++q)
  {
    double l0 = 1.0, xq[TDIM];
#pragma unroll
    for (int t = 0; t < TDIM - 1; ++t) l0 -= pref[q * (TDIM - 1) + t];
#pragma unroll
    for (int d = 0; d < TDIM; ++d)
    {
      double v = l0 * xf[0][d];
#pragma unroll
      for (int t = 0; t < TDIM - 1; ++t) v += pref[q * (TDIM - 1) + t] * xf[t + 1][d];
      xq[d] = v;
    }
    double X0[TDIM], X1[TDIM];
#pragma unroll
    for (int t = 0; t < TDIM; ++t)
    {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int d = 0; d < TDIM; ++d)
      {
        a += g0.K[t][d] * (xq[d] - g0.x[0][d]);
        b += g1.K[t][d] * (xq[d] - g1.x[0][d]);
      }
      X0[t] = a; X1[t] = b;
    }
    double N0[ND], dN0[ND][TDIM], N1[ND], dN1[ND][TDIM];
    tabulate<TDIM, DEG>(X0, N0, dN0);
    tabulate<TDIM, DEG>(X1, N1, dN1);
    const double w = wref[q] * scale * A.params[0] * havg;
    // normal-derivative jump of every macro basis function
    double jn[2 * ND];
#pragma unroll
    for (int j = 0; j < ND; ++j)
    {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int d = 0; d < TDIM; ++d)
#pragma unroll
        for (int t = 0; t < TDIM; ++t)
        {
          a += g0.K[t][d] * dN0[j][t] * nrm[d];
          b += g1.K[t][d] * dN1[j][t] * nrm[d];
        }
      jn[j] = a; jn[ND + j] = -b;
    }
    double ji = 0.0;
#pragma unroll
    for (int j = 0; j < 2 * ND; ++j) ji = (j == ia) ? jn[j] : ji;
    if (A.kernel == CFX_K_GHOST_GRADJUMP)
    {
#pragma unroll
      for (int j = 0; j < 2 * ND; ++j) acc[j * BS + ik] += w * ji * jn[j];
    }
  }

  if (A.dump)
  {
#pragma unroll
    for (int j = 0; j < NLOC; ++j) A.dump[(f * NLOC + I) * NLOC + j] = acc[j];
    return;
  }

  const int64_t crow = (ia < ND) ? c0 : c1;
  const int la = (ia < ND) ? ia : ia - ND;
  const int32_t row = BS * A.dofmap[crow * ND + la] + ik;
  const bool row_bc = A.bc0 && A.bc0[row];
  const int64_t rb = A.indptr[row], re = A.indptr[row + 1];
#pragma unroll
  for (int j = 0; j < 2 * ND; ++j)
  {
    const int64_t ccol = (j < ND) ? c0 : c1;
    const int lj = (j < ND) ? j : j - ND;
    const int32_t col0 = BS * A.dofmap[ccol * ND + lj];
    const int64_t pos = csr_find(A.indices, rb, re, col0);
    if (pos < 0) { *A.error = 1; continue; }
#pragma unroll
    for (int b = 0; b < BS; ++b)
    {
      double v = acc[j * BS + b];
      if (row_bc || (A.bc1 && A.bc1[col0 + b])) v = 0.0;
      atomicAdd(A.values + pos + b, v);
    }
  }
}

template <int TDIM, int DEG, int BS>
void launch_integral_t(const cfx_form_s* a, const cfx_integral_dev& I, AsmArgs A, int64_t only_index, int use_rule)
{
  constexpr int ND = Elem<TDIM, DEG>::ND;
  const bool single = only_index >= 0;
  if (I.type == CFX_INTERIOR_FACET)
  {
    require(a->rank == 2, CFX_ERR_INVALID_ARGUMENT, "interior-facet integrals are implemented for bilinear forms");
    A.n = single ? 1 : I.n_entities;
    A.entities = I.entities.p + (single ? 4 * only_index : 0);
    launch("assemble_facets", assemble_facets_kernel<TDIM, DEG, BS>, grid_for(A.n * 2 * ND * BS), dim3(kBlock), 0, A);
    return;
  }
  if (!single || !use_rule)
  {
    A.n = single ? 1 : I.n_entities;
    A.entities = I.entities.p + (single ? only_index : 0);
    if (A.n > 0)
    {
      if (a->rank == 2)
        launch("assemble_cells_std", assemble_cells_kernel<TDIM, DEG, BS, 2, false>, grid_for(A.n * ND * BS),
               dim3(kBlock), 0, A);
      else
        launch("assemble_vec_std", assemble_cells_kernel<TDIM, DEG, BS, 1, false>, grid_for(A.n * ND * BS),
               dim3(kBlock), 0, A);
    }
  }
  if (I.rules && (!single || use_rule))
  {
    const cfx_rules_s* R = I.rules;
    A.n = single ? 1 : R->nr;
    A.offsets = R->offsets.p + (single ? only_index : 0);
    A.parent_map = R->parent_map.p + (single ? only_index : 0);
    A.points = R->points.p;
    A.weights = R->weights.p;
    A.point_data = I.point_data.n > 0 ? I.point_data.p : nullptr;
    if (A.n > 0)
    {
      if (a->rank == 2)
        launch("assemble_cells_cut", assemble_cells_kernel<TDIM, DEG, BS, 2, true>, grid_for(A.n * ND * BS),
               dim3(kBlock), 0, A);
      else
        launch("assemble_vec_cut", assemble_cells_kernel<TDIM, DEG, BS, 1, true>, grid_for(A.n * ND * BS),
               dim3(kBlock), 0, A);
    }
  }
}

void launch_integral(const cfx_form_s* a, const cfx_integral_dev& I, AsmArgs A, int64_t only_index = -1,
                     int use_rule = 0)
{
  const cfx_space_s* V = a->V;
  const int tdim = V->mesh->tdim;
  A.kernel = I.kernel; A.qdegree = I.qdegree; A.point_stride = I.point_stride;
  for (int k = 0; k < 8; ++k) A.params[k] = I.params[k];
  const int key = tdim * 100 + V->degree * 10 + V->bs;
  switch (key)
  {
  case 211: launch_integral_t<2, 1, 1>(a, I, A, only_index, use_rule); break;
  case 221: launch_integral_t<2, 2, 1>(a, I, A, only_index, use_rule); break;
  case 212: launch_integral_t<2, 1, 2>(a, I, A, only_index, use_rule); break;
  case 222: launch_integral_t<2, 2, 2>(a, I, A, only_index, use_rule); break;
  case 311: launch_integral_t<3, 1, 1>(a, I, A, only_index, use_rule); break;
  case 321: launch_integral_t<3, 2, 1>(a, I, A, only_index, use_rule); break;
  case 313: launch_integral_t<3, 1, 3>(a, I, A, only_index, use_rule); break;
  case 323: launch_integral_t<3, 2, 3>(a, I, A, only_index, use_rule); break;
  default: throw Error(CFX_ERR_INVALID_ARGUMENT, "unsupported (tdim, degree, block size) combination");
  }
}

// ---------------------------------------------------------------------------
// a9 sparsity, row-centric.  A cell integral contributes, to row r, the dofs of
// every marked cell incident to r (static dof->cells incidence of the space);
// a facet integral contributes the dofs of both cells of every incident facet
// (dof->facets incidence rebuilt per pattern, the facet band is O(N^2)).
// Every row also holds its diagonal (assembler.h:538-560).
// Each thread keeps its row as a sorted list in LDS; two passes (count, fill).
// ---------------------------------------------------------------------------
__global__ void mark_cells_kernel(int64_t n, const int32_t* __restrict__ cells, int stride, uint8_t* mark)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) mark[cells[i * stride]] = 1;
}

__global__ void facet_dof_count_kernel(int64_t nf, const int32_t* __restrict__ rows, const int32_t* __restrict__ dofmap,
                                       int nd, int32_t* counts)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nf * 2 * nd) return;
  const int64_t f = i / (2 * nd);
  const int k = (int)(i - f * 2 * nd);
  const int64_t c = rows[4 * f + (k < nd ? 0 : 2)];
  atomicAdd(&counts[dofmap[c * nd + (k < nd ? k : k - nd)]], 1);
}

__global__ void facet_dof_fill_kernel(int64_t nf, const int32_t* __restrict__ rows, const int32_t* __restrict__ dofmap,
                                      int nd, const int64_t* __restrict__ offs, int32_t* cursor, int32_t* facets)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nf * 2 * nd) return;
  const int64_t f = i / (2 * nd);
  const int k = (int)(i - f * 2 * nd);
  const int64_t c = rows[4 * f + (k < nd ? 0 : 2)];
  const int32_t dof = dofmap[c * nd + (k < nd ? k : k - nd)];
  facets[offs[dof] + atomicAdd(&cursor[dof], 1)] = (int32_t)f;
}

template <int CAP>
__device__ __forceinline__ bool list_insert(int32_t* list, int& len, int32_t v)
{
  int lo = 0, hi = len;
  while (lo < hi)
  {
    const int mid = (lo + hi) >> 1;
    if (list[mid] < v) lo = mid + 1; else hi = mid;
  }
  if (lo < len && list[lo] == v) return true;
  if (len >= CAP) return false;
  for (int k = len; k > lo; --k) list[k] = list[k - 1];
  list[lo] = v;
  ++len;
  return true;
}

struct SparsityArgs
{
  int64_t ndofs;
  int nd, bs;
  const int32_t* dofmap;
  const int64_t* d2c_off;
  const int32_t* d2c;
  const uint8_t* cellmark; // or null
  int64_t nfacets;
  const int32_t* facet_rows;
  const int64_t* d2f_off;  // or null
  const int32_t* d2f;
  int32_t* counts;         // pass 0: [ndofs*bs] expanded row lengths
  const int64_t* indptr;   // pass 1
  int32_t* indices;
  int* overflow;
};

template <int THREADS, int CAP, bool FILL>
__global__ void __launch_bounds__(THREADS) sparsity_rows_kernel(SparsityArgs S)
{
  __shared__ int32_t s_list[THREADS][CAP + 1];
  const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  if (r >= S.ndofs) return;
  int32_t* list = s_list[threadIdx.x];
  int len = 1;
  list[0] = (int32_t)r; // diagonal of every row
  bool ok = true;
  if (S.cellmark)
    for (int64_t k = S.d2c_off[r]; k < S.d2c_off[r + 1]; ++k)
    {
      const int64_t c = S.d2c[k];
      if (!S.cellmark[c]) continue;
      for (int j = 0; j < S.nd; ++j) ok = list_insert<CAP>(list, len, S.dofmap[c * S.nd + j]) && ok;
    }
  if (S.d2f_off)
    for (int64_t k = S.d2f_off[r]; k < S.d2f_off[r + 1]; ++k)
    {
      const int64_t f = S.d2f[k];
      for (int s = 0; s < 2; ++s)
      {
        const int64_t c = S.facet_rows[4 * f + 2 * s];
        for (int j = 0; j < S.nd; ++j) ok = list_insert<CAP>(list, len, S.dofmap[c * S.nd + j]) && ok;
      }
    }
  if (!ok) { *S.overflow = 1; return; }
  if constexpr (!FILL)
  {
    for (int a = 0; a < S.bs; ++a) S.counts[r * S.bs + a] = len * S.bs;
  }
  else
  {
    for (int a = 0; a < S.bs; ++a)
    {
      int64_t o = S.indptr[r * S.bs + a];
      for (int k = 0; k < len; ++k)
        for (int b = 0; b < S.bs; ++b) S.indices[o++] = list[k] * S.bs + b;
    }
  }
}

// ---------------------------------------------------------------------------
// a11 deactivation helpers
// ---------------------------------------------------------------------------
__global__ void mark_dofs_kernel(int64_t ncells_active, const int32_t* __restrict__ cells,
                                 const int32_t* __restrict__ dofmap, int nd, int bs, uint8_t* indicator)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncells_active * nd) return;
  const int64_t c = cells[i / nd];
  const int32_t dof = dofmap[c * nd + (int)(i % nd)];
  for (int k = 0; k < bs; ++k) indicator[(int64_t)dof * bs + k] = 1;
}

struct FlagSet
{
  const uint8_t* f;
  __device__ bool operator()(int64_t i) const { return f[i] != 0; }
};
struct FlagClear
{
  const uint8_t* f;
  __device__ bool operator()(int64_t i) const { return f[i] == 0; }
};

__global__ void deactivate_kernel(int64_t n, const int32_t* __restrict__ rows, const int64_t* __restrict__ indptr,
                                  const int32_t* __restrict__ indices, double* values, double* b, double diagonal,
                                  double rhs_value, int* error)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t r = rows[i];
  if (values)
  {
    const int64_t pos = csr_find(indices, indptr[r], indptr[r + 1], r);
    if (pos < 0) *error = 1; else values[pos] = diagonal; // set, not add (set_diagonal via mat_set_values)
  }
  if (b) b[r] = rhs_value;
}

void collect_cell_marks(const cfx_form_s* a, bool include_facets, DevArray<uint8_t>& mark, bool& any)
{
  const int64_t nc = a->V->mesh->ncells;
  mark.alloc(nc);
  mark.zero();
  any = false;
  for (const auto& I : a->integrals)
  {
    if (I.type == CFX_CELL)
    {
      if (I.n_entities > 0)
      {
        launch("mark_cells", mark_cells_kernel, grid_for(I.n_entities), dim3(kBlock), 0, I.n_entities, I.entities.p, 1,
               mark.p);
        any = true;
      }
      if (I.rules && I.rules->nr > 0)
      {
        launch("mark_cells", mark_cells_kernel, grid_for(I.rules->nr), dim3(kBlock), 0, I.rules->nr,
               I.rules->parent_map.p, 1, mark.p);
        any = true;
      }
    }
    else if (include_facets && I.type == CFX_INTERIOR_FACET && I.n_entities > 0)
    {
      launch("mark_cells", mark_cells_kernel, grid_for(I.n_entities), dim3(kBlock), 0, I.n_entities, I.entities.p, 4,
             mark.p);
      launch("mark_cells", mark_cells_kernel, grid_for(I.n_entities), dim3(kBlock), 0, I.n_entities,
             I.entities.p + 2, 4, mark.p);
      any = true;
    }
  }
}

} // namespace

extern "C" {

int cfx_space_create(cfx_mesh_t mesh, int degree, int bs, int64_t ndofs, const int32_t* dofmap, int ndofs_cell,
                     cfx_space_t* out)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(mesh && dofmap && out, CFX_ERR_INVALID_ARGUMENT, "cfx_space_create: null argument");
  require(degree == 1 || degree == 2, CFX_ERR_INVALID_ARGUMENT, "cfx_space_create: Lagrange degree must be 1 or 2");
  const int expect = degree == 1 ? mesh->tdim + 1 : (mesh->tdim == 2 ? 6 : 10);
  require(ndofs_cell == expect, CFX_ERR_INVALID_ARGUMENT, "cfx_space_create: dofs per cell do not match the element");
  require(bs == 1 || bs == mesh->gdim, CFX_ERR_INVALID_ARGUMENT, "cfx_space_create: block size must be 1 or gdim");
  require(ndofs > 0 && ndofs * bs < 2147483647LL, CFX_ERR_INVALID_ARGUMENT, "cfx_space_create: dof count must fit int32");
  auto V = std::make_unique<cfx_space_s>();
  V->mesh = mesh; V->degree = degree; V->bs = bs; V->ndofs = ndofs; V->ndofs_cell = ndofs_cell;
  V->dofmap = to_device(dofmap, mesh->ncells * (int64_t)ndofs_cell);
  *out = V.release();
  CFX_API_END
}

int cfx_space_destroy(cfx_space_t V)
{
  CFX_API_BEGIN
  delete V;
  CFX_API_END
}

int cfx_form_create(cfx_space_t V, int rank, int n_integrals, const cfx_integral* integrals, cfx_form_t* out)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(V && out && (integrals || n_integrals == 0), CFX_ERR_INVALID_ARGUMENT, "cfx_form_create: null argument");
  require(rank == 1 || rank == 2, CFX_ERR_INVALID_ARGUMENT, "cfx_form_create: rank must be 1 or 2");
  auto a = std::make_unique<cfx_form_s>();
  a->V = V; a->rank = rank;
  for (int i = 0; i < n_integrals; ++i)
  {
    const cfx_integral& in = integrals[i];
    cfx_integral_dev I;
    I.type = in.type; I.kernel = in.kernel; I.qdegree = in.qdegree; I.point_stride = in.point_stride;
    require(in.type == CFX_CELL || in.type == CFX_INTERIOR_FACET, CFX_ERR_INVALID_ARGUMENT,
            "cfx_form_create: integral type must be cell or interior_facet");
    const bool bilinear = in.kernel < 100;
    require(bilinear == (rank == 2), CFX_ERR_INVALID_ARGUMENT, "cfx_form_create: kernel rank does not match the form");
    require(in.qdegree >= 0 && in.qdegree <= CFX_QUAD_MAX_DEGREE, CFX_ERR_INVALID_ARGUMENT,
            "cfx_form_create: quadrature degree out of range");
    if (in.type == CFX_INTERIOR_FACET)
      require(in.kernel == CFX_K_GHOST_GRADJUMP && in.rules == nullptr, CFX_ERR_INVALID_ARGUMENT,
              "cfx_form_create: interior-facet integrals support the ghost-penalty kernel with standard quadrature");
    else
      require(in.kernel == CFX_K_MASS || in.kernel == CFX_K_STIFFNESS || in.kernel == CFX_K_NITSCHE
                  || in.kernel == CFX_K_ELASTICITY || in.kernel == CFX_L_SOURCE || in.kernel == CFX_L_NITSCHE_RHS,
              CFX_ERR_INVALID_ARGUMENT, "cfx_form_create: unknown cell kernel id");
    if (in.kernel == CFX_K_NITSCHE || in.kernel == CFX_L_NITSCHE_RHS)
    {
      require(V->bs == 1, CFX_ERR_INVALID_ARGUMENT, "Nitsche kernels are scalar");
      require(in.n_entities == 0 && in.rules && in.point_data && in.point_stride >= V->mesh->gdim,
              CFX_ERR_INVALID_ARGUMENT,
              "Nitsche kernels need runtime interface rules and per-point normals (point_data)");
    }
    if (in.kernel == CFX_K_ELASTICITY)
      require(V->bs == V->mesh->gdim, CFX_ERR_INVALID_ARGUMENT, "elasticity needs a vector space (bs == gdim)");
    if (in.kernel == CFX_L_SOURCE) require(V->bs == 1, CFX_ERR_INVALID_ARGUMENT, "source kernel is scalar");
    I.n_entities = in.n_entities;
    const int64_t width = in.type == CFX_INTERIOR_FACET ? 4 : 1;
    I.entities = to_device(in.entities, in.n_entities * width);
    I.rules = in.rules;
    if (in.rules)
    {
      require(in.rules->mesh == V->mesh, CFX_ERR_INVALID_ARGUMENT, "rules belong to a different mesh");
      if (in.point_data) I.point_data = to_device(in.point_data, in.rules->nq * (int64_t)in.point_stride);
    }
    for (int k = 0; k < 8; ++k) I.params[k] = in.params[k];
    a->integrals.push_back(std::move(I));
  }
  *out = a.release();
  CFX_API_END
}

int cfx_form_destroy(cfx_form_t a)
{
  CFX_API_BEGIN
  delete a;
  CFX_API_END
}

int cfx_create_sparsity(cfx_form_t a, cfx_pattern_t* out)
{
  CFX_API_BEGIN
  require(a && out, CFX_ERR_INVALID_ARGUMENT, "cfx_create_sparsity: null argument");
  // assembler.h:570-574
  require(a->rank == 2, CFX_ERR_RUNTIME, "Cannot create sparsity pattern. Form is not a bilinear.");
  cfx_space_s* V = a->V;
  const int nd = V->ndofs_cell;
  SparsityArgs S{};
  S.ndofs = V->ndofs; S.nd = nd; S.bs = V->bs; S.dofmap = V->dofmap.p;

  DevArray<uint8_t> mark;
  bool any_cells = false;
  collect_cell_marks(a, false, mark, any_cells);
  if (any_cells)
  {
    const Adjacency& adj = V->dof_cells();
    S.d2c_off = adj.offsets.p; S.d2c = adj.cells.p; S.cellmark = mark.p;
  }
  // concatenate the facet rows of all interior-facet integrals
  int64_t nf = 0;
  for (const auto& I : a->integrals)
    if (I.type == CFX_INTERIOR_FACET) nf += I.n_entities;
  DevArray<int32_t> frows, fcount, d2f;
  DevArray<int64_t> d2f_off;
  if (nf > 0)
  {
    frows.alloc(nf * 4);
    int64_t o = 0;
    for (const auto& I : a->integrals)
      if (I.type == CFX_INTERIOR_FACET && I.n_entities > 0)
      {
        CFX_HIP(hipMemcpyAsync(frows.p + 4 * o, I.entities.p, sizeof(int32_t) * 4 * (size_t)I.n_entities,
                               hipMemcpyDeviceToDevice, ctx().stream));
        o += I.n_entities;
      }
    fcount.alloc(V->ndofs);
    fcount.zero();
    launch("facet_dof_count", facet_dof_count_kernel, grid_for(nf * 2 * nd), dim3(kBlock), 0, nf, frows.p,
           V->dofmap.p, nd, fcount.p);
    d2f_off.alloc(V->ndofs + 1);
    exclusive_scan(fcount.p, d2f_off.p, V->ndofs);
    d2f.alloc(nf * 2 * nd);
    fcount.zero();
    launch("facet_dof_fill", facet_dof_fill_kernel, grid_for(nf * 2 * nd), dim3(kBlock), 0, nf, frows.p, V->dofmap.p,
           nd, d2f_off.p, fcount.p, d2f.p);
    S.nfacets = nf; S.facet_rows = frows.p; S.d2f_off = d2f_off.p; S.d2f = d2f.p;
  }

  auto P = std::make_unique<cfx_pattern_s>();
  P->nrows = V->ndofs * V->bs;
  DevArray<int32_t> counts(P->nrows);
  DevArray<int> overflow(1);
  overflow.zero();
  S.counts = counts.p; S.overflow = overflow.p;
  bool big = false;
  launch("sparsity_count", sparsity_rows_kernel<128, 96, false>, grid_for(V->ndofs, 128), dim3(128), 0, S);
  if (read_scalar(overflow.p))
  {
    big = true;
    overflow.zero();
    launch("sparsity_count_big", sparsity_rows_kernel<64, 480, false>, grid_for(V->ndofs, 64), dim3(64), 0, S);
    require(!read_scalar(overflow.p), CFX_ERR_RUNTIME, "sparsity: a row couples more than 480 dofs");
  }
  P->indptr.alloc(P->nrows + 1);
  exclusive_scan(counts.p, P->indptr.p, P->nrows);
  P->nnz = read_scalar(P->indptr.p + P->nrows);
  P->indices.alloc(P->nnz);
  S.indptr = P->indptr.p; S.indices = P->indices.p;
  if (!big)
    launch("sparsity_fill", sparsity_rows_kernel<128, 96, true>, grid_for(V->ndofs, 128), dim3(128), 0, S);
  else
    launch("sparsity_fill_big", sparsity_rows_kernel<64, 480, true>, grid_for(V->ndofs, 64), dim3(64), 0, S);
  CFX_HIP(hipStreamSynchronize(ctx().stream));
  *out = P.release();
  CFX_API_END
}

int cfx_pattern_view_get(cfx_pattern_t p, cfx_pattern_view* v)
{
  CFX_API_BEGIN
  require(p && v, CFX_ERR_INVALID_ARGUMENT, "cfx_pattern_view_get: null argument");
  v->nrows = p->nrows; v->nnz = p->nnz; v->indptr = p->indptr.p; v->indices = p->indices.p;
  CFX_API_END
}

int cfx_pattern_destroy(cfx_pattern_t p)
{
  CFX_API_BEGIN
  delete p;
  CFX_API_END
}

int cfx_assemble_matrix(cfx_form_t a, cfx_pattern_t P, const int8_t* bc0, const int8_t* bc1, double* values)
{
  CFX_API_BEGIN
  require(a && P && values, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_matrix: null argument");
  require(a->rank == 2, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_matrix: form is not bilinear");
  cfx_space_s* V = a->V;
  require(P->nrows == V->ndofs * V->bs, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_matrix: pattern/space size mismatch");
  DevArray<int8_t> dbc0 = to_device(bc0, bc0 ? P->nrows : 0), dbc1 = to_device(bc1, bc1 ? P->nrows : 0);
  OutArray<double> out(values, P->nnz, true);
  DevArray<int> err(1);
  err.zero();
  AsmArgs A{};
  A.x = V->mesh->x.p; A.conn = V->mesh->conn.p; A.dofmap = V->dofmap.p;
  A.bc0 = bc0 ? dbc0.p : nullptr; A.bc1 = bc1 ? dbc1.p : nullptr;
  A.indptr = P->indptr.p; A.indices = P->indices.p; A.values = out.dev; A.dump = nullptr; A.error = err.p;
  for (const auto& I : a->integrals) launch_integral(a, I, A);
  require(!read_scalar(err.p), CFX_ERR_RUNTIME, "assemble_matrix: entry not in the sparsity pattern");
  out.finish();
  CFX_API_END
}

int cfx_assemble_vector(cfx_form_t L, double* b)
{
  CFX_API_BEGIN
  require(L && b, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_vector: null argument");
  require(L->rank == 1, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_vector: form is not linear");
  cfx_space_s* V = L->V;
  OutArray<double> out(b, V->ndofs * V->bs, true);
  DevArray<int> err(1);
  err.zero();
  AsmArgs A{};
  A.x = V->mesh->x.p; A.conn = V->mesh->conn.p; A.dofmap = V->dofmap.p;
  A.values = out.dev; A.error = err.p;
  for (const auto& I : L->integrals) launch_integral(L, I, A);
  out.finish();
  if (out.dev == b) { /* device output: leave the stream running */ }
  CFX_API_END
}

int cfx_tabulate_entity(cfx_form_t a, int integral, int64_t index, int use_rule, double* Ae)
{
  CFX_API_BEGIN
  require(a && Ae, CFX_ERR_INVALID_ARGUMENT, "cfx_tabulate_entity: null argument");
  require(integral >= 0 && integral < (int)a->integrals.size(), CFX_ERR_OUT_OF_RANGE, "integral index out of range");
  const cfx_integral_dev& I = a->integrals[integral];
  cfx_space_s* V = a->V;
  const int64_t limit = (I.type == CFX_CELL && use_rule) ? (I.rules ? I.rules->nr : 0) : I.n_entities;
  require(index >= 0 && index < limit, CFX_ERR_OUT_OF_RANGE, "entity index out of range");
  const int nloc = V->ndofs_cell * V->bs * (I.type == CFX_INTERIOR_FACET ? 2 : 1);
  const int64_t n = a->rank == 2 ? (int64_t)nloc * nloc : nloc;
  OutArray<double> out(Ae, n, false);
  DevArray<int> err(1);
  err.zero();
  AsmArgs A{};
  A.x = V->mesh->x.p; A.conn = V->mesh->conn.p; A.dofmap = V->dofmap.p;
  A.dump = out.dev; A.error = err.p;
  launch_integral(a, I, A, index, use_rule);
  out.finish();
  CFX_API_END
}

int cfx_active_domain(cfx_form_t a, cfx_active_t* out)
{
  CFX_API_BEGIN
  require(a && out, CFX_ERR_INVALID_ARGUMENT, "cfx_active_domain: null argument");
  // deactivate.h:80-85
  require(a->rank == 2, CFX_ERR_INVALID_ARGUMENT, "cutfemx.fem.active_domain requires a rank-2 bilinear CutForm");
  cfx_space_s* V = a->V;
  DevArray<uint8_t> mark;
  bool any = false;
  collect_cell_marks(a, true, mark, any);
  auto d = std::make_unique<cfx_active_s>();
  d->V = V;
  d->n_active = compact("active_cells", V->mesh->ncells, FlagSet{mark.p}, d->active_cells);
  // deactivate.h:155-160
  require(d->n_active > 0, CFX_ERR_INVALID_ARGUMENT, "cutfemx.fem.active_domain found no active background cells");
  const int64_t nrows = V->ndofs * V->bs;
  DevArray<uint8_t> ind(nrows);
  ind.zero();
  launch("mark_dofs", mark_dofs_kernel, grid_for(d->n_active * V->ndofs_cell), dim3(kBlock), 0, d->n_active,
         d->active_cells.p, V->dofmap.p, V->ndofs_cell, V->bs, ind.p);
  d->n_inactive = compact("inactive_dofs", nrows, FlagClear{ind.p}, d->inactive_dofs);
  CFX_HIP(hipStreamSynchronize(ctx().stream));
  *out = d.release();
  CFX_API_END
}

int cfx_active_view(cfx_active_t d, const int32_t** active_cells, int64_t* n_active, const int32_t** inactive_dofs,
                    int64_t* n_inactive)
{
  CFX_API_BEGIN
  require(d != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_active_view: null handle");
  if (active_cells) *active_cells = d->active_cells.p;
  if (n_active) *n_active = d->n_active;
  if (inactive_dofs) *inactive_dofs = d->inactive_dofs.p;
  if (n_inactive) *n_inactive = d->n_inactive;
  CFX_API_END
}

int cfx_deactivate_outside(cfx_active_t d, cfx_pattern_t P, double* values, double* b, double diagonal,
                           double rhs_value)
{
  CFX_API_BEGIN
  require(d && (values == nullptr || P), CFX_ERR_INVALID_ARGUMENT, "cfx_deactivate_outside: null argument");
  const int64_t nrows = d->V->ndofs * d->V->bs;
  std::unique_ptr<OutArray<double>> ov, ob;
  if (values) ov = std::make_unique<OutArray<double>>(values, P->nnz, true);
  if (b) ob = std::make_unique<OutArray<double>>(b, nrows, true);
  DevArray<int> err(1);
  err.zero();
  if (d->n_inactive > 0)
    launch("deactivate", deactivate_kernel, grid_for(d->n_inactive), dim3(kBlock), 0, d->n_inactive,
           d->inactive_dofs.p, P ? P->indptr.p : nullptr, P ? P->indices.p : nullptr, values ? ov->dev : nullptr,
           b ? ob->dev : nullptr, diagonal, rhs_value, err.p);
  // deactivate.h: validate_matrix_rows
  require(!read_scalar(err.p), CFX_ERR_RUNTIME, "Deactivated matrix row has no diagonal entry.");
  if (ov) ov->finish();
  if (ob) ob->finish();
  CFX_API_END
}

int cfx_active_destroy(cfx_active_t d)
{
  CFX_API_BEGIN
  delete d;
  CFX_API_END
}

} // extern "C"
