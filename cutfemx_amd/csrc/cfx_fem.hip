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
#include <cstdlib>

#include "cfx_elem.h"

using namespace cfx;

namespace
{

// arguments shared by the assembly kernels
struct AsmArgs
{
  const double* x;
  const int32_t* conn;
  const int32_t* dofmap;
  // entity description
  DevN n;                    // entities in this launch (length in HBM inside a sync-free step)
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
  // lifting mode (lift_bc_impl, assemble_vector_impl.h:383-436): if set, `values` is the vector b and
  // the row of Ae is contracted with alpha (g - x0) over the Dirichlet columns instead of scattered
  const int8_t* lift_markers;
  const double* lift_values;
  const double* lift_x0;
  double lift_alpha;
  const double* coeff; // dof values of a CFX_F_COEFFICIENT field (rank-1 source terms)
  // interior-facet integrals with facet-hosted rules (8f-4): entity f of the integral with f >= n_std integrates
  // over rule f - n_std (offsets / points / weights above, host vertices below); f0 = first entity of this launch
  int64_t n_std, f0, dump0; // dump0: entity written to slot 0 of `dump`
  const int32_t* host_verts;
  // P1 facet tensors for the row gather: store the macro tensor folded over the dofs the two cells share --
  // (ND + 1)^2 doubles per facet instead of (2 ND)^2 (plan.fold_ok: every facet joins two cells that share all dofs
  // but one each).  Macro dofs: cell 0's, then the dof of cell 1 that cell 0 does not have.
  int dump_fold;
};

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
  if (e >= dev_n(A.n)) return;
  const int i = (int)(tid - e * NLOC);
  const int ia = i / BS, ik = i - ia * BS;
  const int64_t cell = RUNTIME ? A.parent_map[e] : A.entities[e];
  if constexpr (RANK == 2)
  {
    // LiftingMode: only entities with a Dirichlet column are tabulated
    if (A.lift_markers)
    {
      bool any = false;
      for (int j = 0; j < ND; ++j)
        for (int b = 0; b < BS; ++b) any = any || A.lift_markers[BS * A.dofmap[cell * ND + j] + b] != 0;
      if (!any) return;
    }
  }

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

  double cw[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j) cw[j] = 0.0;
  if (A.coeff) // pack_coefficients (pack_form.h:32-170): the cell's coefficient dofs
  {
    // rank 1 on a vector space: component ik of the vector-valued source; rank 2: a scalar coefficient
    const int cbs = RANK == 1 ? BS : 1, comp = RANK == 1 ? ik : 0;
#pragma unroll
    for (int j = 0; j < ND; ++j) cw[j] = A.coeff[(int64_t)A.dofmap[cell * ND + j] * cbs + comp];
  }
  cell_local_row<TDIM, DEG, BS, RANK>(A.kernel, A.params, A.point_stride, g, h, npts, pts, wts, wscale, pdata, ia, ik, acc,
                                      A.coeff ? cw : nullptr);

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
    if (A.lift_markers)
    {
      double s = 0.0;
      bool any = false;
#pragma unroll
      for (int j = 0; j < ND; ++j)
#pragma unroll
        for (int b = 0; b < BS; ++b)
        {
          const int32_t col = BS * cd[j] + b;
          if (A.lift_markers[col])
          {
            s += acc[j * BS + b] * A.lift_alpha * (A.lift_values[col] - (A.lift_x0 ? A.lift_x0[col] : 0.0));
            any = true;
          }
        }
      if (any) atomicAdd(A.values + row, -s);
      return;
    }
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
// RT: the launch covers facet-hosted runtime rules (8f-4): entity f0 + f of the integral integrates over rule
// f0 + f - n_std; the standard launch is compiled without that code
template <int TDIM, int DEG, int BS, bool RT, int KC>
__global__ void __launch_bounds__(kBlock) assemble_facets_kernel(AsmArgs A)
{
  constexpr int ND = Elem<TDIM, DEG>::ND;
  constexpr int NLOC = 2 * ND * BS;
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t fl = tid / NLOC;
  if (fl >= dev_n(A.n)) return;
  const int I = (int)(tid - fl * NLOC);
  const int ia = I / BS, ik = I - ia * BS; // macro basis index, component
  const int64_t f = fl + A.f0;            // entity of the integral; A.entities / point_data / dump are indexed by it
  const int4 row4 = *reinterpret_cast<const int4*>(A.entities + 4 * f);
  const int64_t c0 = row4.x, c1 = row4.z;
  const int lf0 = row4.y;
  if (A.lift_markers)
  {
    bool any = false;
    for (int j = 0; j < ND; ++j)
      for (int b = 0; b < BS; ++b)
        any = any || A.lift_markers[BS * A.dofmap[c0 * ND + j] + b] != 0 || A.lift_markers[BS * A.dofmap[c1 * ND + j] + b] != 0;
    if (!any) return;
  }

  Geo<TDIM> g0, g1;
  load_cell<TDIM>(A.x, A.conn, c0, g0);
  load_cell<TDIM>(A.x, A.conn, c1, g1);
  jacobian<TDIM>(g0);
  jacobian<TDIM>(g1);
  double acc[NLOC];
#pragma unroll
  for (int j = 0; j < NLOC; ++j) acc[j] = 0.0;
  if constexpr (RT)
  {
    const int64_t r = f - A.n_std;
    const int32_t q0 = A.offsets[r], q1 = A.offsets[r + 1];
    double xhost[TDIM][TDIM];
#pragma unroll
    for (int j = 0; j < TDIM; ++j)
    {
      const int64_t v = A.host_verts[r * TDIM + j];
#pragma unroll
      for (int d = 0; d < TDIM; ++d) xhost[j][d] = A.x[3 * v + d];
    }
    facet_local_row<TDIM, DEG, BS, KC>(A.kernel, A.params, A.qdegree, g0, g1, lf0, ia, ik, acc, q1 - q0,
                                   A.points + (int64_t)q0 * (TDIM - 1), A.weights + q0, xhost);
  }
  else
    facet_local_row<TDIM, DEG, BS, KC>(A.kernel, A.params, A.qdegree, g0, g1, lf0, ia, ik, acc);
  if (A.kernel == CFX_K_EXTENSION_L2 && A.point_data)
  {
    const double factor = A.point_data[f]; // cellwise beta of the pair's bad cell
#pragma unroll
    for (int j = 0; j < NLOC; ++j) acc[j] *= factor;
  }

  if constexpr (DEG == 1 && BS == 1 && TDIM == 3)
  {
    if (A.dump && A.dump_fold)
    {
      constexpr int WF = ND + 1;
      constexpr int kWave = 64;
      static_assert(kWave % NLOC == 0, "the 8 rows of a facet sit in 8 consecutive lanes of one wavefront");
      int32_t d0[ND], d1[ND];
#pragma unroll
      for (int j = 0; j < ND; ++j) { d0[j] = A.dofmap[c0 * ND + j]; d1[j] = A.dofmap[c1 * ND + j]; }
      int p1[ND]; // local index in cell 0 of cell 1's j-th dof, -1 for the one cell 0 does not have
      int q0 = -1; // this thread's partner: the local index in cell 1 of cell 0's dof I (threads I < ND)
      int jfree = 0;
#pragma unroll
      for (int j = 0; j < ND; ++j)
      {
        p1[j] = -1;
#pragma unroll
        for (int i = 0; i < ND; ++i) p1[j] = d1[j] == d0[i] ? i : p1[j];
        q0 = p1[j] == I ? j : q0;
        jfree = p1[j] < 0 ? j : jfree;
      }
      // columns: cell 1's shared dofs onto cell 0's, the free one last
      double accf[WF];
#pragma unroll
      for (int i = 0; i < ND; ++i)
      {
        accf[i] = acc[i];
#pragma unroll
        for (int j = 0; j < ND; ++j) accf[i] += p1[j] == i ? acc[ND + j] : 0.0;
      }
      accf[ND] = 0.0;
#pragma unroll
      for (int j = 0; j < ND; ++j) accf[ND] += p1[j] < 0 ? acc[ND + j] : 0.0;
      // rows: thread I < ND adds the row of its partner in cell 1 (lane + ND - I + q0); the thread of cell 1's free
      // dof holds macro row ND as it is; the other threads of cell 1 have handed their rows over
      const int lane = threadIdx.x % kWave;
      const int src = (lane - I) + ND + (q0 >= 0 ? q0 : 0);
      const bool add = I < ND && q0 >= 0;
#pragma unroll
      for (int k = 0; k < WF; ++k)
      {
        const double other = __shfl(accf[k], src, kWave);
        accf[k] += add ? other : 0.0;
      }
      const int m = I < ND ? I : (I - ND == jfree ? ND : -1);
      if (m >= 0)
      {
        double* out = A.dump + ((f - A.dump0) * WF + m) * WF;
#pragma unroll
        for (int k = 0; k < WF; ++k) out[k] = accf[k];
      }
      return;
    }
  }
  if (A.dump)
  {
    if constexpr (NLOC <= 16)
    {
      // a thread holds one row (NLOC doubles) of the facet tensor; the block's rows are contiguous in `dump`:
      // through LDS the stores are coalesced (lane i writes double i of a 256-double line, not its own 64 B run)
      __shared__ double s_row[kBlock * NLOC];
#pragma unroll
      for (int j = 0; j < NLOC; ++j) s_row[threadIdx.x * NLOC + j] = acc[j];
      __syncthreads(); // (threads past the end of the launch have left: the barrier counts the live waves)
      const int64_t first = ((int64_t)blockIdx.x * kBlock + (A.f0 - A.dump0) * NLOC) * NLOC;
      const int nthreads = (int)min((int64_t)kBlock, dev_n(A.n) * NLOC - (int64_t)blockIdx.x * kBlock); // threads 0..nthreads-1 are here
      for (int i = threadIdx.x; i < nthreads * NLOC; i += nthreads) A.dump[first + i] = s_row[i];
    }
    else
    {
#pragma unroll
      for (int j = 0; j < NLOC; ++j) A.dump[((f - A.dump0) * NLOC + I) * NLOC + j] = acc[j];
    }
    return;
  }

  const int64_t crow = (ia < ND) ? c0 : c1;
  const int la = (ia < ND) ? ia : ia - ND;
  const int32_t row = BS * A.dofmap[crow * ND + la] + ik;
  if (A.lift_markers)
  {
    double s = 0.0;
    bool any = false;
#pragma unroll
    for (int j = 0; j < 2 * ND; ++j)
    {
      const int64_t ccol = (j < ND) ? c0 : c1;
      const int lj = (j < ND) ? j : j - ND;
#pragma unroll
      for (int b = 0; b < BS; ++b)
      {
        const int32_t col = BS * A.dofmap[ccol * ND + lj] + b;
        if (A.lift_markers[col])
        {
          s += acc[j * BS + b] * A.lift_alpha * (A.lift_values[col] - (A.lift_x0 ? A.lift_x0[col] : 0.0));
          any = true;
        }
      }
    }
    if (any) atomicAdd(A.values + row, -s);
    return;
  }
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

// ---------------------------------------------------------------------------
// stage-1 tensors of the cut cells of a P1 scalar space: one thread per runtime rule forms the whole
// (tdim+1)^2 tensor.  P1 gradients are constant on the cell, so the stiffness part is
// (sum of the rule's weights) * G_i . G_j and only Nitsche / mass terms walk the points -- the generic
// kernel spends one thread per (rule, row) and re-tabulates at every point.
// ---------------------------------------------------------------------------
#ifndef CFX_CUT_LANES
#define CFX_CUT_LANES 4 // 512^3: 16 -> 830 us per launch, 8 -> 500, 4 -> 385, 2 -> 380
#endif
constexpr int kCutLanes = CFX_CUT_LANES; // lanes per rule: the points are dealt round-robin (coalesced, balanced), partial tensors folded by shuffles
#ifndef CFX_CUTP1_WAVES
#define CFX_CUTP1_WAVES 4 // waves per SIMD (126 registers in 3-D without a bound: 4)
#endif
template <int TDIM>
__global__ void __launch_bounds__(kBlock, CFX_CUTP1_WAVES) cut_tensors_p1_kernel(AsmArgs A)
{
  constexpr int ND = TDIM + 1;
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t e = tid / kCutLanes;
  const int sub = (int)(tid - e * kCutLanes);
  if (e >= dev_n(A.n)) return;
  const int64_t cell = A.parent_map[e];
  Geo<TDIM> g;
  load_cell<TDIM>(A.x, A.conn, cell, g);
  jacobian<TDIM>(g);
  // physical gradients of the barycentric basis: G_j = K^T dN_j, dN_0 = -1, dN_t = e_t
  double G[ND][TDIM];
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < TDIM; ++t) { G[t + 1][d] = g.K[t][d]; s += g.K[t][d]; }
    G[0][d] = -s;
  }
  double T[ND][ND];
#pragma unroll
  for (int i = 0; i < ND; ++i)
#pragma unroll
    for (int j = 0; j < ND; ++j) T[i][j] = 0.0;
  const int32_t q0 = A.offsets[e], q1 = A.offsets[e + 1];
  if (A.kernel == CFX_K_STIFFNESS)
  {
    double wsum = 0.0;
    for (int32_t q = q0 + sub; q < q1; q += kCutLanes) wsum += A.weights[q];
#pragma unroll
    for (int o = kCutLanes / 2; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o, kCutLanes);
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
      for (int j = 0; j < ND; ++j)
      {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < TDIM; ++d) s += G[i][d] * G[j][d];
        T[i][j] = wsum * s;
      }
  }
  else
  {
    const double gam = A.kernel == CFX_K_NITSCHE ? A.params[0] / cell_diameter<TDIM>(g) : 0.0;
    for (int32_t q = q0 + sub; q < q1; q += kCutLanes)
    {
      const double w = A.weights[q];
      double N[ND], l0 = 1.0;
#pragma unroll
      for (int t = 0; t < TDIM; ++t) { N[t + 1] = A.points[(int64_t)q * TDIM + t]; l0 -= N[t + 1]; }
      N[0] = l0;
      if (A.kernel == CFX_K_MASS)
      {
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
          for (int j = 0; j < ND; ++j) T[i][j] += w * N[i] * N[j];
      }
      else // Nitsche: -dn(u) v - dn(v) u + gamma/h u v with the per-point normal
      {
        const double* nrm = A.point_data + (int64_t)q * A.point_stride;
        double dn[ND];
#pragma unroll
        for (int j = 0; j < ND; ++j)
        {
          double s = 0.0;
#pragma unroll
          for (int d = 0; d < TDIM; ++d) s += G[j][d] * nrm[d];
          dn[j] = s;
        }
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
          for (int j = 0; j < ND; ++j) T[i][j] += w * (-dn[j] * N[i] - dn[i] * N[j] + gam * N[j] * N[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
      for (int j = 0; j < ND; ++j)
#pragma unroll
        for (int o = kCutLanes / 2; o > 0; o >>= 1) T[i][j] += __shfl_xor(T[i][j], o, kCutLanes);
  }
  if (sub != 0) return;
  double* out = A.dump + e * (ND * ND);
#pragma unroll
  for (int i = 0; i < ND; ++i)
#pragma unroll
    for (int j = 0; j < ND; ++j) out[i * ND + j] = T[i][j];
}

// ---------------------------------------------------------------------------
// Batched B^T D B on the FP64 matrix cores: the element tensors of sigma(u):eps(v) on vector spaces (bs = 3,
// BASELINE config 5), one wavefront per cell, written entity-major for the row gather.
// With c = 3 i + a numbering the (dof, component) pairs and g_c(q) = d N_i / d x_a at point q,
//   H[c][c'] = sum_q w_q g_c(q) g_c'(q)                       (a NLOC x nq times nq x NLOC product: MFMA),
//   Ae[(i,a),(j,b)] = lambda H[3i+a][3j+b] + mu H[3i+b][3j+a] + mu delta_ab sum_d H[3i+d][3j+d]
// -- the strain-displacement matrices B_q (6 x NLOC) and D never appear: B^T D B is this combination of the
// gradient Gram matrix H, whose K dimension is the quadrature point (4 per v_mfma_f64_16x16x4_f64).
// Lane l of the MFMA holds A[i = l & 15][k = l >> 4] = g_{16 tr + i}(q_k) and B[k][j = l & 15] = w_k g_{16 tc + j}(q_k):
// the same two gradient values serve as A and B operands of the 2 x 2 (P2) tiles.  H goes through LDS once
// (the lambda / mu combination reads entries held by other lanes) and the 900 doubles leave as full lines.
// ---------------------------------------------------------------------------
typedef double cfx_f64x4 __attribute__((ext_vector_type(4)));

template <int DEG, bool RUNTIME>
__global__ void __launch_bounds__(64) elasticity_tensors_mfma_kernel(AsmArgs A)
{
  constexpr int TDIM = 3, BS = 3;
  constexpr int ND = Elem<TDIM, DEG>::ND, NLOC = ND * BS;
  constexpr int NT = (NLOC + 15) / 16;  // 16 x 16 tiles per side
  constexpr int NB = ND * ND;           // 3 x 3 blocks (dof i, dof j) of the tensor
  // (one wavefront per workgroup: the barriers below order the wave's own LDS traffic and cost nothing)
  __shared__ __align__(16) double s_H[NLOC * NLOC]; // H, then Ae in place: the image that is streamed out
  __shared__ double s_L[4][3];        // physical gradients of the barycentric coordinates
  const int lane = threadIdx.x;
  const int64_t nwaves = gridDim.x;
  const double E = A.params[0], nu = A.params[1];
  const double mu = E / (2.0 * (1.0 + nu));
  const double lmbda = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu));
  // this lane's NT columns of G: dof i, component d, and the two barycentric coordinates of the basis function
  // (vertex function lam_a (2 lam_a - 1): a == b; edge function 4 lam_a lam_b)
  int ca[NT], cb[NT], cd[NT];
  bool cok[NT];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt)
  {
    const int c = 16 * tt + (lane & 15);
    cok[tt] = c < NLOC;
    const int i = cok[tt] ? c / 3 : 0;
    cd[tt] = cok[tt] ? c - 3 * i : 0;
    if (DEG == 1 || i < 4) { ca[tt] = i; cb[tt] = i; }
    else
    {
      const int e = i - 4; // Basix edge order (2,3),(1,3),(1,2),(0,3),(0,2),(0,1)
      ca[tt] = e == 0 ? 2 : (e <= 2 ? 1 : 0);
      cb[tt] = (e == 0 || e == 1 || e == 3) ? 3 : ((e == 2 || e == 4) ? 2 : 1);
    }
  }
  const int k = lane >> 4; // the quadrature point of a group of 4 this lane feeds
  // cell id -> connectivity row -> vertex coordinates are three dependent loads: a software pipeline three deep
  // keeps them ahead of the arithmetic (cell e is computed while the coordinates of e + nwaves, the connectivity
  // row of e + 2 nwaves and the id of e + 3 nwaves are in flight); indices past the end load the last entity again
  const int64_t An = dev_n(A.n);
  if (An == 0) return;
  const int64_t last = An - 1;
  auto cell_of = [&](int64_t i) { i = i < last ? i : last; return RUNTIME ? A.parent_map[i] : A.entities[i]; };
  auto conn_of = [&](int32_t c, int32_t* v)
  {
    const int4 r = *reinterpret_cast<const int4*>(A.conn + (int64_t)c * 4);
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
  };
  int32_t cell_c = cell_of(blockIdx.x + 2 * nwaves);
  int32_t vb[4], va[4];
  conn_of(cell_of(blockIdx.x), va);
  conn_of(cell_of(blockIdx.x + nwaves), vb);
  double xa[4][3], xb[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int d = 0; d < 3; ++d) xa[i][d] = A.x[3 * (int64_t)va[i] + d];
  for (int64_t e = blockIdx.x; e < An; e += nwaves)
  {
    const int32_t cell_d = cell_of(e + 3 * nwaves);
    int32_t vc[4];
    conn_of(cell_c, vc);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int d = 0; d < 3; ++d) xb[i][d] = A.x[3 * (int64_t)vb[i] + d];
    Geo<TDIM> g;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int d = 0; d < 3; ++d) g.x[i][d] = xa[i][d];
    jacobian<TDIM>(g);
    if (lane < 12)
    {
      const int m = lane / 3, d = lane - 3 * m;
      // grad lam_m = row m-1 of K for m >= 1, minus their sum for m = 0
      // (static indices only: a runtime-indexed register array would live in scratch memory)
      double v = 0.0;
#pragma unroll
      for (int t = 0; t < TDIM; ++t)
      {
        double ktd = g.K[t][0];
#pragma unroll
        for (int dd = 1; dd < TDIM; ++dd) ktd = (d == dd) ? g.K[t][dd] : ktd;
        v += (m == 0) ? -ktd : ((m - 1 == t) ? ktd : 0.0);
      }
      s_L[m][d] = v;
    }
    __syncthreads();
    double La[NT], Lb[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) { La[tt] = s_L[ca[tt]][cd[tt]]; Lb[tt] = s_L[cb[tt]][cd[tt]]; }
    int npts;
    const double* pts;
    const double* wts;
    double wscale = 1.0;
    if constexpr (RUNTIME)
    {
      const int32_t q0 = A.offsets[e];
      npts = A.offsets[e + 1] - q0;
      pts = A.points + (int64_t)q0 * TDIM;
      wts = A.weights + q0;
    }
    else
    {
      pts = ref_rule(TDIM, A.qdegree, npts, wts);
      wscale = fabs(g.detJ);
    }
    cfx_f64x4 acc[NT][NT];
#pragma unroll
    for (int tr = 0; tr < NT; ++tr)
#pragma unroll
      for (int tc = 0; tc < NT; ++tc) acc[tr][tc] = cfx_f64x4{0.0, 0.0, 0.0, 0.0};
    for (int q0 = 0; q0 < npts; q0 += 4)
    {
      const int q = q0 + k;
      const bool in = q < npts;
      double lam[4];
      lam[0] = 1.0;
#pragma unroll
      for (int t = 0; t < TDIM; ++t)
      {
        lam[t + 1] = in ? pts[(int64_t)q * TDIM + t] : 0.0;
        lam[0] -= lam[t + 1];
      }
      const double w = in ? wts[q] * wscale : 0.0;
      double gv[NT];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
      {
        double la = lam[0], lb = lam[0];
#pragma unroll
        for (int m = 1; m < 4; ++m) { la = (ca[tt] == m) ? lam[m] : la; lb = (cb[tt] == m) ? lam[m] : lb; }
        double v;
        if (DEG == 1) v = La[tt];
        else v = (ca[tt] == cb[tt]) ? (4.0 * la - 1.0) * La[tt] : 4.0 * (lb * La[tt] + la * Lb[tt]);
        gv[tt] = cok[tt] ? v : 0.0;
      }
#pragma unroll
      for (int tr = 0; tr < NT; ++tr)
#pragma unroll
        for (int tc = 0; tc < NT; ++tc)
          acc[tr][tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(gv[tr], w * gv[tc], acc[tr][tc], 0, 0, 0);
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: register r of lane l is H[(l >> 4) + 4 r][l & 15] of the tile
#pragma unroll
    for (int tr = 0; tr < NT; ++tr)
#pragma unroll
      for (int tc = 0; tc < NT; ++tc)
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
          const int row = 16 * tr + k + 4 * r, col = 16 * tc + (lane & 15);
          if (row < NLOC && col < NLOC) s_H[row * NLOC + col] = acc[tr][tc][r];
        }
    __syncthreads();
    // a lane owns the 3 x 3 blocks (i, j) = lane, lane + 64: Ae_blk = lambda H_blk + mu H_blk^T + mu tr(H_blk) I, in place
#pragma unroll
    for (int pass = 0; pass < (NB + 63) / 64; ++pass)
    {
      const int blk = lane + 64 * pass;
      if (blk < NB)
      {
        const int i = blk / ND, j = blk - i * ND;
        double* hb = s_H + (3 * i) * NLOC + 3 * j;
        double h[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) h[a][b] = hb[a * NLOC + b];
        const double tr = mu * (h[0][0] + h[1][1] + h[2][2]);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) hb[a * NLOC + b] = lmbda * h[a][b] + mu * h[b][a] + (a == b ? tr : 0.0);
      }
    }
    __syncthreads();
    double* out = A.dump + e * (int64_t)(NLOC * NLOC);
    // 16 B per lane: 1 KiB per store instruction (a tensor starts on a 16 B boundary: NLOC^2 is even)
    typedef double cfx_d2 __attribute__((ext_vector_type(2)));
    for (int o = lane; o < NLOC * NLOC / 2; o += 64)
      reinterpret_cast<cfx_d2*>(out)[o] = reinterpret_cast<const cfx_d2*>(s_H)[o];
    __syncthreads(); // s_H / s_L are reused by the next cell
    // rotate the pipeline registers here, behind the arithmetic: the moves wait for this iteration's loads
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
#pragma unroll
      for (int d = 0; d < 3; ++d) xa[i][d] = xb[i][d];
      vb[i] = vc[i];
    }
    cell_c = cell_d;
  }
}

// Entity-parallel scatter of staged local tensors ([n][ND x ND] or [n][ND], entity-major): the atomic path of the
// integrands that are compiled at run time (cfx_rtc.hip).  One thread per (entity, local row).
// (local index I = dof i * bs + component a: the blocked layout of the reference's dofmaps)
__global__ void __launch_bounds__(kBlock) scatter_staged_kernel(DevN n_d, const int32_t* __restrict__ cells, const int32_t* __restrict__ dofmap,
                                                                int nd, int bs, int rank, const double* __restrict__ staged,
                                                                const int8_t* __restrict__ bc0, const int8_t* __restrict__ bc1,
                                                                const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                                double* __restrict__ values, int* error)
{
  const int64_t n = dev_n(n_d);
  const int nloc = nd * bs;
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t e = t / nloc;
  if (e >= n) return;
  const int I = (int)(t - e * nloc);
  const int64_t c = cells[e];
  const int32_t row = dofmap[c * nd + I / bs] * bs + I % bs;
  if (rank == 1)
  {
    atomicAdd(&values[row], staged[e * nloc + I]);
    return;
  }
  // zero BC rows / columns: assemble_matrix_impl.h:151-185 (the diagonal is the caller's: set_diagonal)
  const bool row_bc = bc0 && bc0[row];
  const int64_t rb = indptr[row], re = indptr[row + 1];
  for (int j = 0; j < nd; ++j)
  {
    const int32_t col0 = dofmap[c * nd + j] * bs;
    const int64_t pos = csr_find(indices, rb, re, col0);
    if (pos < 0) { *error = 1; continue; }
    for (int b = 0; b < bs; ++b)
    {
      double v = staged[(e * nloc + I) * nloc + j * bs + b];
      if (row_bc || (bc1 && bc1[col0 + b])) v = 0.0;
      atomicAdd(&values[pos + b], v);
    }
  }
}

// ... of staged facet macro tensors [n][2 nloc][2 nloc]: one thread per (facet, macro row); macro dofs = [cell 0, cell 1]
__global__ void __launch_bounds__(kBlock) scatter_staged_facets_kernel(DevN n_d, const int32_t* __restrict__ rows,
                                                                       const int32_t* __restrict__ dofmap, int nd, int bs,
                                                                       const double* __restrict__ staged,
                                                                       const int8_t* __restrict__ bc0, const int8_t* __restrict__ bc1,
                                                                       const int64_t* __restrict__ indptr,
                                                                       const int32_t* __restrict__ indices,
                                                                       double* __restrict__ values, int* error)
{
  const int64_t n = dev_n(n_d);
  const int nloc = nd * bs, nm = 2 * nloc;
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t f = t / nm;
  if (f >= n) return;
  const int I = (int)(t - f * nm);
  const int64_t c[2] = {rows[4 * f], rows[4 * f + 2]};
  const int si = I / nloc, Ii = I - si * nloc;
  const int32_t row = dofmap[c[si] * nd + Ii / bs] * bs + Ii % bs;
  const bool row_bc = bc0 && bc0[row];
  const int64_t rb = indptr[row], re = indptr[row + 1];
  for (int s = 0; s < 2; ++s)
    for (int j = 0; j < nd; ++j)
    {
      const int32_t col0 = dofmap[c[s] * nd + j] * bs;
      const int64_t pos = csr_find(indices, rb, re, col0);
      if (pos < 0) { *error = 1; continue; }
      for (int b = 0; b < bs; ++b)
      {
        double v = staged[(f * nm + I) * nm + s * nloc + j * bs + b];
        if (row_bc || (bc1 && bc1[col0 + b])) v = 0.0;
        atomicAdd(&values[pos + b], v);
      }
    }
}

// a cell integral whose integrand was registered at run time: stage 1 through the compiled wrapper, then -- unless the
// caller wants the staged tensors themselves (A.dump: the row gather, tabulate_entity) -- the scatter above
void launch_user_integral(const cfx_form_s* a, const cfx_integral_dev& I, AsmArgs A, int64_t only_index, int use_rule, int parts)
{
  const cfx_space_s* V = a->V;
  require(A.lift_markers == nullptr, CFX_ERR_INVALID_ARGUMENT, "apply_lifting is not available for user integrands");
  const bool single = only_index >= 0;
  const int nd = V->ndofs_cell, bs = V->bs;
  if (I.type == CFX_INTERIOR_FACET)
  {
    // macro tensors [facet][2 nloc][2 nloc]: to the caller (row gather, tabulate_entity), or staged and scattered
    const DevN n = single ? DevN(1) : I.n_entities.devn();
    if (n.cap == 0) return;
    if (A.dump) { user_stage1_facets(a, I, A.dump, only_index); return; }
    const int64_t nm = 2 * (int64_t)nd * bs;
    DevArray<double> staged(n.cap * nm * nm);
    user_stage1_facets(a, I, staged.p, only_index);
    launch("scatter_staged", scatter_staged_facets_kernel, grid_for(n.cap * nm), dim3(kBlock), 0, n,
           I.entities.p + 4 * (single ? only_index : 0), V->dofmap.p, nd, bs, (const double*)staged.p, A.bc0, A.bc1, A.indptr,
           A.indices, A.values, A.error);
    return;
  }
  require(I.type == CFX_CELL, CFX_ERR_INVALID_ARGUMENT, "user integrands are cell or interior-facet integrals");
  const int64_t nloc = (int64_t)nd * bs;
  const int64_t nt = a->rank == 2 ? nloc * nloc : nloc;
  for (int part = 1; part <= 2; ++part)
  {
    if (!(parts & part)) continue;
    const bool runtime = part == 2;
    if (single && (runtime != (use_rule != 0))) continue;
    if (runtime && !I.rules) continue;
    const DevN n = single ? DevN(1) : (runtime ? I.rules->nr.devn() : I.n_entities.devn());
    if (n.cap == 0) continue;
    if (A.dump) { user_stage1(a, I, runtime, A.dump, 0, 0, only_index); continue; }
    DevArray<double> staged(n.cap * nt);
    user_stage1(a, I, runtime, staged.p, 0, 0, only_index);
    const int32_t* cells = runtime ? I.rules->parent_map.p : I.entities.p;
    launch("scatter_staged", scatter_staged_kernel, grid_for(n.cap * nloc), dim3(kBlock), 0, n, cells + (single ? only_index : 0),
           V->dofmap.p, nd, bs, a->rank, (const double*)staged.p, A.bc0, A.bc1, A.indptr, A.indices, A.values, A.error);
  }
}

template <int TDIM, int DEG, int BS>
void launch_integral_t(const cfx_form_s* a, const cfx_integral_dev& I, AsmArgs A, int64_t only_index, int use_rule,
                       int parts)
{
  constexpr int ND = Elem<TDIM, DEG>::ND;
  const bool single = only_index >= 0;
  if (user_integrand_known(I.kernel)) { launch_user_integral(a, I, A, only_index, use_rule, parts); return; }
  if (I.type == CFX_INTERIOR_FACET)
  {
    require(a->rank == 2, CFX_ERR_INVALID_ARGUMENT, "interior-facet integrals are implemented for bilinear forms");
    A.entities = I.entities.p;
    A.point_data = I.point_data.n > 0 ? I.point_data.p : nullptr; // per-pair factors
    A.n_std = I.n_std < 0 ? INT64_MAX : I.n_std; // (no facet-hosted rules: every entity is a standard facet)
    A.dump0 = single ? only_index : 0;
    if (I.rules)
    {
      A.offsets = I.rules->offsets.p; A.points = I.rules->points.p; A.weights = I.rules->weights.p;
      A.host_verts = I.rules->host_verts.p;
    }
    const bool dg = A.kernel == CFX_K_JUMP || A.kernel == CFX_K_SIP;
    if (!single && !I.rules)
    {
      // standard facets only (the ghost-penalty band): the whole list, whose length may still be in HBM
      A.f0 = 0; A.n = I.n_entities;
      if (A.n.cap > 0)
      {
        if (dg)
          launch("assemble_facets", assemble_facets_kernel<TDIM, DEG, BS, false, 1>, grid_for(A.n.cap * 2 * ND * BS), dim3(kBlock), 0, A);
        else
          launch("assemble_facets", assemble_facets_kernel<TDIM, DEG, BS, false, 0>, grid_for(A.n.cap * 2 * ND * BS), dim3(kBlock), 0, A);
      }
      return;
    }
    // standard entities [0, n_std), then the rules' entities [n_std, n_entities)
    const int64_t lo = single ? only_index : 0, hi = single ? only_index + 1 : I.n_entities.value();
    const int64_t mid = std::min(std::max(I.n_std < 0 ? hi : I.n_std, lo), hi);
    if (mid > lo)
    {
      A.f0 = lo; A.n = mid - lo;
      if (dg)
        launch("assemble_facets", assemble_facets_kernel<TDIM, DEG, BS, false, 1>, grid_for(A.n.cap * 2 * ND * BS), dim3(kBlock), 0, A);
      else
        launch("assemble_facets", assemble_facets_kernel<TDIM, DEG, BS, false, 0>, grid_for(A.n.cap * 2 * ND * BS), dim3(kBlock), 0, A);
    }
    if (hi > mid)
    {
      A.f0 = mid; A.n = hi - mid;
      if (dg)
        launch("assemble_facets_cut", assemble_facets_kernel<TDIM, DEG, BS, true, 1>, grid_for(A.n.cap * 2 * ND * BS), dim3(kBlock),
               0, A);
      else
        launch("assemble_facets_cut", assemble_facets_kernel<TDIM, DEG, BS, true, 0>, grid_for(A.n.cap * 2 * ND * BS), dim3(kBlock),
               0, A);
    }
    return;
  }
  // staged element tensors of the elasticity term on 3-D vector spaces: the MFMA kernel (CFX_MFMA=0: generic rows)
  const char* mf = getenv("CFX_MFMA");
  const bool mfma_tensors = TDIM == 3 && BS == 3 && a->rank == 2 && A.dump != nullptr && !single && !A.lift_markers
                            && A.kernel == CFX_K_ELASTICITY && !A.coeff && !(mf && mf[0] == '0');
  // a resident grid of wavefronts, each walking its share of the cells
  auto mfma_grid = [](int64_t n) { return dim3((unsigned)std::min<int64_t>(n, 256 * 32)); };
  if ((parts & 1) && (!single || !use_rule))
  {
    A.n = single ? DevN(1) : I.n_entities.devn();
    A.entities = I.entities.p + (single ? only_index : 0);
    if (A.n.cap > 0)
    {
      if (mfma_tensors)
      {
        if constexpr (TDIM == 3 && BS == 3)
          launch("elasticity_tensors_mfma", elasticity_tensors_mfma_kernel<DEG, false>, mfma_grid(A.n.cap), dim3(64), 0, A);
      }
      else if (a->rank == 2)
        launch("assemble_cells_std", assemble_cells_kernel<TDIM, DEG, BS, 2, false>, grid_for(A.n.cap * ND * BS),
               dim3(kBlock), 0, A);
      else
        launch("assemble_vec_std", assemble_cells_kernel<TDIM, DEG, BS, 1, false>, grid_for(A.n.cap * ND * BS),
               dim3(kBlock), 0, A);
    }
  }
  if ((parts & 2) && I.rules && (!single || use_rule))
  {
    const cfx_rules_s* R = I.rules;
    A.n = single ? DevN(1) : R->nr.devn();
    A.offsets = R->offsets.p + (single ? only_index : 0);
    A.parent_map = R->parent_map.p + (single ? only_index : 0);
    A.points = R->points.p;
    A.weights = R->weights.p;
    A.point_data = I.point_data.n > 0 ? I.point_data.p : nullptr;
    if (A.n.cap > 0)
    {
      const char* spec = getenv("CFX_CUT_TENSORS_P1");
      if (a->rank == 2 && DEG == 1 && BS == 1 && A.dump && !single && !A.coeff && !(spec && spec[0] == '0')
          && (A.kernel == CFX_K_STIFFNESS || A.kernel == CFX_K_MASS || A.kernel == CFX_K_NITSCHE))
      {
        if constexpr (DEG == 1 && BS == 1)
          launch("cut_tensors_p1", cut_tensors_p1_kernel<TDIM>, grid_for(A.n.cap * kCutLanes), dim3(kBlock), 0, A);
      }
      else if (mfma_tensors)
      {
        if constexpr (TDIM == 3 && BS == 3)
          launch("elasticity_tensors_mfma_cut", elasticity_tensors_mfma_kernel<DEG, true>, mfma_grid(A.n.cap), dim3(64), 0, A);
      }
      else if (a->rank == 2)
        launch("assemble_cells_cut", assemble_cells_kernel<TDIM, DEG, BS, 2, true>, grid_for(A.n.cap * ND * BS),
               dim3(kBlock), 0, A);
      else
        launch("assemble_vec_cut", assemble_cells_kernel<TDIM, DEG, BS, 1, true>, grid_for(A.n.cap * ND * BS),
               dim3(kBlock), 0, A);
    }
  }
}

void launch_integral(const cfx_form_s* a, const cfx_integral_dev& I, AsmArgs A, int64_t only_index = -1,
                     int use_rule = 0, int parts = 3)
{
  const cfx_space_s* V = a->V;
  const int tdim = V->mesh->tdim;
  A.kernel = I.kernel; A.qdegree = I.qdegree; A.point_stride = I.point_stride;
  for (int k = 0; k < 8; ++k) A.params[k] = I.params[k];
  A.coeff = I.coefficient.n > 0 ? I.coefficient.p : nullptr;
  const int key = tdim * 100 + V->degree * 10 + V->bs;
  switch (key)
  {
  case 211: launch_integral_t<2, 1, 1>(a, I, A, only_index, use_rule, parts); break;
  case 221: launch_integral_t<2, 2, 1>(a, I, A, only_index, use_rule, parts); break;
  case 212: launch_integral_t<2, 1, 2>(a, I, A, only_index, use_rule, parts); break;
  case 222: launch_integral_t<2, 2, 2>(a, I, A, only_index, use_rule, parts); break;
  case 311: launch_integral_t<3, 1, 1>(a, I, A, only_index, use_rule, parts); break;
  case 321: launch_integral_t<3, 2, 1>(a, I, A, only_index, use_rule, parts); break;
  case 313: launch_integral_t<3, 1, 3>(a, I, A, only_index, use_rule, parts); break;
  case 323: launch_integral_t<3, 2, 3>(a, I, A, only_index, use_rule, parts); break;
  default: throw Error(CFX_ERR_INVALID_ARGUMENT, "unsupported (tdim, degree, block size) combination");
  }
}

// ---------------------------------------------------------------------------
// Cell integrals of a form whose test and trial spaces differ (cfx_form_create2; assemble_matrix_impl.h:68-189 with
// dofmap0 / bs0 != dofmap1 / bs1).  One thread per (entity, test row): row (ia, ik) of the [(nd0 bs0) x (nd1 bs1)]
// element tensor in registers (acc[j][b], fixed strides: the sizes are run-time values here -- one kernel serves
// every pair of Lagrange spaces instead of one instantiation per pair), scattered into its CSR row with one search
// per trial dof and FP64 atomics, or contracted with the Dirichlet data (lifting), or dumped (tabulate_entity).
// ---------------------------------------------------------------------------
template <int TDIM, bool RUNTIME>
__global__ void __launch_bounds__(kBlock) assemble_cells2_kernel(AsmArgs A, RectArgs R)
{
  constexpr int MAXND = RectRow<TDIM>::MAXND, MAXBS = RectRow<TDIM>::MAXBS;
  const int nloc0 = R.nd0 * R.bs0, nloc1 = R.nd1 * R.bs1;
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t e = tid / nloc0;
  if (e >= dev_n(A.n)) return;
  const int I0 = (int)(tid - e * nloc0);
  const int ia = I0 / R.bs0, ik = I0 - ia * R.bs0;
  const int64_t cell = RUNTIME ? A.parent_map[e] : A.entities[e];
  const int32_t* d1 = R.dofmap1 + cell * R.nd1;
  if (A.lift_markers)
  {
    bool any = false;
    for (int j = 0; j < R.nd1; ++j)
      for (int b = 0; b < R.bs1; ++b) any = any || A.lift_markers[R.bs1 * d1[j] + b] != 0;
    if (!any) return;
  }
  Geo<TDIM> g;
  load_cell<TDIM>(A.x, A.conn, cell, g);
  jacobian<TDIM>(g);
  int npts;
  const double* pts;
  const double* wts;
  double wscale = 1.0;
  if constexpr (RUNTIME)
  {
    const int32_t q0 = A.offsets[e], q1 = A.offsets[e + 1];
    npts = q1 - q0;
    pts = A.points + (int64_t)q0 * TDIM;
    wts = A.weights + q0;
  }
  else
  {
    pts = ref_rule(TDIM, A.qdegree, npts, wts);
    wscale = fabs(g.detJ);
  }
  double acc[MAXND][MAXBS];
#pragma unroll
  for (int j = 0; j < MAXND; ++j)
#pragma unroll
    for (int b = 0; b < MAXBS; ++b) acc[j][b] = 0.0;
  RectRow<TDIM>::accumulate(R, A.kernel, A.params[0], g, ia, ik, npts, pts, wts, wscale, acc);
  if (A.dump)
  {
    double* out = A.dump + (e * nloc0 + I0) * (int64_t)nloc1;
#pragma unroll
    for (int j = 0; j < MAXND; ++j)
#pragma unroll
      for (int b = 0; b < MAXBS; ++b)
        if (j < R.nd1 && b < R.bs1) out[j * R.bs1 + b] = acc[j][b];
    return;
  }
  const int32_t row = R.bs0 * A.dofmap[cell * R.nd0 + ia] + ik;
  if (A.lift_markers)
  {
    double sum = 0.0;
    bool any = false;
#pragma unroll
    for (int j = 0; j < MAXND; ++j)
#pragma unroll
      for (int b = 0; b < MAXBS; ++b)
        if (j < R.nd1 && b < R.bs1)
        {
          const int32_t col = R.bs1 * d1[j] + b;
          if (A.lift_markers[col])
          {
            sum += acc[j][b] * A.lift_alpha * (A.lift_values[col] - (A.lift_x0 ? A.lift_x0[col] : 0.0));
            any = true;
          }
        }
    if (any) atomicAdd(A.values + row, -sum);
    return;
  }
  const bool row_bc = A.bc0 && A.bc0[row];
  const int64_t rb = A.indptr[row], re = A.indptr[row + 1];
#pragma unroll
  for (int j = 0; j < MAXND; ++j)
  {
    if (j >= R.nd1) continue;
    const int32_t col0 = R.bs1 * d1[j];
    const int64_t pos = csr_find(A.indices, rb, re, col0);
    if (pos < 0) { *A.error = 1; continue; }
#pragma unroll
    for (int b = 0; b < MAXBS; ++b)
      if (b < R.bs1)
      {
        double v = acc[j][b];
        if (row_bc || (A.bc1 && A.bc1[col0 + b])) v = 0.0;
        atomicAdd(A.values + pos + b, v);
      }
  }
}

// Interior-facet integrals of a rectangular form (two scalar spaces): one thread per (facet, macro test row).  The
// macro row -- [cell0, cell1] columns of the TRIAL dofmap -- is scattered with one search per trial dof and FP64 atomics,
// contracted with the Dirichlet data (lifting) or dumped (tabulate_entity).  Facet terms between different spaces are
// rare (no demo or test of the reference has one off the block diagonal): they keep the entity-parallel path.
template <int TDIM>
__global__ void __launch_bounds__(kBlock) assemble_facets2_kernel(AsmArgs A, RectArgs R)
{
  constexpr int MAXND = RectRow<TDIM>::MAXND;
  const int nm0 = 2 * R.nd0;
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t e = tid / nm0;
  if (e >= dev_n(A.n)) return;
  const int ia = (int)(tid - e * nm0);
  const int64_t c0 = A.entities[4 * e], c1 = A.entities[4 * e + 2];
  const int lf0 = A.entities[4 * e + 1];
  const int side = ia >= R.nd0 ? 1 : 0;
  const int32_t* d1[2] = {R.dofmap1 + c0 * R.nd1, R.dofmap1 + c1 * R.nd1};
  if (A.lift_markers)
  {
    bool any = false;
    for (int sd = 0; sd < 2; ++sd)
      for (int j = 0; j < R.nd1; ++j) any = any || A.lift_markers[d1[sd][j]] != 0;
    if (!any) return;
  }
  Geo<TDIM> g0, g1;
  load_cell<TDIM>(A.x, A.conn, c0, g0);
  load_cell<TDIM>(A.x, A.conn, c1, g1);
  jacobian<TDIM>(g0);
  jacobian<TDIM>(g1);
  double acc[2 * MAXND];
#pragma unroll
  for (int j = 0; j < 2 * MAXND; ++j) acc[j] = 0.0;
  facet_local_row2<TDIM>(R, A.kernel, A.params, A.qdegree, g0, g1, lf0, ia, acc);
  if (A.dump)
  {
    double* out = A.dump + (e * nm0 + ia) * (int64_t)(2 * R.nd1);
#pragma unroll
    for (int j = 0; j < MAXND; ++j)
      if (j < R.nd1) { out[j] = acc[j]; out[R.nd1 + j] = acc[MAXND + j]; }
    return;
  }
  const int32_t row = A.dofmap[(side ? c1 : c0) * R.nd0 + (ia - side * R.nd0)];
  if (A.lift_markers)
  {
    double sum = 0.0;
    bool any = false;
#pragma unroll
    for (int j = 0; j < MAXND; ++j)
      if (j < R.nd1)
      {
        for (int sd = 0; sd < 2; ++sd)
        {
          const int32_t col = d1[sd][j];
          if (A.lift_markers[col])
          {
            sum += acc[sd * MAXND + j] * A.lift_alpha * (A.lift_values[col] - (A.lift_x0 ? A.lift_x0[col] : 0.0));
            any = true;
          }
        }
      }
    if (any) atomicAdd(A.values + row, -sum);
    return;
  }
  const bool row_bc = A.bc0 && A.bc0[row];
  const int64_t rb = A.indptr[row], re = A.indptr[row + 1];
#pragma unroll
  for (int j = 0; j < MAXND; ++j)
  {
    if (j >= R.nd1) continue;
    for (int sd = 0; sd < 2; ++sd)
    {
      const int32_t col = d1[sd][j];
      const int64_t pos = csr_find(A.indices, rb, re, col);
      if (pos < 0) { *A.error = 1; continue; }
      double v = acc[sd * MAXND + j];
      if (row_bc || (A.bc1 && A.bc1[col])) v = 0.0;
      atomicAdd(A.values + pos, v);
    }
  }
}

// every integral of a rectangular form, or one entity of one of them (only_index >= 0)
void launch_rectangular(const cfx_form_s* a, const cfx_integral_dev& I, AsmArgs A, int64_t only_index = -1, int use_rule = 0)
{
  const cfx_space_s* V0 = a->V;
  const cfx_space_s* V1 = a->V1;
  RectArgs R{V1->dofmap.p, V0->degree, V0->bs, V0->ndofs_cell, V1->degree, V1->bs, V1->ndofs_cell};
  A.kernel = I.kernel; A.qdegree = I.qdegree;
  for (int k = 0; k < 8; ++k) A.params[k] = I.params[k];
  const bool single = only_index >= 0;
  const int nloc0 = R.nd0 * R.bs0;
  const int tdim = V0->mesh->tdim;
  if (I.type == CFX_INTERIOR_FACET)
  {
    A.n = single ? 1 : I.n_entities.value();
    A.entities = I.entities.p + (single ? 4 * only_index : 0);
    if (A.n.cap > 0)
    {
      if (tdim == 2) launch("assemble_facets2", assemble_facets2_kernel<2>, grid_for(A.n.cap * 2 * R.nd0), dim3(kBlock), 0, A, R);
      else launch("assemble_facets2", assemble_facets2_kernel<3>, grid_for(A.n.cap * 2 * R.nd0), dim3(kBlock), 0, A, R);
    }
    return;
  }
  if (!single || !use_rule)
  {
    A.n = single ? 1 : I.n_entities.value();
    A.entities = I.entities.p + (single ? only_index : 0);
    if (A.n.cap > 0)
    {
      if (tdim == 2) launch("assemble_cells2_std", assemble_cells2_kernel<2, false>, grid_for(A.n.cap * nloc0), dim3(kBlock), 0, A, R);
      else launch("assemble_cells2_std", assemble_cells2_kernel<3, false>, grid_for(A.n.cap * nloc0), dim3(kBlock), 0, A, R);
    }
  }
  if (I.rules && (!single || use_rule))
  {
    const cfx_rules_s* Q = I.rules;
    A.n = single ? 1 : Q->nr.value();
    A.offsets = Q->offsets.p + (single ? only_index : 0);
    A.parent_map = Q->parent_map.p + (single ? only_index : 0);
    A.points = Q->points.p; A.weights = Q->weights.p;
    if (A.n.cap > 0)
    {
      if (tdim == 2) launch("assemble_cells2_cut", assemble_cells2_kernel<2, true>, grid_for(A.n.cap * nloc0), dim3(kBlock), 0, A, R);
      else launch("assemble_cells2_cut", assemble_cells2_kernel<3, true>, grid_for(A.n.cap * nloc0), dim3(kBlock), 0, A, R);
    }
  }
}

__global__ void mark_cells_kernel(DevN n_d, const int32_t* __restrict__ cells, int stride, uint8_t* mark)
{
  const int64_t n = dev_n(n_d);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) mark[cells[i * stride]] = 1;
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

// CFX_ASSEMBLY=atomic selects the entity-parallel FP64-atomic kernels (tests cover both paths)
bool force_atomic()
{
  const char* e = getenv("CFX_ASSEMBLY");
  return e && strcmp(e, "atomic") == 0;
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

__global__ void deactivate_kernel(DevN n_d, const int32_t* __restrict__ rows, const int64_t* __restrict__ indptr,
                                  const int32_t* __restrict__ indices, double* values, double* b, double diagonal,
                                  double rhs_value, int* error)
{
  const int64_t n = dev_n(n_d);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t r = rows[i];
  if (values)
  {
    const int64_t rb = indptr[r], re = indptr[r + 1];
    // a deactivated row is normally its diagonal alone (assembler.h:538-560): position known without the search
    const int64_t pos = (re - rb == 1 && indices[rb] == r) ? rb : csr_find(indices, rb, re, r);
    if (pos < 0) *error = 1; else values[pos] = diagonal; // set, not add (set_diagonal via mat_set_values)
  }
  if (b) b[r] = rhs_value;
}

// Deactivation straight from the row marks.  A tile of kByteTile rows that are ALL inactive (most inactive rows: the
// region outside the domain) holds one entry per row, its diagonal (assembler.h:538-560: the all-rows diagonal of every
// pattern this library builds; checked here: one entry per row through the tile's two row pointers, each entry its
// row's diagonal through one coalesced read): row r0 + k sits k entries behind the tile's first row pointer, so the tile
// is two contiguous fills.  The other tiles set their unmarked rows one by one.  tile_counts: active rows per tile
// (special | plain << 32, cfx_row_plan::row_tile_counts).
__global__ void __launch_bounds__(kBlock) deactivate_marks_kernel(int64_t nrows, const uint8_t* __restrict__ rowmark,
                                                                  const int64_t* __restrict__ tile_counts,
                                                                  const int64_t* __restrict__ indptr,
                                                                  const int32_t* __restrict__ indices, double* __restrict__ values,
                                                                  double* __restrict__ b, double diagonal, double rhs_value, int* error,
                                                                  DevN n_active_d, DevN nnz_d)
{
  // (lengths still in HBM: nothing to do in a void step -- marks and row pointers are not those of this step then)
  if (n_active_d.dev && dev_n(n_active_d) == 0) return;
  if (nnz_d.dev && dev_n(nnz_d) == 0) return;
  const int64_t r0 = (int64_t)blockIdx.x * kByteTile;
  const int tl = (int)min((int64_t)kByteTile, nrows - r0);
  const int64_t tc = tile_counts[blockIdx.x];
  const int nact = (int)((tc & 0xffffffffll) + (tc >> 32));
  if (nact == 0 && (!values || indptr[r0 + tl] - indptr[r0] == tl))
  {
    int ok = 1;
    const int64_t e0 = values ? indptr[r0] : 0;
    if (values)
      for (int k = threadIdx.x; k < tl; k += kBlock) ok &= indices[e0 + k] == (int32_t)(r0 + k) ? 1 : 0;
    if (__syncthreads_and(ok))
    {
      if (values) block_fill_run(values + e0, tl, diagonal);
      if (b) block_fill_run(b + r0, tl, rhs_value);
      return;
    }
  }
  if (nact == tl) return;
  for (int k = threadIdx.x; k < tl; k += kBlock)
  {
    const int64_t r = r0 + k;
    if (rowmark[r]) continue;
    if (values)
    {
      const int64_t rb = indptr[r], re = indptr[r + 1];
      const int64_t pos = (re - rb == 1 && indices[rb] == r) ? rb : csr_find(indices, rb, re, (int32_t)r);
      if (pos < 0) *error = 1; else values[pos] = diagonal;
    }
    if (b) b[r] = rhs_value;
  }
}

__global__ void inactive_tile_counts_kernel(int64_t ntiles, int64_t n, const int64_t* __restrict__ active, int32_t* __restrict__ zeros)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntiles) return;
  const int64_t len = min((int64_t)kByteTile, n - t * kByteTile);
  zeros[t] = (int32_t)(len - (active[t] & 0xffffffffll) - (active[t] >> 32));
}

__global__ void facet_cells_covered_kernel(DevN nf_d, const int32_t* __restrict__ rows, const uint8_t* __restrict__ cellmark,
                                           int* uncovered)
{
  const int64_t nf = dev_n(nf_d);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * nf) return;
  if (cellmark[rows[4 * (i >> 1) + 2 * (i & 1)]] == 0) atomicOr(uncovered, 1);
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
      if (I.n_entities.cap() > 0)
      {
        launch("mark_cells", mark_cells_kernel, grid_for(I.n_entities.cap()), dim3(kBlock), 0, I.n_entities, I.entities.p, 1,
               mark.p);
        any = true;
      }
      if (I.rules && I.rules->nr.cap() > 0)
      {
        launch("mark_cells", mark_cells_kernel, grid_for(I.rules->nr.cap()), dim3(kBlock), 0, I.rules->nr,
               I.rules->parent_map.p, 1, mark.p);
        any = true;
      }
    }
    else if (include_facets && I.type == CFX_INTERIOR_FACET && I.n_entities.cap() > 0)
    {
      launch("mark_cells", mark_cells_kernel, grid_for(I.n_entities.cap()), dim3(kBlock), 0, I.n_entities, I.entities.p, 4,
             mark.p);
      launch("mark_cells", mark_cells_kernel, grid_for(I.n_entities.cap()), dim3(kBlock), 0, I.n_entities,
             I.entities.p + 2, 4, mark.p);
      any = true;
    }
  }
}

} // namespace

namespace
{
// P1 gradient-jump ghost penalty, stage 1 for the row gather: the facet tensor gamma h_avg |F| [dn N_i][dn N_j] is
// RANK ONE (P1 gradients are constant), and folded over the dofs the two cells share it is w jf jf^T with
// jf = (jump of cell 0's basis i + that of its twin in cell 1, ..., jump of cell 1's free dof).  One thread per facet
// writes the 80-byte record (jf[0..ND], w, the macro column ids) instead of eight threads writing 64 (folded: 25)
// doubles; the gather forms row r as w jf[m(r)] jf[.] without touching the facet row or the dofmap.  facet_local_row() is the generic statement of the same integrand.
template <int TDIM>
__global__ void __launch_bounds__(kBlock) facet_jump_p1_kernel(DevN n_d, const int32_t* __restrict__ rows,
                                                               const double* __restrict__ x, const int32_t* __restrict__ conn,
                                                               const int32_t* __restrict__ dofmap, double gamma, double hpow,
                                                               int qdegree, double* __restrict__ out, int* error)
{
  const int64_t n = dev_n(n_d);
  constexpr int ND = TDIM + 1;
  const int64_t f = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (f >= n) return;
  const int4 row4 = *reinterpret_cast<const int4*>(rows + 4 * f);
  const int64_t c0 = row4.x, c1 = row4.z;
  const int lf0 = row4.y;
  Geo<TDIM> g0, g1;
  load_cell<TDIM>(x, conn, c0, g0);
  load_cell<TDIM>(x, conn, c1, g1);
  jacobian<TDIM>(g0);
  jacobian<TDIM>(g1);
  const double havg = 0.5 * (cell_diameter<TDIM>(g0) + cell_diameter<TDIM>(g1));
  double nrm[TDIM], nn = 0.0;
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    double v = 0.0;
#pragma unroll
    for (int t = 0; t < TDIM; ++t) v -= g0.K[t][d] * ((lf0 == 0) ? -1.0 : ((lf0 - 1 == t) ? 1.0 : 0.0));
    nrm[d] = v;
    nn += v * v;
  }
  nn = sqrt(nn);
#pragma unroll
  for (int d = 0; d < TDIM; ++d) nrm[d] /= nn;
  // measure of the facet: cell 0's vertices except lf0
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
  int nref;
  const double* wref;
  (void)ref_rule(TDIM - 1, qdegree, nref, wref);
  double wsum = 0.0;
  for (int q = 0; q < nref; ++q) wsum += wref[q];
  // normal-derivative jumps of the macro basis: grad N_0 = -sum_t K[t][.], grad N_{t+1} = K[t][.]
  double j0[ND], j1[ND];
  j0[0] = 0.0; j1[0] = 0.0;
#pragma unroll
  for (int t = 0; t < TDIM; ++t)
  {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int d = 0; d < TDIM; ++d) { a += g0.K[t][d] * nrm[d]; b += g1.K[t][d] * nrm[d]; }
    j0[t + 1] = a; j0[0] -= a;
    j1[t + 1] = -b; j1[0] += b;
  }
  int32_t d0[ND], d1[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j) { d0[j] = dofmap[c0 * ND + j]; d1[j] = dofmap[c1 * ND + j]; }
  double jf[ND + 1];
#pragma unroll
  for (int i = 0; i < ND; ++i) jf[i] = j0[i];
  jf[ND] = 0.0;
  int nfree = 0;
#pragma unroll
  for (int j = 0; j < ND; ++j)
  {
    bool shared = false;
#pragma unroll
    for (int i = 0; i < ND; ++i)
    {
      const bool same = d1[j] == d0[i];
      jf[i] += same ? j1[j] : 0.0;
      shared = shared || same;
    }
    jf[ND] += shared ? 0.0 : j1[j];
    nfree += shared ? 0 : 1;
  }
  if (error && nfree != 1) *error = 4; // not an interior facet of a conforming mesh with a continuous space
  // 80-byte record: 6 doubles (jf[0..ND], w; 2-D: one unused), then 8 int32: the ND + 1 macro columns (cell 0's dofs,
  // cell 1's free dof), -1 padding, entry 7 = number of dofs of cell 1 that cell 0 lacks (1 on a conforming mesh)
  double rec[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) rec[k] = 0.0;
#pragma unroll
  for (int k = 0; k <= ND; ++k) rec[k] = jf[k];
  rec[ND + 1] = wsum * scale * gamma * havg * (hpow != 0.0 ? pow(havg, hpow) : 1.0);
  int32_t cm[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) cm[k] = -1;
#pragma unroll
  for (int i = 0; i < ND; ++i) cm[i] = d0[i];
#pragma unroll
  for (int j = 0; j < ND; ++j)
  {
    bool shared = false;
#pragma unroll
    for (int i = 0; i < ND; ++i) shared = shared || d1[j] == d0[i];
    cm[ND] = shared ? cm[ND] : d1[j];
  }
  cm[7] = nfree;
  double2* o = reinterpret_cast<double2*>(out + 10 * f);
#pragma unroll
  for (int k = 0; k < 3; ++k) o[k] = make_double2(rec[2 * k], rec[2 * k + 1]);
  int4* oc = reinterpret_cast<int4*>(out + 10 * f + 6);
  oc[0] = make_int4(cm[0], cm[1], cm[2], cm[3]);
  oc[1] = make_int4(cm[4], cm[5], cm[6], cm[7]);
}
// Degree-2 gradient-jump ghost penalty, stage 1 for the row gather: at one quadrature point q the integrand is the
// rank-one matrix w_q [dn N_i][dn N_j], so the facet tensor is a sum of nq rank-one terms (nq = 3 for degree-2
// integrands on a triangle).  One thread per (facet, point) writes the record (jf_q[0..WF), w_q) with jf_q folded
// over the dofs the two cells share: nq x 16 doubles per facet instead of (2 ND)^2 = 400, and the gather forms its row
// as sum_q w_q jf_q[m] jf_q[.].  Macro dofs: cell 0's, then the dofs of cell 1 that cell 0 does not have in cell 1's
// local order.  facet_local_row() is the generic statement of the same integrand.
template <int TDIM>
__global__ void __launch_bounds__(kBlock) facet_jumps_p2_kernel(int64_t n, int nq, const int32_t* __restrict__ rows,
                                                                const double* __restrict__ x, const int32_t* __restrict__ conn,
                                                                const int32_t* __restrict__ dofmap, double gamma, double hpow,
                                                                int qdegree, double* __restrict__ out)
{
  constexpr int ND = Elem<TDIM, 2>::ND, WF = Elem<TDIM, 2>::WF, NX = WF - ND;
  static_assert(WF + 1 <= 16 && WF <= 15, "record of 16 doubles, 16 column ids");
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t f = t / nq;
  if (f >= n) return;
  const int q = (int)(t - f * nq);
  const int4 row4 = *reinterpret_cast<const int4*>(rows + 4 * f);
  const int64_t c0 = row4.x, c1 = row4.z;
  const int lf0 = row4.y;
  Geo<TDIM> g0, g1;
  load_cell<TDIM>(x, conn, c0, g0);
  load_cell<TDIM>(x, conn, c1, g1);
  jacobian<TDIM>(g0);
  jacobian<TDIM>(g1);
  const double havg = 0.5 * (cell_diameter<TDIM>(g0) + cell_diameter<TDIM>(g1));
  double nrm[TDIM], nn = 0.0;
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < TDIM; ++k) v -= g0.K[k][d] * ((lf0 == 0) ? -1.0 : ((lf0 - 1 == k) ? 1.0 : 0.0));
    nrm[d] = v;
    nn += v * v;
  }
  nn = sqrt(nn);
#pragma unroll
  for (int d = 0; d < TDIM; ++d) nrm[d] /= nn;
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
  int nref;
  const double* wref;
  const double* pref = ref_rule(TDIM - 1, qdegree, nref, wref);
  double l0 = 1.0, xq[TDIM];
#pragma unroll
  for (int k = 0; k < TDIM - 1; ++k) l0 -= pref[q * (TDIM - 1) + k];
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    double v = l0 * xf[0][d];
#pragma unroll
    for (int k = 0; k < TDIM - 1; ++k) v += pref[q * (TDIM - 1) + k] * xf[k + 1][d];
    xq[d] = v;
  }
  double X0[TDIM], X1[TDIM], a0[TDIM], a1[TDIM];
#pragma unroll
  for (int k = 0; k < TDIM; ++k)
  {
    double u = 0.0, v = 0.0, p = 0.0, r = 0.0;
#pragma unroll
    for (int d = 0; d < TDIM; ++d)
    {
      u += g0.K[k][d] * (xq[d] - g0.x[0][d]);
      v += g1.K[k][d] * (xq[d] - g1.x[0][d]);
      p += g0.K[k][d] * nrm[d];
      r += g1.K[k][d] * nrm[d];
    }
    X0[k] = u; X1[k] = v; a0[k] = p; a1[k] = r;
  }
  double N[ND], dN[ND][TDIM], j0[ND], j1[ND];
  tabulate<TDIM, 2>(X0, N, dN);
#pragma unroll
  for (int j = 0; j < ND; ++j)
  {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < TDIM; ++k) v += a0[k] * dN[j][k];
    j0[j] = v;
  }
  tabulate<TDIM, 2>(X1, N, dN);
#pragma unroll
  for (int j = 0; j < ND; ++j)
  {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < TDIM; ++k) v += a1[k] * dN[j][k];
    j1[j] = -v;
  }
  // fold: shared dofs of cell 1 onto cell 0's, the others behind them in cell 1's order
  double jf[WF];
#pragma unroll
  for (int i = 0; i < ND; ++i) jf[i] = j0[i];
#pragma unroll
  for (int e = 0; e < NX; ++e) jf[ND + e] = 0.0;
  int nfree = 0;
#pragma unroll
  for (int j = 0; j < ND; ++j)
  {
    const int32_t dj = dofmap[c1 * ND + j];
    bool shared = false;
#pragma unroll
    for (int i = 0; i < ND; ++i)
    {
      const bool same = dj == dofmap[c0 * ND + i];
      jf[i] += same ? j1[j] : 0.0;
      shared = shared || same;
    }
#pragma unroll
    for (int e = 0; e < NX; ++e) jf[ND + e] += (!shared && nfree == e) ? j1[j] : 0.0;
    nfree += shared ? 0 : 1;
  }
  double rec[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) rec[k] = 0.0;
#pragma unroll
  for (int k = 0; k < WF; ++k) rec[k] = jf[k];
  rec[WF] = wref[q] * scale * gamma * havg * (hpow != 0.0 ? pow(havg, hpow) : 1.0);
  // facet block: 16 int32 (the WF macro columns, -1 padding, entry 15 = number of cell 1's dofs cell 0 lacks),
  // then nq records of 16 doubles
  double* blk = out + f * (int64_t)(8 + 16 * nq);
  double2* o = reinterpret_cast<double2*>(blk + 8 + 16 * q);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = make_double2(rec[2 * k], rec[2 * k + 1]);
  if (q == 0)
  {
    int32_t cm[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) cm[k] = -1;
#pragma unroll
    for (int i = 0; i < ND; ++i) cm[i] = dofmap[c0 * ND + i];
    int nf2 = 0;
#pragma unroll
    for (int j = 0; j < ND; ++j)
    {
      const int32_t dj = dofmap[c1 * ND + j];
      bool shared = false;
#pragma unroll
      for (int i = 0; i < ND; ++i) shared = shared || dj == cm[i];
#pragma unroll
      for (int e = 0; e < NX; ++e) cm[ND + e] = (!shared && nf2 == e) ? dj : cm[ND + e];
      nf2 += shared ? 0 : 1;
    }
    cm[15] = nf2;
    int4* oc = reinterpret_cast<int4*>(blk);
#pragma unroll
    for (int k = 0; k < 4; ++k) oc[k] = make_int4(cm[4 * k], cm[4 * k + 1], cm[4 * k + 2], cm[4 * k + 3]);
  }
}
// Degree-2 stiffness on cut cells, stage 1 for the row gather: the moments of the barycentric coordinates over the
// rule (m0, m1_x, m2_xy: 15 numbers in 3-D) -- p2_stiffness_row_moments() forms any row of the 10 x 10 tensor from
// them, so a rule costs 16 doubles instead of 100 and one pass over its points instead of ten.
template <int TDIM>
__global__ void __launch_bounds__(kBlock) cut_moments_kernel(int64_t nr, const int32_t* __restrict__ offsets,
                                                             const double* __restrict__ points, const double* __restrict__ weights,
                                                             double* __restrict__ out)
{
  constexpr int NV = TDIM + 1, NM = 1 + NV + NV * (NV + 1) / 2;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= nr) return;
  double m[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) m[k] = 0.0;
  const int32_t q0 = offsets[e], q1 = offsets[e + 1];
  for (int32_t q = q0; q < q1; ++q)
  {
    double lam[NV];
    lam[0] = 1.0;
#pragma unroll
    for (int t = 0; t < TDIM; ++t) { lam[t + 1] = points[(int64_t)q * TDIM + t]; lam[0] -= lam[t + 1]; }
    const double w = weights[q];
    m[0] += w;
    int idx = 1 + NV;
#pragma unroll
    for (int x = 0; x < NV; ++x)
    {
      const double wx = w * lam[x];
      m[1 + x] += wx;
#pragma unroll
      for (int y = x; y < NV; ++y) m[idx++] += wx * lam[y];
    }
  }
  static_assert(NM <= 16, "record of 16 doubles");
  double2* o = reinterpret_cast<double2*>(out + 16 * e);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = make_double2(m[2 * k], m[2 * k + 1]);
}
} // namespace

namespace cfx
{
// local tensors of all standard (parts=1) or runtime (parts=2) entities of one
// integral, written entity-major to `out` (used by the row-gather assembly)
void dump_integral(cfx_form_s* a, int integral, int parts, double* out, bool fold_facets)
{
  cfx_space_s* V = a->V;
  ZeroFlag err;
  AsmArgs A{};
  A.x = V->mesh->x.p; A.conn = V->mesh->conn.p; A.dofmap = V->dofmap.p;
  A.dump = out; A.error = err.p;
  A.dump_fold = fold_facets ? 1 : 0;
  launch_integral(a, a->integrals[integral], A, -1, 0, parts);
}
// stage 1 of the P1 gradient-jump facets for the row gather: 10 doubles per facet (facet_jump_p1_kernel)
void dump_facet_jumps_p1(cfx_form_s* a, int integral, double* out, int* error)
{
  cfx_space_s* V = a->V;
  const cfx_integral_dev& I = a->integrals[integral];
  if (I.n_entities.cap() == 0) return;
  if (V->mesh->tdim == 2)
    launch("assemble_facets", facet_jump_p1_kernel<2>, grid_for(I.n_entities.cap()), dim3(kBlock), 0, I.n_entities, I.entities.p,
           V->mesh->x.p, V->mesh->conn.p, V->dofmap.p, I.params[0], I.params[1], I.qdegree, out, error);
  else
    launch("assemble_facets", facet_jump_p1_kernel<3>, grid_for(I.n_entities.cap()), dim3(kBlock), 0, I.n_entities, I.entities.p,
           V->mesh->x.p, V->mesh->conn.p, V->dofmap.p, I.params[0], I.params[1], I.qdegree, out, error);
}
// stage 1 of a degree-2 stiffness integral over runtime rules: 16 doubles per rule (cut_moments_kernel)
void dump_cut_moments(cfx_form_s* a, int integral, double* out)
{
  const cfx_integral_dev& I = a->integrals[integral];
  if (!I.rules || I.rules->nr.value() == 0) return;
  const int64_t nrl = I.rules->nr.value();
  if (a->V->mesh->tdim == 2)
    launch("assemble_cells_cut", cut_moments_kernel<2>, grid_for(nrl), dim3(kBlock), 0, nrl,
           I.rules->offsets.p, I.rules->points.p, I.rules->weights.p, out);
  else
    launch("assemble_cells_cut", cut_moments_kernel<3>, grid_for(nrl), dim3(kBlock), 0, nrl,
           I.rules->offsets.p, I.rules->points.p, I.rules->weights.p, out);
}

// stage 1 of the degree-2 gradient-jump facets for the row gather: nq x 16 doubles per facet (facet_jumps_p2_kernel)
void dump_facet_jumps_p2(cfx_form_s* a, int integral, int nq, double* out)
{
  cfx_space_s* V = a->V;
  const cfx_integral_dev& I = a->integrals[integral];
  const int64_t nf = I.n_entities.value();
  if (nf == 0) return;
  if (V->mesh->tdim == 2)
    launch("assemble_facets", facet_jumps_p2_kernel<2>, grid_for(nf * nq), dim3(kBlock), 0, nf, nq,
           I.entities.p, V->mesh->x.p, V->mesh->conn.p, V->dofmap.p, I.params[0], I.params[1], I.qdegree, out);
  else
    launch("assemble_facets", facet_jumps_p2_kernel<3>, grid_for(nf * nq), dim3(kBlock), 0, nf, nq,
           I.entities.p, V->mesh->x.p, V->mesh->conn.p, V->dofmap.p, I.params[0], I.params[1], I.qdegree, out);
}
} // namespace cfx

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
  V->dofmap = to_device_aligned(dofmap, mesh->ncells * (int64_t)ndofs_cell);
  *out = V.release();
  CFX_API_END
}

int cfx_space_static_bytes(cfx_space_t V, int64_t bytes[4])
{
  CFX_API_BEGIN
  require(V && bytes, CFX_ERR_INVALID_ARGUMENT, "cfx_space_static_bytes: null argument");
  const bool shared = V->dofmap.p == V->mesh->conn.p && V->ndofs == V->mesh->nnodes;
  const cfx::Adjacency& adj = shared ? V->mesh->v2c : V->d2c;
  bytes[0] = adj.built ? 8 * adj.offsets.n + 4 * adj.cells.n : 0;
  const cfx::Stencil& S = V->stencil;
  bytes[1] = 8 * S.offsets.n + 4 * S.nbr.n + 4 * S.slot4.n + S.diagpos.n + S.cpos.n + S.slotn.n; // (slotn: degree-2 slot records)
  const cfx::VecBlocks& B = V->vblocks;
  bytes[2] = 8 * S.tile_voff.n + 4 * S.tile_verts.n + 2 * S.st_loc.n
             + 8 * B.u_off.n + 2 * B.slot.n + 2 * B.seg.n + 8 * B.p_off.n + 8 * B.p_pos.n; // (+ the cell blocks of the linear forms)
  bytes[3] = (V->mesh->c2c_built ? 4 * V->mesh->c2c.n : 0)
             + (V->mesh->class_built ? 4 * V->mesh->class_nruns.n + 8 * (V->mesh->class_runs.n + V->mesh->class_sub_runs.n) : 0); // (+ the vertex runs of the culled classification)
  CFX_API_END
}

int cfx_space_destroy(cfx_space_t V)
{
  CFX_API_BEGIN
  delete V;
  CFX_API_END
}

static int form_create_impl(cfx_space_t V, cfx_space_t V1, int rank, int n_integrals, const cfx_integral* integrals,
                            cfx_form_t* out);

int cfx_form_create(cfx_space_t V, int rank, int n_integrals, const cfx_integral* integrals, cfx_form_t* out)
{
  return form_create_impl(V, V, rank, n_integrals, integrals, out);
}

int cfx_form_create2(cfx_space_t V_test, cfx_space_t V_trial, int n_integrals, const cfx_integral* integrals, cfx_form_t* out)
{
  CFX_API_BEGIN
  require(V_test && V_trial, CFX_ERR_INVALID_ARGUMENT, "cfx_form_create2: null argument");
  require(V_test->mesh == V_trial->mesh, CFX_ERR_INVALID_ARGUMENT, "cfx_form_create2: the two spaces live on different meshes");
  CFX_API_END_NO_RETURN
  return form_create_impl(V_test, V_trial, 2, n_integrals, integrals, out);
}

static int form_create_impl(cfx_space_t V, cfx_space_t V1, int rank, int n_integrals, const cfx_integral* integrals,
                            cfx_form_t* out)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(V && out && (integrals || n_integrals == 0), CFX_ERR_INVALID_ARGUMENT, "cfx_form_create: null argument");
  require(rank == 1 || rank == 2, CFX_ERR_INVALID_ARGUMENT, "cfx_form_create: rank must be 1 or 2");
  auto a = std::make_unique<cfx_form_s>();
  a->V = V; a->V1 = V1; a->rank = rank;
  const bool rect = V1 != V;
  for (int i = 0; i < n_integrals; ++i)
  {
    const cfx_integral& in = integrals[i];
    cfx_integral_dev I;
    I.type = in.type; I.kernel = in.kernel; I.qdegree = in.qdegree; I.point_stride = in.point_stride;
    require(in.type == CFX_CELL || in.type == CFX_INTERIOR_FACET, CFX_ERR_INVALID_ARGUMENT,
            "cfx_form_create: integral type must be cell or interior_facet");
    const bool user = user_integrand_known(in.kernel);
    const bool bilinear = user ? user_integrand_rank(in.kernel) == 2 : in.kernel < 100;
    require(bilinear == (rank == 2), CFX_ERR_INVALID_ARGUMENT, "cfx_form_create: kernel rank does not match the form");
    require(in.qdegree >= 0 && in.qdegree <= CFX_QUAD_MAX_DEGREE, CFX_ERR_INVALID_ARGUMENT,
            "cfx_form_create: quadrature degree out of range");
    if (in.type == CFX_INTERIOR_FACET && user)
      require(user_integrand_kind(in.kernel) == 1 && !rect && V->degree <= 2 && in.rules == nullptr && 2 * V->ndofs_cell * V->bs <= 24,
              CFX_ERR_INVALID_ARGUMENT,
              "cfx_form_create: an interior-facet integral takes an integrand registered with cfx_integrand_register_facet "
              "(standard facets, spaces of degree 1 or 2 with at most 24 macro dofs)");
    else if (in.type == CFX_INTERIOR_FACET)
    {
      require(in.kernel == CFX_K_GHOST_GRADJUMP || in.kernel == CFX_K_EXTENSION_L2 || in.kernel == CFX_K_JUMP
                  || in.kernel == CFX_K_SIP,
              CFX_ERR_INVALID_ARGUMENT,
              "cfx_form_create: interior-facet integrals support the ghost-penalty, value-jump, interior-penalty and "
              "extension-penalty kernels");
      if (in.rules)
        require(in.kernel != CFX_K_EXTENSION_L2 && in.rules->host_width == 4 && in.point_data == nullptr,
                CFX_ERR_INVALID_ARGUMENT,
                "cfx_form_create: runtime rules of an interior-facet integral are facet-hosted rules over interior rows "
                "(cfx_cut_create_facets with row_width 4)");
    }
    else if (in.rules && in.rules->host_width != 0)
      throw Error(CFX_ERR_INVALID_ARGUMENT,
                  "cfx_form_create: a cell integral takes cell-hosted rules (pass facet-hosted rules through "
                  "cfx_facet_rules_to_cells)");
    else if (user)
      require(in.type == CFX_CELL && !rect && V->degree <= 2 && user_integrand_kind(in.kernel) == 0, CFX_ERR_INVALID_ARGUMENT,
              "cfx_form_create: a cell integral takes an integrand registered with cfx_integrand_register (spaces of degree 1 or 2)");
    else
      require(in.kernel == CFX_K_MASS || in.kernel == CFX_K_STIFFNESS || in.kernel == CFX_K_NITSCHE
                  || in.kernel == CFX_K_ELASTICITY || in.kernel == CFX_L_SOURCE || in.kernel == CFX_L_NITSCHE_RHS
                  || (rect && (in.kernel == CFX_K_DIV_TEST || in.kernel == CFX_K_DIV_TRIAL)),
              CFX_ERR_INVALID_ARGUMENT,
              "cfx_form_create: unknown cell kernel id (the divergence blocks need different test and trial spaces: "
              "cfx_form_create2)");
    if (rect)
    {
      // test space != trial space: cell integrals of the kernels assemble_cells2_kernel knows, shapes that match
      require(in.coefficient == nullptr && in.point_data == nullptr, CFX_ERR_INVALID_ARGUMENT,
              "cfx_form_create2: coefficients / per-point data are not supported on rectangular blocks");
      const int gd = V->mesh->gdim;
      if (in.type == CFX_INTERIOR_FACET)
        // interior-facet terms between two spaces (assemble_matrix_impl.h:462-606 with dofmap0 != dofmap1): the
        // gradient-jump and value-jump kernels over standard facets, scalar spaces of degree 1 or 2
        require((in.kernel == CFX_K_GHOST_GRADJUMP || in.kernel == CFX_K_JUMP) && V->bs == 1 && V1->bs == 1 && in.rules == nullptr,
                CFX_ERR_INVALID_ARGUMENT,
                "cfx_form_create2: cell integrals, and interior-facet integrals (gradient jump, value jump) between "
                "scalar spaces over standard facets");
      else if (in.kernel == CFX_K_DIV_TEST)
        require(V->bs == gd && V1->bs == 1, CFX_ERR_INVALID_ARGUMENT, "div(v) p: vector test space, scalar trial space");
      else if (in.kernel == CFX_K_DIV_TRIAL)
        require(V->bs == 1 && V1->bs == gd, CFX_ERR_INVALID_ARGUMENT, "q div(u): scalar test space, vector trial space");
      else
        require((in.kernel == CFX_K_MASS || in.kernel == CFX_K_STIFFNESS) && V->bs == V1->bs, CFX_ERR_INVALID_ARGUMENT,
                "cfx_form_create2: mass / stiffness blocks need equal block sizes; other kernels are square-only");
    }
    if (in.kernel == CFX_K_NITSCHE || in.kernel == CFX_L_NITSCHE_RHS)
    {
      require(V->bs == 1, CFX_ERR_INVALID_ARGUMENT, "Nitsche kernels are scalar");
      require(in.n_entities == 0 && in.rules && in.point_data && in.point_stride >= V->mesh->gdim,
              CFX_ERR_INVALID_ARGUMENT,
              "Nitsche kernels need runtime interface rules and per-point normals (point_data)");
    }
    if (in.kernel == CFX_K_ELASTICITY)
      require(V->bs == V->mesh->gdim, CFX_ERR_INVALID_ARGUMENT, "elasticity needs a vector space (bs == gdim)");
    if (in.kernel == CFX_L_SOURCE)
      require(V->bs == 1 || (int)in.params[0] == CFX_F_COEFFICIENT, CFX_ERR_INVALID_ARGUMENT,
              "the source term of a vector space takes a vector-valued Function (field id CFX_F_COEFFICIENT)");
    if (bilinear)
    {
      // a scalar coefficient of the form's element multiplies the integrand (kappa grad u . grad v, rho u v, the
      // density-weighted elasticity of python/demo/demo_compliance_optimization.py)
      require(in.coefficient == nullptr || in.kernel == CFX_K_MASS || in.kernel == CFX_K_STIFFNESS
                  || in.kernel == CFX_K_ELASTICITY,
              CFX_ERR_INVALID_ARGUMENT, "cfx_form_create: a coefficient is accepted by the mass, stiffness and elasticity terms");
      if (in.coefficient) I.coefficient = to_device(in.coefficient, V->ndofs);
    }
    else
    {
      const bool wants = (in.kernel == CFX_L_SOURCE && (int)in.params[0] == CFX_F_COEFFICIENT)
                         || (in.kernel == CFX_L_NITSCHE_RHS && (int)in.params[1] == CFX_F_COEFFICIENT);
      require(!(in.kernel == CFX_L_NITSCHE_RHS && wants), CFX_ERR_INVALID_ARGUMENT,
              "the Nitsche datum takes an analytic field id");
      require(wants == (in.coefficient != nullptr), CFX_ERR_INVALID_ARGUMENT,
              "cfx_form_create: `coefficient` goes with the field id CFX_F_COEFFICIENT (and only with it)");
      if (wants) I.coefficient = to_device(in.coefficient, V->ndofs * V->bs); // vector spaces: bs values per dof
    }
    // (a library list whose length is still in HBM comes back by address with the capacity the ABI returned)
    I.n_entities = list_lookup(in.entities, in.n_entities);
    const int64_t width = in.type == CFX_INTERIOR_FACET ? 4 : 1;
    I.entities = to_device(in.entities, in.n_entities * width);
    I.rules = in.rules;
    I.n_std = -1; // all entities are standard ones (see below: facet-hosted rules append theirs)
    if (in.rules && in.type == CFX_INTERIOR_FACET)
    {
      // one entity list for the pattern / row plan / staging: [standard rows, the rules' rows]
      require(in.rules->mesh == V->mesh, CFX_ERR_INVALID_ARGUMENT, "rules belong to a different mesh");
      const int64_t nr = in.rules->nr.value();
      const int64_t n_in = I.n_entities.value(); // (a concatenated list: exact lengths)
      require(n_in == in.n_entities || !I.n_entities.cell, CFX_ERR_INVALID_ARGUMENT,
              "cfx_form_create: end the step before mixing a pending facet list with facet-hosted rules");
      DevArray<int32_t> all((in.n_entities + nr) * 4);
      if (in.n_entities > 0)
        CFX_HIP(hipMemcpyAsync(all.p, I.entities.p, sizeof(int32_t) * 4 * (size_t)in.n_entities, hipMemcpyDeviceToDevice,
                               ctx().stream));
      if (nr > 0)
        CFX_HIP(hipMemcpyAsync(all.p + 4 * in.n_entities, in.rules->host_rows.p, sizeof(int32_t) * 4 * (size_t)nr,
                               hipMemcpyDeviceToDevice, ctx().stream));
      CFX_HIP(hipStreamSynchronize(ctx().stream));
      I.entities = std::move(all);
      I.n_entities = in.n_entities + nr;
      I.n_std = in.n_entities;
    }
    else if (in.rules)
    {
      require(in.rules->mesh == V->mesh, CFX_ERR_INVALID_ARGUMENT, "rules belong to a different mesh");
      if (in.point_data) I.point_data = to_device(in.point_data, in.rules->nq.cap() * (int64_t)in.point_stride);
    }
    else if (in.kernel == CFX_K_EXTENSION_L2 && in.point_data)
      I.point_data = to_device(in.point_data, in.n_entities); // one factor per pair (cellwise beta)
    for (int k = 0; k < 8; ++k) I.params[k] = in.params[k];
    I.entities_serial = I.n_entities.cap() > 0 ? dev_block_serial(I.entities.p) : 0;
    I.rules_serial = I.rules ? I.rules->serial : 0;
    a->integrals.push_back(std::move(I));
  }
  *out = a.release();
  CFX_API_END
}

int cfx_form_prepare(cfx_form_t a)
{
  CFX_API_BEGIN
  require(a != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_form_prepare: null argument");
  validate_form(a);
  if (!force_atomic() && !a->rectangular()) prepare_form_tables(a);
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
  validate_form(a);
  auto P = std::make_unique<cfx_pattern_s>();
  build_pattern(a, P.get());
  end_of_call_sync();
  *out = P.release();
  CFX_API_END
}

int cfx_pattern_view_get(cfx_pattern_t p, cfx_pattern_view* v)
{
  CFX_API_BEGIN
  require(p && v, CFX_ERR_INVALID_ARGUMENT, "cfx_pattern_view_get: null argument");
  v->nrows = p->nrows; v->nnz = p->nnz.cap(); v->indptr = p->indptr.p; v->indices = p->indices.p; v->ncols = p->ncols;
  CFX_API_END
}

int cfx_pattern_reuse_stats(cfx_pattern_t p, int64_t* hashed_rows, int64_t* reused_rows)
{
  CFX_API_BEGIN
  require(p != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_pattern_reuse_stats: null handle");
  if (hashed_rows) *hashed_rows = p->n_hashed_rows;
  if (reused_rows) *reused_rows = p->n_reused_rows;
  CFX_API_END
}

int cfx_pattern_destroy(cfx_pattern_t p)
{
  CFX_API_BEGIN
  delete p;
  CFX_API_END
}

// assemble_matrix, optionally preceded by la::MatrixCSR::set_value(0) in the same call (`zero_first`): the
// zeroing is then the library's to schedule, and rows with a single writer are stored instead of accumulated
static void assemble_matrix_impl(cfx_form_t a, cfx_pattern_t P, const int8_t* bc0, const int8_t* bc1, double* values,
                                 bool zero_first)
{
  require(a && P && values, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_matrix: null argument");
  require(a->rank == 2, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_matrix: form is not bilinear");
  validate_form(a);
  cfx_space_s* V = a->V;
  require(P->nrows == V->ndofs * V->bs && P->ncols == a->V1->ndofs * a->V1->bs, CFX_ERR_INVALID_ARGUMENT,
          "cfx_assemble_matrix: pattern/space size mismatch");
  DevArray<int8_t> dbc0 = to_device(bc0, bc0 ? P->nrows : 0), dbc1 = to_device(bc1, bc1 ? P->ncols : 0);
  OutArray<double> out(values, count_for_buffer(P->nnz, values), !zero_first);
  const char* kMissing = "assemble_matrix: entry not in the sparsity pattern";
  ErrorFlag err(CFX_ERR_RUNTIME, kMissing);
  AsmArgs A{};
  A.x = V->mesh->x.p; A.conn = V->mesh->conn.p; A.dofmap = V->dofmap.p;
  A.bc0 = bc0 ? dbc0.p : nullptr; A.bc1 = bc1 ? dbc1.p : nullptr;
  A.indptr = P->indptr.p; A.indices = P->indices.p; A.values = out.dev; A.dump = nullptr; A.error = err.p;
  // row-gather path (deterministic, no global atomics) when the form allows it (with zero_first it does the zeroing
  // itself: rows with a single writer are stored, only the rest is filled); otherwise the whole fill and the
  // entity-parallel kernels with FP64 atomics
  if (a->rectangular())
  {
    // test space != trial space: the row gather of the rectangular blocks (round 4; one writer per row, no global
    // atomics), or the entity-parallel kernel (CFX_ASSEMBLY=atomic, rows beyond 256 columns)
    if (zero_first) dev_fill(out.dev, 0, sizeof(double) * (size_t)P->nnz.value());
    if (force_atomic() || !assemble_rect_rows(a, P, A.bc0, A.bc1, out.dev, A.error))
      for (const auto& I : a->integrals) launch_rectangular(a, I, A);
    err.check(CFX_ERR_RUNTIME, kMissing);
    out.finish();
    return;
  }
  if (!force_atomic() && assemble_matrix_rows(a, P, A.bc0, A.bc1, out.dev, zero_first)) { out.finish(); return; }
  // (entity-parallel kernels: a pattern whose nnz is still in HBM is filled up to its capacity)
  if (zero_first) dev_fill(out.dev, 0, sizeof(double) * (size_t)P->nnz.cap());
  for (const auto& I : a->integrals) launch_integral(a, I, A);
  err.check(CFX_ERR_RUNTIME, kMissing);
  out.finish();
}

int cfx_assemble_matrix(cfx_form_t a, cfx_pattern_t P, const int8_t* bc0, const int8_t* bc1, double* values)
{
  CFX_API_BEGIN
  assemble_matrix_impl(a, P, bc0, bc1, values, false);
  CFX_API_END
}

int cfx_assemble_matrix_zeroed(cfx_form_t a, cfx_pattern_t P, const int8_t* bc0, const int8_t* bc1, double* values)
{
  CFX_API_BEGIN
  assemble_matrix_impl(a, P, bc0, bc1, values, true);
  CFX_API_END
}

int cfx_assemble_vector(cfx_form_t L, double* b)
{
  CFX_API_BEGIN
  require(L && b, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_vector: null argument");
  require(L->rank == 1, CFX_ERR_INVALID_ARGUMENT, "cfx_assemble_vector: form is not linear");
  validate_form(L);
  cfx_space_s* V = L->V;
  OutArray<double> out(b, V->ndofs * V->bs, true);
  ZeroFlag err;
  AsmArgs A{};
  A.x = V->mesh->x.p; A.conn = V->mesh->conn.p; A.dofmap = V->dofmap.p;
  A.values = out.dev; A.error = err.p;
  if (force_atomic() || !assemble_vector_rows(L, out.dev))
    for (const auto& I : L->integrals) launch_integral(L, I, A);
  out.finish();
  if (out.dev == b) { /* device output: leave the stream running */ }
  CFX_API_END
}

int cfx_apply_lifting(cfx_form_t a, const int8_t* bc_markers, const double* bc_values, const double* x0, double alpha,
                      double* b)
{
  CFX_API_BEGIN
  require(a && bc_markers && bc_values && b, CFX_ERR_INVALID_ARGUMENT, "cfx_apply_lifting: null argument");
  require(a->rank == 2, CFX_ERR_INVALID_ARGUMENT, "cfx_apply_lifting: form is not bilinear");
  validate_form(a);
  cfx_space_s* V = a->V;
  const int64_t n = V->ndofs * V->bs;               // b lives on the test space,
  const int64_t n1 = a->V1->ndofs * a->V1->bs;      // the Dirichlet data on the trial space (lift_bc_impl: columns)
  DevArray<int8_t> dm = to_device(bc_markers, n1);
  DevArray<double> dv = to_device(bc_values, n1), dx0 = to_device(x0, x0 ? n1 : 0);
  OutArray<double> out(b, n, true);
  ZeroFlag err;
  AsmArgs A{};
  A.x = V->mesh->x.p; A.conn = V->mesh->conn.p; A.dofmap = V->dofmap.p;
  A.values = out.dev; A.error = err.p;
  A.lift_markers = dm.p; A.lift_values = dv.p; A.lift_x0 = x0 ? dx0.p : nullptr; A.lift_alpha = alpha;
  for (const auto& I : a->integrals)
  {
    if (a->rectangular()) launch_rectangular(a, I, A); else launch_integral(a, I, A);
  }
  out.finish();
  CFX_API_END
}

namespace
{
__global__ void set_bc_kernel(int64_t n, const int8_t* __restrict__ markers, const double* __restrict__ g,
                              const double* __restrict__ x0, double alpha, double* __restrict__ b)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && markers[i]) b[i] = alpha * (g[i] - (x0 ? x0[i] : 0.0));
}
} // namespace

int cfx_set_bc(int64_t n, const int8_t* bc_markers, const double* bc_values, const double* x0, double alpha, double* b)
{
  CFX_API_BEGIN
  require(n >= 0 && bc_markers && bc_values && b, CFX_ERR_INVALID_ARGUMENT, "cfx_set_bc: null argument");
  ctx().ensure();
  DevArray<int8_t> dm = to_device(bc_markers, n);
  DevArray<double> dv = to_device(bc_values, n), dx0 = to_device(x0, x0 ? n : 0);
  OutArray<double> out(b, n, true);
  launch("set_bc", set_bc_kernel, grid_for(n), dim3(kBlock), 0, n, dm.p, dv.p, x0 ? dx0.p : nullptr, alpha, out.dev);
  out.finish();
  CFX_API_END
}

namespace
{
struct RowAllZero
{
  const int64_t* indptr;
  const double* values;
  double tol;
  __device__ bool operator()(int64_t r) const
  {
    for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k)
      if (fabs(values[k]) > tol) return false;
    return true;
  }
};
} // namespace

int cfx_zero_rows(cfx_pattern_t P, const double* values, double tol, int32_t** rows, int64_t* n_rows)
{
  CFX_API_BEGIN
  require(P && values && rows && n_rows, CFX_ERR_INVALID_ARGUMENT, "cfx_zero_rows: null argument");
  DevArray<double> dv = to_device(values, P->nnz.value());
  DevArray<int32_t> list;
  const int64_t n = compact("zero_rows", P->nrows, RowAllZero{P->indptr.p, dv.p, tol}, list);
  int32_t* out = static_cast<int32_t*>(dev_alloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1)));
  if (n > 0)
    CFX_HIP(hipMemcpyAsync(out, list.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, ctx().stream));
  *rows = out;
  *n_rows = n;
  CFX_API_END
}

namespace
{
// ---- a CSR matrix in another numbering (cfx_csr_permute): row r -> row_perm[r], column c -> col_perm[c] ----
__global__ void permute_check_kernel(int64_t n, const int32_t* __restrict__ perm, int32_t* hits, int* bad)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t t = perm[i];
  if (t < 0 || t >= n) { *bad = 1; return; }
  atomicAdd(&hits[t], 1);
}
__global__ void permute_verify_kernel(int64_t n, const int32_t* __restrict__ hits, int* bad)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && hits[i] != 1) *bad = 1;
}
__global__ void permute_len_kernel(int64_t nrows, const int64_t* __restrict__ indptr, const int32_t* __restrict__ row_perm,
                                   int64_t* __restrict__ len)
{
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < nrows) len[row_perm[r]] = indptr[r + 1] - indptr[r];
}
// one wavefront per row: the row's (mapped column, value) pairs are sorted by column in LDS (bitonic over the next power
// of two, padded with INT_MAX) and written at the row's new place
constexpr int kPermuteCap = 2048;
__global__ void __launch_bounds__(64) permute_rows_kernel(int64_t nrows, const int64_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ indices, const double* __restrict__ values,
                                                          const int32_t* __restrict__ row_perm, const int32_t* __restrict__ col_perm,
                                                          const int64_t* __restrict__ out_indptr, int32_t* __restrict__ out_indices,
                                                          double* __restrict__ out_values, int* error)
{
  __shared__ int32_t s_key[kPermuteCap];
  __shared__ double s_val[kPermuteCap];
  const int lane = threadIdx.x;
  for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x)
  {
    const int64_t b = indptr[r];
    const int len = (int)(indptr[r + 1] - b);
    if (len > kPermuteCap) { *error = 1; continue; }
    int n2 = 1;
    while (n2 < len) n2 <<= 1;
    for (int k = lane; k < n2; k += 64)
    {
      s_key[k] = k < len ? col_perm[indices[b + k]] : 0x7fffffff;
      s_val[k] = (k < len && values) ? values[b + k] : 0.0;
    }
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1)
      {
        for (int i = lane; i < n2; i += 64)
        {
          const int p = i ^ j;
          if (p > i)
          {
            const int32_t a = s_key[i], c = s_key[p];
            const bool up = (i & k) == 0;
            if ((a > c) == up)
            {
              s_key[i] = c; s_key[p] = a;
              const double t = s_val[i]; s_val[i] = s_val[p]; s_val[p] = t;
            }
          }
        }
        __syncthreads();
      }
    const int64_t o = out_indptr[row_perm[r]];
    for (int k = lane; k < len; k += 64)
    {
      out_indices[o + k] = s_key[k];
      if (out_values) out_values[o + k] = s_val[k];
    }
    __syncthreads(); // the LDS image is reused by the next row
  }
}

constexpr int kMergeMaxBlocks = 8; // blocks per block row
struct MergeCols { int64_t n[8]; };
struct MergeRow
{
  int nb;                              // blocks of this block row (empty ones have indptr == nullptr)
  const int64_t* indptr[kMergeMaxBlocks];
  const int32_t* indices[kMergeMaxBlocks];
  const double* values[kMergeMaxBlocks];
  int64_t col_offset[kMergeMaxBlocks];
};

// entries of row r of the block row: the sum of the blocks' row lengths
// (the merged row is sorted only if every block's row is: checked here, `bad` is raised by the caller; columns must also
// lie inside their block column)
__global__ void block_merge_len_kernel(int64_t nrows, MergeRow B, int64_t* __restrict__ len, MergeCols ncols, int* bad)
{
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  int64_t l = 0;
  for (int j = 0; j < B.nb; ++j)
    if (B.indptr[j])
    {
      const int64_t b = B.indptr[j][r], e = B.indptr[j][r + 1];
      l += e - b;
      for (int64_t k = b; k < e; ++k)
      {
        const int32_t c = B.indices[j][k];
        if (c < 0 || c >= ncols.n[j] || (k > b && B.indices[j][k - 1] >= c)) *bad = 1;
      }
    }
  len[r] = l;
}

// 8 lanes per row: block after block, columns shifted by the block column's offset (each block's columns ascend and
// the offsets ascend with the block column: the merged row is sorted)
__global__ void block_merge_fill_kernel(int64_t nrows, MergeRow B, const int64_t* __restrict__ out_indptr,
                                        int32_t* __restrict__ out_indices, double* __restrict__ out_values)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = t >> 3;
  const int lane = (int)(t & 7);
  if (r >= nrows) return;
  int64_t o = out_indptr[r];
  for (int j = 0; j < B.nb; ++j)
  {
    if (!B.indptr[j]) continue;
    const int64_t b = B.indptr[j][r], e = B.indptr[j][r + 1];
    for (int64_t k = b + lane; k < e; k += 8)
    {
      out_indices[o + (k - b)] = (int32_t)(B.indices[j][k] + B.col_offset[j]);
      if (out_values) out_values[o + (k - b)] = B.values[j] ? B.values[j][k] : 0.0;
    }
    o += e - b;
  }
}
} // namespace

int cfx_csr_block_merge(int nbr, int nbc, const int64_t* const* indptr, const int32_t* const* indices,
                        const double* const* values, const int64_t* nrows, const int64_t* ncols, int64_t** out_indptr,
                        int32_t** out_indices, double** out_values, int64_t* out_nnz)
{
  CFX_API_BEGIN
  require(indptr && indices && nrows && ncols && out_indptr && out_indices && out_nnz, CFX_ERR_INVALID_ARGUMENT,
          "cfx_csr_block_merge: null argument");
  require(nbr >= 1 && nbc >= 1 && nbc <= kMergeMaxBlocks, CFX_ERR_INVALID_ARGUMENT,
          "cfx_csr_block_merge: 1 .. 8 block columns");
  int64_t total_rows = 0, total_cols = 0;
  for (int i = 0; i < nbr; ++i) { require(nrows[i] >= 0, CFX_ERR_INVALID_ARGUMENT, "cfx_csr_block_merge: negative row count"); total_rows += nrows[i]; }
  for (int j = 0; j < nbc; ++j) { require(ncols[j] >= 0, CFX_ERR_INVALID_ARGUMENT, "cfx_csr_block_merge: negative column count"); total_cols += ncols[j]; }
  require(total_cols <= 2147483647LL, CFX_ERR_OUT_OF_RANGE, "cfx_csr_block_merge: more than 2^31 - 1 columns");
  // the blocks on the device (host arrays are uploaded; device arrays are used where they lie)
  std::vector<DevArray<int64_t>> d_ip((size_t)nbr * nbc);
  std::vector<DevArray<int32_t>> d_ix((size_t)nbr * nbc);
  std::vector<DevArray<double>> d_v((size_t)nbr * nbc);
  std::vector<MergeRow> rows((size_t)nbr);
  for (int i = 0; i < nbr; ++i)
  {
    MergeRow& B = rows[i];
    B = MergeRow{};
    B.nb = nbc;
    int64_t co = 0;
    for (int j = 0; j < nbc; ++j)
    {
      const size_t k = (size_t)i * nbc + j;
      B.col_offset[j] = co;
      co += ncols[j];
      if (!indptr[k] || nrows[i] == 0) continue;
      require(indices[k] != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_csr_block_merge: a block has row offsets but no column indices");
      d_ip[k] = to_device(indptr[k], nrows[i] + 1);
      const int64_t nnz_k = read_scalar(d_ip[k].p + nrows[i]);
      d_ix[k] = to_device(indices[k], nnz_k);
      B.indptr[j] = d_ip[k].p; B.indices[j] = d_ix[k].p;
      if (values && values[k]) { d_v[k] = to_device(values[k], nnz_k); B.values[j] = d_v[k].p; }
    }
  }
  // (outputs are owned here until the call has succeeded: a later throw frees them)
  DevArray<int64_t> ip_out(total_rows + 1);
  int64_t* ip = ip_out.p;
  DevArray<int64_t> len(total_rows);
  ZeroFlag unsorted;
  MergeCols mcols{};
  for (int j = 0; j < nbc; ++j) mcols.n[j] = ncols[j];
  int64_t r0 = 0;
  for (int i = 0; i < nbr; ++i)
  {
    if (nrows[i] > 0)
      launch("block_merge", block_merge_len_kernel, grid_for(nrows[i]), dim3(kBlock), 0, nrows[i], rows[i], len.p + r0, mcols, unsorted.p);
    r0 += nrows[i];
  }
  exclusive_scan(len.p, ip, total_rows);
  const int64_t nnz = read_scalar(ip + total_rows);
  require(!read_scalar(unsorted.p), CFX_ERR_INVALID_ARGUMENT,
          "cfx_csr_block_merge: a block has a row whose columns do not ascend strictly inside its block column");
  DevArray<int32_t> ix_out(nnz);
  DevArray<double> vals_out;
  if (out_values) vals_out.alloc(nnz);
  int32_t* ix = ix_out.p;
  double* vals = vals_out.p;
  r0 = 0;
  for (int i = 0; i < nbr; ++i)
  {
    if (nrows[i] > 0)
      launch("block_merge", block_merge_fill_kernel, grid_for(nrows[i] * 8), dim3(kBlock), 0, nrows[i], rows[i], ip + r0, ix,
             vals);
    r0 += nrows[i];
  }
  // the fill kernels are done before the caller sees the arrays (torch may read them on another stream)
  end_of_call_sync();
  ip_out.owned = false; ix_out.owned = false; vals_out.owned = false;   // handed to the caller (cfx_device_free)
  *out_indptr = ip;
  *out_indices = ix;
  if (out_values) *out_values = vals;
  *out_nnz = nnz;
  CFX_API_END
}

int cfx_csr_permute(int64_t nrows, int64_t ncols, const int64_t* indptr, const int32_t* indices, const double* values,
                    const int32_t* row_perm, const int32_t* col_perm, int64_t** out_indptr, int32_t** out_indices,
                    double** out_values)
{
  CFX_API_BEGIN
  require(indptr && indices && row_perm && col_perm && out_indptr && out_indices && nrows >= 0 && ncols >= 0,
          CFX_ERR_INVALID_ARGUMENT, "cfx_csr_permute: null argument");
  require(nrows <= 2147483647LL && ncols <= 2147483647LL, CFX_ERR_OUT_OF_RANGE, "cfx_csr_permute: more than 2^31 - 1 rows or columns");
  DevArray<int64_t> d_ip = to_device(indptr, nrows + 1);
  const int64_t nnz = nrows > 0 ? read_scalar(d_ip.p + nrows) : 0;
  DevArray<int32_t> d_ix = to_device(indices, nnz);
  DevArray<double> d_v;
  if (values) d_v = to_device(values, nnz);
  DevArray<int32_t> d_rp = to_device(row_perm, nrows), d_cp = to_device(col_perm, ncols);
  // both maps must be permutations: every target hit exactly once
  {
    ZeroFlag bad;
    DevArray<int32_t> hits(std::max(nrows, ncols));
    hits.zero();
    launch("csr_permute", permute_check_kernel, grid_for(nrows), dim3(kBlock), 0, nrows, d_rp.p, hits.p, bad.p);
    launch("csr_permute", permute_verify_kernel, grid_for(nrows), dim3(kBlock), 0, nrows, hits.p, bad.p);
    hits.zero();
    launch("csr_permute", permute_check_kernel, grid_for(ncols), dim3(kBlock), 0, ncols, d_cp.p, hits.p, bad.p);
    launch("csr_permute", permute_verify_kernel, grid_for(ncols), dim3(kBlock), 0, ncols, hits.p, bad.p);
    require(!read_scalar(bad.p), CFX_ERR_INVALID_ARGUMENT, "cfx_csr_permute: row_perm / col_perm is not a permutation");
  }
  DevArray<int64_t> ip_out(nrows + 1);
  DevArray<int64_t> len(nrows);
  launch("csr_permute", permute_len_kernel, grid_for(nrows), dim3(kBlock), 0, nrows, d_ip.p, d_rp.p, len.p);
  exclusive_scan(len.p, ip_out.p, nrows);
  DevArray<int32_t> ix_out(nnz);
  DevArray<double> v_out;
  if (out_values) v_out.alloc(nnz);
  ErrorFlag err(CFX_ERR_RUNTIME, "cfx_csr_permute: a row holds more than 2048 entries");
  if (nrows > 0)
    launch("csr_permute", permute_rows_kernel, wave_grid(nrows), dim3(64), 0, nrows, d_ip.p, d_ix.p, values ? d_v.p : (const double*)nullptr,
           d_rp.p, d_cp.p, ip_out.p, ix_out.p, out_values ? v_out.p : (double*)nullptr, err.p);
  err.check(CFX_ERR_RUNTIME, "cfx_csr_permute: a row holds more than 2048 entries");
  end_of_call_sync();
  ip_out.owned = false; ix_out.owned = false; v_out.owned = false; // handed to the caller (cfx_device_free)
  *out_indptr = ip_out.p;
  *out_indices = ix_out.p;
  if (out_values) *out_values = v_out.p;
  CFX_API_END
}

int cfx_tabulate_entity(cfx_form_t a, int integral, int64_t index, int use_rule, double* Ae)
{
  CFX_API_BEGIN
  require(a && Ae, CFX_ERR_INVALID_ARGUMENT, "cfx_tabulate_entity: null argument");
  require(integral >= 0 && integral < (int)a->integrals.size(), CFX_ERR_OUT_OF_RANGE, "integral index out of range");
  validate_form(a);
  const cfx_integral_dev& I = a->integrals[integral];
  cfx_space_s* V = a->V;
  const int64_t limit = (I.type == CFX_CELL && use_rule) ? (I.rules ? I.rules->nr.value() : 0) : I.n_entities.value();
  require(index >= 0 && index < limit, CFX_ERR_OUT_OF_RANGE, "entity index out of range");
  const int nloc = V->ndofs_cell * V->bs * (I.type == CFX_INTERIOR_FACET ? 2 : 1);
  const int nloc1 = a->rectangular() ? a->V1->ndofs_cell * a->V1->bs * (I.type == CFX_INTERIOR_FACET ? 2 : 1) : nloc;
  const int64_t n = a->rank == 2 ? (int64_t)nloc * nloc1 : nloc;
  OutArray<double> out(Ae, n, false);
  ZeroFlag err;
  AsmArgs A{};
  A.x = V->mesh->x.p; A.conn = V->mesh->conn.p; A.dofmap = V->dofmap.p;
  A.dump = out.dev; A.error = err.p;
  if (a->rectangular()) launch_rectangular(a, I, A, index, use_rule); else launch_integral(a, I, A, index, use_rule);
  out.finish();
  CFX_API_END
}

int cfx_active_domain(cfx_form_t a, cfx_active_t* out)
{
  CFX_API_BEGIN
  require(a && out, CFX_ERR_INVALID_ARGUMENT, "cfx_active_domain: null argument");
  // deactivate.h:80-85
  require(a->rank == 2, CFX_ERR_INVALID_ARGUMENT, "cutfemx.fem.active_domain requires a rank-2 bilinear CutForm");
  require(!a->rectangular(), CFX_ERR_INVALID_ARGUMENT,
          "active_domain / deactivation act on square systems: the form's test and trial spaces differ");
  validate_form(a);
  cfx_space_s* V = a->V;
  auto d = std::make_unique<cfx_active_s>();
  d->V = V;
  // The row plan of the form already holds both indicators: cellmark (cells of
  // every cell integral, standard or runtime) and rowmark (dofs touched by any
  // entity) -- collect_active_cells / build_active_indicator of deactivate.h:103-183.
  cfx_row_plan& plan = row_plan(a);
  d->plan = a->plan;
  const int64_t nrows = V->ndofs * V->bs;
  (void)nrows;
  // deactivate.h:155-160: no active cell <=> no active row (a cell of a cell integral marks its dofs)
  step_require_positive(plan.n_active_rows, CFX_ERR_INVALID_ARGUMENT, "cutfemx.fem.active_domain found no active background cells");
  end_of_call_sync();
  *out = d.release();
  CFX_API_END
}

} // extern "C"

namespace cfx
{
// ActiveDomain::active_cells / inactive_dofs as arrays (deactivate.h:387-400), compacted from the plan's marks
void active_lists(cfx_active_s* d)
{
  if (d->lists_built) return;
  cfx_space_s* V = d->V;
  cfx_row_plan& plan = *d->plan;
  const int64_t nc = V->mesh->ncells;
  const uint8_t* cell_ind = plan.cellmark.p;
  if (plan.nfacets.value() > 0 && !plan.facet_rows.owned)
  {
    // the facet rows of a plan with one facet list ARE the form's entity array: gone with the list's owner (the form and
    // the ghost-facet handle of a step that has ended) -- say so instead of reading a recycled block
    for (const auto& k : plan.key_facets)
      if (k[1] == (int64_t)(uintptr_t)plan.facet_rows.p && k[5] != 0 && dev_block_serial(plan.facet_rows.p) != (uint64_t)k[5])
        throw Error(CFX_ERR_RUNTIME, "stale active domain: the facet list of the form it was made from was released; ask for "
                                     "the lists while the form is alive (the Python wrapper keeps it)");
  }
  if (plan.nfacets.value() > 0)
  {
    // the cells of the facet integrals are almost always cells of the cell integrals too (the ghost-penalty band
    // lies in the cut and inside cells): then the cell marks alone are the indicator; else both cells of every
    // facet join it (deactivate.h:138-146)
    ZeroFlag uncovered;
    const int64_t nf = plan.nfacets.value();
    launch("active_cells", facet_cells_covered_kernel, grid_for(nf * 2), dim3(kBlock), 0, nf, plan.facet_rows.p,
           plan.cellmark.p, uncovered.p);
    if (read_scalar(uncovered.p))
    {
      d->cell_indicator.alloc((nc + 3) & ~3LL);
      CFX_HIP(hipMemcpyAsync(d->cell_indicator.p, plan.cellmark.p, (size_t)nc, hipMemcpyDeviceToDevice, ctx().stream));
      launch("mark_cells", mark_cells_kernel, grid_for(nf), dim3(kBlock), 0, nf, plan.facet_rows.p, 4, d->cell_indicator.p);
      launch("mark_cells", mark_cells_kernel, grid_for(nf), dim3(kBlock), 0, nf, plan.facet_rows.p + 2, 4, d->cell_indicator.p);
      cell_ind = d->cell_indicator.p;
    }
  }
  const int32_t* known_tiles = (cell_ind == plan.cellmark.p && plan.cell_tile_counts.n == (nc + kByteTile - 1) / kByteTile)
                                   ? plan.cell_tile_counts.p : nullptr;
  d->n_active = Count(compact_bytes("active_cells", nc, cell_ind, ByteNonZero{}, d->active_cells, known_tiles));
  const int64_t nrows = V->ndofs * V->bs;
  if (V->bs == 1)
    d->n_inactive = Count(compact_bytes("inactive_dofs", nrows, plan.rowmark.p, ByteZero{}, d->inactive_dofs, nullptr,
                                        nrows - plan.n_active_rows.value()));
  else
  {
    DevArray<uint8_t> ind(nrows);
    ind.zero();
    const int64_t nact = d->n_active.value();
    launch("mark_dofs", mark_dofs_kernel, grid_for(nact * V->ndofs_cell), dim3(kBlock), 0, nact,
           d->active_cells.p, V->dofmap.p, V->ndofs_cell, V->bs, ind.p);
    d->n_inactive = Count(compact_bytes("inactive_dofs", nrows, ind.p, ByteZero{}, d->inactive_dofs));
  }
  d->lists_built = true;
}
} // namespace cfx

extern "C" {

int cfx_active_view(cfx_active_t d, const int32_t** active_cells, int64_t* n_active, const int32_t** inactive_dofs,
                    int64_t* n_inactive)
{
  CFX_API_BEGIN
  require(d != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_active_view: null handle");
  // the number of inactive dofs alone (ActiveDomain.num_active_dofs) needs no list; inside a step it is the capacity
  // derived from the row plan while the exact count is still in HBM
  if (!active_cells && !n_active && !inactive_dofs)
  {
    // rows off the plan's active set (a dof is active iff an entity of the form touches it: the row marks)
    if (n_inactive) *n_inactive = d->V->ndofs * d->V->bs - d->plan->n_active_rows.cap() * d->V->bs;
    return CFX_OK;
  }
  active_lists(d);
  if (active_cells) *active_cells = d->active_cells.p;
  if (n_active) *n_active = d->n_active.cap();
  if (inactive_dofs) *inactive_dofs = d->inactive_dofs.p;
  if (n_inactive) *n_inactive = d->n_inactive.cap();
  CFX_API_END
}

int cfx_deactivate_outside(cfx_active_t d, cfx_pattern_t P, double* values, double* b, double diagonal,
                           double rhs_value)
{
  CFX_API_BEGIN
  require(d && (values == nullptr || P), CFX_ERR_INVALID_ARGUMENT, "cfx_deactivate_outside: null argument");
  const int64_t nrows = d->V->ndofs * d->V->bs;
  std::unique_ptr<OutArray<double>> ov, ob;
  if (values) ov = std::make_unique<OutArray<double>>(values, count_for_buffer(P->nnz, values), true);
  if (b) ob = std::make_unique<OutArray<double>>(b, nrows, true);
  const char* kNoDiag = "Deactivated matrix row has no diagonal entry.";
  ErrorFlag err(CFX_ERR_RUNTIME, kNoDiag);
  cfx_row_plan& plan = *d->plan;
  const int64_t ntiles = (nrows + kByteTile - 1) / kByteTile;
  if (d->V->bs == 1 && plan.row_tile_counts.n == ntiles && (!values || P->nrows == nrows))
  {
    // straight from the row marks, tile by tile: a tile of kByteTile inactive rows is two contiguous fills, the other
    // tiles set their inactive rows one by one -- no list of inactive dofs is built
    launch("deactivate", deactivate_marks_kernel, dim3((unsigned)ntiles), dim3(kBlock), 0, nrows, plan.rowmark.p,
           plan.row_tile_counts.p, values ? P->indptr.p : (const int64_t*)nullptr, values ? P->indices.p : (const int32_t*)nullptr,
           values ? ov->dev : (double*)nullptr, b ? ob->dev : (double*)nullptr, diagonal, rhs_value, err.p,
           plan.n_active_rows.devn(), values ? P->nnz.devn() : DevN());
  }
  else
  {
    active_lists(d);
    const int64_t n_inactive = d->n_inactive.value();
    if (n_inactive > 0)
      launch("deactivate", deactivate_kernel, grid_for(n_inactive), dim3(kBlock), 0, n_inactive,
             d->inactive_dofs.p, P ? P->indptr.p : nullptr, P ? P->indices.p : nullptr, values ? ov->dev : nullptr,
             b ? ob->dev : nullptr, diagonal, rhs_value, err.p);
  }
  // deactivate.h: validate_matrix_rows
  err.check(CFX_ERR_RUNTIME, kNoDiag);
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
