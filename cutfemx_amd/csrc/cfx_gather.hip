// cutfemx_amd: row-centric kernels, part 2 -- gather assembly of matrices and
// vectors without global atomics.
//
// Two stages, both uniform in work per lane:
//  1. "tensor" kernels: the expensive local tensors -- cut cells (runtime rules,
//     10-40 points each), ghost-penalty facets, and rank-1 source terms (a few
//     transcendental evaluations per point) -- are formed entity-parallel and
//     stored entity-major in HBM (one 128 B line per P1 tet tensor).
//  2. "rows" kernels: a group of G lanes owns one matrix row r.  Its items are
//     the marked cells incident to r (dof->cells incidence of the space) and the
//     facets incident to r (dof->facets incidence of the form).  A lane either
//     computes the item's local row inline (uncut P1 stiffness: one point) or
//     reads it from the stage-1 tensors, then adds it to the row's CSR slots held
//     in LDS.  One coalesced `values[row] += ...` per row at the end.
// This is the reference's cell loop + mat_add (assemble_matrix_impl.h:103-188,
// :462-606) re-ordered by row.  The LDS reduction is either LDS FP64 atomics
// (default) or a lane-ordered, bitwise-reproducible sequence
// (CFX_DETERMINISTIC=1).
#include <cstdlib>
#include <type_traits>

#include "cfx_elem.h"

using namespace cfx;

namespace cfx
{
void dump_integral(cfx_form_s* a, int integral, int parts, double* out, bool fold_facets = false); // cfx_fem.hip
void dump_facet_jumps_p1(cfx_form_s* a, int integral, double* out, int* error);                    // cfx_fem.hip
void dump_facet_jumps_p2(cfx_form_s* a, int integral, int nq, double* out);                        // cfx_fem.hip
void dump_cut_moments(cfx_form_s* a, int integral, double* out);                                   // cfx_fem.hip
int quad_npoints(int dim, int degree);                                                              // cfx_quadhost.cpp
}

namespace
{

constexpr int kWave = 64;
// Row block of a workgroup.  Workgroups go round-robin to the 8 XCDs (block b -> XCD b % 8), each
// with its own L2.  CFX_XCD_TILE = T > 0 hands every XCD runs of T consecutive row blocks, so that
// neighbouring rows (which share cells, coordinates and staged tensors) meet in one L2.
#ifndef CFX_XCD_TILE
#define CFX_XCD_TILE 128 // measured at 512^3: vector gather -7 %, matrix kernels -2 %; 8192 (one chunk per XCD) +5 %
#endif
__device__ __forceinline__ int64_t row_block_id()
{
#if CFX_XCD_TILE > 0
  constexpr int64_t T = CFX_XCD_TILE;
  const int64_t b = blockIdx.x;
  return (b / (8 * T)) * (8 * T) + (b % 8) * T + (b / 8) % T;
#else
  return blockIdx.x;
#endif
}
#ifndef CFX_ROW_BLOCK
#define CFX_ROW_BLOCK row_block_id()
#endif
inline dim3 row_grid(int64_t nblocks)
{
#if CFX_XCD_TILE > 0
  const int64_t q = 8 * (int64_t)CFX_XCD_TILE;
  return dim3((unsigned)((nblocks + q - 1) / q * q));
#else
  return xcd_grid(nblocks);
#endif
}
#ifndef CFX_CUT_G
#define CFX_CUT_G 8 // lanes per interface row in assemble_rows_kernel (P1)
#endif
#ifndef CFX_VEC_CUT_LANES
#define CFX_VEC_CUT_LANES 4 // lanes per runtime rule in stage 1 of the linear forms (512^3: 16 -> 887 us, 8 -> 580, 4 -> 469, 2 -> 541)
#endif
#ifndef CFX_ROWS_CUT_WAVES
#define CFX_ROWS_CUT_WAVES 8 // ... and its form without the uncut-cell items (74 registers in 3-D: 5 or 6 -> 1.25, 8 -> 1.16 ms at 512^3)
#endif
#ifndef CFX_ROWS_WAVES
#define CFX_ROWS_WAVES 5 // waves per SIMD the gather kernel is compiled for (measured: 4 -> 2.27, 5 -> 2.09, 6 -> 2.58 ms at 256^3)
#endif

// whether assemble_rows_kernel carries the generic inline local-row code for a degree
#ifndef CFX_ROWS_INLINE_DEG2
#define CFX_ROWS_INLINE_DEG2 0 // degree 2 stages its uncut tensors: the inline code costs 60 VGPRs (config 4: 215 -> 151 ms)
#endif
template <int DEG>
constexpr bool kRowsInline = DEG == 1 || CFX_ROWS_INLINE_DEG2 != 0;

struct RowIntegral
{
  int kernel, qdegree, point_stride;
  int std_inline;            // uncut entities: read std_tensors (0), generic inline row (1), P1 stiffness row from vertex coordinates (2),
                             // degree-2 stiffness row in closed form (3)
  const unsigned long long* std_bits; // bitset of the uncut entities
  const int32_t* std_rank;            // entities before each 64-cell word
  const double* std_tensors; // [n_entities][ND*ND] (rank 2) or [ND][n_entities] (rank 1)
  int64_t n_std;             // n_entities
  int std_by_cell;           // rank 1: std_tensors is [ND][ncells], indexed by the cell itself (n_std = ncells; unmarked cells unwritten)
  const int32_t* parent_map; // sorted rule parents
  DevN nr;                   // (length in HBM inside a sync-free step: the entries behind it are not rules)
  const double* rule_tensors; // [nr][ND*ND] or [nr][ND]; rule_moments: [nr][16]
  int rule_moments;           // degree-2 stiffness: rule_tensors holds the barycentric moments of every rule (cut_moments_kernel)
  const int32_t* rule_keys;   // parent cell -> first rule (open addressing, plan.rule_keys / rule_first)
  const int32_t* rule_first;
  unsigned rule_mask;
  double params[8];
};

struct RowArgs
{
  const double* x;
  const int32_t* conn;
  const int32_t* dofmap;
  DevN n_active;             // rows of this launch (length in HBM inside a sync-free step)
  const int32_t* active_rows;
  const int64_t* d2c_off;
  const int32_t* d2c;
  const uint8_t* cellmark;
  const int64_t* d2f_off;      // indexed by special_pos[r] where special_mark[r]
  const uint8_t* special_mark;
  const int32_t* special_pos;
  const int32_t* d2f;
  const int32_t* facet_rows;
  const double* facet_tensors; // [nfacets][(2ND)^2]
  int n_cell;
  RowIntegral cell[4];
  const int8_t* bc0;
  const int8_t* bc1;
  const int64_t* indptr;
  const int32_t* indices;
  double* values;
  int* error;
  int iso_geometry; // the space's dofmap is the geometry dofmap (P1): dofs are vertex ids
  unsigned mark_mask;   // cell-mark bits this launch handles: 0x0F uncut entities, 0xF0 runtime rules (+ facets)
  int fold_facets;      // every facet-type entity joins two cells across a shared facet (no extension pairs): 1 fold in the
                        // gather (P1), 2 stage 1 stored the folded tensor (P1), 3 stage 1 stored rank-one records
  int facet_nq;         // fold_facets = 3, degree 2: records (quadrature points) per facet
  unsigned inline_bits; // p1 kernel: mark bits of the inline P1 stiffness integrals
  const uint8_t* slotn;    // degree 2: 12-byte slot records per (dof, cell) entry (cfx::Stencil::slotn), or null
  const int32_t* vec_skip; // linear forms, second pass over the plain rows: skip row r when vec_skip[r] > 0 (it has a segment)
  DevN vec_n_odd;          // ... and leave at once when no plain row is without one
  const uint32_t* slot4;   // plain kernel: cfx::Stencil tables of the space
  const uint8_t* diagpos;
  const int64_t* st_off;
  const int32_t* st_nbr;
  const unsigned long long* plain_masks; // plan.plain_masks / plain_uniform, indexed like active_rows
  const uint8_t* plain_uniform;
  int fresh; // values holds zeros on entry (cfx_assemble_matrix_zeroed): single-writer rows are stored, not accumulated
  // degree 2, scalar: ONE tensor per cut cell, summed over every rule of every cell integral (cut_tensors_p2_kernel),
  // keyed by the cell's position in plan.cut_cells (bitset + rank); null: per-integral rule tensors / moments
  const double* cut_tensors;
  const unsigned long long* cut_bits;
  const int32_t* cut_rank;
};

// index of cell c in the sorted entity list described by (bits, rank)
__device__ __forceinline__ int64_t entity_index(const unsigned long long* __restrict__ bits,
                                                const int32_t* __restrict__ rank, int64_t c)
{
  const int64_t w = c >> 6;
  return (int64_t)rank[w] + __popcll(bits[w] & ((1ull << (c & 63)) - 1ull));
}

// first rule hosted by cut cell c (the cell is known to host at least one)
__device__ __forceinline__ int64_t first_rule(const int32_t* __restrict__ keys, const int32_t* __restrict__ first,
                                              unsigned mask, int32_t c)
{
  unsigned h = cfx_hash32((uint32_t)c) & mask;
  for (unsigned probe = 0; probe <= mask; ++probe)
  {
    const int32_t k = keys[h];
    if (k == c) return first[h];
    if (k == -1) break;
    h = (h + 1) & mask;
  }
  return 0x7fffffff; // absent: the caller's range loop ends immediately
}

__device__ __forceinline__ int64_t lower_bound_i32(const int32_t* __restrict__ a, int64_t n, int32_t v)
{
  int64_t lo = 0, hi = n;
  while (lo < hi)
  {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// 24 B vertex record (stride 3 doubles, 8 B aligned): one 16 B + one 8 B load
typedef double cfx_d2u __attribute__((ext_vector_type(2), aligned(8)));

template <int TDIM>
__device__ __forceinline__ void load_vertex(const double* __restrict__ x, int64_t v, double* out)
{
  const cfx_d2u a = *reinterpret_cast<const cfx_d2u*>(x + 3 * v);
  out[0] = a.x; out[1] = a.y;
  if constexpr (TDIM == 3) out[2] = x[3 * v + 2];
}

// Row of the P1 stiffness tensor that belongs to vertex xr of a simplex, from xr and the
// other TDIM vertices xo (any order).  With e_k = xo_k - xr and the cofactor vectors c_k
// (grad lambda_k = c_k / det), grad lambda_r = -(sum c_k) / det and
//   A[r][k] = |K| grad lambda_r . grad lambda_k = (c_r . c_k) / (TDIM! |det|),
// the exact integral for any quadrature degree (constant gradients).
template <int TDIM>
__device__ __forceinline__ void p1_stiffness_row(const double* xr, const double (*xo)[TDIM], double& diag, double* off)
{
  double e[TDIM][TDIM], c[TDIM][TDIM], cr[TDIM];
#pragma unroll
  for (int k = 0; k < TDIM; ++k)
#pragma unroll
    for (int d = 0; d < TDIM; ++d) e[k][d] = xo[k][d] - xr[d];
  double det;
  if constexpr (TDIM == 3)
  {
#pragma unroll
    for (int k = 0; k < 3; ++k)
    {
      const double* a = e[(k + 1) % 3];
      const double* b = e[(k + 2) % 3];
      c[k][0] = a[1] * b[2] - a[2] * b[1];
      c[k][1] = a[2] * b[0] - a[0] * b[2];
      c[k][2] = a[0] * b[1] - a[1] * b[0];
    }
    det = e[0][0] * c[0][0] + e[0][1] * c[0][1] + e[0][2] * c[0][2];
  }
  else
  {
    c[0][0] = e[1][1];  c[0][1] = -e[1][0];
    c[1][0] = -e[0][1]; c[1][1] = e[0][0];
    det = e[0][0] * e[1][1] - e[0][1] * e[1][0];
  }
  // reciprocal by v_rcp_f64 + one Newton step (the IEEE division sequence is 11 instructions; error < 1 ulp)
  const double den = (TDIM == 3 ? 6.0 : 2.0) * fabs(det);
  double scale = __builtin_amdgcn_rcp(den);
  scale = fma(scale, fma(-den, scale, 1.0), scale);
  // cr = -grad lambda_r det: its sign only enters the off-diagonal products, where it goes into the scale
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    double t = c[0][d];
#pragma unroll
    for (int k = 1; k < TDIM; ++k) t += c[k][d];
    cr[d] = t;
  }
  double dd = cr[0] * cr[0];
#pragma unroll
  for (int d = 1; d < TDIM; ++d) dd = fma(cr[d], cr[d], dd);
  diag = dd * scale;
  const double nscale = -scale;
#pragma unroll
  for (int k = 0; k < TDIM; ++k)
  {
    double t = cr[0] * c[k][0];
#pragma unroll
    for (int d = 1; d < TDIM; ++d) t = fma(cr[d], c[k][d], t);
    off[k] = t * nscale;
  }
}

// ---------------------------------------------------------------------------
// stage 1 for linear forms: all ND entries of be per entity, the source /
// boundary datum evaluated once per point.  One thread per entity.
// ---------------------------------------------------------------------------
struct VecArgs
{
  const double* x;
  const int32_t* conn;
  DevN n;                    // entities of this launch (n.cap: the stride of the [ND][n] layout)
  const int32_t* entities;   // uncut: cells
  const int32_t* offsets;    // runtime
  const int32_t* parent_map;
  const double* points;
  const double* weights;
  const double* point_data;
  int point_stride, kernel, qdegree;
  double params[8];
  const int32_t* dofmap;
  const double* coeff; // dof values of a CFX_F_COEFFICIENT source, or null
  double* out; // runtime rules: [n][ND]; uncut entities: [ND][n], or [ND][out_cells] indexed by the cell when out_cells > 0
  int64_t out_cells;
  // P1 row-ordered staging (cfx_row_plan::vec_t2off): entry i of the cell goes to t2[t2off[dof_i] + cpos[cell][i]]
  // when dof_i is a plain row; the per-cell record is only written for cells with a dof off the plain rows
  const int32_t* t2off;
  const uint8_t* cpos;
  double* t2;
};

// element vector of uncut entity e: one array per local index ([ND][n]).  The rows gather (cell, local index)
// pairs; cells numbered together around a vertex often hold it at the same local index (6 + 6 + 6 x 2 of the 24
// tets around a vertex of a Kuhn box mesh), so their entries share 32 B sectors -- the [n][ND] records never do.
// P1 with a row-ordered staging (A.t2): entry i goes to the segment of row dof_i; the per-cell record is only
// written for cells with a dof off the plain rows.
template <int ND, bool P1>
__device__ __forceinline__ void store_std_vector(const VecArgs& A, int64_t e, int64_t cell, const double* be)
{
  bool record = true;
  if constexpr (P1)
  {
    if (A.t2)
    {
      record = false;
#pragma unroll
      for (int i = 0; i < ND; ++i)
      {
        const int32_t o = A.t2off[A.dofmap[cell * ND + i]] - 1; // (stored + 1: 0 = the row has no segment)
        if (o >= 0) A.t2[(int64_t)o + A.cpos[cell * ND + i]] = be[i];
        else record = true;
      }
    }
  }
  if (record)
  {
    const int64_t stride = A.out_cells > 0 ? A.out_cells : A.n.cap, at = A.out_cells > 0 ? cell : e;
#pragma unroll
    for (int i = 0; i < ND; ++i) A.out[(int64_t)i * stride + at] = be[i];
  }
}

// LANES > 1 (runtime rules): a group of lanes shares one rule, the points of the rule are dealt
// round-robin to the lanes (coalesced point / weight / normal reads, balanced 6-42 point rules) and
// the ND partial sums are folded with shuffles.
// element vector of one entity (an uncut cell, or rule e of a cut cell): lane `sub` of LANES takes the points sub, sub + LANES, ...
// vids: the cell's vertex ids when the caller has them already (its connectivity row was requested ahead)
// KSEL: the integrand is known when the kernel is compiled (1 the source term, 2 the Nitsche datum; 0: A.kernel) -- the
// source term then holds neither K nor the derivatives (degree 2, runtime rules: 158 -> under 128 registers)
template <int TDIM, int DEG, bool RUNTIME, int LANES, int KSEL = 0>
__device__ __forceinline__ void entity_vector(const VecArgs& A, int64_t e, int64_t cell, int sub, double* be,
                                              const int32_t* vids = nullptr)
{
  constexpr int ND = Elem<TDIM, DEG>::ND;
  const int kern = KSEL == 0 ? A.kernel : (KSEL == 1 ? CFX_L_SOURCE : CFX_L_NITSCHE_RHS);
  Geo<TDIM> g;
  if (vids)
  {
#pragma unroll
    for (int i = 0; i <= TDIM; ++i) load_vertex<TDIM>(A.x, vids[i], g.x[i]);
  }
  else
    load_cell<TDIM>(A.x, A.conn, cell, g);
  const bool nitsche = kern == CFX_L_NITSCHE_RHS;
  if (!RUNTIME || nitsche) jacobian<TDIM>(g); // (runtime rules carry physical weights: the source term needs no Jacobian)
  const double h = nitsche ? cell_diameter<TDIM>(g) : 1.0; // only the Nitsche datum needs h (and K)
  int npts;
  const double *pts, *wts, *pdata = nullptr;
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
#pragma unroll
  for (int i = 0; i < ND; ++i) be[i] = 0.0;
  // edges from vertex 0: x(X) = x_0 + sum_t X_t (x_{t+1} - x_0), TDIM fused multiply-adds per coordinate
  double ed[TDIM][TDIM];
#pragma unroll
  for (int t = 0; t < TDIM; ++t)
#pragma unroll
    for (int d = 0; d < TDIM; ++d) ed[t][d] = g.x[t + 1][d] - g.x[0][d];
  for (int q = sub; q < npts; q += LANES)
  {
    double X[TDIM], xq[TDIM];
#pragma unroll
    for (int t = 0; t < TDIM; ++t) X[t] = pts[(int64_t)q * TDIM + t];
#pragma unroll
    for (int d = 0; d < TDIM; ++d)
    {
      double v = g.x[0][d];
#pragma unroll
      for (int t = 0; t < TDIM; ++t) v = fma(X[t], ed[t][d], v);
      xq[d] = v;
    }
    const double w = wts[q] * wscale;
    double N[ND];
    if (kern == CFX_L_SOURCE)
    {
      tabulate_values<TDIM, DEG>(X, N);
      double fv;
      if (A.coeff)
      {
        fv = 0.0;
#pragma unroll
        for (int j = 0; j < ND; ++j) fv += N[j] * A.coeff[A.dofmap[cell * ND + j]];
      }
      else
        fv = field_eval<TDIM>((int)A.params[0], xq);
      const double f = w * A.params[1] * fv;
#pragma unroll
      for (int i = 0; i < ND; ++i) be[i] += f * N[i];
    }
    else if (kern == CFX_L_NITSCHE_RHS)
    {
      const double* nrm = pdata + (int64_t)q * A.point_stride;
      const double gam = A.params[0] / h;
      const double gv = A.params[2] * field_eval<TDIM>((int)A.params[1], xq);
      // n . grad N_i = sum_d n_d sum_t K[t][d] dN_i/dX_t = (K n) . dN_i/dX
      double kn[TDIM], dn[ND];
#pragma unroll
      for (int t = 0; t < TDIM; ++t)
      {
        double v = 0.0;
#pragma unroll
        for (int d = 0; d < TDIM; ++d) v += g.K[t][d] * nrm[d];
        kn[t] = v;
      }
      tabulate_dot<TDIM, DEG>(X, kn, N, dn);
#pragma unroll
      for (int i = 0; i < ND; ++i) be[i] += w * (-dn[i] * gv + gam * gv * N[i]);
    }
  }
}

#ifndef CFX_VECCUT_WAVES
#define CFX_VECCUT_WAVES 1
#endif
template <int TDIM, int DEG, bool RUNTIME, int LANES = 1, int KSEL = 0>
__global__ void __launch_bounds__(kBlock, (DEG == 2 && RUNTIME && KSEL != 0) ? 4 : ((DEG == 1 && RUNTIME) ? CFX_VECCUT_WAVES : 1)) vec_tensors_kernel(VecArgs A)
{
  constexpr int ND = Elem<TDIM, DEG>::ND;
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t e = tid / LANES;
  // LANES == 1: the point index is the same in every lane, so the rule's points / weights (and the P1 basis
  // values) are scalar loads and scalar operands
  const int sub = LANES == 1 ? 0 : (int)(tid - e * LANES);
  if (e >= dev_n(A.n)) return;
  const int64_t cell = RUNTIME ? A.parent_map[e] : A.entities[e];
  double be[ND];
  entity_vector<TDIM, DEG, RUNTIME, LANES, KSEL>(A, e, cell, sub, be);
  if constexpr (LANES > 1)
  {
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
      for (int o = LANES / 2; o > 0; o >>= 1) be[i] += __shfl_xor(be[i], o, LANES);
    if (sub != 0) return;
  }
  if constexpr (RUNTIME)
  {
#pragma unroll
    for (int i = 0; i < ND; ++i) A.out[e * ND + i] = be[i];
  }
  else
    store_std_vector<ND, DEG == 1>(A, e, cell, be);
}

// The source term f v with f = c prod_d sin(pi x_d) (CFX_F_SINPROD / CFX_F_POISSON_RHS) on the UNCUT cells of a P1
// space: the kernel the generic one above spends its time in (11 points x 3 sinpi per tet since the round-3 degree-4 rule, FP64-VALU bound).
// sin(pi (a + delta)) = S cos(pi delta) + C sin(pi delta) with a = the coordinate of vertex 0 (one sincospi per
// axis and cell) and delta the offset of the point inside the cell: for |pi delta| <= 0.03 the two series up to
// u^7 are exact to 2e-17, so a point costs 7 fused multiply-adds per axis instead of a range reduction and a
// degree-17 polynomial.  Larger cells (coarse test meshes) take the exact evaluation.
// Memory side: entity id -> connectivity row -> 4 vertex / 4 segment-offset gathers are three dependent levels, ~8 us
// per wave under load against ~1 us of arithmetic, so a thread walks its cells with a software pipeline three
// deep (cell j is computed while the gathers of j+1, the connectivity row of j+2 and the id of j+3 are in flight).
#ifndef CFX_SOURCE_BLOCKS_PER_CU
#define CFX_SOURCE_BLOCKS_PER_CU 16
#endif
template <int TDIM>
struct SourceCell
{
  int64_t e;
  int32_t cell;
  int32_t v[TDIM + 1];
  double x[TDIM + 1][TDIM];
  int32_t t2o[TDIM + 1];
  uint32_t cp;
};

template <int TDIM>
__device__ __forceinline__ void source_load_conn(const VecArgs& A, SourceCell<TDIM>& c)
{
  if constexpr (TDIM == 3)
  {
    const int4 r = *reinterpret_cast<const int4*>(A.conn + (int64_t)c.cell * 4);
    c.v[0] = r.x; c.v[1] = r.y; c.v[2] = r.z; c.v[3] = r.w;
    if (A.t2) c.cp = *reinterpret_cast<const uint32_t*>(A.cpos + (int64_t)c.cell * 4);
  }
  else
  {
#pragma unroll
    for (int i = 0; i <= TDIM; ++i) c.v[i] = A.conn[(int64_t)c.cell * (TDIM + 1) + i];
    c.cp = 0;
    if (A.t2)
    {
#pragma unroll
      for (int i = 0; i <= TDIM; ++i) c.cp |= (uint32_t)A.cpos[(int64_t)c.cell * (TDIM + 1) + i] << (8 * i);
    }
  }
}

template <int TDIM>
__device__ __forceinline__ void source_load_vertices(const VecArgs& A, SourceCell<TDIM>& c)
{
#pragma unroll
  for (int i = 0; i <= TDIM; ++i)
  {
    load_vertex<TDIM>(A.x, c.v[i], c.x[i]);
    c.t2o[i] = A.t2 ? A.t2off[c.v[i]] - 1 : -1; // (stored + 1: 0 = the row has no segment)
  }
}

template <int TDIM>
__device__ __forceinline__ void source_vector(const SourceCell<TDIM>& c, int npts, const double* __restrict__ pts,
                                              const double* __restrict__ wts, double fscale, double* be)
{
  constexpr int ND = TDIM + 1;
  double ed[TDIM][TDIM]; // (x_{t+1} - x_0)_d
  double hmax = 0.0;
#pragma unroll
  for (int t = 0; t < TDIM; ++t)
#pragma unroll
    for (int d = 0; d < TDIM; ++d)
    {
      ed[t][d] = c.x[t + 1][d] - c.x[0][d];
      hmax = fmax(hmax, fabs(ed[t][d]));
    }
  double det;
  if constexpr (TDIM == 2) det = ed[0][0] * ed[1][1] - ed[0][1] * ed[1][0];
  else
    det = ed[0][0] * (ed[1][1] * ed[2][2] - ed[1][2] * ed[2][1]) - ed[0][1] * (ed[1][0] * ed[2][2] - ed[1][2] * ed[2][0])
          + ed[0][2] * (ed[1][0] * ed[2][1] - ed[1][1] * ed[2][0]);
  const double cscale = fscale * fabs(det);
#pragma unroll
  for (int i = 0; i < ND; ++i) be[i] = 0.0;
  const double umax = kPi * hmax;
  if (umax <= 0.03)
  {
    // sin(pi a + u) = S + C u - S u^2/2 - C u^3/6 + S u^4/24 + C u^5/120 - S u^6/720 - C u^7/5040: the first
    // neglected term is below 1.1e-16 after u^5 for |u| <= 0.0065 (h <= 1/484) and after u^7 for |u| <= 0.03
    double S[TDIM], C[TDIM];
#pragma unroll
    for (int d = 0; d < TDIM; ++d)
    {
      cfx_sincospi(c.x[0][d], S[d], C[d]);
#pragma unroll
      for (int t = 0; t < TDIM; ++t) ed[t][d] *= kPi;
    }
    auto series = [&](auto deg_tag)
    {
      constexpr int DEGS = decltype(deg_tag)::value;
      double co[TDIM][DEGS + 1];
#pragma unroll
      for (int d = 0; d < TDIM; ++d)
      {
        const double inv[8] = {1.0, 1.0, -1.0 / 2.0, -1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, -1.0 / 720.0, -1.0 / 5040.0};
#pragma unroll
        for (int k = 0; k <= DEGS; ++k) co[d][k] = ((k & 1) ? C[d] : S[d]) * inv[k];
      }
      // be[0] is carried as F = sum_q f_q (N_0 = 1 - sum_t X_t: be_0 = F - sum_t be_{t+1}) and the cell's scale is
      // applied once at the end: 4 instructions per point less in a kernel that is bound by FP64 issue
      for (int q = 0; q < npts; ++q) // (the point index is wave-uniform: points, weights and basis values are scalars)
      {
        double X[TDIM];
#pragma unroll
        for (int t = 0; t < TDIM; ++t) X[t] = pts[q * TDIM + t];
        double f = wts[q];
#pragma unroll
        for (int d = 0; d < TDIM; ++d)
        {
          double u = X[0] * ed[0][d];
#pragma unroll
          for (int t = 1; t < TDIM; ++t) u = fma(X[t], ed[t][d], u);
          double p = co[d][DEGS];
#pragma unroll
          for (int k = DEGS - 1; k >= 0; --k) p = fma(p, u, co[d][k]);
          f *= p;
        }
        be[0] += f;
#pragma unroll
        for (int t = 0; t < TDIM; ++t) be[t + 1] = fma(f, X[t], be[t + 1]);
      }
#pragma unroll
      for (int t = 0; t < TDIM; ++t) be[0] -= be[t + 1];
#pragma unroll
      for (int i = 0; i < ND; ++i) be[i] *= cscale;
    };
    if (umax <= 0.0065) series(std::integral_constant<int, 5>{});
    else series(std::integral_constant<int, 7>{});
  }
  else
  {
    for (int q = 0; q < npts; ++q)
    {
      double X[TDIM], l0 = 1.0;
#pragma unroll
      for (int t = 0; t < TDIM; ++t) { X[t] = pts[q * TDIM + t]; l0 -= X[t]; }
      double f = wts[q] * cscale;
#pragma unroll
      for (int d = 0; d < TDIM; ++d)
      {
        double v = c.x[0][d];
#pragma unroll
        for (int t = 0; t < TDIM; ++t) v = fma(X[t], ed[t][d], v);
        f *= cfx_sinpi(v);
      }
      be[0] = fma(f, l0, be[0]);
#pragma unroll
      for (int t = 0; t < TDIM; ++t) be[t + 1] = fma(f, X[t], be[t + 1]);
    }
  }
}

template <int TDIM>
__device__ __forceinline__ void source_compute(const VecArgs& A, const SourceCell<TDIM>& c, bool valid, int npts,
                                               const double* __restrict__ pts, const double* __restrict__ wts, double fscale)
{
  constexpr int ND = TDIM + 1;
  double be[ND];
  source_vector<TDIM>(c, npts, pts, wts, fscale, be);
  if (!valid) return;
  // entry i to the segment of row dof_i (store_std_vector, with the gathers already in registers)
  bool record = A.t2 == nullptr;
#pragma unroll
  for (int i = 0; i < ND; ++i)
  {
    if (c.t2o[i] >= 0) A.t2[(int64_t)c.t2o[i] + ((c.cp >> (8 * i)) & 0xffu)] = be[i];
    else record = true;
  }
  if (record)
  {
#pragma unroll
    for (int i = 0; i < ND; ++i) A.out[A.out_cells > 0 ? (int64_t)i * A.out_cells + c.cell : (int64_t)i * A.n.cap + c.e] = be[i];
  }
}

#ifndef CFX_SOURCE_WAVES
#define CFX_SOURCE_WAVES 4 // (waves per SIMD the source kernels are compiled for: 3 -> 4 took 3.14 -> 2.96 ms at 512^3 in round 4; 5 spills: 4.4 ms)
#endif
template <int TDIM>
__global__ void __launch_bounds__(kBlock, CFX_SOURCE_WAVES) vec_source_sin_p1_kernel(VecArgs A)
{
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t e0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t An = dev_n(A.n);
  if (An == 0) return;
  const int64_t last = An - 1;
  int npts;
  const double* wts;
  const double* pts = ref_rule(TDIM, A.qdegree, npts, wts);
  const double fscale = A.params[1] * ((int)A.params[0] == CFX_F_POISSON_RHS ? (double)TDIM * kPi * kPi : 1.0);
  auto ent = [&](int64_t e) { return A.entities[e < last ? e : last]; }; // (clamped: the tail of the pipeline loads the last cell again)
  SourceCell<TDIM> a, b;
  int32_t cell_c, cell_d;
  a.e = e0; b.e = e0 + stride;
  a.cell = ent(a.e); b.cell = ent(b.e); cell_c = ent(e0 + 2 * stride);
  source_load_conn<TDIM>(A, a);
  source_load_conn<TDIM>(A, b);
  source_load_vertices<TDIM>(A, a);
  for (int64_t e = e0; e < An; e += stride)
  {
    cell_d = ent(e + 3 * stride);
    SourceCell<TDIM> c;
    c.e = e + 2 * stride; c.cell = cell_c;
    source_load_conn<TDIM>(A, c);
    source_load_vertices<TDIM>(A, b);
    source_compute<TDIM>(A, a, true, npts, pts, wts, fscale);
    a = b;
    b = c;
    cell_c = cell_d;
  }
}

// ---------------------------------------------------------------------------
// Linear forms by cell block (cfx::VecBlocks): a workgroup takes B consecutive cells, one per thread, puts their
// element vectors at the entries' sorted slots in LDS, and thread t adds up the segment of the t-th dof of the
// block's union -- one partial per (block, dof), stored coalesced.  The rows then add the partials of the blocks
// around them (vec_blocks_rows_kernel).  Cells without the integral's mark contribute zeros.
// ---------------------------------------------------------------------------
struct VecBlockArgs
{
  int64_t n_active;          // blocks with a marked cell
  const int32_t* active;
  const int64_t* base;       // [nblocks] first partial of the block
  const int64_t* u_off;
  const uint16_t* slot;
  const uint16_t* seg;
  const uint8_t* cellmark;
  uint8_t mark;              // the uncut entities of the one integral that has them (0: none)
  int64_t ncells;
  double* part;
  // the runtime rules of cell slot s (mark bit 16 << s): staged element vectors [nr][ND], rules of a cell consecutive
  struct Rules
  {
    const int32_t* parent_map;
    DevN nr;
    const double* tensors;
    const int32_t* keys;
    const int32_t* first;
    unsigned mask;
  } rules[4];
  int n_slots;
};

// ... plus the staged vectors of the cell's runtime rules, slot by slot, rule by rule
template <int ND>
__device__ __forceinline__ void add_rule_vectors(const VecBlockArgs& P, int64_t c, unsigned cm, double* be)
{
  if ((cm & 0xF0u) == 0) return;
  for (int s = 0; s < P.n_slots; ++s)
    if (cm & (16u << s))
    {
      const VecBlockArgs::Rules& R = P.rules[s];
      for (int64_t e = first_rule(R.keys, R.first, R.mask, (int32_t)c); e < dev_len(R.nr) && R.parent_map[e] == c; ++e)
#pragma unroll
        for (int j = 0; j < ND; ++j) be[j] += R.tensors[e * ND + j];
    }
}

// the ND sorted slots of a cell (2 B each; an even ND is read as 4 B or 8 B words)
template <int ND>
__device__ __forceinline__ void load_slots(const uint16_t* __restrict__ slot, int64_t c, uint16_t* sl)
{
  if constexpr (ND % 4 == 0)
  {
    const uint2* w = reinterpret_cast<const uint2*>(slot + c * ND);
#pragma unroll
    for (int j = 0; j < ND / 4; ++j)
    {
      const uint2 v = w[j];
      sl[4 * j] = (uint16_t)(v.x & 0xffffu); sl[4 * j + 1] = (uint16_t)(v.x >> 16);
      sl[4 * j + 2] = (uint16_t)(v.y & 0xffffu); sl[4 * j + 3] = (uint16_t)(v.y >> 16);
    }
  }
  else if constexpr (ND % 2 == 0)
  {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(slot + c * ND);
#pragma unroll
    for (int j = 0; j < ND / 2; ++j)
    {
      const uint32_t v = w[j];
      sl[2 * j] = (uint16_t)(v & 0xffffu); sl[2 * j + 1] = (uint16_t)(v >> 16);
    }
  }
  else
  {
#pragma unroll
    for (int j = 0; j < ND; ++j) sl[j] = slot[c * ND + j];
  }
}

// RULES = 0: the uncut entities (mark P.mark) of the blocks of P.active; RULES = 1: the staged rule vectors of
// the blocks that hold a rule parent, added to the block's partials (stored when the block holds no uncut entity:
// bit 31 of the list entry) -- the partials of a block belong to that block alone, so the second pass is race-free;
// RULES = 2: both in one pass (a cell is an uncut entity or a rule parent)
#ifndef CFX_VB_WAVES
#define CFX_VB_WAVES 5 // wavefronts per SIMD the degree-2 block kernel is compiled for
#endif
template <int TDIM, int DEG, int B, int RULES>
__global__ void __launch_bounds__(B, DEG == 2 ? CFX_VB_WAVES : 1) vec_blocks_kernel(VecArgs A, VecBlockArgs P)
{
  constexpr int ND = Elem<TDIM, DEG>::ND;
  __shared__ double s_val[B * ND];
  const int32_t entry = P.active[blockIdx.x];
  const int64_t k = entry & 0x7fffffff;
  const int64_t c = k * B + threadIdx.x;
  const int nb = (int)min((int64_t)B, P.ncells - k * B);
  const bool inb = (int)threadIdx.x < nb;
  // Everything addressed by the block number alone is requested here, in one batch: the union's offsets and the first
  // partial (scalar loads), the cell's slots, mark and connectivity row.  The kernel is bound by the length of its
  // chain of dependent loads x the blocks a CU holds (configs[3]: 786 k blocks of 2 wavefronts, ~10 resident per CU),
  // not by bytes or arithmetic: block -> {mark, slots} -> connectivity -> vertices -> ... -> offsets -> segment bounds
  // was six levels, this is three
  const int64_t ub = P.u_off[k], pb = P.base[k];
  const int nu = (int)(P.u_off[k + 1] - ub);
  // (the slots stay packed two to a register until they are used: 100 -> 96 registers is a fifth wavefront per SIMD)
  constexpr int NW = (ND + 1) / 2;
  uint32_t slw[NW];
  if (inb)
  {
    if constexpr (ND % 2 == 0)
    {
      const uint32_t* w = reinterpret_cast<const uint32_t*>(P.slot + c * ND);
#pragma unroll
      for (int j = 0; j < NW; ++j) slw[j] = w[j];
    }
    else
    {
#pragma unroll
      for (int j = 0; j < NW; ++j)
        slw[j] = (uint32_t)P.slot[c * ND + 2 * j] | (2 * j + 1 < ND ? (uint32_t)P.slot[c * ND + 2 * j + 1] << 16 : 0u);
    }
  }
  const unsigned cm = inb ? P.cellmark[c] : 0u;
  int32_t vids[TDIM + 1];
  if constexpr (RULES != 1)
  {
    if constexpr (TDIM == 3)
    {
      const int4 r = *reinterpret_cast<const int4*>(A.conn + (inb ? c : k * B) * 4);
      vids[0] = r.x; vids[1] = r.y; vids[2] = r.z; vids[3] = r.w;
    }
    else
    {
#pragma unroll
      for (int i = 0; i <= TDIM; ++i) vids[i] = A.conn[(inb ? c : k * B) * (TDIM + 1) + i];
    }
  }
  // the segment of this thread's first dof of the union (the later rounds are requested after the arithmetic, when
  // registers are free)
  const int t0 = threadIdx.x;
  const uint32_t f01 = (t0 < nu ? (uint32_t)P.seg[ub + t0] : 0u) | ((t0 + 1 < nu ? (uint32_t)P.seg[ub + t0 + 1] : (uint32_t)(nb * ND)) << 16);
  double be[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j) be[j] = 0.0;
  if constexpr (RULES != 0) add_rule_vectors<ND>(P, c, cm, be);
  if constexpr (RULES != 1)
  {
    if (cm & P.mark) entity_vector<TDIM, DEG, false, 1>(A, 0, c, 0, be, vids);
  }
  if (inb)
  {
#pragma unroll
    for (int j = 0; j < ND; ++j) s_val[(slw[j / 2] >> (16 * (j & 1))) & 0xffffu] = be[j];
  }
  __syncthreads();
  const bool add = RULES == 1 && entry >= 0;
  const int f0 = (int)(f01 & 0xffffu), f1 = (int)(f01 >> 16);
  constexpr int T = 3; // further rounds whose bounds are requested together
  int g0[T], g1[T];
#pragma unroll
  for (int q = 0; q < T; ++q)
  {
    const int t = t0 + (q + 1) * B;
    g0[q] = t < nu ? (int)P.seg[ub + t] : 0;
    g1[q] = t + 1 < nu ? (int)P.seg[ub + t + 1] : nb * ND;
  }
  if (t0 < nu)
  {
    double sum = 0.0;
    for (int i = f0; i < f1; ++i) sum += s_val[i];
    if (add) P.part[pb + t0] += sum; else P.part[pb + t0] = sum;
  }
#pragma unroll
  for (int q = 0; q < T; ++q)
  {
    const int t = t0 + (q + 1) * B;
    if (t < nu)
    {
      double sum = 0.0;
      for (int i = g0[q]; i < g1[q]; ++i) sum += s_val[i];
      if (add) P.part[pb + t] += sum; else P.part[pb + t] = sum;
    }
  }
  for (int t = t0 + (T + 1) * B; t < nu; t += B)
  {
    const int a0 = P.seg[ub + t], a1 = t + 1 < nu ? (int)P.seg[ub + t + 1] : nb * ND;
    double sum = 0.0;
    for (int i = a0; i < a1; ++i) sum += s_val[i];
    if (add) P.part[pb + t] += sum; else P.part[pb + t] = sum;
  }
}

// ... the sin-product source term on a P1 space (vec_source_sin_p1_kernel's arithmetic and software pipeline: the
// workgroup walks its blocks with the vertices + segment bounds of the next block and the connectivity + offsets of
// the one after in flight; the LDS values are double-buffered, one barrier per block)
template <int TDIM>
__global__ void __launch_bounds__(kBlock, CFX_SOURCE_WAVES) vec_blocks_sin_p1_kernel(VecArgs A, VecBlockArgs P)
{
  constexpr int ND = TDIM + 1, B = kBlock;
  __shared__ double s_val[2][B * ND];
  int npts;
  const double* wts;
  const double* pts = ref_rule(TDIM, A.qdegree, npts, wts);
  const double fscale = A.params[1] * ((int)A.params[0] == CFX_F_POISSON_RHS ? (double)TDIM * kPi * kPi : 1.0);
  const int64_t nt = P.n_active, stride = gridDim.x, last = P.ncells - 1;
  auto blk = [&](int64_t i) { return (int64_t)P.active[i < nt ? i : nt - 1]; }; // (clamped: the tail loads the last block again)
  struct Extra
  {
    bool on;
    uint16_t sl[ND];
    int64_t k, ub, pb; // block, first dof of its union, first partial
    int nu, nb, s0, s1; // dofs of the union, cells of the block, this thread's segment
  };
  auto stage = [&](int64_t k, SourceCell<TDIM>& c, Extra& x)
  {
    const int64_t cell = k * B + threadIdx.x;
    c.cell = (int32_t)(cell < last ? cell : last);
    source_load_conn<TDIM>(A, c);
    x.on = cell <= last && (P.cellmark[c.cell] & P.mark) != 0;
    load_slots<ND>(P.slot, c.cell, x.sl);
    x.k = k; x.ub = P.u_off[k]; x.nu = (int)(P.u_off[k + 1] - x.ub); x.pb = P.base[k];
    x.nb = (int)min((int64_t)B, P.ncells - k * B);
  };
  auto stage2 = [&](SourceCell<TDIM>& c, Extra& x)
  {
    source_load_vertices<TDIM>(A, c);
    const int t = threadIdx.x;
    x.s0 = t < x.nu ? (int)P.seg[x.ub + t] : 0;
    x.s1 = t + 1 < x.nu ? (int)P.seg[x.ub + t + 1] : x.nb * ND;
  };
  SourceCell<TDIM> a, b;
  Extra xa, xb;
  const int64_t i0 = blockIdx.x;
  int64_t k_c = blk(i0 + 2 * stride);
  stage(blk(i0), a, xa);
  stage(blk(i0 + stride), b, xb);
  stage2(a, xa);
  int buf = 0;
  for (int64_t i = i0; i < nt; i += stride)
  {
    const int64_t k_d = blk(i + 3 * stride);
    SourceCell<TDIM> c;
    Extra xc;
    stage(k_c, c, xc);
    stage2(b, xb);
    double be[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) be[j] = 0.0;
    if (__ballot(xa.on) != 0) source_vector<TDIM>(a, npts, pts, wts, fscale, be); // (a wavefront without a marked cell: the ends of a block)
    double* sv = s_val[buf];
    if ((int)threadIdx.x < xa.nb)
    {
#pragma unroll
      for (int j = 0; j < ND; ++j) sv[xa.sl[j]] = xa.on ? be[j] : 0.0;
    }
    __syncthreads(); // (the other buffer is written by the next trip: its readers of the previous trip have passed this barrier)
    if ((int)threadIdx.x < xa.nu)
    {
      double sum = 0.0;
      for (int q = xa.s0; q < xa.s1; ++q) sum += sv[q];
      P.part[xa.pb + threadIdx.x] = sum;
    }
    for (int t = threadIdx.x + B; t < xa.nu; t += B) // a union longer than the block (not on box meshes)
    {
      const int s0 = P.seg[xa.ub + t], s1 = t + 1 < xa.nu ? (int)P.seg[xa.ub + t + 1] : xa.nb * ND;
      double sum = 0.0;
      for (int q = s0; q < s1; ++q) sum += sv[q];
      P.part[xa.pb + t] = sum;
    }
    buf ^= 1;
    a = b; xa = xb;
    b = c; xb = xc;
    k_c = k_d;
  }
}

// stage 2: b[r] += the partials of the blocks around row r, in ascending block order (then a fixed tree over the
// row's lanes: bitwise reproducible).  base[] = -1 for a block without a marked cell (nothing was written for it)
template <int G>
__global__ void __launch_bounds__(kWave) vec_blocks_rows_kernel(DevN n_d, const int32_t* __restrict__ rows,
                                                                const int64_t* __restrict__ p_off,
                                                                const int64_t* __restrict__ p_pos,
                                                                const int64_t* __restrict__ base,
                                                                const double* __restrict__ part, double* __restrict__ b)
{
  const int64_t n = dev_n(n_d);
  const int lane = threadIdx.x, gl = lane % G;
  const int64_t i = CFX_ROW_BLOCK * (kWave / G) + lane / G;
  const bool live = i < n;
  const int64_t r = live ? rows[i] : 0;
  const int64_t o0 = live ? p_off[r] : 0;
  const int np = live ? (int)(p_off[r + 1] - o0) : 0;
  double sum = 0.0;
  constexpr int Q = 2; // the first Q entries of the lane are requested together
  int64_t e[Q], bb[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) e[q] = gl + q * G < np ? p_pos[o0 + gl + q * G] : -1;
#pragma unroll
  for (int q = 0; q < Q; ++q) bb[q] = e[q] >= 0 ? base[e[q] >> 11] : -1;
#pragma unroll
  for (int q = 0; q < Q; ++q)
    if (bb[q] >= 0) sum += part[bb[q] + (e[q] & 2047)];
  for (int q = gl + Q * G; q < np; q += G)
  {
    const int64_t ee = p_pos[o0 + q], b0 = base[ee >> 11];
    if (b0 >= 0) sum += part[b0 + (ee & 2047)];
  }
#pragma unroll
  for (int d = G / 2; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, G);
  if (live && gl == 0) b[r] += sum;
}

// ---------------------------------------------------------------------------
// stage 2, bilinear forms
// ---------------------------------------------------------------------------
// (degree 2: 10 columns per cell item, 20 per facet item: 2 waves/SIMD leave 256 VGPRs)
// CUTS = false compiles the rule-tensor and facet items out: the lean form that serves the uncut
// items of every row of a degree-2 space (the facet section alone holds 100 registers there).
// STD = false compiles the uncut-cell items out: the form that serves only the rule / facet items
// of the interface rows (mark_mask 0xF0) next to a lean kernel for everything else.
// CUTT: the cut cells' contributions come as one staged tensor per cut cell (A.cut_tensors): the per-integral rule
// lookups, the moments and the closed-form code are not compiled in (degree 2: 157 -> under 128 VGPRs)
template <int TDIM, int DEG, int G, int CAP, bool ORDERED, bool CUTS = true, bool STD = true, bool CUTT = false>
__global__ void __launch_bounds__(kWave, DEG > 1 ? (CUTS ? (CUTT ? (STD ? 3 : 4) : 2) : 3) : (STD ? CFX_ROWS_WAVES : CFX_ROWS_CUT_WAVES)) assemble_rows_kernel(RowArgs A)
{
  constexpr int ND = Elem<TDIM, DEG>::ND;
  constexpr int W = 2 * ND; // widest item: a facet's macro row
  constexpr int RPW = kWave / G;
  // one element of padding per row: the groups of a wavefront probe the same position of
  // their own rows, which would otherwise sit in the same LDS banks
  __shared__ int32_t s_col[RPW][CAP + 1];
  __shared__ double s_val[RPW][CAP + 1];
  const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
  const int64_t ri = CFX_ROW_BLOCK * RPW + grp;
  const bool live = ri < dev_n(A.n_active);
  const int64_t r = live ? A.active_rows[ri] : 0;
  const int64_t rb = live ? A.indptr[r] : 0;
  int len = live ? (int)(A.indptr[r + 1] - rb) : 0;
  if (len > CAP) { *A.error = 2; len = 0; }
  // Row prologue: this lane's slice of the row's columns, requested in one batch
  constexpr int KMAX = CAP / G;
  int32_t mycol[KMAX];
#pragma unroll
  for (int q = 0; q < KMAX; ++q)
  {
    const int k = gl + q * G;
    mycol[q] = k < len ? A.indices[rb + k] : -1;
  }
#pragma unroll
  for (int q = 0; q < KMAX; ++q)
  {
    const int k = gl + q * G;
    if (k < len)
    {
      s_col[grp][k] = mycol[q];
      s_val[grp][k] = 0.0;
    }
  }
  __syncthreads();
  const bool cells = live && A.cellmark != nullptr;
  const int64_t cb = cells ? A.d2c_off[r] : 0;
  const int nc = cells ? (int)(A.d2c_off[r + 1] - cb) : 0;
  const bool facets = live && A.d2f_off != nullptr && (A.mark_mask & 0xF0u) != 0;
  const int64_t fpos = (facets && A.special_mark[r]) ? (int64_t)A.special_pos[r] : -1; // incidence is per special row
  const int64_t fb = fpos >= 0 ? A.d2f_off[fpos] : 0;
  const int nf = fpos >= 0 ? (int)(A.d2f_off[fpos + 1] - fb) : 0;
  const bool row_bc = live && A.bc0 && A.bc0[r];
  // every cell item adds to the diagonal: kept in a register per lane and reduced over the
  // group once, instead of ~24 LDS atomics on one address per row
  double dsum = 0.0;
  const bool diag_bc = row_bc || (live && A.bc1 != nullptr && A.bc1[r] != 0);

  // add one item (ncols columns) of every group to its row; ORDERED keeps item order
  // CSR slot of column `col` of this row (-1 and the error flag if absent)
  auto find_slot = [&](int32_t col) -> int
  {
    int lo = 0, hi = len;
    while (lo < hi)
    {
      const int mid = (lo + hi) >> 1;
      if (s_col[grp][mid] < col) lo = mid + 1; else hi = mid;
    }
    if (lo < len && s_col[grp][lo] == col) return lo;
    *A.error = 1;
    return -1;
  };
  // add one item of NC columns to the row; `sl` holds the CSR slot of each column
  auto add_item = [&](auto nc_tag, bool has, const int32_t* cols, double* acc, const int* sl)
  {
    constexpr int NC = decltype(nc_tag)::value;
    // (slot, value) pairs of this lane's item; BC rows / columns are zeroed here
    // (assemble_matrix_impl.h:151-185)
    double v[NC];
    int s[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j)
    {
      s[j] = has ? sl[j] : -1;
      const bool bc = has && (row_bc || (A.bc1 != nullptr && A.bc1[cols[j]] != 0));
      v[j] = bc ? 0.0 : acc[j];
    }
    if constexpr (ORDERED)
    {
      for (int turn = 0; turn < G; ++turn) // one lane of each group at a time: item order
      {
        if (gl == turn)
        {
#pragma unroll
          for (int j = 0; j < NC; ++j)
            if (s[j] >= 0) s_val[grp][s[j]] += v[j];
        }
        __syncthreads(); // block = one wavefront: orders the LDS read-modify-writes of successive turns
      }
    }
    else
    {
#pragma unroll
      for (int j = 0; j < NC; ++j)
        if (s[j] >= 0) atomicAdd(&s_val[grp][s[j]], v[j]);
    }
  };

  // ---- cell items, R per lane per pass.  The index loads of a pass (incidence
  // list, marks, dof rows) are issued together so that their latencies overlap:
  // the kernel is bound by dependent gathers, not by bandwidth or flops.
  constexpr int R = DEG > 1 ? 1 : (G <= 4 ? 6 : 3); // P1: 3 x 8 lanes cover the 24 tets around a Kuhn-mesh vertex in one pass
  const int ncl = len > 0 ? nc : 0;
  for (int base = 0;; base += R * G)
  {
    if (__ballot(base + gl < ncl) == 0) break;
    int32_t cell[R];
    uint8_t mk[R];
    int32_t cd[R][ND];
#pragma unroll
    for (int k = 0; k < R; ++k)
    {
      const int t = base + k * G + gl;
      cell[k] = t < ncl ? A.d2c[cb + t] : -1;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) mk[k] = cell[k] >= 0 ? (uint8_t)(A.cellmark[cell[k]] & A.mark_mask) : (uint8_t)0;
#pragma unroll
    for (int k = 0; k < R; ++k)
    {
      if (!mk[k]) continue;
      if constexpr (ND == 4)
      {
        const int4 v = *reinterpret_cast<const int4*>(A.dofmap + (int64_t)cell[k] * 4);
        cd[k][0] = v.x; cd[k][1] = v.y; cd[k][2] = v.z; cd[k][3] = v.w;
      }
      else
      {
#pragma unroll
        for (int j = 0; j < ND; ++j) cd[k][j] = A.dofmap[(int64_t)cell[k] * ND + j];
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k)
    {
      if (__ballot(mk[k] != 0) == 0) continue;
#ifndef CFX_ROWS_NO_SCHED_BARRIER
      // keep the items sequential: hoisting the coordinate loads of all R items above the
      // first item's arithmetic costs more registers than the 5 waves/SIMD budget has
      __builtin_amdgcn_sched_barrier(0);
#endif
      const int64_t c = cell[k];
      const uint8_t mark = mk[k];
      // local column of r itself (cd[k][lr] == r for every marked item)
      int lr = 0;
#pragma unroll
      for (int j = 0; j < ND; ++j) lr = (cd[k][j] == (int32_t)r) ? j : lr;
      {
        double acc[ND];
        int csl[ND];
#pragma unroll
        for (int j = 0; j < ND; ++j) { acc[j] = 0.0; csl[j] = -1; }
        if (mk[k])
        {
#pragma unroll
          for (int j = 0; j < ND; ++j) csl[j] = find_slot(cd[k][j]);
          for (int i = 0; i < A.n_cell; ++i)
          {
            const RowIntegral& I = A.cell[i];
            if (STD && (mark & (1u << i)))
            {
              if (DEG == 2 && I.std_inline == 3)
              {
                if constexpr (DEG == 2)
                {
                  // degree-2 stiffness on the affine cell in closed form (p2_stiffness_row): nothing staged
                  Geo<TDIM> g;
                  load_cell<TDIM>(A.x, A.conn, c, g);
                  jacobian<TDIM>(g);
                  p2_stiffness_row<TDIM>(g, lr, 1.0, acc);
                }
              }
              else if (kRowsInline<DEG> && I.std_inline)
              {
                if constexpr (kRowsInline<DEG>)
                {
                  Geo<TDIM> g;
                  load_cell<TDIM>(A.x, A.conn, c, g);
                  jacobian<TDIM>(g);
                  int npts;
                  const double* wts;
                  const double* pts = ref_rule(TDIM, I.qdegree, npts, wts);
                  cell_local_row<TDIM, DEG, 1, 2>(I.kernel, I.params, I.point_stride, g, 0.0, npts, pts, wts,
                                                  fabs(g.detJ), nullptr, lr, 0, acc);
                }
              }
              else
              {
                const int64_t e = entity_index(I.std_bits, I.std_rank, c);
                const double* T = I.std_tensors + (e * ND + lr) * ND;
#pragma unroll
                for (int j = 0; j < ND; ++j) acc[j] += T[j];
              }
            }
            if (CUTS && !CUTT && (mark & (16u << i)))
            {
              if (DEG == 2 && I.rule_moments)
              {
                if constexpr (DEG == 2)
                {
                  // degree-2 stiffness over the cut part from the rule's moments (p2_stiffness_row_moments)
                  Geo<TDIM> g;
                  load_cell<TDIM>(A.x, A.conn, c, g);
                  jacobian<TDIM>(g);
                  for (int64_t e = first_rule(I.rule_keys, I.rule_first, I.rule_mask, (int32_t)c); e < dev_len(I.nr) && I.parent_map[e] == c; ++e)
                  {
                    const double2* mp = reinterpret_cast<const double2*>(I.rule_tensors + e * 16);
                    double mom[16];
#pragma unroll
                    for (int k2 = 0; k2 < 8; ++k2)
                    {
                      const double2 v = mp[k2];
                      mom[2 * k2] = v.x; mom[2 * k2 + 1] = v.y;
                    }
                    p2_stiffness_row_moments<TDIM>(g, lr, mom, acc);
                  }
                }
              }
              else
              for (int64_t e = first_rule(I.rule_keys, I.rule_first, I.rule_mask, (int32_t)c); e < dev_len(I.nr) && I.parent_map[e] == c; ++e)
              {
                const double* T = I.rule_tensors + (e * ND + lr) * ND;
#pragma unroll
                for (int j = 0; j < ND; ++j) acc[j] += T[j];
              }
            }
          }
          if constexpr (CUTS && CUTT)
          {
            if (mark & 0xF0u)
            {
              const int64_t e = entity_index(A.cut_bits, A.cut_rank, c);
              const double2* T = reinterpret_cast<const double2*>(A.cut_tensors + (e * ND + lr) * ND);
              static_assert(ND % 2 == 0, "rows of the cut tensors are read as 16 B pairs");
#pragma unroll
              for (int j = 0; j < ND / 2; ++j)
              {
                const double2 v = T[j];
                acc[2 * j] += v.x; acc[2 * j + 1] += v.y;
              }
            }
          }
#pragma unroll
          for (int j = 0; j < ND; ++j)
            if (j == lr) { dsum += diag_bc ? 0.0 : acc[j]; csl[j] = -1; }
        }
        add_item(std::integral_constant<int, ND>{}, mk[k] != 0, cd[k], acc, csl);
      }
    }
  }
  {
    double d = dsum;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, G);
    if (live && gl == 0 && len > 0 && nc > 0)
    {
      const int slot = find_slot((int32_t)r);
      if (slot >= 0) atomicAdd(&s_val[grp][slot], d);
    }
  }

  // ---- facet items (rows next to the interface only)
  const int nfl = (CUTS && len > 0) ? nf : 0;
  for (int base = 0; CUTS; base += G)
  {
    const int t = base + gl;
    const bool has = t < nfl;
    if (__ballot(has) == 0) break;
    if (DEG == 1 && A.fold_facets == 3)
    {
      if constexpr (DEG == 1)
      {
        // P1 gradient jump: one 80-byte record per facet -- (jf[0..ND], w) and the ND + 1 macro columns; row m(r) of
        // the folded tensor = w jf[m] jf[.].  Neither the facet row nor the dofmap is read
        constexpr int WF = ND + 1;
        double acc[WF];
        int32_t cm[WF];
        int sl[WF];
#pragma unroll
        for (int j = 0; j < WF; ++j) { acc[j] = 0.0; cm[j] = -1; sl[j] = -1; }
        if (has)
        {
          const int64_t f = A.d2f[fb + t];
          const double2* rec = reinterpret_cast<const double2*>(A.facet_tensors + f * 10);
          const double2 r0 = rec[0], r1 = rec[1], r2 = rec[2];
          const int4 c0 = *reinterpret_cast<const int4*>(rec + 3), c1 = *reinterpret_cast<const int4*>(rec + 4);
          const double jf[6] = {r0.x, r0.y, r1.x, r1.y, r2.x, r2.y};
          const int32_t c8[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
          int m = -1;
#pragma unroll
          for (int j = 0; j < WF; ++j) { cm[j] = c8[j]; m = (c8[j] == (int32_t)r) ? j : m; }
          if (c8[7] != 1) *A.error = 4; // not an interior facet of a conforming mesh with a continuous space
          double jm = 0.0;
#pragma unroll
          for (int j = 0; j < WF; ++j) jm = (j == m) ? jf[j] : jm;
          jm *= jf[ND + 1];
#pragma unroll
          for (int j = 0; j < WF; ++j) acc[j] = jm * jf[j];
#pragma unroll
          for (int j = 0; j < WF; ++j) sl[j] = find_slot(cm[j]);
        }
        add_item(std::integral_constant<int, WF>{}, has, cm, acc, sl);
      }
      continue;
    }
    if (DEG == 2 && A.fold_facets == 3)
    {
      if constexpr (DEG == 2)
      {
        // degree-2 gradient jump: stage 1 stored facet_nq rank-one records (jf_q[0..WF), w_q) per facet, folded over
        // the dofs the two cells share; row of macro dof m = sum_q w_q jf_q[m] jf_q[.], WF columns instead of 2 ND
        constexpr int WF = Elem<TDIM, DEG>::WF, NX = WF - ND;
        double acc[WF];
        int32_t cm[WF];
        int sl[WF];
#pragma unroll
        for (int j = 0; j < WF; ++j) { acc[j] = 0.0; cm[j] = -1; sl[j] = -1; }
        if (has)
        {
          const int64_t f = A.d2f[fb + t];
          // the facet's block: its macro columns (16 int32), then the records
          const double* blk = A.facet_tensors + f * (int64_t)(8 + 16 * A.facet_nq);
          const int4* cmp = reinterpret_cast<const int4*>(blk);
          const int4 q0 = cmp[0], q1 = cmp[1], q2 = cmp[2], q3 = cmp[3];
          const int32_t c16[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
          int m = -1;
#pragma unroll
          for (int j = 0; j < WF; ++j) { cm[j] = c16[j]; m = (c16[j] == (int32_t)r) ? j : m; }
          if (c16[15] != NX) *A.error = 4; // not an interior facet of a conforming mesh with a continuous space
          const double2* rec = reinterpret_cast<const double2*>(blk + 8);
          for (int q = 0; q < A.facet_nq; ++q)
          {
            double v[16];
#pragma unroll
            for (int k = 0; k < 8; ++k)
            {
              const double2 p = rec[q * 8 + k];
              v[2 * k] = p.x; v[2 * k + 1] = p.y;
            }
            double jm = 0.0;
#pragma unroll
            for (int j = 0; j < WF; ++j) jm = (j == m) ? v[j] : jm;
            jm *= v[WF];
#pragma unroll
            for (int j = 0; j < WF; ++j) acc[j] += jm * v[j];
          }
#pragma unroll
          for (int j = 0; j < WF; ++j) sl[j] = find_slot(cm[j]);
        }
        add_item(std::integral_constant<int, WF>{}, has, cm, acc, sl);
      }
      continue;
    }
    double acc[W];
    int32_t cols[W];
#pragma unroll
    for (int j = 0; j < W; ++j) { acc[j] = 0.0; cols[j] = -1; }
    if (has)
    {
      const int64_t f = A.d2f[fb + t];
      const int4 row4 = *reinterpret_cast<const int4*>(A.facet_rows + 4 * f);
      if constexpr (ND == 4)
      {
        const int4 d0 = *reinterpret_cast<const int4*>(A.dofmap + (int64_t)row4.x * 4);
        const int4 d1 = *reinterpret_cast<const int4*>(A.dofmap + (int64_t)row4.z * 4);
        cols[0] = d0.x; cols[1] = d0.y; cols[2] = d0.z; cols[3] = d0.w;
        cols[4] = d1.x; cols[5] = d1.y; cols[6] = d1.z; cols[7] = d1.w;
      }
      else
      {
#pragma unroll
        for (int j = 0; j < ND; ++j)
        {
          cols[j] = A.dofmap[(int64_t)row4.x * ND + j];
          cols[ND + j] = A.dofmap[(int64_t)row4.z * ND + j];
        }
      }
      // r may be a dof of both cells: both macro rows land in global row r
      int i0 = -1, i1 = -1;
#pragma unroll
      for (int j = 0; j < ND; ++j)
      {
        i0 = cols[j] == (int32_t)r ? j : i0;
        i1 = cols[ND + j] == (int32_t)r ? ND + j : i1;
      }
      if (DEG == 1 && A.fold_facets == 2)
      {
        // stage 1 stored the tensor folded over the shared dofs: macro row of r = its index in cell 0, or ND
        // (the dof of cell 1 that cell 0 does not have); one row of ND + 1 entries, the free column last
        constexpr int WF = ND + 1;
        const double* T = A.facet_tensors + f * (WF * WF) + (i0 >= 0 ? i0 : ND) * WF;
#pragma unroll
        for (int j = 0; j < WF; ++j) acc[j] = T[j];
      }
      else
      {
      const double* T = A.facet_tensors + f * (W * W);
      if (i0 >= 0)
      {
#pragma unroll
        for (int j = 0; j < W; ++j) acc[j] += T[i0 * W + j];
      }
      if (i1 >= 0)
      {
#pragma unroll
        for (int j = 0; j < W; ++j) acc[j] += T[i1 * W + j];
      }
      }
    }
    if (DEG == 1 && A.fold_facets == 2)
    {
     if constexpr (DEG == 1)
     {
      // columns: cell 0's dofs, then cell 1's free dof
      int32_t c5[ND + 1];
      double a5[ND + 1];
      int s5[ND + 1];
      int32_t ocol = -1;
      int nfree = 0;
#pragma unroll
      for (int j = 0; j < ND; ++j)
      {
        bool shared = false;
#pragma unroll
        for (int i = 0; i < ND; ++i) shared = shared || cols[ND + j] == cols[i];
        if (has && !shared) { ocol = cols[ND + j]; ++nfree; }
      }
      if (has && nfree != 1) *A.error = 4;
#pragma unroll
      for (int j = 0; j < ND; ++j) { c5[j] = cols[j]; a5[j] = acc[j]; s5[j] = -1; }
      c5[ND] = ocol; a5[ND] = acc[ND]; s5[ND] = -1;
      if (has)
      {
#pragma unroll
        for (int j = 0; j <= ND; ++j) s5[j] = find_slot(c5[j]);
      }
      add_item(std::integral_constant<int, ND + 1>{}, has, c5, a5, s5);
     }
    }
    else if (DEG == 1 && A.fold_facets)
    {
     if constexpr (DEG == 1)
     {
      // P1: the two cells share the facet's TDIM vertices, so the macro row has ND + 1
      // distinct columns: fold the shared dofs of cell 1 onto cell 0's, 5 slot searches
      // and LDS adds instead of 8
      int32_t c5[ND + 1];
      double a5[ND + 1];
      int s5[ND + 1];
      int nfree = 0;
      int32_t ocol = -1;
      double oacc = 0.0;
#pragma unroll
      for (int j = 0; j < ND; ++j)
      {
        bool shared = false;
#pragma unroll
        for (int i = 0; i < ND; ++i)
          if (has && cols[ND + j] == cols[i]) { acc[i] += acc[ND + j]; shared = true; }
        if (has && !shared) { ocol = cols[ND + j]; oacc = acc[ND + j]; ++nfree; }
      }
      if (has && nfree != 1) *A.error = 4; // not an interior facet of a conforming simplicial mesh
#pragma unroll
      for (int j = 0; j < ND; ++j) { c5[j] = cols[j]; a5[j] = acc[j]; s5[j] = -1; }
      c5[ND] = ocol; a5[ND] = oacc; s5[ND] = -1;
      if (has)
      {
#pragma unroll
        for (int j = 0; j <= ND; ++j) s5[j] = find_slot(c5[j]);
      }
      add_item(std::integral_constant<int, ND + 1>{}, has, c5, a5, s5);
     }
    }
    else
    {
      int fsl[W];
#pragma unroll
      for (int j = 0; j < W; ++j) fsl[j] = has ? find_slot(cols[j]) : -1;
      add_item(std::integral_constant<int, W>{}, has, cols, acc, fsl);
    }
  }
  __syncthreads();
  if (A.fresh == 2)
  {
    // first and only writer of these rows on a matrix that holds nothing yet (set_value(0) fused): stored
#pragma unroll
    for (int q = 0; q < KMAX; ++q)
    {
      const int k = gl + q * G;
      if (k < len) A.values[rb + k] = s_val[grp][k];
    }
    return;
  }
  // Row epilogue: values[row] += reduced row, loads batched
  double myval[KMAX];
#pragma unroll
  for (int q = 0; q < KMAX; ++q)
  {
    const int k = gl + q * G;
    myval[q] = k < len ? A.values[rb + k] : 0.0;
  }
#pragma unroll
  for (int q = 0; q < KMAX; ++q)
  {
    const int k = gl + q * G;
    if (k < len) A.values[rb + k] = myval[q] + s_val[grp][k];
  }
}

// ---------------------------------------------------------------------------
// stage 2, bilinear forms, degree 2 (scalar): the INTERFACE rows in one pass -- the uncut items (closed-form stiffness
// row), the cut-cell items (one row of the cell's combined tensor, cut_tensors_p2_kernel) and the gradient-jump facet
// items (rank-one records) of a row, stored when the matrix holds nothing yet.  assemble_rows_kernel walks the cell
// items and then the facet items, and inside an item one load after the other as the branches ask for them: at three
// wavefronts per SIMD that kernel spent two thirds of its time parked on ~8 + 4 dependent gathers per pass (47 -> 40 ms
// at BASELINE config 4).  Here a lane takes the t-th cell item AND the t-th facet item of its row in the same pass and
// everything that depends on the same index is requested together:
//   row -> {row pointer, incidence ranges} -> {cell id, facet id} -> {mark, dof row, connectivity row, cut bit word /
//   rank, facet column ids} -> {vertices, tensor row} -> cell item -> {facet records} -> facet item.
// ---------------------------------------------------------------------------
#ifndef CFX_IFC_WAVES
#define CFX_IFC_WAVES 4 // waves per SIMD the interface kernel is compiled for (128 registers, 11 spilled: 25.4 -> 24.6 ms; the
                        // form with the probe loops alone was indifferent to 2 / 3 / 4)
#endif
template <int TDIM, int G, int CAP, bool ORDERED>
__global__ void __launch_bounds__(kWave, CFX_IFC_WAVES) assemble_rows_p2_interface_kernel(RowArgs A)
{
  constexpr int DEG = 2, ND = Elem<TDIM, DEG>::ND, NV = TDIM + 1, WF = Elem<TDIM, DEG>::WF, NX = WF - ND;
  constexpr int RPW = kWave / G;
  // the row's columns as an open-addressing hash map column -> CSR slot (2 CAP keys per row: load factor <= 1/2): a
  // lookup is a hash and one or two LDS reads where the binary search over the sorted column list took 8 dependent
  // compare steps -- 24 lookups per pass were 770 of the ~1000 VALU instructions of a pass, and the kernel's VALU
  // issue alone was half its run time (profiles/r03: 2964 VALU wave-instructions per wavefront)
  constexpr int HS = 2 * CAP;
  static_assert((HS & (HS - 1)) == 0, "power of two");
  constexpr int kHashShift = HS == 128 ? 25 : (HS == 256 ? 24 : (HS == 512 ? 23 : 22)); // multiplicative hash: the top log2(HS) bits
  static_assert(HS == 128 || HS == 256 || HS == 512 || HS == 1024, "hash shift");
  __shared__ int32_t s_key[RPW][HS];
  __shared__ uint16_t s_slot[RPW][HS];
  __shared__ double s_val[RPW][CAP + 1];
  const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
  const int64_t ri = CFX_ROW_BLOCK * RPW + grp;
  const bool live = ri < dev_n(A.n_active);
  const int64_t r = live ? A.active_rows[ri] : 0;
  const int64_t rb = live ? A.indptr[r] : 0;
  int len = live ? (int)(A.indptr[r + 1] - rb) : 0;
  if (len > CAP) { *A.error = 2; len = 0; }
  const int64_t cb = live ? A.d2c_off[r] : 0;
  const int nc = (live && len > 0) ? (int)(A.d2c_off[r + 1] - cb) : 0;
  const bool facets = live && len > 0 && A.d2f_off != nullptr;
  const int64_t fpos = (facets && A.special_mark[r]) ? (int64_t)A.special_pos[r] : -1;
  const int64_t fb = fpos >= 0 ? A.d2f_off[fpos] : 0;
  const int nf = fpos >= 0 ? (int)(A.d2f_off[fpos + 1] - fb) : 0;
  constexpr int KMAX = CAP / G;
  for (int k = gl; k < HS; k += G) s_key[grp][k] = -1;
  __syncthreads();
  {
    int32_t mycol[KMAX];
#pragma unroll
    for (int q = 0; q < KMAX; ++q)
    {
      const int k = gl + q * G;
      mycol[q] = k < len ? A.indices[rb + k] : -1;
    }
#pragma unroll
    for (int q = 0; q < KMAX; ++q)
    {
      const int k = gl + q * G;
      if (k < len)
      {
        s_val[grp][k] = 0.0;
        unsigned h = ((uint32_t)mycol[q] * 2654435761u) >> kHashShift;
        while (atomicCAS(&s_key[grp][h], -1, mycol[q]) != -1) h = (h + 1) & (HS - 1); // (columns of a row are distinct)
        s_slot[grp][h] = (uint16_t)k;
      }
    }
  }
  __syncthreads();
  const bool row_bc = live && A.bc0 && A.bc0[r];
  const bool diag_bc = row_bc || (live && A.bc1 != nullptr && A.bc1[r] != 0);
  double dsum = 0.0;
  auto find_slot = [&](int32_t col) -> int
  {
    // the first two positions without a branch (load factor <= 1/2: ~9 of 10 lookups end there), both keys in flight
    // together; the probe loop only behind them (26 loops with data-dependent trip counts per pass were a good part
    // of the kernel's scalar instructions: as many as vector ones)
    const unsigned h0 = ((uint32_t)col * 2654435761u) >> kHashShift, h1 = (h0 + 1) & (HS - 1);
    const int32_t k0 = s_key[grp][h0], k1 = s_key[grp][h1];
    const int v0 = s_slot[grp][h0], v1 = s_slot[grp][h1]; // (requested with the keys: no dependent LDS read on a hit)
    if (k0 == col) return v0;
    if (k1 == col) return v1;
    if (k0 != -1 && k1 != -1)
    {
      unsigned h = (h1 + 1) & (HS - 1);
      for (int probe = 2; probe < HS; ++probe)
      {
        const int32_t key = s_key[grp][h];
        if (key == col) return (int)s_slot[grp][h];
        if (key == -1) break;
        h = (h + 1) & (HS - 1);
      }
    }
    *A.error = 1;
    return -1;
  };
  auto add_item = [&](auto nc_tag, bool has, const int32_t* cols, const double* acc, const int* sl)
  {
    constexpr int NC = decltype(nc_tag)::value;
    double v[NC];
    int sidx[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j)
    {
      sidx[j] = has ? sl[j] : -1;
      const bool bc = has && (row_bc || (A.bc1 != nullptr && A.bc1[cols[j]] != 0));
      v[j] = bc ? 0.0 : acc[j];
    }
    if constexpr (ORDERED)
    {
      for (int turn = 0; turn < G; ++turn)
      {
        if (gl == turn)
        {
#pragma unroll
          for (int j = 0; j < NC; ++j)
            if (sidx[j] >= 0) s_val[grp][sidx[j]] += v[j];
        }
        __syncthreads();
      }
    }
    else
    {
#pragma unroll
      for (int j = 0; j < NC; ++j)
        if (sidx[j] >= 0) atomicAdd(&s_val[grp][sidx[j]], v[j]);
    }
  };
  int npass = max((nc + G - 1) / G, (nf + G - 1) / G);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) npass = max(npass, __shfl_xor(npass, o, 64)); // (ORDERED: the groups' barriers match)
  const int64_t last_c = nc > 0 ? cb : 0; // a valid incidence entry for the lanes without an item
  for (int p = 0; p < npass; ++p)
  {
    const int t = p * G + gl;
    const bool hasc = t < nc, hasf = t < nf;
    // ---- level 1: the item ids (lanes without an item re-read a valid entry: every load below is unconditional)
    const int64_t c = A.d2c[hasc ? cb + t : last_c];
    const int64_t f = hasf ? (int64_t)A.d2f[fb + t] : 0;
    // ---- level 2: everything addressed by the ids
    const uint8_t mark = hasc ? (uint8_t)(A.cellmark[c] & A.mark_mask) : (uint8_t)0;
    int32_t cd[ND];
    if constexpr (ND % 2 == 0)
    {
      // (a dof row of an even number of entries starts on an 8 B boundary: half as many load instructions)
      const int2* d2 = reinterpret_cast<const int2*>(A.dofmap + c * ND);
#pragma unroll
      for (int j = 0; j < ND / 2; ++j) { const int2 v2 = d2[j]; cd[2 * j] = v2.x; cd[2 * j + 1] = v2.y; }
    }
    else
    {
#pragma unroll
      for (int j = 0; j < ND; ++j) cd[j] = A.dofmap[c * ND + j];
    }
    int32_t cn[NV];
    if constexpr (TDIM == 3)
    {
      const int4 v4 = *reinterpret_cast<const int4*>(A.conn + c * 4);
      cn[0] = v4.x; cn[1] = v4.y; cn[2] = v4.z; cn[3] = v4.w;
    }
    else
    {
#pragma unroll
      for (int i = 0; i < NV; ++i) cn[i] = A.conn[c * NV + i];
    }
    const unsigned long long bw = A.cut_bits ? A.cut_bits[c >> 6] : 0ull;
    const int32_t rk = A.cut_rank ? A.cut_rank[c >> 6] : 0;
    const double* blk = A.facet_tensors + f * (int64_t)(8 + 16 * A.facet_nq);
    int32_t cm[WF];
    int nfree = NX;
    if (nf > 0) // (wave-divergent only between rows with and without facets)
    {
      const int4* cmp = reinterpret_cast<const int4*>(blk);
      const int4 q0 = cmp[0], q1 = cmp[1], q2 = cmp[2], q3 = cmp[3];
      const int32_t c16[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
      for (int j = 0; j < WF; ++j) cm[j] = c16[j];
      nfree = c16[15];
    }
    else
    {
#pragma unroll
      for (int j = 0; j < WF; ++j) cm[j] = -1;
    }
    // ---- level 3: vertices of the cell, the row of the cut cell's tensor
    int lr = 0;
#pragma unroll
    for (int j = 0; j < ND; ++j) lr = (cd[j] == (int32_t)r) ? j : lr;
    Geo<TDIM> g;
#pragma unroll
    for (int i = 0; i < NV; ++i) load_vertex<TDIM>(A.x, cn[i], g.x[i]);
    const bool is_cut = (mark & 0xF0u) != 0;
    const int64_t e = (int64_t)rk + __popcll(bw & ((1ull << (c & 63)) - 1ull));
    double trow[ND];
    {
      // (a lane whose cell is not cut reads row 0 of tensor 0: the load stays unconditional)
      const double2* T = reinterpret_cast<const double2*>(A.cut_tensors + (is_cut ? (e * ND + lr) * ND : 0));
#pragma unroll
      for (int j = 0; j < ND / 2; ++j)
      {
        const double2 v = T[j];
        trow[2 * j] = v.x; trow[2 * j + 1] = v.y;
      }
    }
    // ---- the cell item
    {
      double acc[ND];
      int csl[ND];
#pragma unroll
      for (int j = 0; j < ND; ++j) { acc[j] = 0.0; csl[j] = -1; }
      if (mark)
      {
#pragma unroll
        for (int j = 0; j < ND; ++j) csl[j] = find_slot(cd[j]);
        if (mark & 0x0Fu)
        {
          jacobian<TDIM>(g);
          p2_stiffness_row<TDIM>(g, lr, 1.0, acc);
        }
        if (is_cut)
        {
#pragma unroll
          for (int j = 0; j < ND; ++j) acc[j] += trow[j];
        }
#pragma unroll
        for (int j = 0; j < ND; ++j)
          if (j == lr) { dsum += diag_bc ? 0.0 : acc[j]; csl[j] = -1; }
      }
      add_item(std::integral_constant<int, ND>{}, mark != 0, cd, acc, csl);
    }
    // ---- the facet item: facet_nq rank-one records (jf_q[0..WF), w_q); row of macro dof m = sum_q w_q jf_q[m] jf_q[.]
    {
      double acc[WF];
      int sl[WF];
#pragma unroll
      for (int j = 0; j < WF; ++j) { acc[j] = 0.0; sl[j] = -1; }
      if (hasf)
      {
        int m = -1;
#pragma unroll
        for (int j = 0; j < WF; ++j) m = (cm[j] == (int32_t)r) ? j : m;
        if (nfree != NX) *A.error = 4; // not an interior facet of a conforming mesh with a continuous space
        const double2* rec = reinterpret_cast<const double2*>(blk + 8);
        for (int q = 0; q < A.facet_nq; ++q)
        {
          double v[16];
#pragma unroll
          for (int k = 0; k < 8; ++k)
          {
            const double2 pq = rec[q * 8 + k];
            v[2 * k] = pq.x; v[2 * k + 1] = pq.y;
          }
          double jm = 0.0;
#pragma unroll
          for (int j = 0; j < WF; ++j) jm = (j == m) ? v[j] : jm;
          jm *= v[WF];
#pragma unroll
          for (int j = 0; j < WF; ++j) acc[j] += jm * v[j];
        }
#pragma unroll
        for (int j = 0; j < WF; ++j) sl[j] = find_slot(cm[j]);
      }
      if (nf > 0 || ORDERED) add_item(std::integral_constant<int, WF>{}, hasf, cm, acc, sl);
    }
  }
  {
    double d = dsum;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, G);
    if (live && gl == 0 && len > 0 && nc > 0)
    {
      const int slot = find_slot((int32_t)r);
      if (slot >= 0) atomicAdd(&s_val[grp][slot], d);
    }
  }
  __syncthreads();
  if (A.fresh == 2)
  {
#pragma unroll
    for (int q = 0; q < KMAX; ++q)
    {
      const int k = gl + q * G;
      if (k < len) A.values[rb + k] = s_val[grp][k];
    }
    return;
  }
  double myval[KMAX];
#pragma unroll
  for (int q = 0; q < KMAX; ++q)
  {
    const int k = gl + q * G;
    myval[q] = k < len ? A.values[rb + k] : 0.0;
  }
#pragma unroll
  for (int q = 0; q < KMAX; ++q)
  {
    const int k = gl + q * G;
    if (k < len) A.values[rb + k] = myval[q] + s_val[grp][k];
  }
}

// ---------------------------------------------------------------------------
// stage 2, bilinear forms, the uncut P1 cells alone.  P1 space on the geometry dofmap:
// an item is the row's vertex plus the TDIM other vertices of the cell, its row of the
// stiffness tensor comes from p1_stiffness_row() (6 coordinate loads, no Jacobian
// inverse, no pass through the connectivity), only the TDIM off-diagonal columns go
// through LDS and the diagonal is reduced in registers.  Everything that is not an
// uncut-cell stiffness item (cut-cell rule tensors, facets) is left to
// assemble_rows_kernel on the rows next to the interface (plan.special_rows).
// ---------------------------------------------------------------------------
#ifndef CFX_BLOCK_WAVES
#define CFX_BLOCK_WAVES 4
#endif
#ifndef CFX_P1_WAVES
#define CFX_P1_WAVES 6
#endif
template <int TDIM, int G, int CAP, bool ORDERED>
__global__ void __launch_bounds__(kWave, CFX_P1_WAVES) assemble_rows_p1_kernel(RowArgs A)
{
  constexpr int ND = TDIM + 1;
  constexpr int RPW = kWave / G;
  __shared__ int32_t s_col[RPW][CAP + 1];
  __shared__ double s_val[RPW][CAP + 1];
  const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
  const int64_t ri = CFX_ROW_BLOCK * RPW + grp;
  const bool live = ri < dev_n(A.n_active);
  const int64_t r = live ? A.active_rows[ri] : 0;
  const int64_t rb = live ? A.indptr[r] : 0;
  int len = live ? (int)(A.indptr[r + 1] - rb) : 0;
  if (len > CAP) { *A.error = 2; len = 0; }
  constexpr int KMAX = CAP / G;
  {
    int32_t mycol[KMAX];
#pragma unroll
    for (int q = 0; q < KMAX; ++q)
    {
      const int k = gl + q * G;
      mycol[q] = k < len ? A.indices[rb + k] : -1;
    }
#pragma unroll
    for (int q = 0; q < KMAX; ++q)
    {
      const int k = gl + q * G;
      if (k < len) { s_col[grp][k] = mycol[q]; s_val[grp][k] = 0.0; }
    }
  }
  __syncthreads();
  const int64_t cb = live ? A.d2c_off[r] : 0;
  const int nc = (live && len > 0) ? (int)(A.d2c_off[r + 1] - cb) : 0;
  const bool row_bc = live && A.bc0 && A.bc0[r];
  const bool diag_bc = row_bc || (live && A.bc1 != nullptr && A.bc1[r] != 0);
  double xr[TDIM];
#pragma unroll
  for (int d = 0; d < TDIM; ++d) xr[d] = 0.0;
  if (live) load_vertex<TDIM>(A.x, r, xr);
  double dsum = 0.0;

  auto find_slot = [&](int32_t col) -> int
  {
    int lo = 0, hi = len;
    while (lo < hi)
    {
      const int mid = (lo + hi) >> 1;
      if (s_col[grp][mid] < col) lo = mid + 1; else hi = mid;
    }
    if (lo < len && s_col[grp][lo] == col) return lo;
    *A.error = 1;
    return -1;
  };

  constexpr int R = 3; // 3 x 8 lanes cover the 24 tets around a Kuhn-mesh vertex in one pass
  for (int base = 0;; base += R * G)
  {
    if (__ballot(base + gl < nc) == 0) break;
    int32_t cell[R];
    int rep[R];
    int32_t cd[R][ND];
#pragma unroll
    for (int k = 0; k < R; ++k)
    {
      const int t = base + k * G + gl;
      cell[k] = t < nc ? A.d2c[cb + t] : -1;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) rep[k] = cell[k] >= 0 ? __popc(A.cellmark[cell[k]] & A.inline_bits) : 0;
#pragma unroll
    for (int k = 0; k < R; ++k)
    {
      if (!rep[k]) continue;
      if constexpr (ND == 4)
      {
        const int4 v = *reinterpret_cast<const int4*>(A.dofmap + (int64_t)cell[k] * 4);
        cd[k][0] = v.x; cd[k][1] = v.y; cd[k][2] = v.z; cd[k][3] = v.w;
      }
      else
      {
#pragma unroll
        for (int j = 0; j < ND; ++j) cd[k][j] = A.dofmap[(int64_t)cell[k] * ND + j];
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k)
    {
      if (__ballot(rep[k] != 0) == 0) continue;
      const bool has = rep[k] != 0;
      int lr = 0;
#pragma unroll
      for (int j = 0; j < ND; ++j) lr = (has && cd[k][j] == (int32_t)r) ? j : lr;
      int32_t oc[TDIM];
      int osl[TDIM];
      double ov[TDIM];
#pragma unroll
      for (int t = 0; t < TDIM; ++t) { oc[t] = has ? (t < lr ? cd[k][t] : cd[k][t + 1]) : (int32_t)r; osl[t] = -1; ov[t] = 0.0; }
      if (has)
      {
        double xo[TDIM][TDIM], dg, off[TDIM];
#pragma unroll
        for (int t = 0; t < TDIM; ++t) load_vertex<TDIM>(A.x, oc[t], xo[t]);
#pragma unroll
        for (int t = 0; t < TDIM; ++t) osl[t] = find_slot(oc[t]);
        p1_stiffness_row<TDIM>(xr, xo, dg, off);
        const double scale = (double)rep[k]; // the same cell in several inline integrals
        dsum += diag_bc ? 0.0 : dg * scale;
#pragma unroll
        for (int t = 0; t < TDIM; ++t)
          ov[t] = (row_bc || (A.bc1 != nullptr && A.bc1[oc[t]] != 0)) ? 0.0 : off[t] * scale;
      }
      if constexpr (ORDERED)
      {
        for (int turn = 0; turn < G; ++turn) // one lane of each group at a time: item order
        {
          if (gl == turn)
          {
#pragma unroll
            for (int t = 0; t < TDIM; ++t)
              if (osl[t] >= 0) s_val[grp][osl[t]] += ov[t];
          }
          __syncthreads();
        }
      }
      else
      {
#pragma unroll
        for (int t = 0; t < TDIM; ++t)
          if (osl[t] >= 0) atomicAdd(&s_val[grp][osl[t]], ov[t]);
      }
    }
  }
  {
    double d = dsum;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, G);
    if (gl == 0 && nc > 0)
    {
      const int slot = find_slot((int32_t)r);
      if (slot >= 0) atomicAdd(&s_val[grp][slot], d);
    }
  }
  __syncthreads();
  if (A.fresh == 2)
  {
    // first writer of these rows on a matrix that holds nothing yet (set_value(0) fused, run_matrix): stored -- the
    // zero fill skipped them, the rule / facet items are added by the kernel that follows
#pragma unroll
    for (int q = 0; q < KMAX; ++q)
    {
      const int k = gl + q * G;
      if (k < len) A.values[rb + k] = s_val[grp][k];
    }
    return;
  }
  double myval[KMAX];
#pragma unroll
  for (int q = 0; q < KMAX; ++q)
  {
    const int k = gl + q * G;
    myval[q] = k < len ? A.values[rb + k] : 0.0;
  }
#pragma unroll
  for (int q = 0; q < KMAX; ++q)
  {
    const int k = gl + q * G;
    if (k < len) A.values[rb + k] = myval[q] + s_val[grp][k];
  }
}

// ---------------------------------------------------------------------------
// stage 2, bilinear forms, PLAIN rows: active rows whose items are uncut cells only.
// Their CSR row is the subset of the mesh-static stencil selected by a 64-bit mask
// (build_pattern laid it out that way), so the CSR slot of a column is
// popcount(mask below its stencil position) and the position comes with the incidence
// entry (Stencil::slot4): no column list in LDS, no search, no comparison against r.
// ---------------------------------------------------------------------------
#ifndef CFX_PLAIN_WAVES
#define CFX_PLAIN_WAVES 6
#endif
#ifndef CFX_VEC_G
#define CFX_VEC_G 8 // lanes per row of the vector gather (512^3: G=4 3.05 ms, G=8 2.89 ms, G=2 4.67 ms)
#endif
#ifndef CFX_PLAIN_G
#define CFX_PLAIN_G 4 // lanes per plain row (measured at 512^3: G=8,R=3 3.55 ms; G=4,R=6 3.14 ms; G=16 5.1 ms)
#endif
#ifndef CFX_PLAIN_R
#define CFX_PLAIN_R 6 // items per lane per pass
#endif
// STAGE (stencils of at most 32 vertices): the coordinates of the row's stencil vertices are
// loaded once per row into LDS (~15 gathers) and every item reads its TDIM other vertices from
// there by stencil position: no dofmap row and no coordinate gathers per item (~170 per row).
template <int TDIM, int G, int CAP, bool ORDERED, bool STAGE>
__global__ void __launch_bounds__(kWave, CFX_PLAIN_WAVES) assemble_rows_plain_kernel(RowArgs A)
{
  constexpr int ND = TDIM + 1;
  constexpr int RPW = kWave / G;
  constexpr int KMAX = CAP / G;
  constexpr int R = CFX_PLAIN_R; // R x G lanes cover the 24 tets around a Kuhn-mesh vertex in one pass
  constexpr int SX = CAP <= 32 ? CAP : 32; // staged stencil entries per row (a plain row is no longer than its stencil)
  __shared__ double s_val[RPW][CAP + 1];
  __shared__ double s_x[STAGE ? RPW : 1][STAGE ? SX + 1 : 1][TDIM];
  const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
  const int64_t ri = CFX_ROW_BLOCK * RPW + grp;
  const bool live = ri < dev_n(A.n_active);
  const int64_t r = live ? A.active_rows[ri] : 0;
  const int64_t rb = live ? A.indptr[r] : 0;
  int len = live ? (int)(A.indptr[r + 1] - rb) : 0;
  if (len > CAP) { *A.error = 2; len = 0; }
#pragma unroll
  for (int q = 0; q < KMAX; ++q)
  {
    const int k = gl + q * G;
    if (k < len) s_val[grp][k] = 0.0;
  }
  const int64_t cb = live ? A.d2c_off[r] : 0;
  const int nc = (live && len > 0) ? (int)(A.d2c_off[r + 1] - cb) : 0;
  const unsigned dpos = live ? A.diagpos[r] : 0u;
  const bool row_bc = live && A.bc0 && A.bc0[r];
  const bool diag_bc = row_bc || (live && A.bc1 != nullptr && A.bc1[r] != 0);
  double xr[TDIM];
#pragma unroll
  for (int d = 0; d < TDIM; ++d) xr[d] = 0.0;
  if constexpr (STAGE)
  {
    const int64_t sb = live ? A.st_off[r] : 0;
    const int slen = live ? (int)(A.st_off[r + 1] - sb) : 0;
    static_assert(!STAGE || SX % G == 0, "stencil staging: SX must be a multiple of the group size");
    int32_t vid[SX / G];
#pragma unroll
    for (int q = 0; q < SX / G; ++q)
    {
      const int p = gl + q * G;
      vid[q] = p < slen ? A.st_nbr[sb + p] : -1;
    }
#pragma unroll
    for (int q = 0; q < SX / G; ++q)
    {
      const int p = gl + q * G;
      if (vid[q] >= 0)
      {
        double xv[TDIM];
        load_vertex<TDIM>(A.x, vid[q], xv);
#pragma unroll
        for (int d = 0; d < TDIM; ++d) s_x[grp][p][d] = xv[d];
      }
    }
  }
  else
  {
    if (live) load_vertex<TDIM>(A.x, r, xr);
  }

  // mask and (where all incident cells carry one mark) the mark itself come from the plan
  const unsigned long long mask = live ? A.plain_masks[ri] : 0ull;
  const uint8_t umark = live ? A.plain_uniform[ri] : (uint8_t)0;
  int32_t cell[R];
  uint32_t s4[R];
  uint8_t mk[R];
  auto load_chunk = [&](int base)
  {
#pragma unroll
    for (int k = 0; k < R; ++k)
    {
      const int t = base + k * G + gl;
      // the cell id is only the key of its mark byte: not read for rows whose incident cells share one mark
      cell[k] = t < nc ? (umark ? 0 : A.d2c[cb + t]) : -1;
      s4[k] = t < nc ? A.slot4[cb + t] : 0u;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) mk[k] = cell[k] >= 0 ? (umark ? umark : A.cellmark[cell[k]]) : (uint8_t)0;
  };
  load_chunk(0); // in flight together with the stencil staging above
  // the pattern row must be exactly this subset (build_pattern from the same plan)
  if (live && len > 0 && __popcll(mask) != len) { *A.error = 5; len = 0; }
  // a row that holds its whole stencil (mask = the low `len` bits): CSR slot = stencil position
  const bool full = (mask & (mask + 1ull)) == 0ull;
  __syncthreads();
  if constexpr (STAGE)
  {
    if (live)
    {
#pragma unroll
      for (int d = 0; d < TDIM; ++d) xr[d] = s_x[grp][dpos][d];
    }
  }

  double dsum = 0.0;
  for (int base = 0;; base += R * G)
  {
    if (__ballot(base + gl < nc) == 0) break;
    if (base > 0) load_chunk(base);
    int rep[R];
    int32_t cd[R][ND];
#pragma unroll
    for (int k = 0; k < R; ++k) rep[k] = (len > 0 && cell[k] >= 0) ? __popc(mk[k] & A.inline_bits) : 0;
    if constexpr (!STAGE)
    {
#pragma unroll
      for (int k = 0; k < R; ++k)
      {
        if (!rep[k]) continue;
        if constexpr (ND == 4)
        {
          const int4 v = *reinterpret_cast<const int4*>(A.dofmap + (int64_t)cell[k] * 4);
          cd[k][0] = v.x; cd[k][1] = v.y; cd[k][2] = v.z; cd[k][3] = v.w;
        }
        else
        {
#pragma unroll
          for (int j = 0; j < ND; ++j) cd[k][j] = A.dofmap[(int64_t)cell[k] * ND + j];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k)
    {
      if (__ballot(rep[k] != 0) == 0) continue;
      const bool has = rep[k] != 0;
      int lr = 0;
#pragma unroll
      for (int j = 0; j < ND; ++j) lr = (((s4[k] >> (8 * j)) & 0xffu) == dpos) ? j : lr;
      unsigned opos[TDIM];
      int32_t oc[TDIM];
      int osl[TDIM];
      double ov[TDIM];
#pragma unroll
      for (int t = 0; t < TDIM; ++t)
      {
        opos[t] = has ? (s4[k] >> (8 * (t < lr ? t : t + 1))) & 0xffu : dpos;
        if constexpr (STAGE) oc[t] = (has && A.bc1 != nullptr) ? A.st_nbr[A.st_off[r] + opos[t]] : (int32_t)r;
        else oc[t] = has ? (t < lr ? cd[k][t] : cd[k][t + 1]) : (int32_t)r;
        osl[t] = -1; ov[t] = 0.0;
      }
      if (has)
      {
        double xo[TDIM][TDIM], dg, off[TDIM];
#pragma unroll
        for (int t = 0; t < TDIM; ++t)
        {
          if constexpr (STAGE)
          {
#pragma unroll
            for (int d = 0; d < TDIM; ++d) xo[t][d] = s_x[grp][opos[t]][d];
          }
          else
            load_vertex<TDIM>(A.x, oc[t], xo[t]);
        }
#pragma unroll
        for (int t = 0; t < TDIM; ++t) osl[t] = full ? (int)opos[t] : __popcll(mask & ((1ull << opos[t]) - 1ull));
        p1_stiffness_row<TDIM>(xr, xo, dg, off);
        const double scale = (double)rep[k]; // the same cell in several inline integrals
        dsum += diag_bc ? 0.0 : dg * scale;
#pragma unroll
        for (int t = 0; t < TDIM; ++t)
          ov[t] = (row_bc || (A.bc1 != nullptr && A.bc1[oc[t]] != 0)) ? 0.0 : off[t] * scale;
      }
      if constexpr (ORDERED)
      {
        for (int turn = 0; turn < G; ++turn) // one lane of each group at a time: item order
        {
          if (gl == turn)
          {
#pragma unroll
            for (int t = 0; t < TDIM; ++t)
              if (osl[t] >= 0) s_val[grp][osl[t]] += ov[t];
          }
          __syncthreads();
        }
      }
      else
      {
#pragma unroll
        for (int t = 0; t < TDIM; ++t)
          if (osl[t] >= 0) atomicAdd(&s_val[grp][osl[t]], ov[t]);
      }
    }
  }
  {
    double d = dsum;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, G);
    if (gl == 0 && nc > 0 && len > 0) atomicAdd(&s_val[grp][__popcll(mask & ((1ull << dpos) - 1ull))], d);
  }
  __syncthreads();
  if (A.fresh)
  {
    // `values` was zeroed by this very call (cfx_assemble_matrix_zeroed) and a plain row has one writer: store
#pragma unroll
    for (int q = 0; q < KMAX; ++q)
    {
      const int k = gl + q * G;
      if (k < len) A.values[rb + k] = s_val[grp][k];
    }
    return;
  }
  double myval[KMAX];
#pragma unroll
  for (int q = 0; q < KMAX; ++q)
  {
    const int k = gl + q * G;
    myval[q] = k < len ? A.values[rb + k] : 0.0;
  }
#pragma unroll
  for (int q = 0; q < KMAX; ++q)
  {
    const int k = gl + q * G;
    if (k < len) A.values[rb + k] = myval[q] + s_val[grp][k];
  }
}

// ---------------------------------------------------------------------------
// stage 2, bilinear forms, plain rows by ROW TILE (Stencil::tile_verts / st_loc): one wavefront per tile of
// kRowTile consecutive dofs, 4 lanes per row.  The rows of a tile sit next to each other in every mesh-static
// array, so the tile's coordinates (sorted union of the rows' stencils), its slot4 range and its st_loc range
// are staged in LDS with coalesced loads: a few requests per 128 B line for the whole tile where the row-wise
// kernel above issues ~100 partial-line gathers per row (it is bound by the request rate of the vector L1, not
// by HBM bytes).  Items then run out of LDS: positions -> tile-local vertex -> coordinates, accumulators at the
// row's stencil segment.  With `fresh` the rows are stored, not read back.
// ---------------------------------------------------------------------------
#ifndef CFX_TILE_WAVES
#define CFX_TILE_WAVES 5
#endif
#ifndef CFX_TILE_R
#define CFX_TILE_R 3 // items per lane and pass (measured with 5 waves per SIMD: 2 -> 2.17 ms, 3 -> 2.17, 4 -> 2.42, 6 -> 3.23; with 4: 6 -> 2.26)
#endif
struct TileArgs
{
  const double* x;
  DevN n_tiles;
  const int32_t* tile_first; // position of the tile's first plain row in `rows`
  const int32_t* tile_id;
  DevN n_plain;
  const int32_t* rows;
  const unsigned long long* masks;
  const uint8_t* uniform;
  const int64_t* d2c_off;
  const int32_t* d2c;
  const uint32_t* slot4;
  const uint8_t* cellmark;
  const int64_t* st_off;
  const int32_t* st_nbr;
  const uint16_t* st_loc;
  const uint8_t* diagpos;
  const int64_t* tile_voff;
  const int32_t* tile_verts;
  const int64_t* indptr;
  double* values;
  const int8_t* bc0;
  const int8_t* bc1;
  unsigned inline_bits;
  int fresh;
  int* error;
  int64_t ndofs;
};

// LDS capacity classes of a tile (vertices of the union, neighbour entries, dof->cells entries).  The arrays are
// separate __shared__ objects so that the compiler knows the accumulator atomics cannot alias the staged tables
// and is free to run the LDS reads of the next items ahead of them.
template <int CLS> struct TileCap;
template <> struct TileCap<0> { static constexpr int V = 192, S = 256, I = 384, VL = 168; };   // Kuhn box meshes: 162 / 240 / 384
// (VL: vertices held in LDS -- 168 keeps a wavefront at 8128 B, 20 wavefronts per CU; V: the register staging width)
template <> struct TileCap<1> { static constexpr int V = 512, S = 512, I = 1024, VL = 512; };
inline int tile_class(const Stencil& st)
{
  for (int c = 0; c < 2; ++c)
  {
    const int V = c == 0 ? TileCap<0>::VL : TileCap<1>::VL, S = c == 0 ? TileCap<0>::S : TileCap<1>::S,
              I = c == 0 ? TileCap<0>::I : TileCap<1>::I;
    if (st.max_tile_verts <= V && st.max_tile_st <= S && st.max_tile_items <= I) return c;
  }
  return -1;
}

// the 3 positions of a cell's other dofs from slot4 (byte j = position of the cell's j-th dof) and the row's own
// position replicated in every byte: the byte equal to it is dropped (positions of one cell are distinct)
__device__ __forceinline__ uint32_t other_positions(uint32_t s4, uint32_t dpos4)
{
  const uint32_t x = s4 ^ dpos4;
  const uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u;     // lowest set bit marks the zero byte
  const uint32_t sh = (uint32_t)(__ffs((int)z) - 1) & ~7u;     // 8 * index of that byte
  const uint32_t m = (1u << sh) - 1u;
  return (s4 & m) | ((s4 >> 8) & ~m);
}

template <int TDIM, bool ORDERED, int CLS>
__global__ void __launch_bounds__(kWave, CFX_TILE_WAVES) assemble_tiles_plain_kernel(TileArgs A)
{
  constexpr int G = 4, R = CFX_TILE_R;
  constexpr int CAPV = TileCap<CLS>::V, CAPS = TileCap<CLS>::S, CAPI = TileCap<CLS>::I;
  constexpr int VR = CAPV / kWave, SR = CAPS / kWave, IR = CAPI / kWave;
  static_assert(kWave / G == kRowTile, "one lane group per row of the tile");
  constexpr int CAPVL = TileCap<CLS>::VL;
  __shared__ double s_x[CAPVL * TDIM];
  __shared__ double s_val[CAPS]; // row g accumulates at its stencil segment
  __shared__ uint32_t s_s4[CAPI];
  __shared__ uint16_t s_loc[CAPS];
  const int lane = threadIdx.x, g = lane / G, gl = lane % G;
  const int64_t w = CFX_ROW_BLOCK;
  if (w >= dev_n(A.n_tiles)) return;
  const int64_t i0 = A.tile_first[w];
  const int64_t t = A.tile_id[w];
  const int64_t r0 = t * kRowTile;
  // ---- every load that does not depend on another one is issued here, ahead of the first wait
  int32_t rl = -1;
  if (lane < kRowTile && i0 + lane < dev_n(A.n_plain)) rl = A.rows[i0 + lane];
  const int64_t rr = r0 + (lane < kRowTile ? lane : kRowTile);
  const int64_t so = A.st_off[rr < A.ndofs ? rr : A.ndofs], co = A.d2c_off[rr < A.ndofs ? rr : A.ndofs];
  const int64_t vb = A.tile_voff[t];
  const int nv = (int)(A.tile_voff[t + 1] - vb);
  int32_t vid[VR];
#pragma unroll
  for (int q = 0; q < VR; ++q) vid[q] = lane + q * kWave < nv ? A.tile_verts[vb + lane + q * kWave] : -1;
  const int64_t r = r0 + g;
  const int64_t rq = r < A.ndofs ? r : A.ndofs - 1;
  const int64_t rb = A.indptr[rq], re = A.indptr[rq + 1];
  const unsigned dpos = A.diagpos[rq];
  // which rows of the tile are plain: their ids are the next entries of the (ascending) plain list
  unsigned pm = (rl >= 0 && rl / kRowTile == t) ? 1u << (rl % kRowTile) : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pm |= __shfl_xor(pm, o, 64);
  const bool live = (pm >> g) & 1u;
  const int64_t pi = i0 + __popc(pm & ((1u << g) - 1u));
  const unsigned long long mask = live ? A.masks[pi] : 0ull;
  const uint8_t umark = live ? A.uniform[pi] : (uint8_t)0;
  const int64_t sb0 = __shfl(so, 0, 64), cb0 = __shfl(co, 0, 64);
  const int nst = (int)(__shfl(so, kRowTile, 64) - sb0), nit = (int)(__shfl(co, kRowTile, 64) - cb0);
  const int st_rel = (int)(__shfl(so, g, 64) - sb0);
  const int c_rel = (int)(__shfl(co, g, 64) - cb0);
  int nc = (int)(__shfl(co, g + 1, 64) - __shfl(co, g, 64));
  if (nst > CAPS || nit > CAPI || nv > CAPVL) { *A.error = 2; return; }
  uint32_t s4r[IR];
#pragma unroll
  for (int q = 0; q < IR; ++q) s4r[q] = lane + q * kWave < nit ? A.slot4[cb0 + lane + q * kWave] : 0u;
  uint16_t locr[SR];
#pragma unroll
  for (int q = 0; q < SR; ++q) locr[q] = lane + q * kWave < nst ? A.st_loc[sb0 + lane + q * kWave] : (uint16_t)0;
  double xv[VR][TDIM];
#pragma unroll
  for (int q = 0; q < VR; ++q)
  {
#pragma unroll
    for (int d = 0; d < TDIM; ++d) xv[q][d] = 0.0;
    if (vid[q] >= 0) load_vertex<TDIM>(A.x, vid[q], xv[q]);
  }
  const bool row_bc = live && A.bc0 && A.bc0[r];
  const bool diag_bc = row_bc || (live && A.bc1 != nullptr && A.bc1[r] != 0);
  // ---- LDS image of the tile
#pragma unroll
  for (int q = 0; q < IR; ++q) s_s4[lane + q * kWave] = s4r[q];
#pragma unroll
  for (int q = 0; q < SR; ++q)
  {
    s_loc[lane + q * kWave] = locr[q];
    s_val[lane + q * kWave] = 0.0;
  }
#pragma unroll
  for (int q = 0; q < VR; ++q)
    if (lane + q * kWave < CAPVL)
    {
#pragma unroll
      for (int d = 0; d < TDIM; ++d) s_x[(lane + q * kWave) * TDIM + d] = xv[q][d];
    }
  int len = live ? (int)(re - rb) : 0;
  if (live && len > 0 && __popcll(mask) != len) { *A.error = 5; len = 0; }
  if (!live || len == 0) nc = 0;
  // a row that holds its whole stencil (mask = the low `len` bits): CSR slot = stencil position
  const bool full = (mask & (mask + 1ull)) == 0ull;
  const bool any_bc = A.bc0 != nullptr || A.bc1 != nullptr; // (kernel-uniform)
  // wave-uniform: every row of the tile has one mark for all its cells and holds its whole stencil
  const bool fast = __ballot(nc > 0 && (umark == 0 || !full)) == 0ull && !any_bc;
  const uint32_t dpos4 = dpos * 0x01010101u;
  const bool nrep1 = __ballot(nc > 0 && __popc(umark & A.inline_bits) != 1) == 0ull; // wave-uniform: one integral per cell
  __syncthreads();
  double xr[TDIM];
  {
    const int vr = nc > 0 ? s_loc[st_rel + dpos] : 0;
#pragma unroll
    for (int d = 0; d < TDIM; ++d) xr[d] = s_x[vr * TDIM + d];
  }
  double dsum = 0.0;
  // lane gl owns the items [gl * per, gl * per + per): lanes that run together work on cells a quarter of the
  // row's list apart, which rarely share a vertex -- fewer LDS atomics that serialise on one address
  const int per = (nc + G - 1) / G;
  for (int base = 0;; base += R)
  {
    if (__ballot(base < per) == 0) break;
    uint32_t o3[R];
    int rep[R];
#pragma unroll
    for (int k = 0; k < R; ++k)
    {
      const int tt = gl * per + base + k;
      const bool in = base + k < per && tt < nc;
      o3[k] = other_positions(in ? s_s4[c_rel + tt] : (dpos4 ^ 0x00010203u), dpos4);
      rep[k] = in ? 1 : 0;
    }
    if (!fast)
    {
#pragma unroll
      for (int k = 0; k < R; ++k)
      {
        const int tt = gl * per + base + k;
        // the cell id is only the key of its mark byte: not read for rows whose incident cells share one mark
        const uint8_t mk = rep[k] ? (umark ? umark : A.cellmark[A.d2c[cb0 + c_rel + tt]]) : (uint8_t)0;
        rep[k] = __popc(mk & A.inline_bits);
      }
    }
    else
    {
      const int nrep = __popc(umark & A.inline_bits); // the same cell in several inline integrals
#pragma unroll
      for (int k = 0; k < R; ++k) rep[k] *= nrep;
    }
    int vo[R][TDIM];
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
      for (int q = 0; q < TDIM; ++q) vo[k][q] = s_loc[st_rel + ((o3[k] >> (8 * q)) & 0xffu)];
#pragma unroll
    for (int k = 0; k < R; ++k)
    {
      if (!ORDERED && __ballot(rep[k] != 0) == 0) continue;
      double xo[TDIM][TDIM], dg, off[TDIM];
#pragma unroll
      for (int q = 0; q < TDIM; ++q)
#pragma unroll
        for (int d = 0; d < TDIM; ++d)
          xo[q][d] = s_x[vo[k][q] * TDIM + d];
      p1_stiffness_row<TDIM>(xr, xo, dg, off);
      int osl[TDIM];
      double ov[TDIM];
#pragma unroll
      for (int q = 0; q < TDIM; ++q)
      {
        osl[q] = (int)((o3[k] >> (8 * q)) & 0xffu);
        ov[q] = off[q];
      }
      if (fast && nrep1)
        dsum += rep[k] ? dg : 0.0; // (lanes without an item computed on a dummy cell)
      else
      {
        const double scale = (double)rep[k]; // the same cell in several inline integrals; 0: no item
        dsum += diag_bc ? 0.0 : dg * scale;
#pragma unroll
        for (int q = 0; q < TDIM; ++q) ov[q] *= scale;
      }
      if (!fast)
      {
#pragma unroll
        for (int q = 0; q < TDIM; ++q)
        {
          if (!full) osl[q] = __popcll(mask & ((1ull << osl[q]) - 1ull));
          if (any_bc && rep[k]
              && (row_bc || (A.bc1 != nullptr && A.bc1[A.st_nbr[sb0 + st_rel + ((o3[k] >> (8 * q)) & 0xffu)]] != 0)))
            ov[q] = 0.0;
        }
      }
      if constexpr (ORDERED)
      {
        for (int turn = 0; turn < G; ++turn) // one lane of each group at a time: item order
        {
          if (gl == turn && rep[k])
          {
#pragma unroll
            for (int q = 0; q < TDIM; ++q) s_val[st_rel + osl[q]] += ov[q];
          }
          __syncthreads();
        }
      }
      else
      {
        if (rep[k])
        {
#pragma unroll
          for (int q = 0; q < TDIM; ++q) atomicAdd(&s_val[st_rel + osl[q]], ov[q]);
        }
      }
    }
  }
  {
    double d = dsum;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, G);
    if (gl == 0 && nc > 0) atomicAdd(&s_val[st_rel + (full ? (int)dpos : __popcll(mask & ((1ull << dpos) - 1ull)))], d);
  }
  __syncthreads();
  if (A.fresh)
  {
    for (int k = gl; k < len; k += G) A.values[rb + k] = s_val[st_rel + k];
  }
  else
  {
    for (int k = gl; k < len; k += G) A.values[rb + k] += s_val[st_rel + k];
  }
}

// ---------------------------------------------------------------------------
// stage 2, bilinear forms on BLOCK spaces (bs = gdim: elasticity, BASELINE config 5).  A group
// of G lanes owns one matrix row R = bs * dof + component; the CSR row is bs-blocked (column
// list = scalar columns expanded by bs), so the group searches scalar columns and adds bs
// values per column.  Items: marked incident cells of the dof -- the local row (dof, component)
// of an uncut cell is recomputed inline (a staged 30 x 30 tensor per uncut P2 cell would be
// 7 KB), the rows of cut cells come from the stage-1 rule tensors.  No facet items: forms with
// facet integrals on block spaces keep the entity-parallel path.
// ---------------------------------------------------------------------------
// INLINE = false (default): uncut tensors are staged, the elasticity arithmetic is not compiled in
// and the kernel fits 4 waves/SIMD.
// INLINE: how the uncut items get their row -- 0 staged tensors, 1 quadrature row (cell_local_row), 2 closed-form
// degree-2 elasticity row (p2_elasticity_row: std_inline == 3)
template <int TDIM, int DEG, int BS, int G, int CAP, bool ORDERED, int INLINE>
__global__ void __launch_bounds__(kWave, INLINE == 1 ? 2 : CFX_BLOCK_WAVES) assemble_rows_block_kernel(RowArgs A)
{
  constexpr int ND = Elem<TDIM, DEG>::ND;
  constexpr int NLOC = ND * BS;
  constexpr int RPW = kWave / G;
  __shared__ int32_t s_col[RPW][CAP + 1];
  __shared__ double s_val[RPW][CAP * BS + 1];
  const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
  const int64_t ri = CFX_ROW_BLOCK * RPW + grp;
  const bool live = ri < dev_n(A.n_active) * BS;
  const int64_t rr = live ? ri / BS : 0;
  const int kc = live ? (int)(ri - rr * BS) : 0;
  const int64_t r = live ? A.active_rows[rr] : 0;
  const int64_t R = r * BS + kc;
  const int64_t rb = live ? A.indptr[R] : 0;
  int lene = live ? (int)(A.indptr[R + 1] - rb) : 0;
  int len = lene / BS;
  if (len > CAP) { *A.error = 2; len = 0; lene = 0; }
  for (int k = gl; k < len; k += G) s_col[grp][k] = A.indices[rb + (int64_t)k * BS] / BS;
  for (int k = gl; k < lene; k += G) s_val[grp][k] = 0.0;
  __syncthreads();
  const bool cells = live && A.cellmark != nullptr && len > 0;
  const int64_t cb = cells ? A.d2c_off[r] : 0;
  const int nc = cells ? (int)(A.d2c_off[r + 1] - cb) : 0;
  const bool row_bc = live && A.bc0 && A.bc0[R];
  auto find_slot = [&](int32_t col) -> int
  {
    int lo = 0, hi = len;
    while (lo < hi)
    {
      const int mid = (lo + hi) >> 1;
      if (s_col[grp][mid] < col) lo = mid + 1; else hi = mid;
    }
    if (lo < len && s_col[grp][lo] == col) return lo;
    *A.error = 1;
    return -1;
  };
  for (int base = 0;; base += G)
  {
    if (__ballot(base + gl < nc) == 0) break;
    const int t = base + gl;
    const int64_t c = t < nc ? (int64_t)A.d2c[cb + t] : -1;
    const uint8_t mark = c >= 0 ? (uint8_t)(A.cellmark[c] & A.mark_mask) : (uint8_t)0;
    double acc[NLOC];
    int csl[ND];
    int32_t cd[ND];
#pragma unroll
    for (int j = 0; j < NLOC; ++j) acc[j] = 0.0;
#pragma unroll
    for (int j = 0; j < ND; ++j) { csl[j] = -1; cd[j] = 0; }
    if (mark)
    {
      int lr = 0;
#pragma unroll
      for (int j = 0; j < ND; ++j)
      {
        cd[j] = A.dofmap[c * ND + j];
        lr = (cd[j] == (int32_t)r) ? j : lr;
      }
#pragma unroll
      for (int j = 0; j < ND; ++j) csl[j] = find_slot(cd[j]);
      // zero BC rows / columns (assemble_matrix_impl.h:151-185): bit j BS + b set = entry dropped.  The mask is taken
      // BEFORE the integrals: with the closed-form row inlined between the dofmap reads and a bc loop placed after it,
      // hipcc 7.2 (gfx950, TDIM = 2) paired the markers with the wrong columns (tests/test_gpu_spaces.py
      // test_vector_elasticity_with_ghost_penalty_and_lifting[2-8-2-*] caught it; the 3-D instantiation was right)
      uint32_t zmask = row_bc ? 0xffffffffu : 0u;
      if (A.bc1 != nullptr)
      {
#pragma unroll
        for (int j = 0; j < ND; ++j)
#pragma unroll
          for (int b = 0; b < BS; ++b) zmask |= A.bc1[(int64_t)cd[j] * BS + b] != 0 ? (1u << (j * BS + b)) : 0u;
      }
      for (int i = 0; i < A.n_cell; ++i)
      {
        const RowIntegral& I = A.cell[i];
        if (mark & (1u << i))
        {
          if (INLINE && I.std_inline)
          {
            if constexpr (INLINE == 2 && DEG == 2 && BS == TDIM)
            {
              Geo<TDIM> g;
              load_cell<TDIM>(A.x, A.conn, c, g);
              jacobian<TDIM>(g);
              const double E = I.params[0], nu = I.params[1];
              p2_elasticity_row<TDIM>(g, lr, kc, E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu)), E / (2.0 * (1.0 + nu)), acc);
            }
            else if constexpr (INLINE == 1)
            {
              Geo<TDIM> g;
              load_cell<TDIM>(A.x, A.conn, c, g);
              jacobian<TDIM>(g);
              int npts;
              const double* wts;
              const double* pts = ref_rule(TDIM, I.qdegree, npts, wts);
              cell_local_row<TDIM, DEG, BS, 2>(I.kernel, I.params, I.point_stride, g, 0.0, npts, pts, wts, fabs(g.detJ),
                                               nullptr, lr, kc, acc);
            }
          }
          else
          {
            const int64_t e = entity_index(I.std_bits, I.std_rank, c);
            const double* T = I.std_tensors + (e * NLOC + lr * BS + kc) * NLOC;
#pragma unroll
            for (int j = 0; j < NLOC; ++j) acc[j] += T[j];
          }
        }
        if (mark & (16u << i))
        {
          for (int64_t e = first_rule(I.rule_keys, I.rule_first, I.rule_mask, (int32_t)c); e < dev_len(I.nr) && I.parent_map[e] == c; ++e)
          {
            const double* T = I.rule_tensors + (e * NLOC + lr * BS + kc) * NLOC;
#pragma unroll
            for (int j = 0; j < NLOC; ++j) acc[j] += T[j];
          }
        }
      }
      static_assert(NLOC <= 32, "one bit per entry of the local row");
#pragma unroll
      for (int k = 0; k < NLOC; ++k) acc[k] = ((zmask >> k) & 1u) ? 0.0 : acc[k];
    }
    if constexpr (ORDERED)
    {
      for (int turn = 0; turn < G; ++turn) // one lane of each group at a time: item order
      {
        if (gl == turn)
        {
#pragma unroll
          for (int j = 0; j < ND; ++j)
            if (csl[j] >= 0)
            {
#pragma unroll
              for (int b = 0; b < BS; ++b) s_val[grp][csl[j] * BS + b] += acc[j * BS + b];
            }
        }
        __syncthreads();
      }
    }
    else
    {
#pragma unroll
      for (int j = 0; j < ND; ++j)
        if (csl[j] >= 0)
        {
#pragma unroll
          for (int b = 0; b < BS; ++b) atomicAdd(&s_val[grp][csl[j] * BS + b], acc[j * BS + b]);
        }
    }
  }
  // ---- facet items (rows next to the interface): ghost penalty / extension terms on block spaces.  The macro
  // row (local dof i of cell 0 or 1, component kc) of the staged (2 NLOC)^2 facet tensor, one cell's half of the
  // columns at a time (NLOC accumulators live, not 2 NLOC)
  const bool facets = live && len > 0 && A.d2f_off != nullptr && (A.mark_mask & 0xF0u) != 0;
  const int64_t fpos = (facets && A.special_mark[r]) ? (int64_t)A.special_pos[r] : -1; // incidence is per special row
  const int64_t fb = fpos >= 0 ? A.d2f_off[fpos] : 0;
  const int nf = fpos >= 0 ? (int)(A.d2f_off[fpos + 1] - fb) : 0;
  for (int base = 0;; base += G)
  {
    if (__ballot(base + gl < nf) == 0) break;
    const int t = base + gl;
    const bool has = t < nf;
    const int64_t f = has ? (int64_t)A.d2f[fb + t] : 0;
    if (DEG == 2 && A.fold_facets == 3)
    {
      if constexpr (DEG == 2)
      {
        // gradient jump, degree 2: facet_nq rank-one records per facet (assemble_rows_kernel has the scalar form);
        // the vector term couples equal components only: row (dof, kc) receives columns (macro dof j, kc)
        constexpr int WF = Elem<TDIM, DEG>::WF, NX = WF - ND;
        double acc[WF];
        int32_t cm[WF];
#pragma unroll
        for (int j = 0; j < WF; ++j) { acc[j] = 0.0; cm[j] = -1; }
        const double* blk = A.facet_tensors + f * (int64_t)(8 + 16 * A.facet_nq);
        if (has)
        {
          const int4* cmp = reinterpret_cast<const int4*>(blk);
          const int4 q0 = cmp[0], q1 = cmp[1], q2 = cmp[2], q3 = cmp[3];
          const int32_t c16[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
          int m = -1;
#pragma unroll
          for (int j = 0; j < WF; ++j) { cm[j] = c16[j]; m = (c16[j] == (int32_t)r) ? j : m; }
          if (c16[15] != NX) *A.error = 4;
          const double2* rec = reinterpret_cast<const double2*>(blk + 8);
          for (int q = 0; q < A.facet_nq; ++q)
          {
            double v[16];
#pragma unroll
            for (int k = 0; k < 8; ++k)
            {
              const double2 p = rec[q * 8 + k];
              v[2 * k] = p.x; v[2 * k + 1] = p.y;
            }
            double jm = 0.0;
#pragma unroll
            for (int j = 0; j < WF; ++j) jm = (j == m) ? v[j] : jm;
            jm *= v[WF];
#pragma unroll
            for (int j = 0; j < WF; ++j) acc[j] += jm * v[j];
          }
        }
        int sl[WF];
#pragma unroll
        for (int j = 0; j < WF; ++j)
        {
          sl[j] = has ? find_slot(cm[j]) : -1;
          if (has && (row_bc || (A.bc1 != nullptr && A.bc1[(int64_t)cm[j] * BS + kc] != 0))) acc[j] = 0.0;
        }
        if constexpr (ORDERED)
        {
          for (int turn = 0; turn < G; ++turn)
          {
            if (gl == turn)
            {
#pragma unroll
              for (int j = 0; j < WF; ++j)
                if (sl[j] >= 0) s_val[grp][sl[j] * BS + kc] += acc[j];
            }
            __syncthreads();
          }
        }
        else
        {
#pragma unroll
          for (int j = 0; j < WF; ++j)
            if (sl[j] >= 0) atomicAdd(&s_val[grp][sl[j] * BS + kc], acc[j]);
        }
      }
      continue;
    }
    int64_t fc[2] = {0, 0};
    int32_t cols[2][ND];
    int irow[2] = {-1, -1};
    if (has)
    {
      fc[0] = A.facet_rows[4 * f]; fc[1] = A.facet_rows[4 * f + 2];
#pragma unroll
      for (int side = 0; side < 2; ++side)
#pragma unroll
        for (int j = 0; j < ND; ++j)
        {
          cols[side][j] = A.dofmap[fc[side] * ND + j];
          irow[side] = cols[side][j] == (int32_t)r ? side * ND + j : irow[side]; // r may be a dof of both cells
        }
    }
    constexpr int W = 2 * NLOC;
    const double* T = A.facet_tensors + f * (int64_t)(W * W);
#pragma unroll
    for (int side = 0; side < 2; ++side)
    {
      double acc[NLOC];
      int csl[ND];
#pragma unroll
      for (int j = 0; j < NLOC; ++j) acc[j] = 0.0;
#pragma unroll
      for (int j = 0; j < ND; ++j) csl[j] = -1;
      if (has)
      {
#pragma unroll
        for (int m = 0; m < 2; ++m) // the macro rows of r in cell 0 and in cell 1 both land in global row R
          if (irow[m] >= 0)
          {
            const double* Tr = T + (int64_t)(irow[m] * BS + kc) * W + side * NLOC;
#pragma unroll
            for (int j = 0; j < NLOC; ++j) acc[j] += Tr[j];
          }
#pragma unroll
        for (int j = 0; j < ND; ++j) csl[j] = find_slot(cols[side][j]);
#pragma unroll
        for (int j = 0; j < ND; ++j)
#pragma unroll
          for (int b = 0; b < BS; ++b)
            if (row_bc || (A.bc1 != nullptr && A.bc1[(int64_t)cols[side][j] * BS + b] != 0)) acc[j * BS + b] = 0.0;
      }
      if constexpr (ORDERED)
      {
        for (int turn = 0; turn < G; ++turn)
        {
          if (gl == turn)
          {
#pragma unroll
            for (int j = 0; j < ND; ++j)
              if (csl[j] >= 0)
              {
#pragma unroll
                for (int b = 0; b < BS; ++b) s_val[grp][csl[j] * BS + b] += acc[j * BS + b];
              }
          }
          __syncthreads();
        }
      }
      else
      {
#pragma unroll
        for (int j = 0; j < ND; ++j)
          if (csl[j] >= 0)
          {
#pragma unroll
            for (int b = 0; b < BS; ++b) atomicAdd(&s_val[grp][csl[j] * BS + b], acc[j * BS + b]);
          }
      }
    }
  }
  __syncthreads();
  if (A.fresh) // single writer on a matrix that holds no earlier contributions: stored, the zero fill skipped these rows
    for (int k = gl; k < lene; k += G) A.values[rb + k] = s_val[grp][k];
  else
    for (int k = gl; k < lene; k += G) A.values[rb + k] += s_val[grp][k];
}

// la::MatrixCSR::set_value(0) restricted to the rows no gather kernel writes: the inactive rows (their diagonal entry)
// (nnz_d: the pattern's entry count, possibly still in HBM -- 0 in a void step, whose marks and row pointers are not
// this step's: nothing is touched then, and never anything at or beyond nnz)
__global__ void __launch_bounds__(kBlock) zero_inactive_rows_kernel(int64_t nrows, int bs, const uint8_t* __restrict__ rowmark,
                                                                    const int64_t* __restrict__ indptr, double* __restrict__ values,
                                                                    DevN nnz_d)
{
  const int64_t nnz = dev_n(nnz_d);
  const int64_t R = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (R >= nrows || nnz == 0 || rowmark[R / bs]) return;
  const int64_t e1 = min(indptr[R + 1], nnz);
  for (int64_t k = max(indptr[R], (int64_t)0); k < e1; ++k) values[k] = 0.0;
}

// ... scalar spaces with the plan's per-tile row counts: a tile of kByteTile rows without an active row is kByteTile
// consecutive diagonal entries -- one contiguous fill from the tile's first row pointer, no per-row reads
__global__ void __launch_bounds__(kBlock) zero_inactive_tiles_kernel(int64_t nrows, const uint8_t* __restrict__ rowmark,
                                                                     const int64_t* __restrict__ tile_counts,
                                                                     const int64_t* __restrict__ indptr, double* __restrict__ values,
                                                                     DevN nnz_d)
{
  const int64_t nnz = dev_n(nnz_d);
  if (nnz == 0) return;
  const int64_t r0 = (int64_t)blockIdx.x * kByteTile;
  const int n = (int)min((int64_t)kByteTile, nrows - r0);
  if (tile_counts[blockIdx.x] == 0)
  {
    const int64_t e0 = indptr[r0];
    if (e0 >= 0 && e0 + n <= nnz) block_fill_run(values + e0, n, 0.0);
    return;
  }
  for (int k = threadIdx.x; k < n; k += kBlock)
    if (!rowmark[r0 + k])
    {
      const int64_t e1 = min(indptr[r0 + k + 1], nnz);
      for (int64_t e = max(indptr[r0 + k], (int64_t)0); e < e1; ++e) values[e] = 0.0;
    }
}

// stage 2, linear forms: b[r] += sum over the marked incident cells of be[local row]
template <int TDIM, int DEG, int G>
__global__ void __launch_bounds__(kWave) assemble_vec_rows_kernel(RowArgs A)
{
  constexpr int ND = Elem<TDIM, DEG>::ND;
  constexpr int RPW = kWave / G;
  const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
  const int64_t ri = CFX_ROW_BLOCK * RPW + grp;
  // second pass of the P1 row-ordered staging: the plain rows WITHOUT a segment (none, usually: the wave leaves at once)
  if (A.vec_skip && dev_n(A.vec_n_odd) == 0) return;
  bool live = ri < dev_n(A.n_active) && A.cellmark != nullptr;
  const int64_t r = live ? A.active_rows[ri] : 0;
  if (live && A.vec_skip && A.vec_skip[r] > 0) live = false; // (segment offsets are stored + 1)
  const int64_t cb = live ? A.d2c_off[r] : 0;
  const int nc = live ? (int)(A.d2c_off[r + 1] - cb) : 0;
  double part = 0.0; // items gl, gl+G, ... in ascending order
  const unsigned dpos = (live && A.slot4) ? A.diagpos[r] : 0u;
  constexpr int R = G <= 4 ? 6 : 4;
  for (int base = 0;; base += R * G)
  {
    if (__ballot(base + gl < nc) == 0) break;
    int64_t cell[R];
    uint8_t mk[R];
    int lr[R];
    uint32_t s4[R];
#pragma unroll
    for (int k = 0; k < R; ++k)
    {
      const int t = base + k * G + gl;
      cell[k] = t < nc ? (int64_t)A.d2c[cb + t] : -1;
      s4[k] = (A.slot4 && t < nc) ? A.slot4[cb + t] : 0u; // static stencil: local index without the dofmap row
      // degree 2: byte 10 of the slot record is the local index
      if (DEG == 2 && A.slotn && t < nc) s4[k] = reinterpret_cast<const uint32_t*>(A.slotn + (cb + t) * 12)[2];
    }
#pragma unroll
    for (int k = 0; k < R; ++k) mk[k] = cell[k] >= 0 ? A.cellmark[cell[k]] : (uint8_t)0;
#pragma unroll
    for (int k = 0; k < R; ++k)
    {
      lr[k] = 0;
      if (!mk[k]) continue;
      if (DEG == 2 && A.slotn) lr[k] = (int)((s4[k] >> 16) & 0xffu);
      else if (A.slot4)
      {
#pragma unroll
        for (int j = 0; j < ND; ++j) lr[k] = (((s4[k] >> (8 * j)) & 0xffu) == dpos) ? j : lr[k];
      }
      else
      {
#pragma unroll
        for (int j = 0; j < ND; ++j) lr[k] = (A.dofmap[cell[k] * ND + j] == (int32_t)r) ? j : lr[k];
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k)
    {
      if (!mk[k]) continue;
      const int64_t c = cell[k];
      for (int i = 0; i < A.n_cell; ++i)
      {
        const RowIntegral& I = A.cell[i];
        if ((mk[k] & (1u << i)) && I.std_tensors)
          part += I.std_tensors[(int64_t)lr[k] * I.n_std + (I.std_by_cell ? c : entity_index(I.std_bits, I.std_rank, c))];
        if (mk[k] & (16u << i))
          for (int64_t e = first_rule(I.rule_keys, I.rule_first, I.rule_mask, (int32_t)c); e < dev_len(I.nr) && I.parent_map[e] == c; ++e)
            part += I.rule_tensors[e * ND + lr[k]];
      }
    }
  }
  // fixed-shape tree over the group's lanes: bitwise reproducible
#pragma unroll
  for (int d = G / 2; d >= 1; d >>= 1) part += __shfl_xor(part, d, G);
  if (live && gl == 0) A.values[r] += part;
}

bool deterministic()
{
  const char* e = getenv("CFX_DETERMINISTIC");
  return e && e[0] == '1';
}

// stage 2, bilinear forms, degree 2: the rows that copied their static neighbour list (cfx_pattern_s::full_rows --
// every incident cell an uncut entity of the one stiffness integral).  An item = incidence entry + its 12-byte
// slot record (cfx::Stencil::slotn): no marks, no dofmap row, no column search, no staged tensor -- the row of the
// element tensor comes from p2_stiffness_row() and lands at the recorded positions of the row's LDS accumulators.
// The row is stored when the matrix was just zeroed.
struct P2PlainArgs
{
  int64_t n;
  const int32_t* rows;
  const double* x;
  const int32_t* conn;
  const int64_t* d2c_off;
  const int32_t* d2c;
  const uint8_t* slotn;
  const int64_t* indptr;
  double* values;
  int fresh;
  int* error;
};

#ifndef CFX_P2PLAIN_WAVES
#define CFX_P2PLAIN_WAVES 4
#endif
template <int TDIM, int G, int CAP, bool ORDERED>
__global__ void __launch_bounds__(kWave, CFX_P2PLAIN_WAVES) assemble_rows_p2_plain_kernel(P2PlainArgs A)
{
  constexpr int ND = Elem<TDIM, 2>::ND, RPW = kWave / G;
  __shared__ double s_val[RPW][CAP];
  const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
  const int64_t ri = CFX_ROW_BLOCK * RPW + grp;
  const bool live = ri < A.n;
  const int64_t r = live ? A.rows[ri] : 0;
  const int64_t rb = live ? A.indptr[r] : 0;
  int len = live ? (int)(A.indptr[r + 1] - rb) : 0;
  if (len > CAP) { *A.error = 2; len = 0; }
  for (int k = gl; k < len; k += G) s_val[grp][k] = 0.0;
  __syncthreads();
  const int64_t cb = live ? A.d2c_off[r] : 0;
  const int nc = (live && len > 0) ? (int)(A.d2c_off[r + 1] - cb) : 0;
  for (int base = 0;; base += G)
  {
    const int t = base + gl;
    const bool has = t < nc;
    if (__ballot(has) == 0) break;
    double acc[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) acc[j] = 0.0;
    uint32_t w0 = 0, w1 = 0, w2 = 0;
    if (has)
    {
      const int64_t c = A.d2c[cb + t];
      const uint32_t* rec = reinterpret_cast<const uint32_t*>(A.slotn + (cb + t) * 12);
      w0 = rec[0]; w1 = rec[1]; w2 = rec[2];
      Geo<TDIM> g;
      load_cell<TDIM>(A.x, A.conn, c, g);
      jacobian<TDIM>(g);
      p2_stiffness_row<TDIM>(g, (int)((w2 >> 16) & 0xffu), 1.0, acc);
    }
    int sl[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j)
    {
      const uint32_t w = j < 4 ? w0 : (j < 8 ? w1 : w2);
      sl[j] = has ? (int)((w >> (8 * (j & 3))) & 0xffu) : -1;
    }
    if constexpr (ORDERED)
    {
      for (int turn = 0; turn < G; ++turn)
      {
        if (gl == turn)
        {
#pragma unroll
          for (int j = 0; j < ND; ++j)
            if (sl[j] >= 0) s_val[grp][sl[j]] += acc[j];
        }
        __syncthreads();
      }
    }
    else
    {
#pragma unroll
      for (int j = 0; j < ND; ++j)
        if (sl[j] >= 0) atomicAdd(&s_val[grp][sl[j]], acc[j]);
    }
  }
  __syncthreads();
  if (A.fresh)
    for (int k = gl; k < len; k += G) A.values[rb + k] = s_val[grp][k];
  else
    for (int k = gl; k < len; k += G) A.values[rb + k] += s_val[grp][k];
}

// stage 2, bilinear forms, vector-valued degree 2: the dofs whose rows copied their static neighbour list
// (cfx_pattern_s::full_rows), BS component rows at a time.  A group of G lanes owns a dof: the items (incident
// cells: id, 12-byte slot record, entity index) are gathered first, one per lane, into LDS; then the group walks
// the items together -- the BS rows of the staged tensor that belong to the dof are (BS x ND BS) contiguous doubles,
// read coalesced, one entry per lane per pass, and added to the LDS rows at the recorded slots (distinct addresses
// within an item, items in list order: no atomics, reproducible).  The generic block kernel repeats the item
// traversal, the dofmap row and ND column searches for each of the BS component rows.
#ifndef CFX_BLOCK_PLAIN_U
#define CFX_BLOCK_PLAIN_U 8
#endif
struct BlockPlainArgs
{
  int64_t n;
  const int32_t* rows;
  const int64_t* d2c_off;
  const int32_t* d2c;
  const uint8_t* slotn;
  const unsigned long long* std_bits;
  const int32_t* std_rank;
  const double* std_tensors;
  const int64_t* indptr;
  double* values;
  int fresh;
  int* error;
};

template <int TDIM, int BS, int G, int CAP>
__global__ void __launch_bounds__(kWave) assemble_rows_block_plain_kernel(BlockPlainArgs A)
{
  constexpr int ND = Elem<TDIM, 2>::ND, NLOC = ND * BS, RPW = kWave / G, NE = BS * NLOC; // NE entries of an item
  constexpr int ROW = CAP * BS;
  __shared__ double s_val[RPW][BS][ROW];
  __shared__ int64_t s_tb[RPW][G];      // per item: offset of its BS tensor rows
  __shared__ uint32_t s_rec[RPW][G][3]; // ... and its slot record
  const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
  const int64_t ri = CFX_ROW_BLOCK * RPW + grp;
  const bool live = ri < A.n;
  const int64_t r = live ? A.rows[ri] : 0;
  const int64_t rb0 = live ? A.indptr[r * BS] : 0;
  int lene = live ? (int)(A.indptr[r * BS + 1] - rb0) : 0; // expanded row length (the BS rows of a dof are equally long)
  if (lene > ROW) { *A.error = 2; lene = 0; }
  for (int k = gl; k < BS * ROW; k += G) (&s_val[grp][0][0])[k] = 0.0;
  const int64_t cb = live ? A.d2c_off[r] : 0;
  const int nc = (live && lene > 0) ? (int)(A.d2c_off[r + 1] - cb) : 0;
  int ncmax = nc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ncmax = max(ncmax, __shfl_xor(ncmax, o, 64)); // the groups loop together
  for (int base = 0; base < ncmax; base += G)
  {
    __syncthreads();
    const int t = base + gl;
    if (t < nc)
    {
      const int64_t c = A.d2c[cb + t];
      const uint32_t* rec = reinterpret_cast<const uint32_t*>(A.slotn + (cb + t) * 12);
      const uint32_t w0 = rec[0], w1 = rec[1], w2 = rec[2];
      const int64_t e = entity_index(A.std_bits, A.std_rank, c);
      s_tb[grp][gl] = (e * NLOC + (int64_t)((w2 >> 16) & 0xffu) * BS) * NLOC;
      s_rec[grp][gl][0] = w0; s_rec[grp][gl][1] = w1; s_rec[grp][gl][2] = w2;
    }
    __syncthreads();
    const int m = min(G, nc - base);
    // U items at a time: their tensor entries are requested together, then added in item order
    constexpr int U = CFX_BLOCK_PLAIN_U, NP = (NE + G - 1) / G;
    for (int it0 = 0; it0 < m; it0 += U)
    {
      double v[U][NP];
#pragma unroll
      for (int u = 0; u < U; ++u)
      {
        const bool on = it0 + u < m;
        const double* T = A.std_tensors + (on ? s_tb[grp][it0 + u] : 0);
#pragma unroll
        for (int p = 0; p < NP; ++p)
        {
          const int k = gl + p * G;
          v[u][p] = (on && k < NE) ? T[k] : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
      {
        if (it0 + u >= m) break;
        const uint32_t w0 = s_rec[grp][it0 + u][0], w1 = s_rec[grp][it0 + u][1], w2 = s_rec[grp][it0 + u][2];
#pragma unroll
        for (int p = 0; p < NP; ++p)
        {
          const int k = gl + p * G;
          if (k < NE)
          {
            const int a = k / NLOC, jb = k - a * NLOC, j = jb / BS, b = jb - j * BS;
            const uint32_t w = j < 4 ? w0 : (j < 8 ? w1 : w2);
            const int slot = (int)((w >> (8 * (j & 3))) & 0xffu);
            s_val[grp][a][slot * BS + b] += v[u][p];
          }
        }
      }
    }
  }
  __syncthreads();
  for (int a = 0; a < BS; ++a)
  {
    const int64_t rb = live ? A.indptr[r * BS + a] : 0;
    if (A.fresh)
      for (int k = gl; k < lene; k += G) A.values[rb + k] = s_val[grp][a][k];
    else
      for (int k = gl; k < lene; k += G) A.values[rb + k] += s_val[grp][a][k];
  }
}

// stage 2, bilinear forms, vector-valued degree 2, elasticity on uncut cells WITHOUT staged tensors: the dofs whose
// rows copied their static neighbour list, all BS component rows at a time.  An item = incidence entry + 12-byte slot
// record, one item per lane; the lane forms the BS x (ND BS) block row of the element tensor in closed form from the
// cell's four vertices (p2_gradient_gram_row: ~50 flops per 3 x 3 block) and adds it to the dof's LDS rows at the
// recorded slots.  Replaces elasticity_tensors_mfma (7.2 KB written per uncut cell) + assemble_rows_block_plain (the
// same bytes read back): the only HBM traffic left is the incidence / slot / geometry gathers and the stored rows.
struct BlockP2Args
{
  int64_t n;
  const int32_t* rows;
  const double* x;
  const int32_t* conn;
  const int64_t* d2c_off;
  const int32_t* d2c;
  const uint8_t* slotn;
  const int64_t* indptr;
  double* values;
  double lmbda, mu;
  int fresh;
  int* error;
};

#ifndef CFX_BLOCK_P2_WAVES
#define CFX_BLOCK_P2_WAVES 2
#endif
// 3 x 3 (2 x 2) gradient Gram block of one (row dof, column dof) pair with RUNTIME indices: the row is described by its
// barycentric pair (a, b) (a vertex row: a == b, Gb = 0), the column by (jc, jd) likewise; Gs = grad(lam_k), Ga / Gb =
// |K| grad(lam_a / lam_b).  Same numbers as p2_gradient_gram_row(), whose column index is a compile-time constant.
template <int TDIM>
__device__ __forceinline__ void p2_gradient_gram_block(const double* __restrict__ Gs, const double* __restrict__ Ga,
                                                       const double* __restrict__ Gb, int a, int b, bool rvert, int jc, int jd,
                                                       bool jvert, double (&H)[TDIM][TDIM])
{
  constexpr double m_d = TDIM == 2 ? 1.0 / 12.0 : 1.0 / 20.0, m_s = 2.0 * m_d;
  constexpr double inv = 1.0 / (TDIM + 1);
  constexpr double vv_d = 16.0 * m_d - 8.0 * inv + 1.0, vv_s = 16.0 * m_s - 8.0 * inv + 1.0;
  constexpr double s_d = 4.0 * m_d - inv, s_s = 4.0 * m_s - inv;
  double kac, kad, kbc, kbd;
  if (rvert)
  {
    kac = jvert ? ((a == jc) ? vv_s : vv_d) : 4.0 * ((a == jd) ? s_s : s_d);
    kad = jvert ? 0.0 : 4.0 * ((a == jc) ? s_s : s_d);
    kbc = 0.0; kbd = 0.0;
  }
  else
  {
    kac = jvert ? 4.0 * ((jc == b) ? s_s : s_d) : 16.0 * ((b == jd) ? m_s : m_d);
    kad = jvert ? 0.0 : 16.0 * ((b == jc) ? m_s : m_d);
    kbc = jvert ? 4.0 * ((jc == a) ? s_s : s_d) : 16.0 * ((a == jd) ? m_s : m_d);
    kbd = jvert ? 0.0 : 16.0 * ((a == jc) ? m_s : m_d);
  }
  double ua[TDIM], ub[TDIM];
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    const double gc = Gs[jc * TDIM + d], gd = Gs[jd * TDIM + d];
    ua[d] = kac * gc + kad * gd;
    ub[d] = kbc * gc + kbd * gd;
  }
#pragma unroll
  for (int p = 0; p < TDIM; ++p)
#pragma unroll
    for (int q = 0; q < TDIM; ++q) H[p][q] = Ga[p] * ua[q] + Gb[p] * ub[q];
}

// A wavefront takes 64 consecutive dofs of the list (one lane each reads the dof's row pointers and incidence range
// into LDS), then walks them RPW = 64 / G at a time.  A group of G lanes owns a dof: phase 1 (one item per lane)
// turns the cell's vertices into grad(lam_k) and the row-side gradients in LDS; phase 2 (one item at a time, one
// COLUMN dof per lane) forms the block of column j and adds its BS x BS entries to the dof's LDS rows with plain
// read-modify-writes -- the columns of an item have distinct slots and the items follow each other in list order, so
// there are no atomics (FP64 LDS atomics from sixteen lanes onto shared slots: 13.7 ms for this kernel) and the
// result is reproducible.  The gathers are three dependent hops (incidence entry -> connectivity row -> vertices)
// at ~2 us each with a handful of wavefronts per CU (the dof's BS x 3 len accumulators bound the occupancy), so they
// run as a software pipeline across the groups of dofs: while group i is in phase 2, the vertices of i + 1, the
// connectivity rows of i + 2 and the incidence entries of i + 3 are in flight; every load is unconditional (clamped
// indices) so that nothing drains the vector-memory counter early.  A dof with more than G incident cells takes
// extra, unpipelined passes.
template <int TDIM, int G, int CAP>
__global__ void __launch_bounds__(kWave, CFX_BLOCK_P2_WAVES) assemble_rows_block_p2_kernel(BlockP2Args A)
{
  constexpr int BS = TDIM, NV = TDIM + 1, ND = Elem<TDIM, 2>::ND, RPW = kWave / G, ROW = CAP * BS;
  constexpr int NGEO = NV * TDIM + 2 * TDIM;
  static_assert(G >= ND, "one lane per column dof");
  __shared__ __align__(16) double s_val[RPW][BS * ROW]; // the dof's BS rows back to back (row stride = its row length):
                                                        // the image of its contiguous run of the CSR value array
  __shared__ double s_geo[RPW][G][NGEO]; // per staged item: grad(lam_k) [NV][TDIM], Ga, Gb
  __shared__ uint32_t s_rec[RPW][G][4];  // ... its slot record and (a | b << 8 | vertex << 16)
  __shared__ int64_t s_rb[kWave], s_cb[kWave];
  __shared__ int32_t s_len[kWave], s_nc[kWave];
  const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
  const int64_t i0 = CFX_ROW_BLOCK * kWave;
  if (i0 >= A.n) return;
  {
    const int64_t i = i0 + lane;
    const bool live = i < A.n;
    const int64_t r = A.rows[live ? i : A.n - 1];
    const int64_t rb = A.indptr[r * BS];
    int len = (int)(A.indptr[r * BS + 1] - rb); // expanded row length (the BS rows of a dof are consecutive and equally long)
    const int64_t cb = A.d2c_off[r];
    int nc = (int)(A.d2c_off[r + 1] - cb);
    if (len > ROW) { *A.error = 2; len = 0; }
    if (!live || len == 0) { len = 0; nc = 0; }
    s_rb[lane] = rb; s_cb[lane] = cb; s_len[lane] = len; s_nc[lane] = nc;
  }
  __syncthreads();
  const int nd_here = (int)min((int64_t)kWave, A.n - i0);
  const int nq = (nd_here + RPW - 1) / RPW;
  const double lmbda = A.lmbda, mu = A.mu;
  // this lane's column dof j = gl: its barycentric pair
  constexpr int ea2[3] = {1, 0, 0}, eb2[3] = {2, 2, 1};
  constexpr int ea3[6] = {2, 1, 1, 0, 0, 0}, eb3[6] = {3, 3, 2, 3, 2, 1};
  const bool jcol = gl < ND, jvert = gl < NV;
  int jc = jvert ? gl : 0, jd = jc;
#pragma unroll
  for (int e = 0; e < ND - NV; ++e)
  {
    jc = (gl == NV + e) ? (TDIM == 2 ? ea2[e % 3] : ea3[e % 6]) : jc;
    jd = (gl == NV + e) ? (TDIM == 2 ? eb2[e % 3] : eb3[e % 6]) : jd;
  }
  // ---- the three load stages (unconditional: a lane without an item reads the dof's first item again)
  struct Item { int32_t c; uint32_t w0, w1, w2; };
  auto load_item = [&](int q, int t) -> Item
  {
    const int d = min(q, nq - 1) * RPW + grp;
    const int nc = s_nc[d];
    const int64_t e = s_cb[d] + ((t < nc) ? t : 0);
    const uint32_t* rec = reinterpret_cast<const uint32_t*>(A.slotn + e * 12);
    Item it;
    it.c = A.d2c[e]; it.w0 = rec[0]; it.w1 = rec[1]; it.w2 = rec[2];
    return it;
  };
  struct Conn { int32_t v[NV]; };
  auto load_conn = [&](int32_t c) -> Conn
  {
    Conn k;
    if constexpr (TDIM == 3)
    {
      const int4 r4 = *reinterpret_cast<const int4*>(A.conn + (int64_t)c * 4);
      k.v[0] = r4.x; k.v[1] = r4.y; k.v[2] = r4.z; k.v[3] = r4.w;
    }
    else
    {
#pragma unroll
      for (int i = 0; i < NV; ++i) k.v[i] = A.conn[(int64_t)c * NV + i];
    }
    return k;
  };
  struct Coords { double x[NV][TDIM]; };
  auto load_coords = [&](const Conn& k) -> Coords
  {
    Coords X;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int d = 0; d < TDIM; ++d) X.x[i][d] = A.x[3 * (int64_t)k.v[i] + d];
    return X;
  };
  // phase 1 of one item: geometry -> LDS
  auto stage_item = [&](const Item& it, const Coords& X)
  {
    const int lr = (int)((it.w2 >> 16) & 0xffu);
    Geo<TDIM> g;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int d = 0; d < TDIM; ++d) g.x[i][d] = X.x[i][d];
    jacobian<TDIM>(g);
    int a = lr, b = lr;
#pragma unroll
    for (int e = 0; e < ND - NV; ++e)
    {
      a = (lr == NV + e) ? (TDIM == 2 ? ea2[e % 3] : ea3[e % 6]) : a;
      b = (lr == NV + e) ? (TDIM == 2 ? eb2[e % 3] : eb3[e % 6]) : b;
    }
    const bool rvert = lr < NV;
    double Gs[NV][TDIM];
#pragma unroll
    for (int d = 0; d < TDIM; ++d)
    {
      double s0 = 0.0;
#pragma unroll
      for (int k = 0; k < TDIM; ++k) { Gs[k + 1][d] = g.K[k][d]; s0 -= g.K[k][d]; }
      Gs[0][d] = s0;
    }
    const double vol = fabs(g.detJ) * (TDIM == 2 ? 0.5 : 1.0 / 6.0);
    double* geo = s_geo[grp][gl];
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
      for (int d = 0; d < TDIM; ++d) geo[k * TDIM + d] = Gs[k][d];
#pragma unroll
    for (int d = 0; d < TDIM; ++d)
    {
      double va = 0.0, vb = 0.0;
#pragma unroll
      for (int k = 0; k < NV; ++k) { va = (a == k) ? Gs[k][d] : va; vb = (b == k) ? Gs[k][d] : vb; }
      geo[NV * TDIM + d] = va * vol;
      geo[NV * TDIM + TDIM + d] = rvert ? 0.0 : vb * vol;
    }
    s_rec[grp][gl][0] = it.w0; s_rec[grp][gl][1] = it.w1; s_rec[grp][gl][2] = it.w2;
    s_rec[grp][gl][3] = (uint32_t)a | ((uint32_t)b << 8) | (rvert ? 0x10000u : 0u);
  };
  // phase 2: the m staged items of this group, one at a time, one column dof per lane
  auto add_items = [&](int m, int lene)
  {
    int mmax = m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mmax = max(mmax, __shfl_xor(mmax, o, 64)); // the groups loop together
    for (int it = 0; it < mmax; ++it)
    {
      if (it < m && jcol)
      {
        const uint32_t w3 = s_rec[grp][it][3];
        const uint32_t w = s_rec[grp][it][gl >> 2];
        const int slot = (int)((w >> (8 * (gl & 3))) & 0xffu);
        const double* geo = s_geo[grp][it];
        double H[TDIM][TDIM];
        p2_gradient_gram_block<TDIM>(geo, geo + NV * TDIM, geo + NV * TDIM + TDIM, (int)(w3 & 0xffu), (int)((w3 >> 8) & 0xffu),
                                     (w3 & 0x10000u) != 0, jc, jd, jvert, H);
        double tr = 0.0;
#pragma unroll
        for (int d = 0; d < TDIM; ++d) tr += H[d][d];
        tr *= mu;
        double* row = &s_val[grp][0] + slot * BS;
#pragma unroll
        for (int a = 0; a < BS; ++a)
#pragma unroll
          for (int b = 0; b < BS; ++b) row[a * lene + b] += lmbda * H[a][b] + mu * H[b][a] + (a == b ? tr : 0.0);
      }
    }
  };
  // ---- prologue: items of groups 0, 1, 2; connectivity of 0, 1; vertices of 0
  Item itA = load_item(0, gl), itB = load_item(1, gl), itC = load_item(2, gl);
  Conn cnA = load_conn(itA.c), cnB = load_conn(itB.c);
  Coords xA = load_coords(cnA);
  for (int q = 0; q < nq; ++q)
  {
    const int d = q * RPW + grp;
    const bool dlive = d < nd_here;
    const int nc = dlive ? s_nc[d] : 0;
    const int lene = dlive ? s_len[d] : 0;
    for (int k = gl; k < BS * lene; k += G) s_val[grp][k] = 0.0;
    if (gl < nc) stage_item(itA, xA);
    // next stages: vertices of q + 1, connectivity of q + 2, incidence entries of q + 3 -- in flight during phase 2
    const Coords xN = load_coords(cnB);
    const Conn cnN = load_conn(itC.c);
    const Item itN = load_item(q + 3, gl);
    __syncthreads();
    add_items(min(G, nc), lene);
    for (int base = G; base < nc; base += G) // (rare: more incident cells than lanes)
    {
      __syncthreads();
      const int t = base + gl;
      if (t < nc)
      {
        const Item it = load_item(q, t);
        stage_item(it, load_coords(load_conn(it.c)));
      }
      __syncthreads();
      add_items(min(G, nc - base), lene);
    }
    __syncthreads();
    // the BS rows of a dof are one contiguous run of BS x lene values: the whole wavefront streams the runs of the
    // RPW dofs one after the other, 16 B per lane (groups of G lanes storing 8 B each left the matrix write at
    // 1.4 TB/s: 8.4 ms of this kernel's 13.3 at BASELINE config 5's share)
#pragma unroll
    for (int g2 = 0; g2 < RPW; ++g2)
    {
      const int d2 = q * RPW + g2;
      if (d2 >= nd_here) break;
      const int n2 = BS * s_len[d2];
      double* out = A.values + s_rb[d2];
      const double* src = s_val[g2];
      // head: one value if the run starts on an odd index (16 B stores need 16 B alignment), then pairs, then the tail
      const int head = (int)((reinterpret_cast<uintptr_t>(out) >> 3) & 1) & (n2 > 0 ? 1 : 0);
      if (A.fresh)
      {
        if (lane == 0 && head) out[0] = src[0];
        const int npair = (n2 - head) >> 1;
        for (int k = lane; k < npair; k += kWave)
        {
          double2 v;
          v.x = src[head + 2 * k]; v.y = src[head + 2 * k + 1];
          *reinterpret_cast<double2*>(out + head + 2 * k) = v;
        }
        if (lane == 0 && ((n2 - head) & 1)) out[n2 - 1] = src[n2 - 1];
      }
      else
        for (int k = lane; k < n2; k += kWave) out[k] += src[k];
    }
    __syncthreads(); // s_val / s_geo are reused by the next group
    itA = itB; itB = itC; itC = itN;
    cnA = cnB; cnB = cnN;
    xA = xN;
  }
}

// degree-2 elasticity on a vector space without coefficient, integrated exactly (qdegree >= 2): the uncut cells need
// no staged tensors -- closed-form rows (p2_gradient_gram_row).  CFX_P2_CLOSED=0 keeps the staged MFMA tensors.
inline bool p2_elasticity_closed(const cfx_form_s* a, const cfx_integral_dev& I)
{
  const cfx_space_s* V = a->V;
  const char* cf = getenv("CFX_P2_CLOSED");
  return a->rank == 2 && V->degree == 2 && V->bs == V->mesh->tdim && I.type == CFX_CELL && I.kernel == CFX_K_ELASTICITY
         && I.coefficient.n == 0 && I.qdegree >= 2 && !(cf && cf[0] == '0');
}

// ---------------------------------------------------------------------------
// stage 1 of the cut cells of a degree-2 scalar space: ONE ND x ND tensor per cut cell, summed over every runtime rule
// of every cell integral of the form (stiffness over the volume rule + Nitsche over the interface rule of the same
// cell, ...).  kCutLanes2 lanes per cut cell share the points of each rule; mass and Nitsche terms accumulate the
// symmetric tensor point by point (N_i and the normal derivatives dn_i straight from the barycentric coordinates:
// vertex (4 lam_i - 1) alpha_i, edge 4 (lam_b alpha_a + lam_a alpha_b), alpha_k = grad(lam_k) . n), stiffness terms
// accumulate the 15 barycentric moments and take the closed form (p2_stiffness_row_moments) at the end.  The gather then
// reads one 80-byte row per (row, cut cell) item -- before: a hash probe, a moment record, a Nitsche tensor row and the
// closed form per integral and item, with the generic tabulation kernel (one thread per rule and row) behind it
// (assemble_cells_cut 12 ms + ~half of assemble_rows_cut's 47 ms at BASELINE config 4).
// ---------------------------------------------------------------------------
struct CutSlot
{
  int kernel, point_stride;
  double params[2];
  const int32_t* offsets;
  const int32_t* parent_map;
  int64_t nr;
  const double* points;
  const double* weights;
  const double* point_data;
  const int32_t* rule_keys;
  const int32_t* rule_first;
  unsigned rule_mask;
  const int32_t* cut_first; // [n cut cells]: first rule of the k-th cut cell in this integral, -1 none (plan.cut_first)
};
struct CutTensorArgs
{
  int64_t n;
  const int32_t* cut_cells;
  const uint8_t* cellmark;
  const double* x;
  const int32_t* conn;
  int n_slots;
  CutSlot slot[4];
  double* out;
};
constexpr int kCutLanes2 = 4;

#ifndef CFX_CUTT_WAVES
#define CFX_CUTT_WAVES 2
#endif
template <int TDIM>
__global__ void __launch_bounds__(kBlock, CFX_CUTT_WAVES) cut_tensors_p2_kernel(CutTensorArgs A)
{
  constexpr int NV = TDIM + 1, NE = TDIM == 2 ? 3 : 6, ND = NV + NE, NP = ND * (ND + 1) / 2;
  constexpr int NO = (NP + kCutLanes2 - 1) / kCutLanes2;
  constexpr int ea2[3] = {1, 0, 0}, eb2[3] = {2, 2, 1};
  constexpr int ea3[6] = {2, 1, 1, 0, 0, 0}, eb3[6] = {3, 3, 2, 3, 2, 1};
  __shared__ double s_T[kBlock / kCutLanes2][NP + 1];
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t k0 = tid / kCutLanes2;
  const int sub = (int)(tid - k0 * kCutLanes2);
  const bool live = k0 < A.n;
  const int64_t k = live ? k0 : A.n - 1;
  const int ci = threadIdx.x / kCutLanes2;
  const int64_t c = A.cut_cells[k];
  // the first rule of the cell in every integral comes from a table parallel to the cut-cell list (coalesced, and in
  // flight together with the cell id) instead of a hash probe behind the cell mark
  int32_t efirst[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) efirst[i] = (live && i < A.n_slots && A.slot[i].cut_first) ? A.slot[i].cut_first[k] : -1;
  Geo<TDIM> g;
  load_cell<TDIM>(A.x, A.conn, c, g);
  jacobian<TDIM>(g);
  const double h = cell_diameter<TDIM>(g);
  // The kCutLanes2 lanes of a cell walk ALL points of its rules (the tabulation of N and dn is ~100 flops) and
  // share the ND (ND + 1) / 2 entries of the symmetric tensor: entry p belongs to lane p mod kCutLanes2.  (One lane
  // per share of the POINTS with the whole tensor in registers needed 480 VGPRs.)  Every lane keeps the 15 moments.
  double To[NO], mom[16];
#pragma unroll
  for (int p = 0; p < NO; ++p) To[p] = 0.0;
#pragma unroll
  for (int p = 0; p < 16; ++p) mom[p] = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
  {
    if (i >= A.n_slots || efirst[i] < 0) continue;
    const CutSlot& S = A.slot[i];
    for (int64_t e = efirst[i]; e < S.nr && (e == efirst[i] || S.parent_map[e] == c); ++e)
    {
      const int32_t q0 = S.offsets[e], q1 = S.offsets[e + 1];
      for (int32_t q = q0; q < q1; ++q)
      {
        double lam[NV];
        lam[0] = 1.0;
#pragma unroll
        for (int t = 0; t < TDIM; ++t) { lam[t + 1] = S.points[(int64_t)q * TDIM + t]; lam[0] -= lam[t + 1]; }
        const double w = S.weights[q];
        if (S.kernel == CFX_K_STIFFNESS)
        {
          mom[0] += w;
          int idx = 1 + NV;
#pragma unroll
          for (int a = 0; a < NV; ++a)
          {
            const double wa = w * lam[a];
            mom[1 + a] += wa;
#pragma unroll
            for (int b = a; b < NV; ++b) mom[idx++] += wa * lam[b];
          }
          continue;
        }
        double N[ND], dn[ND];
#pragma unroll
        for (int a = 0; a < NV; ++a) N[a] = lam[a] * (2.0 * lam[a] - 1.0);
#pragma unroll
        for (int e2 = 0; e2 < NE; ++e2)
        {
          const int a = TDIM == 2 ? ea2[e2 % 3] : ea3[e2 % 6], b = TDIM == 2 ? eb2[e2 % 3] : eb3[e2 % 6];
          N[NV + e2] = 4.0 * lam[a] * lam[b];
        }
        double gam = 0.0;
        if (S.kernel == CFX_K_NITSCHE) // -dn(u) v - dn(v) u + gamma / h u v with the per-point normal
        {
          const double* nrm = S.point_data + (int64_t)q * S.point_stride;
          double al[NV]; // alpha_k = grad(lam_k) . n,  grad(lam_0) = -sum_t K[t][.], grad(lam_{t+1}) = K[t][.]
          al[0] = 0.0;
#pragma unroll
          for (int t = 0; t < TDIM; ++t)
          {
            double v = 0.0;
#pragma unroll
            for (int d = 0; d < TDIM; ++d) v += g.K[t][d] * nrm[d];
            al[t + 1] = v; al[0] -= v;
          }
#pragma unroll
          for (int a = 0; a < NV; ++a) dn[a] = (4.0 * lam[a] - 1.0) * al[a];
#pragma unroll
          for (int e2 = 0; e2 < NE; ++e2)
          {
            const int a = TDIM == 2 ? ea2[e2 % 3] : ea3[e2 % 6], b = TDIM == 2 ? eb2[e2 % 3] : eb3[e2 % 6];
            dn[NV + e2] = 4.0 * (lam[b] * al[a] + lam[a] * al[b]);
          }
          gam = S.params[0] / h;
        }
        else // CFX_K_MASS: the same accumulation with gamma = 1 and no normal derivatives
        {
#pragma unroll
          for (int a = 0; a < ND; ++a) dn[a] = 0.0;
          gam = 1.0;
        }
        int p = 0;
#pragma unroll
        for (int a = 0; a < ND; ++a)
        {
          const double ga = gam * N[a] - dn[a]; // (gamma N_a - dn_a) N_b - N_a dn_b
#pragma unroll
          for (int b = a; b < ND; ++b)
          {
            const double v = w * (ga * N[b] - N[a] * dn[b]);
            To[p / kCutLanes2] += ((p % kCutLanes2) == sub) ? v : 0.0;
            ++p;
          }
        }
      }
    }
  }
  // the symmetric sums go through LDS: a row picks its entries with a runtime index
#pragma unroll
  for (int p = 0; p < NP; ++p)
    if ((p % kCutLanes2) == sub) s_T[ci][p] = To[p / kCutLanes2];
  __syncthreads();
  if (!live) return;
  // rows sub, sub + kCutLanes2, ... per lane: the sums plus the stiffness row from the moments
  double* out = A.out + k * (int64_t)(ND * ND);
  for (int i = sub; i < ND; i += kCutLanes2)
  {
    double row[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j)
    {
      const int a = i < j ? i : j, b = i < j ? j : i;
      row[j] = s_T[ci][a * ND - a * (a - 1) / 2 + (b - a)];
    }
    if (mom[0] != 0.0) p2_stiffness_row_moments<TDIM>(g, i, mom, row);
    double2* o2 = reinterpret_cast<double2*>(out + i * ND);
#pragma unroll
    for (int j = 0; j < ND / 2; ++j) o2[j] = make_double2(row[2 * j], row[2 * j + 1]);
  }
}

struct Stage1
{
  std::vector<DevArray<double>> buffers;
  DevArray<double> t2;      // linear forms, P1: row-ordered staging of the uncut cells (run_vector), else empty
  double* vec_t2 = nullptr;
  bool vec_blocks = false;  // linear forms: the uncut cells go by cell block (run_vector), nothing is staged per cell
  DevArray<double> part;    // ... the partials of the step
};

// bound of the walks "for (e = first rule of c; e < bound && parent_map[e] == c; ++e)": the number of rules -- or, when
// it is still in HBM and the list ends with a sentinel, the capacity + 1 as a plain number (no load in the loop)
inline DevN rule_bound(const cfx_rules_s* R)
{
  if (R->nr.pending() && R->parent_sentinel) return DevN(R->nr.cap() + 1);
  return R->nr.devn();
}

inline VecArgs vec_args(cfx_form_s* L, const cfx_integral_dev& I)
{
  cfx_space_s* V = L->V;
  VecArgs A{};
  A.x = V->mesh->x.p; A.conn = V->mesh->conn.p;
  A.kernel = I.kernel; A.qdegree = I.qdegree; A.point_stride = I.point_stride;
  for (int k = 0; k < 8; ++k) A.params[k] = I.params[k];
  A.dofmap = V->dofmap.p;
  A.coeff = I.coefficient.n > 0 ? I.coefficient.p : nullptr;
  return A;
}

// the sin-product source term on a P1 space over the geometry dofmap: the series kernels serve it
template <int DEG>
inline bool source_series_ok(const cfx_space_s* V, const cfx_integral_dev& I)
{
  const int field = (int)I.params[0];
  const char* fs = getenv("CFX_SOURCE_SERIES");
  return DEG == 1 && I.kernel == CFX_L_SOURCE && I.coefficient.n == 0 && (field == CFX_F_SINPROD || field == CFX_F_POISSON_RHS)
         && V->dofmap.p == V->mesh->conn.p && !(fs && fs[0] == '0');
}

// stage 1 of a linear form by cell block: the partials of the uncut cells of integral I (mark bit `mark`)
template <int TDIM, int DEG>
void vec_block_partials(cfx_form_s* L, const cfx_integral_dev* Istd, uint8_t mark, const RowArgs& R, double* part)
{
  cfx_space_s* V = L->V;
  cfx_row_plan& plan = row_plan(L);
  const VecBlocks& S = space_vec_blocks(V);
  constexpr int ND = Elem<TDIM, DEG>::ND, B = ND <= 8 ? 256 : 128;
  if (S.B != B) throw Error(CFX_ERR_RUNTIME, "vec_block_partials: block size of the space's cell blocks");
  VecArgs A{};
  A.x = V->mesh->x.p; A.conn = V->mesh->conn.p; // (a form of rule integrals alone: the kernel still requests the connectivity rows)
  if (Istd) A = vec_args(L, *Istd);
  VecBlockArgs P{};
  P.n_active = plan.n_vb_active; P.active = plan.vb_active.p; P.base = plan.vb_base.p; P.u_off = S.u_off.p;
  P.slot = S.slot.p; P.seg = S.seg.p; P.cellmark = plan.cellmark.p; P.mark = mark; P.ncells = V->mesh->ncells; P.part = part;
  P.n_slots = R.n_cell;
  for (int s = 0; s < R.n_cell; ++s)
  {
    const RowIntegral& I = R.cell[s];
    P.rules[s] = {I.parent_map, I.rule_tensors ? I.nr : DevN(0), I.rule_tensors, I.rule_keys, I.rule_first, I.rule_mask};
  }
  if (P.n_active == 0) {}
  else if (Istd && source_series_ok<DEG>(V, *Istd))
  {
    if constexpr (DEG == 1)
    {
      const int64_t blocks = std::min<int64_t>(P.n_active, 256 * CFX_SOURCE_BLOCKS_PER_CU);
      launch("vec_blocks_std", vec_blocks_sin_p1_kernel<TDIM>, dim3((unsigned)blocks), dim3(kBlock), 0, A, P);
    }
  }
  else if (plan.vb_merged)
    launch("vec_blocks_std", vec_blocks_kernel<TDIM, DEG, B, 2>, dim3((unsigned)P.n_active), dim3(B), 0, A, P);
  else
    launch("vec_blocks_std", vec_blocks_kernel<TDIM, DEG, B, 0>, dim3((unsigned)P.n_active), dim3(B), 0, A, P);
  if (plan.n_vb_cut > 0)
  {
    P.n_active = plan.n_vb_cut; P.active = plan.vb_cut.p;
    launch("vec_blocks_cut", vec_blocks_kernel<TDIM, DEG, B, 1>, dim3((unsigned)P.n_active), dim3(B), 0, A, P);
  }
}

template <int TDIM, int DEG>
void vec_tensors(cfx_form_s* L, const cfx_integral_dev& I, bool runtime, double* out, double* t2 = nullptr, int64_t out_cells = 0)
{
  cfx_space_s* V = L->V;
  if (user_integrand_known(I.kernel))
  {
    // an integrand compiled at run time (cfx_rtc.hip): its wrapper writes the layout the rows read -- rules [nr][ND],
    // uncut cells [ND][n] by entity or [ND][ncells] by cell
    if (runtime) user_stage1(L, I, true, out, 0, 0);
    else user_stage1(L, I, false, out, out_cells > 0 ? 2 : 1, out_cells > 0 ? out_cells : I.n_entities.cap());
    return;
  }
  VecArgs A = vec_args(L, I);
  A.out_cells = out_cells;
  if (t2 && !runtime)
  {
    A.t2 = t2; A.t2off = row_plan(L).vec_t2off.p; A.cpos = space_stencil(V).cpos.p;
  }
  A.out = out;
  if (!runtime)
  {
    A.n = I.n_entities; A.entities = I.entities.p;
    if (source_series_ok<DEG>(V, I))
    {
      if constexpr (DEG == 1)
      {
        // a resident grid: every thread walks ~n / (256 CUs x blocks x 256) cells through its software pipeline
        const int64_t blocks = std::min<int64_t>((A.n.cap + kBlock - 1) / kBlock, 256 * CFX_SOURCE_BLOCKS_PER_CU);
        launch("vec_tensors_std", vec_source_sin_p1_kernel<TDIM>, dim3((unsigned)blocks), dim3(kBlock), 0, A);
      }
    }
    else
      launch("vec_tensors_std", vec_tensors_kernel<TDIM, DEG, false>, grid_for(A.n.cap), dim3(kBlock), 0, A);
  }
  else
  {
    const cfx_rules_s* R = I.rules;
    A.n = R->nr; A.offsets = R->offsets.p; A.parent_map = R->parent_map.p; A.points = R->points.p;
    A.weights = R->weights.p; A.point_data = I.point_data.n > 0 ? I.point_data.p : nullptr;
    const dim3 grid = grid_for(A.n.cap * CFX_VEC_CUT_LANES);
    if (I.kernel == CFX_L_SOURCE)
      launch("vec_tensors_cut", vec_tensors_kernel<TDIM, DEG, true, CFX_VEC_CUT_LANES, 1>, grid, dim3(kBlock), 0, A);
    else if (I.kernel == CFX_L_NITSCHE_RHS)
      launch("vec_tensors_cut", vec_tensors_kernel<TDIM, DEG, true, CFX_VEC_CUT_LANES, 2>, grid, dim3(kBlock), 0, A);
    else
      launch("vec_tensors_cut", vec_tensors_kernel<TDIM, DEG, true, CFX_VEC_CUT_LANES>, grid, dim3(kBlock), 0, A);
  }
}

// stage 1 + RowArgs for a form
// the rule integrals of a degree-2 scalar bilinear form that cut_tensors_p2_kernel can sum into one tensor per cut cell
inline bool p2_cut_tensors_ok(const cfx_form_s* a)
{
  const char* e = getenv("CFX_P2_CUT_TENSORS");
  if (e && e[0] == '0') return false;
  if (a->rank != 2 || a->V->degree != 2 || a->V->bs != 1) return false;
  bool any = false;
  for (const auto& I : a->integrals)
    if (I.type == CFX_CELL && I.rules && I.rules->nr.cap() > 0)
    {
      if (!(I.kernel == CFX_K_STIFFNESS || I.kernel == CFX_K_MASS || I.kernel == CFX_K_NITSCHE) || I.coefficient.n > 0) return false;
      any = true;
    }
  return any;
}

template <int TDIM, int DEG, int BS = 1>
RowArgs prepare(cfx_form_s* a, Stage1& st, bool combine_cuts = false)
{
  constexpr int ND = Elem<TDIM, DEG>::ND * BS; // local tensor dimension (scalar dofs x block size)
  cfx_row_plan& plan = row_plan(a);
  cfx_space_s* V = a->V;
  RowArgs A{};
  A.x = V->mesh->x.p; A.conn = V->mesh->conn.p; A.dofmap = V->dofmap.p;
  A.iso_geometry = (DEG == 1 && V->dofmap.p == V->mesh->conn.p) ? 1 : 0;
  A.n_active = plan.n_active_rows; A.active_rows = plan.active_rows.p;
  if (plan.any_cells)
  {
    const Adjacency& adj = V->dof_cells();
    A.d2c_off = adj.offsets.p; A.d2c = adj.cells.p; A.cellmark = plan.cellmark.p;
  }
  const int64_t tsize = a->rank == 2 ? ND * ND : ND;
  A.n_cell = plan.n_cell_slots;
  for (int s = 0; s < plan.n_cell_slots; ++s)
  {
    const int ii = plan.cell_slot_integral[s];
    const cfx_integral_dev& I = a->integrals[ii];
    RowIntegral& R = A.cell[s];
    R.kernel = I.kernel; R.qdegree = I.qdegree; R.point_stride = I.point_stride;
    for (int k = 0; k < 8; ++k) R.params[k] = I.params[k];
    const int64_t n_ent = I.n_entities.cap(); // (capacity while the list's length is in HBM: strides and buffers)
    if (n_ent > 0)
    {
      R.std_bits = reinterpret_cast<const unsigned long long*>(plan.std_bits[s].p);
      R.std_rank = plan.std_rank[s].p;
    }
    // uncut P1 stiffness is one point: cheaper to recompute than to stage
    const char* inl = getenv("CFX_STD_INLINE");
    // (degree 2: staging 100 doubles per uncut cell would be 38 GB at config 4 -- the row of the
    // local tensor is recomputed per (row, cell) item instead, 4-14 quadrature points)
    const bool inline_ok = BS > 1 ? (I.kernel == CFX_K_ELASTICITY || I.kernel == CFX_K_MASS || I.kernel == CFX_K_STIFFNESS)
                                  : (DEG == 1 ? I.kernel == CFX_K_STIFFNESS
                                              : (I.kernel == CFX_K_STIFFNESS || I.kernel == CFX_K_MASS));
    R.std_inline = (a->rank == 2 && inline_ok && kRowsInline<DEG> && I.coefficient.n == 0 && !(inl && inl[0] == '0')) ? 1 : 0;
    if (BS > 1 && !(inl && inl[0] == '1')) R.std_inline = 0; // block spaces stage their uncut tensors (see assemble_matrix_rows)
    if (R.std_inline && A.iso_geometry && !(inl && inl[0] == '1')) R.std_inline = 2;
    if (R.std_inline == 1) A.iso_geometry = 0; // a generic inline integral: the ISO kernel cannot serve this form
    {
      // degree-2 scalar stiffness without coefficient: closed-form row per item, no staged tensors
      const char* cf = getenv("CFX_P2_CLOSED");
      if (DEG == 2 && BS == 1 && a->rank == 2 && I.kernel == CFX_K_STIFFNESS && I.coefficient.n == 0 && I.qdegree >= 2
          && !(cf && cf[0] == '0'))
        R.std_inline = 3;
      // ... and the vector-valued elasticity term likewise (closed-form block rows)
      if (DEG == 2 && BS == TDIM && p2_elasticity_closed(a, I)) R.std_inline = 3;
    }
    if (a->rank == 1 && st.vec_blocks && n_ent > 0)
      R.std_tensors = nullptr; // (the uncut cells reach the rows as block partials, run_vector)
    else if (!R.std_inline && n_ent > 0)
    {
      // linear forms whose entity list covers a good part of the mesh stage [ND][ncells]: the rows index the record by
      // the cell they hold anyway, without the bitset + rank lookup of the entity index (two gathers per item)
      const char* bc = getenv("CFX_VEC_BY_CELL");
      const bool by_cell = a->rank == 1 && BS == 1 && !st.vec_t2 && n_ent * 4 >= V->mesh->ncells && !(bc && bc[0] == '0');
      st.buffers.emplace_back((by_cell ? V->mesh->ncells : n_ent) * tsize);
      R.std_tensors = st.buffers.back().p;
      R.n_std = by_cell ? V->mesh->ncells : n_ent;
      R.std_by_cell = by_cell ? 1 : 0;
      if (a->rank == 2) dump_integral(a, ii, 1, st.buffers.back().p);
      else if constexpr (BS == 1) vec_tensors<TDIM, DEG>(a, I, false, st.buffers.back().p, st.vec_t2, by_cell ? V->mesh->ncells : 0);
    }
    const int64_t n_rules = I.rules ? I.rules->nr.cap() : 0;
    if (n_rules > 0 && combine_cuts)
    {
      // (the rule lookups stay valid for the kernels that ask for them; nothing is staged per integral)
      R.parent_map = I.rules->parent_map.p; R.nr = rule_bound(I.rules);
      R.rule_keys = plan.rule_keys[s].p; R.rule_first = plan.rule_first[s].p; R.rule_mask = plan.rule_mask[s];
    }
    else if (n_rules > 0)
    {
      R.parent_map = I.rules->parent_map.p; R.nr = rule_bound(I.rules);
      R.rule_keys = plan.rule_keys[s].p; R.rule_first = plan.rule_first[s].p; R.rule_mask = plan.rule_mask[s];
      const char* cm = getenv("CFX_P2_MOMENTS");
      const bool moments = DEG == 2 && BS == 1 && a->rank == 2 && I.kernel == CFX_K_STIFFNESS && I.coefficient.n == 0
                           && !(cm && cm[0] == '0');
      st.buffers.emplace_back(n_rules * (moments ? 16 : tsize));
      R.rule_tensors = st.buffers.back().p;
      R.rule_moments = moments ? 1 : 0;
      if (moments) dump_cut_moments(a, ii, st.buffers.back().p);
      else if (a->rank == 2) dump_integral(a, ii, 2, st.buffers.back().p);
      else if constexpr (BS == 1) vec_tensors<TDIM, DEG>(a, I, true, st.buffers.back().p);
    }
  }
  if constexpr (DEG == 2 && BS == 1)
  {
    if (combine_cuts)
    {
      plan_cut_cells(a);
      if (plan.n_cut_cells > 0)
      {
        CutTensorArgs C{};
        C.n = plan.n_cut_cells; C.cut_cells = plan.cut_cells.p; C.cellmark = plan.cellmark.p;
        C.x = A.x; C.conn = A.conn; C.n_slots = plan.n_cell_slots;
        for (int s = 0; s < plan.n_cell_slots; ++s)
        {
          const cfx_integral_dev& I = a->integrals[plan.cell_slot_integral[s]];
          CutSlot& S = C.slot[s];
          S.kernel = I.kernel; S.point_stride = I.point_stride; S.params[0] = I.params[0]; S.params[1] = I.params[1];
          if (I.rules && I.rules->nr.value() > 0)
          {
            S.offsets = I.rules->offsets.p; S.parent_map = I.rules->parent_map.p; S.nr = I.rules->nr.value();
            S.points = I.rules->points.p; S.weights = I.rules->weights.p;
            S.point_data = I.point_data.n > 0 ? I.point_data.p : nullptr;
            S.rule_keys = plan.rule_keys[s].p; S.rule_first = plan.rule_first[s].p; S.rule_mask = plan.rule_mask[s];
            S.cut_first = plan.cut_first[s].p;
          }
        }
        st.buffers.emplace_back(plan.n_cut_cells * (int64_t)(ND * ND));
        C.out = st.buffers.back().p;
        launch("cut_tensors_p2", cut_tensors_p2_kernel<TDIM>, grid_for(C.n * kCutLanes2), dim3(kBlock), 0, C);
        A.cut_tensors = C.out;
        A.cut_bits = reinterpret_cast<const unsigned long long*>(plan.cut_bits.p);
        A.cut_rank = plan.cut_rank.p;
      }
    }
  }
  bool has_facets = false;
  A.fold_facets = plan.fold_ok ? 1 : 0;
  for (const auto& I : a->integrals)
  {
    has_facets = has_facets || I.type == CFX_INTERIOR_FACET;
    if (I.type == CFX_INTERIOR_FACET && I.kernel == CFX_K_EXTENSION_L2) A.fold_facets = 0; // (bad, root) pairs
  }
  if (has_facets && plan.nfacets.cap() > 0)
  {
    A.d2f_off = plan.d2f_offsets.p; A.d2f = plan.d2f.p; A.facet_rows = plan.facet_rows.p;
    A.special_mark = plan.special_mark.p; A.special_pos = plan.special_pos.p;
    // 3-D P1 scalar spaces: stage 1 stores the macro tensor already folded over the shared dofs, 25 doubles per
    // facet instead of 64 (fold_facets = 2); elsewhere the P1 fold happens in the gather (1) or not at all (0)
    const char* fe = getenv("CFX_FACET_FOLD_STAGE1");
    bool user_facets = false; // (integrands compiled at run time stage the whole macro tensor: folded in the gather, or not)
    for (int s = 0; s < plan.n_facet_slots; ++s)
      user_facets = user_facets || user_integrand_known(a->integrals[plan.facet_slot_integral[s]].kernel);
    if (TDIM == 3 && DEG == 1 && BS == 1 && A.fold_facets && a->rank == 2 && !(fe && fe[0] == '0') && !user_facets) A.fold_facets = 2;
    // ... and when every facet term is the gradient jump over standard facets, the folded tensor is rank one:
    // stage 1 stores its vector and weight (8 doubles per facet, fold_facets = 3)
    bool rank_one = DEG == 1 && BS == 1 && A.fold_facets && a->rank == 2 && !(fe && (fe[0] == '0' || fe[0] == '2'));
    for (int s = 0; s < plan.n_facet_slots; ++s)
    {
      const cfx_integral_dev& I = a->integrals[plan.facet_slot_integral[s]];
      rank_one = rank_one && I.kernel == CFX_K_GHOST_GRADJUMP && I.rules == nullptr && I.n_std < 0;
    }
    if (rank_one) A.fold_facets = 3;
    // degree 2, scalar: nq rank-one records per facet when every facet term is the gradient jump at one degree
    if (DEG == 2 && A.fold_facets && a->rank == 2 && !(fe && fe[0] == '0'))
    {
      bool low = true;
      int qd = -1;
      for (int s = 0; s < plan.n_facet_slots; ++s)
      {
        const cfx_integral_dev& I = a->integrals[plan.facet_slot_integral[s]];
        low = low && I.kernel == CFX_K_GHOST_GRADJUMP && I.rules == nullptr && I.n_std < 0
              && (qd < 0 || qd == I.qdegree);
        qd = I.qdegree;
      }
      const int nq = low ? quad_npoints(TDIM - 1, qd) : 0;
      if (low && nq >= 1 && nq <= 6) { A.fold_facets = 3; A.facet_nq = nq; }
    }
    const int64_t fsize = A.fold_facets == 3 ? (DEG == 2 ? 8 + 16 * A.facet_nq : 10)
                                             : (A.fold_facets == 2 ? (ND + 1) * (ND + 1) : 4 * ND * ND);
    st.buffers.emplace_back(plan.nfacets.cap() * fsize);
    A.facet_tensors = st.buffers.back().p;
    int64_t o = 0;
    for (int s = 0; s < plan.n_facet_slots; ++s)
    {
      const int ii = plan.facet_slot_integral[s];
      if (A.fold_facets == 3 && DEG == 2) dump_facet_jumps_p2(a, ii, A.facet_nq, st.buffers.back().p + o * fsize);
      else if (A.fold_facets == 3) dump_facet_jumps_p1(a, ii, st.buffers.back().p + o * fsize, nullptr); // (a non-conforming row is reported by the gather)
      else dump_integral(a, ii, 1, st.buffers.back().p + o * fsize, A.fold_facets == 2);
      o += a->integrals[ii].n_entities.cap(); // (several facet lists: exact lengths, cfx::row_plan)
    }
  }
  return A;
}

// the gather kernels' error word: what each value means (raised at once, or by cfx_step_end inside a step)
constexpr const char* kEntryMissing = "assemble_matrix: entry not in the sparsity pattern";
inline void raise_gather_error(int err)
{
  require(err != 1, CFX_ERR_RUNTIME, kEntryMissing);
  require(err != 2, CFX_ERR_RUNTIME, "assemble_matrix: row longer than the gather kernel's capacity");
  require(err != 5, CFX_ERR_RUNTIME, "assemble_matrix: a stencil-subset row does not match its sparsity pattern");
  require(err != 4, CFX_ERR_INVALID_ARGUMENT, "assemble_matrix: a facet row does not join two cells across a shared facet");
  require(err == 0, CFX_ERR_RUNTIME, "assemble_matrix: the row gather reported an error");
}

template <int TDIM, int DEG>
int run_matrix(cfx_form_s* a, cfx_pattern_s* P, const int8_t* bc0, const int8_t* bc1, double* values, bool fresh)
{
  Stage1 st;
  // degree 2: the split path (below) takes the cut cells' contributions as one tensor per cut cell
  bool combine_cuts = false;
  if constexpr (DEG == 2)
  {
    cfx_row_plan& plan0 = row_plan(a);
    const char* fs0 = getenv("CFX_ROWS_SPLIT");
    const int mr0 = P->max_row_len;
    combine_cuts = P->split_plan == plan0.serial && mr0 > 64 && mr0 <= 256
                   && (plan0.n_special_rows.value() * 2 <= plan0.n_active_rows.value() || (fs0 && fs0[0] == '1'))
                   && plan0.n_special_rows.value() > 0
                   && p2_cut_tensors_ok(a);
  }
  RowArgs A = prepare<TDIM, DEG>(a, st, combine_cuts);
  A.fresh = fresh ? 1 : 0;
  A.bc0 = bc0; A.bc1 = bc1; A.indptr = P->indptr.p; A.indices = P->indices.p; A.values = values;
  ErrorFlag err(CFX_ERR_RUNTIME, kEntryMissing, raise_gather_error);
  A.error = err.p;
  // `fresh` = MatrixCSR.set_value(0) fused into this call (nothing has written to `values` yet: stage 1 fills its own
  // buffers).  P1 on the split path: every active row has a FIRST writer that stores (the tile / plain kernels their
  // rows, assemble_rows_p1 the rows it is given), so only the inactive rows (one diagonal entry each) are zeroed --
  // 0.2 instead of 0.7 ms at 512^3; every other path fills the whole array first.  CFX_LAZY_ZERO=0: always fill.
  bool lazy_zero = false;
  auto fill_all = [&]() { if (fresh) dev_fill(values, 0, sizeof(double) * (size_t)P->nnz.cap()); };
  if constexpr (DEG > 1)
  {
    // degree 2: the host sizes work by the row counts -- read them back if they are still in HBM
    cfx_row_plan& pl = row_plan(a);
    (void)pl.n_active_rows.value(); (void)pl.n_special_rows.value(); (void)pl.n_plain_rows.value();
    A.n_active = pl.n_active_rows;
  }
  if (A.n_active.cap == 0) fill_all();
  if (A.n_active.cap > 0)
  {
    const bool det = deterministic();
    const int mr = P->max_row_len;
    cfx_row_plan& plan = row_plan(a);
#define CFX_ROWS(GG, CAPP, NAME, ARGS)                                                                    \
  do                                                                                                      \
  {                                                                                                       \
    const dim3 grid = row_grid(((ARGS).n_active.cap + (kWave / GG) - 1) / (kWave / GG));                  \
    if (det) launch(NAME, assemble_rows_kernel<TDIM, DEG, GG, CAPP, true>, grid, dim3(kWave), 0, ARGS);   \
    else launch(NAME, assemble_rows_kernel<TDIM, DEG, GG, CAPP, false>, grid, dim3(kWave), 0, ARGS);      \
  } while (0)
    // P1 space on the geometry dofmap whose uncut-cell integrals are all inline stiffness:
    // lean kernel for the uncut items of every row, generic kernel for the rule / facet
    // items of the rows next to the interface
    bool split = false;
    if constexpr (DEG == 1)
    {
      unsigned inline_bits = 0;
      bool all_inline = A.iso_geometry != 0 && mr <= 64;
      for (int s = 0; s < A.n_cell; ++s)
      {
        if (A.cell[s].std_inline == 2) inline_bits |= 1u << s;
        else if (A.cell[s].std_bits) all_inline = false; // staged uncut tensors: generic path
      }
      const char* fs1 = getenv("CFX_ROWS_SPLIT");
      const bool split_p1 = all_inline && inline_bits && (2 * plan.n_special_rows.hint() <= plan.n_active_rows.hint() || (fs1 && fs1[0] == '1'));
      const char* lz = getenv("CFX_LAZY_ZERO");
      // (only when P was laid out from this very plan: a pattern of a larger form -- several forms assembled into one
      // matrix -- has full-length rows where this plan has none, and they must all be zeroed)
      lazy_zero = fresh && split_p1 && P->built_plan == plan.serial && !(lz && lz[0] == '0');
      if (!lazy_zero) fill_all();
      if (split_p1)
      {
        split = true;
        RowArgs F = A;
        F.inline_bits = inline_bits;
        if (lazy_zero)
        {
          F.fresh = 2; // assemble_rows_p1 is the first writer of the rows it is given (all active rows, or the interface rows)
          const int64_t nr = P->nrows, ntiles = (nr + kByteTile - 1) / kByteTile;
          if (plan.row_tile_counts.n == ntiles)
            launch("zero_inactive_rows", zero_inactive_tiles_kernel, dim3((unsigned)ntiles), dim3(kBlock), 0, nr, plan.rowmark.p,
                   plan.row_tile_counts.p, P->indptr.p, values, P->nnz.devn());
          else
            launch("zero_inactive_rows", zero_inactive_rows_kernel, grid_for(nr), dim3(kBlock), 0, nr, 1, plan.rowmark.p,
                   P->indptr.p, values, P->nnz.devn());
        }
        // rows laid out as stencil subsets by build_pattern from this very plan: slots by popcount
        const Stencil& stn = a->V->stencil;
        if (stn.usable && plan.n_plain_rows.cap() > 0 && P->stencil_plan == plan.serial)
        {
          RowArgs Q = F;
          Q.n_active = plan.n_plain_rows; Q.active_rows = plan.plain_rows.p;
          Q.slot4 = stn.slot4.p; Q.diagpos = stn.diagpos.p; Q.st_off = stn.offsets.p; Q.st_nbr = stn.nbr.p;
          plain_row_masks(a);
          Q.plain_masks = plan.plain_masks.p; Q.plain_uniform = plan.plain_uniform.p;
          // row tiles (coalesced staging of the mesh-static tables), when every tile fits an LDS capacity class
          const Stencil& stt = space_stencil_tiles(a->V);
          const int tcls = stt.tiles_usable ? tile_class(stt) : -1;
          const bool use_tiles = tcls >= 0 && plan.n_plain_tiles.cap() > 0;
          if (use_tiles)
          {
            TileArgs T{};
            T.x = Q.x; T.n_tiles = plan.n_plain_tiles; T.tile_first = plan.plain_tile_first.p; T.tile_id = plan.plain_tile_id.p;
            T.n_plain = plan.n_plain_rows; T.rows = plan.plain_rows.p; T.masks = plan.plain_masks.p; T.uniform = plan.plain_uniform.p;
            T.d2c_off = Q.d2c_off; T.d2c = Q.d2c; T.slot4 = stn.slot4.p; T.cellmark = Q.cellmark;
            T.st_off = stn.offsets.p; T.st_nbr = stn.nbr.p; T.st_loc = stt.st_loc.p; T.diagpos = stn.diagpos.p;
            T.tile_voff = stt.tile_voff.p; T.tile_verts = stt.tile_verts.p;
            T.indptr = Q.indptr; T.values = Q.values; T.bc0 = Q.bc0; T.bc1 = Q.bc1;
            T.inline_bits = inline_bits; T.fresh = Q.fresh; T.error = Q.error; T.ndofs = a->V->ndofs;
            const dim3 gt = row_grid(plan.n_plain_tiles.cap());
            if (tcls == 0)
            {
              if (det) launch("assemble_tiles_plain", assemble_tiles_plain_kernel<TDIM, true, 0>, gt, dim3(kWave), 0, T);
              else launch("assemble_tiles_plain", assemble_tiles_plain_kernel<TDIM, false, 0>, gt, dim3(kWave), 0, T);
            }
            else
            {
              if (det) launch("assemble_tiles_plain", assemble_tiles_plain_kernel<TDIM, true, 1>, gt, dim3(kWave), 0, T);
              else launch("assemble_tiles_plain", assemble_tiles_plain_kernel<TDIM, false, 1>, gt, dim3(kWave), 0, T);
            }
          }
          const dim3 gq = row_grid((Q.n_active.cap + (kWave / CFX_PLAIN_G) - 1) / (kWave / CFX_PLAIN_G));
          const char* stage_env = getenv("CFX_PLAIN_STAGE");
          const bool stage = stn.max_len <= 32 && !(stage_env && stage_env[0] == '0');
          // plain rows are subsets of their stencil: the LDS footprint follows the longest stencil
          // (16 covers the 15-vertex stencil of Kuhn meshes: 4.4 KB per block instead of 8.4 KB)
          const int plain_cap = stn.max_len <= 16 ? 16 : (stn.max_len <= 32 ? 32 : 64);
#define CFX_PLAIN(CAPP)                                                                                                \
  do                                                                                                                   \
  {                                                                                                                    \
    if (det && stage) launch("assemble_rows_plain", assemble_rows_plain_kernel<TDIM, CFX_PLAIN_G, CAPP, true, true>, gq, dim3(kWave), 0, Q);   \
    else if (det) launch("assemble_rows_plain", assemble_rows_plain_kernel<TDIM, CFX_PLAIN_G, CAPP, true, false>, gq, dim3(kWave), 0, Q);     \
    else if (stage) launch("assemble_rows_plain", assemble_rows_plain_kernel<TDIM, CFX_PLAIN_G, CAPP, false, true>, gq, dim3(kWave), 0, Q);   \
    else launch("assemble_rows_plain", assemble_rows_plain_kernel<TDIM, CFX_PLAIN_G, CAPP, false, false>, gq, dim3(kWave), 0, Q);             \
  } while (0)
          if (use_tiles) {}
          else if (plain_cap == 16) CFX_PLAIN(16); else if (plain_cap == 32) CFX_PLAIN(32); else CFX_PLAIN(64);
#undef CFX_PLAIN
          // the uncut items of the interface rows keep the searching kernel
          F.n_active = plan.n_special_rows; F.active_rows = plan.special_rows.p;
        }
        const dim3 grid = row_grid((F.n_active.cap + 7) / 8);
        if (mr <= 32)
        {
          if (det) launch("assemble_rows_p1", assemble_rows_p1_kernel<TDIM, 8, 32, true>, grid, dim3(kWave), 0, F);
          else launch("assemble_rows_p1", assemble_rows_p1_kernel<TDIM, 8, 32, false>, grid, dim3(kWave), 0, F);
        }
        else
        {
          if (det) launch("assemble_rows_p1", assemble_rows_p1_kernel<TDIM, 8, 64, true>, grid, dim3(kWave), 0, F);
          else launch("assemble_rows_p1", assemble_rows_p1_kernel<TDIM, 8, 64, false>, grid, dim3(kWave), 0, F);
        }
        RowArgs S = A;
        S.n_active = plan.n_special_rows; S.active_rows = plan.special_rows.p; S.mark_mask = 0xF0u;
        if (S.n_active.cap > 0)
        {
#define CFX_ROWS_CUT(GG, CAPP)                                                                                       \
  do                                                                                                                 \
  {                                                                                                                  \
    const dim3 grid = row_grid((S.n_active.cap + (kWave / GG) - 1) / (kWave / GG));                                  \
    if (det) launch("assemble_rows_cut", assemble_rows_kernel<TDIM, DEG, GG, CAPP, true, true, false>, grid, dim3(kWave), 0, S);   \
    else launch("assemble_rows_cut", assemble_rows_kernel<TDIM, DEG, GG, CAPP, false, true, false>, grid, dim3(kWave), 0, S);      \
  } while (0)
          if (mr <= 32) CFX_ROWS_CUT(CFX_CUT_G, 32);
          else CFX_ROWS_CUT(CFX_CUT_G, 64);
#undef CFX_ROWS_CUT
        }
      }
    }
    if constexpr (DEG > 1)
    {
      // one pass over the interface rows (uncut + cut-cell + facet items in one kernel, rows stored) when the cut
      // cells come as combined tensors and the pattern kept the other hashed rows apart: then every active row has
      // exactly one writer and only the inactive rows are zeroed
      const bool one_pass = combine_cuts && A.cut_tensors != nullptr && P->odd_plan == plan.serial && P->full_plan == plan.serial;
      const char* lz = getenv("CFX_LAZY_ZERO");
      lazy_zero = fresh && one_pass && P->built_plan == plan.serial && !(lz && lz[0] == '0');
      if (!lazy_zero) fill_all();
      else
        launch("zero_inactive_rows", zero_inactive_rows_kernel, grid_for(P->nrows), dim3(kBlock), 0, P->nrows, 1, plan.rowmark.p,
               P->indptr.p, values, P->nnz.devn());
      // degree 2: (a) the uncut items of every row with the lean kernel, short rows (<= 64 columns:
      // the edge dofs, ~5 cells each) 8 lanes per row, long rows 16; (b) rule + facet items of the
      // interface rows with the full kernel.  Needs the row partition made with the pattern.
      const char* fs = getenv("CFX_ROWS_SPLIT"); // '1': split whatever the share of the interface rows (tests on small meshes)
      if (P->split_plan == plan.serial && mr > 64 && mr <= 256
          && (plan.n_special_rows.value() * 2 <= plan.n_active_rows.value() || (fs && fs[0] == '1')))
      {
        split = true;
#define CFX_LEAN(GG, CAPP, ROWS, NROWS)                                                                              \
  do                                                                                                                 \
  {                                                                                                                  \
    RowArgs Q = A;                                                                                                   \
    Q.mark_mask = 0x0Fu; Q.active_rows = (ROWS); Q.n_active = (NROWS); Q.fresh = lazy_zero ? 2 : Q.fresh;           \
    const dim3 grid = row_grid((Q.n_active.cap + (kWave / GG) - 1) / (kWave / GG));                                  \
    if (Q.n_active.cap > 0)                                                                                          \
    {                                                                                                                \
      if (det) launch("assemble_rows_uncut", assemble_rows_kernel<TDIM, DEG, GG, CAPP, true, false>, grid, dim3(kWave), 0, Q);  \
      else launch("assemble_rows_uncut", assemble_rows_kernel<TDIM, DEG, GG, CAPP, false, false>, grid, dim3(kWave), 0, Q);     \
    }                                                                                                                \
  } while (0)
        // (decided here, used below: the dedicated interface kernel takes the pattern's short / long lists whole)
        bool lean_ok = one_pass && (plan.nfacets.value() == 0 || (A.fold_facets == 3 && A.facet_tensors != nullptr));
        for (int q = 0; q < A.n_cell; ++q)
          lean_ok = lean_ok && (A.cell[q].std_bits == nullptr || A.cell[q].std_inline == 3);
        if (const char* li = getenv("CFX_P2_INTERFACE")) lean_ok = lean_ok && li[0] != '0';

        if (lean_ok) {}
        else if (one_pass)
          CFX_LEAN(16, 256, P->odd_rows.p, P->n_odd_rows);
        else
        {
          CFX_LEAN(8, 64, P->short_rows.p, P->n_short_rows);
          CFX_LEAN(16, 256, P->mid_rows.p, P->n_mid_rows);
          CFX_LEAN(16, 256, P->long_rows.p, P->n_long_rows);
        }
        if (P->full_plan == plan.serial && P->n_full_rows > 0)
        {
          // the rows that copied their static list (not in the two lists above)
          const Stencil& stn = a->V->stencil;
          bool closed = !bc0 && !bc1 && stn.slotn_ok && stn.max_len <= 128;
          for (int s = 0; s < A.n_cell; ++s) closed = closed && (A.cell[s].std_bits == nullptr || A.cell[s].std_inline == 3);
          if (closed)
          {
            P2PlainArgs Q{};
            Q.n = P->n_full_rows; Q.rows = P->full_rows.p; Q.x = A.x; Q.conn = A.conn; Q.d2c_off = A.d2c_off; Q.d2c = A.d2c;
            Q.slotn = stn.slotn.p; Q.indptr = A.indptr; Q.values = A.values; Q.fresh = A.fresh; Q.error = A.error;
            // Lists of at most 72 columns (Kuhn meshes: 65): four lanes per row, sixteen rows per wavefront -- most rows are
            // edge dofs with 4-8 incident cells, which leave a quarter to a half of eight lanes idle in a kernel that spends
            // half its time on VALU issue (geometry + closed-form row per item).  (Splitting the list by row length for a
            // 32-column form of the short rows gains as much here and costs 1.8 ms in the pattern.)
            if (stn.max_len <= 72)
            {
              const dim3 grid = row_grid((Q.n + 15) / 16);
              if (det) launch("assemble_rows_p2_plain", assemble_rows_p2_plain_kernel<TDIM, 4, 72, true>, grid, dim3(kWave), 0, Q);
              else launch("assemble_rows_p2_plain", assemble_rows_p2_plain_kernel<TDIM, 4, 72, false>, grid, dim3(kWave), 0, Q);
            }
            else
            {
              const dim3 grid = row_grid((Q.n + 7) / 8);
              if (det) launch("assemble_rows_p2_plain", assemble_rows_p2_plain_kernel<TDIM, 8, 128, true>, grid, dim3(kWave), 0, Q);
              else launch("assemble_rows_p2_plain", assemble_rows_p2_plain_kernel<TDIM, 8, 128, false>, grid, dim3(kWave), 0, Q);
            }
          }
          else
            CFX_LEAN(16, 256, P->full_rows.p, P->n_full_rows);
        }
#undef CFX_LEAN
        RowArgs S = A;
        S.n_active = plan.n_special_rows; S.active_rows = plan.special_rows.p; S.mark_mask = 0xF0u;
        if (S.n_active.cap > 0)
        {
          const dim3 grid = row_grid((S.n_active.cap + 3) / 4);
          // every uncut-cell integral the closed-form stiffness row, every facet term as rank-one records (or none):
          // the dedicated interface kernel; else the general one
          if (lean_ok)
          {
            S.mark_mask = 0xFFu; S.fresh = lazy_zero ? 2 : S.fresh;
            if (plan.nfacets.value() == 0) { S.d2f_off = nullptr; S.facet_nq = 0; }
            // the rows of at most 64 columns (edge dofs: most of them) 8 lanes per row with a quarter of the LDS, the
            // others 16 lanes and 256 columns; the two lists hold every hashed row, the few plain ones among them included
            RowArgs S1 = S, S2 = S, S3 = S;
            S1.n_active = P->n_short_rows; S1.active_rows = P->short_rows.p;
            S2.n_active = P->n_long_rows; S2.active_rows = P->long_rows.p;
            S3.n_active = P->n_mid_rows; S3.active_rows = P->mid_rows.p;
            const dim3 g1 = row_grid((S1.n_active.cap + 7) / 8), g2 = row_grid((S2.n_active.cap + 3) / 4);
            const dim3 g3 = row_grid((S3.n_active.cap + 3) / 4);
            if (S3.n_active.cap > 0)
            {
              if (det) launch("assemble_rows_cut", assemble_rows_p2_interface_kernel<TDIM, 16, 128, true>, g3, dim3(kWave), 0, S3);
              else launch("assemble_rows_cut", assemble_rows_p2_interface_kernel<TDIM, 16, 128, false>, g3, dim3(kWave), 0, S3);
            }
            if (S1.n_active.cap > 0)
            {
              if (det) launch("assemble_rows_cut", assemble_rows_p2_interface_kernel<TDIM, 8, 64, true>, g1, dim3(kWave), 0, S1);
              else launch("assemble_rows_cut", assemble_rows_p2_interface_kernel<TDIM, 8, 64, false>, g1, dim3(kWave), 0, S1);
            }
            if (S2.n_active.cap > 0)
            {
              if (det) launch("assemble_rows_cut", assemble_rows_p2_interface_kernel<TDIM, 16, 256, true>, g2, dim3(kWave), 0, S2);
              else launch("assemble_rows_cut", assemble_rows_p2_interface_kernel<TDIM, 16, 256, false>, g2, dim3(kWave), 0, S2);
            }

          }
          else if (one_pass)
          {
            S.mark_mask = 0xFFu; S.fresh = lazy_zero ? 2 : S.fresh;
            if (det) launch("assemble_rows_cut", assemble_rows_kernel<TDIM, DEG, 16, 256, true, true, true, true>, grid, dim3(kWave), 0, S);
            else launch("assemble_rows_cut", assemble_rows_kernel<TDIM, DEG, 16, 256, false, true, true, true>, grid, dim3(kWave), 0, S);
          }
          else if (combine_cuts && S.cut_tensors)
          {
            if (det) launch("assemble_rows_cut", assemble_rows_kernel<TDIM, DEG, 16, 256, true, true, false, true>, grid, dim3(kWave), 0, S);
            else launch("assemble_rows_cut", assemble_rows_kernel<TDIM, DEG, 16, 256, false, true, false, true>, grid, dim3(kWave), 0, S);
          }
          else if (det) launch("assemble_rows_cut", assemble_rows_kernel<TDIM, DEG, 16, 256, true, true, false>, grid, dim3(kWave), 0, S);
          else launch("assemble_rows_cut", assemble_rows_kernel<TDIM, DEG, 16, 256, false, true, false>, grid, dim3(kWave), 0, S);
        }
      }
    }
    if (!split)
    {
      A.mark_mask = 0xFFu;
      if (mr <= 32) CFX_ROWS(8, 32, "assemble_rows", A);
      else if (mr <= 64) CFX_ROWS(8, 64, "assemble_rows", A);
      else if (DEG > 1 && mr <= 256)
      {
        if constexpr (DEG > 1) CFX_ROWS(16, 256, "assemble_rows", A); // P2 rows: 20-70 columns, up to ~200 with facets
      }
      else CFX_ROWS(64, 512, "assemble_rows_wide", A);
    }
#undef CFX_ROWS
  }
  return err.deferred ? 0 : read_scalar(err.p);
}

template <int TDIM, int DEG, int BS>
int run_matrix_block(cfx_form_s* a, cfx_pattern_s* P, const int8_t* bc0, const int8_t* bc1, double* values, bool fresh)
{
  Stage1 st;
  RowArgs A = prepare<TDIM, DEG, BS>(a, st);
  A.bc0 = bc0; A.bc1 = bc1; A.indptr = P->indptr.p; A.indices = P->indices.p; A.values = values;
  A.mark_mask = 0xFFu;
  ErrorFlag err(CFX_ERR_RUNTIME, kEntryMissing, raise_gather_error);
  A.error = err.p;
  {
    // block spaces: the host sizes work by the row counts -- read them back if they are still in HBM
    cfx_row_plan& pl = row_plan(a);
    (void)pl.n_active_rows.value(); (void)pl.n_special_rows.value(); (void)pl.n_plain_rows.value();
    A.n_active = pl.n_active_rows;
  }
  // `fresh` = MatrixCSR.set_value(0) fused into this call: every active row is written by exactly one kernel below
  // (the dof-at-a-time kernel or the searching one), which then STORES its rows; only the inactive rows (one
  // diagonal entry each) are zeroed -- the full fill of the value array was 1.8 of 10.8 ms at BASELINE config 5's share
  A.fresh = fresh ? 1 : 0;
  if (fresh)
  {
    cfx_row_plan& plan0 = row_plan(a);
    // (a pattern built from another, larger form: its rows off this plan's active set are not single diagonal
    // entries written by nobody -- fill everything)
    if (P->built_plan == plan0.serial)
      launch("zero_inactive_rows", zero_inactive_rows_kernel, grid_for(P->nrows), dim3(kBlock), 0, P->nrows, BS, plan0.rowmark.p,
             P->indptr.p, values, P->nnz.devn());
    else
      dev_fill(values, 0, sizeof(double) * (size_t)P->nnz.value());
  }
  if (A.n_active.cap > 0)
  {
    const bool det = deterministic();
    const int mr = P->max_row_len; // scalar columns per row
    bool inl = false, closed = false;
    for (int s = 0; s < A.n_cell; ++s)
    {
      inl = inl || (A.cell[s].std_inline != 0 && A.cell[s].std_inline != 3);
      closed = closed || A.cell[s].std_inline == 3;
    }
    require(!(inl && closed), CFX_ERR_RUNTIME, "assemble_matrix: mixed inline modes on a block space");
#define CFX_BLOCK_V(GG, CAPP, ORD, INL) \
  launch("assemble_rows_block", assemble_rows_block_kernel<TDIM, DEG, BS, GG, CAPP, ORD, INL>, grid, dim3(kWave), 0, A)
#define CFX_BLOCK(GG, CAPP)                                                                                          \
  do                                                                                                                 \
  {                                                                                                                  \
    const dim3 grid = row_grid((A.n_active.cap * BS + (kWave / GG) - 1) / (kWave / GG));                             \
    if (closed) { if (det) CFX_BLOCK_V(GG, CAPP, true, 2); else CFX_BLOCK_V(GG, CAPP, false, 2); }                   \
    else if (inl) { if (det) CFX_BLOCK_V(GG, CAPP, true, 1); else CFX_BLOCK_V(GG, CAPP, false, 1); }                 \
    else { if (det) CFX_BLOCK_V(GG, CAPP, true, 0); else CFX_BLOCK_V(GG, CAPP, false, 0); }                          \
  } while (0)
    // degree 2: the dofs whose rows copied their static list, all BS component rows at a time
    if constexpr (DEG == 2)
    {
      cfx_row_plan& plan = row_plan(a);
      const Stencil& stn = a->V->stencil;
      const char* bp = getenv("CFX_BLOCK_PLAIN");
      int slot = -1, n_std = 0;
      for (int s = 0; s < A.n_cell; ++s)
        if (A.cell[s].std_bits) { slot = s; ++n_std; }
      const bool full_ok = P->full_plan == plan.serial && P->n_full_rows > 0 && !bc0 && !bc1 && !det && stn.slotn_ok
                           && stn.max_len <= 72 && n_std == 1 && !(bp && bp[0] == '0');
      if (full_ok && A.cell[slot].std_inline == 3)
      {
        if constexpr (BS == TDIM)
        {
          // closed-form block rows from the cells' vertices: no staged tensors at all for these dofs
          BlockP2Args Q{};
          Q.n = P->n_full_rows; Q.rows = P->full_rows.p; Q.x = A.x; Q.conn = A.conn; Q.d2c_off = A.d2c_off; Q.d2c = A.d2c;
          Q.slotn = stn.slotn.p; Q.indptr = A.indptr; Q.values = A.values; Q.fresh = fresh ? 1 : 0; Q.error = A.error;
          const double E = A.cell[slot].params[0], nu = A.cell[slot].params[1];
          Q.lmbda = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu)); Q.mu = E / (2.0 * (1.0 + nu));
          // lanes per dof >= dofs per cell (one column dof per lane) and >= incident cells (one item per lane and pass):
          // short lists (edge dofs: 4 - 6 cells) 16 lanes and 32 columns, the others (vertex dofs: 24 cells) 32 and 72
          constexpr int GS = TDIM == 3 ? 16 : 8, GL = TDIM == 3 ? 32 : 16;
          const int64_t ns = P->n_full_short, nl = P->n_full_rows - ns;
          if (ns > 0)
          {
            Q.n = ns; Q.rows = P->full_rows.p;
            launch("assemble_rows_block_p2", assemble_rows_block_p2_kernel<TDIM, GS, 32>, row_grid((ns + kWave - 1) / kWave),
                   dim3(kWave), 0, Q);
          }
          if (nl > 0)
          {
            Q.n = nl; Q.rows = P->full_rows.p + ns;
            launch("assemble_rows_block_p2", assemble_rows_block_p2_kernel<TDIM, GL, 72>, row_grid((nl + kWave - 1) / kWave),
                   dim3(kWave), 0, Q);
          }
          A.n_active = P->n_rest_rows; A.active_rows = P->rest_rows.p;
        }
      }
      else if (full_ok && A.cell[slot].std_tensors && !A.cell[slot].std_inline)
      {
        BlockPlainArgs Q{};
        Q.n = P->n_full_rows; Q.rows = P->full_rows.p; Q.d2c_off = A.d2c_off; Q.d2c = A.d2c; Q.slotn = stn.slotn.p;
        Q.std_bits = A.cell[slot].std_bits; Q.std_rank = A.cell[slot].std_rank; Q.std_tensors = A.cell[slot].std_tensors;
        Q.indptr = A.indptr; Q.values = A.values; Q.fresh = fresh ? 1 : 0; Q.error = A.error;
        launch("assemble_rows_block_plain", assemble_rows_block_plain_kernel<TDIM, BS, 32, 72>, row_grid((Q.n + 1) / 2),
               dim3(kWave), 0, Q);
        A.n_active = P->n_rest_rows; A.active_rows = P->rest_rows.p;
      }
    }
    if (A.n_active.cap > 0)
    {
    if (mr <= 32) CFX_BLOCK(8, 32);
    else if (mr <= 128) CFX_BLOCK(16, 128);
    else CFX_BLOCK(32, 256);
    }
#undef CFX_BLOCK_V
#undef CFX_BLOCK
  }
  return err.deferred ? 0 : read_scalar(err.p);
}

// stage 2 of a linear form on the plain rows of a P1 space: the row's entries are contiguous in the
// row-ordered staging (cfx_row_plan::vec_t2off) -- no incidence list, no marks, no local-index search
#ifndef CFX_VEC_PLAIN_G
#define CFX_VEC_PLAIN_G 4
#endif
#ifndef CFX_VEC_BLOCK_G
#define CFX_VEC_BLOCK_G 1 // lanes per row of vec_blocks_rows_kernel (configs[3]: 2.13 ms at 4, 1.51 at 2, 1.35 at 1)
#endif
template <int G>
__global__ void __launch_bounds__(kWave) assemble_vec_plain_kernel(DevN n_plain_d, const int32_t* __restrict__ rows,
                                                                   const int64_t* __restrict__ d2c_off,
                                                                   const int32_t* __restrict__ t2off,
                                                                   const double* __restrict__ t2, double* __restrict__ b)
{
  const int64_t n_plain = dev_n(n_plain_d);
  const int lane = threadIdx.x, gl = lane % G;
  const int64_t i = (int64_t)blockIdx.x * (kWave / G) + lane / G;
  const bool live = i < n_plain;
  const int64_t r = live ? rows[i] : 0;
  const int nc = live ? (int)(d2c_off[r + 1] - d2c_off[r]) : 0;
  const int64_t o = live ? (int64_t)t2off[r] - 1 : -1; // -1: the row has no segment (stored + 1: 0 = none)
  double part = 0.0; // entries gl, gl + G, ... in ascending order, then a fixed tree: bitwise reproducible
  if (o >= 0)
  {
    // the first Q entries of the lane are requested together (a Kuhn-mesh vertex has 24 cells: 6 per lane at G = 4)
    constexpr int Q = 8;
    double v[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      const int k = gl + q * G;
      v[q] = k < nc ? t2[o + k] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) part += v[q];
    for (int k = gl + Q * G; k < nc; k += G) part += t2[o + k];
  }
#pragma unroll
  for (int d = G / 2; d >= 1; d >>= 1) part += __shfl_xor(part, d, G);
  if (live && gl == 0 && o >= 0) b[r] += part;
}

// Which way the uncut cells and rules of a linear form reach the rows.  The series source term on a P1 space keeps the
// row-ordered staging when the plan offers it (512^3: 2.65 + 0.73 ms against 2.8 + 0.55 ms by cell block, and the
// rule items by row 0.6 ms against 0.85 ms by block); every other form goes by cell block, uncut entities and rule
// parents in one pass.  CFX_VEC_BLOCKS=0: never by block; =2: always.
struct BlockChoice { bool use, merged; };
template <int DEG>
BlockChoice vec_block_choice(cfx_form_s* L, cfx_row_plan& plan, int slot, int count)
{
  const char* e = getenv("CFX_VEC_BLOCKS");
  if ((e && e[0] == '0') || count > 1 || L->V->bs != 1 || !plan.usable) return {false, false};
  for (const auto& I : L->integrals) // (integrands compiled at run time are staged per cell, cfx_rtc.hip)
    if (user_integrand_known(I.kernel)) return {false, false};
  const bool series = count == 1 && source_series_ok<DEG>(L->V, L->integrals[plan.cell_slot_integral[slot]]);
  if (series && !(e && e[0] == '2'))
  {
    const char* ro = getenv("CFX_VEC_ROWORDER");
    if (!(ro && ro[0] == '0') && plain_vec_offsets(L, (uint8_t)(1u << slot))) return {false, false};
  }
  return {true, !series};
}

template <int TDIM, int DEG>
void run_vector(cfx_form_s* L, double* b)
{
  Stage1 st;
  cfx_row_plan& plan = row_plan(L);
  // one integral with uncut entities (the volume term)
  int slot = -1, count = 0;
  for (int s = 0; s < plan.n_cell_slots; ++s)
    if (L->integrals[plan.cell_slot_integral[s]].n_entities.cap() > 0) { slot = s; ++count; }
  // ... by cell block: partials per (block, dof) of the uncut cells and the runtime rules, then the rows add the
  // partials of the blocks around them
  const uint8_t std_mark = count == 1 ? (uint8_t)(1u << slot) : (uint8_t)0;
  uint8_t rule_marks = 0;
  for (int s = 0; s < plan.n_cell_slots; ++s)
  {
    const cfx_integral_dev& I = L->integrals[plan.cell_slot_integral[s]];
    if (I.rules && I.rules->nr.cap() > 0) rule_marks |= (uint8_t)(16u << s);
  }
  const BlockChoice bc = vec_block_choice<DEG>(L, plan, slot, count);
  if (bc.use && (std_mark | rule_marks) != 0 && vec_block_plan(L, (uint8_t)(std_mark | rule_marks), bc.merged))
    st.vec_blocks = true;
  if constexpr (DEG == 1)
  {
    // ... or (CFX_VEC_BLOCKS=0) its element vectors go to the plain rows directly
    const char* ro = getenv("CFX_VEC_ROWORDER");
    bool has_user = false;
    for (const auto& I : L->integrals) has_user = has_user || user_integrand_known(I.kernel);
    if (!st.vec_blocks && count == 1 && L->V->bs == 1 && plan.usable && !(ro && ro[0] == '0') && !has_user
        && plain_vec_offsets(L, (uint8_t)(1u << slot)))
    {
      st.t2.alloc(plan.vec_t2_total.cap());
      st.vec_t2 = st.t2.p;
    }
  }
  RowArgs A = prepare<TDIM, DEG>(L, st);
  A.values = b;
  if (st.vec_blocks)
  {
    st.part.alloc(plan.vb_total);
    vec_block_partials<TDIM, DEG>(L, count == 1 ? &L->integrals[plan.cell_slot_integral[slot]] : nullptr, std_mark, A, st.part.p);
    if (plan.n_active_rows.cap() > 0 && plan.vb_total > 0)
    {
      const VecBlocks& S = space_vec_blocks(L->V);
      constexpr int G = CFX_VEC_BLOCK_G;
      launch("vec_blocks_rows", vec_blocks_rows_kernel<G>, row_grid((plan.n_active_rows.cap() + (kWave / G) - 1) / (kWave / G)), dim3(kWave),
             0, plan.n_active_rows, plan.active_rows.p, S.p_off.p, S.p_pos.p, plan.vb_base.p, st.part.p, b);
    }
    return;
  }
  if constexpr (DEG == 1)
  {
    const Stencil& stn = space_stencil(L->V);
    if (stn.usable) { A.slot4 = stn.slot4.p; A.diagpos = stn.diagpos.p; }
  }
  else
  {
    const Stencil& stn = space_stencil_slotn(L->V); // local index of the row's dof in every incident cell
    if (stn.slotn_ok && L->V->bs == 1) A.slotn = stn.slotn.p;
  }
  if (st.vec_t2)
  {
    constexpr int G = CFX_VEC_PLAIN_G;
    launch("assemble_vec_plain", assemble_vec_plain_kernel<G>,
           dim3((unsigned)((plan.n_plain_rows.cap() + (kWave / G) - 1) / (kWave / G))), dim3(kWave), 0, plan.n_plain_rows,
           plan.plain_rows.p, A.d2c_off, plan.vec_t2off.p, st.vec_t2, b);
    // everything else gathers the per-cell records: the rows next to the interface ...
    A.n_active = plan.n_special_rows; A.active_rows = plan.special_rows.p;
    if (plan.n_vec_odd_rows.cap() > 0)
    {
      // ... and the plain rows without a segment (their cells do not all carry the mark: the edge of a restricted
      // entity list): a pass over the plain rows that skips the others
      RowArgs O = A;
      O.n_active = plan.n_plain_rows; O.active_rows = plan.plain_rows.p;
      O.vec_skip = plan.vec_t2off.p; O.vec_n_odd = plan.n_vec_odd_rows;
      launch("assemble_vec_rows", assemble_vec_rows_kernel<TDIM, DEG, CFX_VEC_G>,
             row_grid((O.n_active.cap + (kWave / CFX_VEC_G) - 1) / (kWave / CFX_VEC_G)), dim3(kWave), 0, O);
    }
  }
  if (A.n_active.cap > 0)
    launch("assemble_vec_rows", assemble_vec_rows_kernel<TDIM, DEG, CFX_VEC_G>,
           row_grid((A.n_active.cap + (kWave / CFX_VEC_G) - 1) / (kWave / CFX_VEC_G)), dim3(kWave), 0,
           A);
}

} // namespace

namespace cfx
{

// ---------------------------------------------------------------------------
// Rectangular blocks (test space != trial space, cfx_form_create2) by row gather: the work units of the entity-parallel
// kernel -- (cell, test row) -> one row of the [(nd0 bs0) x (nd1 bs1)] element tensor -- ordered by destination row.
// Sixteen lanes per matrix row; the row's columns sit in LDS, a lane takes the incident cells t = gl, gl + 16, ... of
// the row's test dof, forms the tensor row of every integral the cell's mark names (standard rule for an uncut
// entity, the cell's runtime rules through the plan's hash map), finds the trial dofs' columns by binary search in LDS
// and adds there; the row is written once.  No global atomics; bitwise reproducible with CFX_DETERMINISTIC=1.  The
// plan is the test space's (marks, rule maps and active rows do not depend on the trial space).
// ---------------------------------------------------------------------------
struct Rect2Slot
{
  int kernel, qdegree;
  double scale;
  uint8_t std_bit, rule_bit;
  DevN nr;
  const int32_t* offsets;
  const int32_t* parent_map;
  const double* points;
  const double* weights;
  const int32_t* rule_keys;
  const int32_t* rule_first;
  unsigned rule_mask;
};
struct Rect2Args
{
  DevN n_active;
  const int32_t* active_rows; // test dofs touched by the form
  const double* x;
  const int32_t* conn;
  const int32_t* dofmap0;
  const int64_t* d2c_off;
  const int32_t* d2c;
  const uint8_t* cellmark;
  const int8_t* bc0;
  const int8_t* bc1;
  const int64_t* indptr;
  const int32_t* indices;
  double* values;
  int* error;
  RectArgs R;
  int n_slots;
  Rect2Slot slot[4];
};

template <int TDIM, bool ORDERED>
__global__ void __launch_bounds__(kWave) assemble_rows2_kernel(Rect2Args A)
{
  constexpr int G = 16, RPW = kWave / G, CAP = 256;
  constexpr int MAXND = RectRow<TDIM>::MAXND, MAXBS = RectRow<TDIM>::MAXBS;
  __shared__ int32_t s_col[RPW][CAP];
  __shared__ double s_val[RPW][CAP];
  const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
  const RectArgs& R = A.R;
  const int64_t w = CFX_ROW_BLOCK * RPW + grp; // scalar rows of the active test dofs, bs0 per dof
  const int64_t ri = w / R.bs0;
  const int ik = (int)(w - ri * R.bs0);
  const bool live = ri < dev_n(A.n_active);
  const int64_t dof = live ? A.active_rows[ri] : 0;
  const int64_t row = R.bs0 * dof + ik;
  const int64_t rb = live ? A.indptr[row] : 0;
  int len = live ? (int)(A.indptr[row + 1] - rb) : 0;
  if (len > CAP) { *A.error = 2; len = 0; }
  for (int k = gl; k < len; k += G) { s_col[grp][k] = A.indices[rb + k]; s_val[grp][k] = 0.0; }
  __syncthreads();
  const bool row_bc = live && A.bc0 && A.bc0[row];
  const int64_t cb = live ? A.d2c_off[dof] : 0;
  const int nc = (live && len > 0) ? (int)(A.d2c_off[dof + 1] - cb) : 0;
  int npass = (nc + G - 1) / G;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) npass = max(npass, __shfl_xor(npass, o, 64)); // (the groups' barriers match)
  for (int p = 0; p < npass; ++p)
  {
    const int t = p * G + gl;
    const bool has = t < nc;
    const int64_t c = has ? (int64_t)A.d2c[cb + t] : 0;
    const uint8_t mark = has ? A.cellmark[c] : (uint8_t)0;
    double acc[MAXND][MAXBS];
#pragma unroll
    for (int j = 0; j < MAXND; ++j)
#pragma unroll
      for (int b = 0; b < MAXBS; ++b) acc[j][b] = 0.0;
    if (mark)
    {
      int ia = 0;
      for (int j = 0; j < R.nd0; ++j) ia = (A.dofmap0[c * R.nd0 + j] == (int32_t)dof) ? j : ia;
      Geo<TDIM> g;
      load_cell<TDIM>(A.x, A.conn, c, g);
      jacobian<TDIM>(g);
      for (int s = 0; s < A.n_slots; ++s)
      {
        const Rect2Slot& S = A.slot[s];
        if (mark & S.std_bit)
        {
          int npts;
          const double* wts;
          const double* pts = ref_rule(TDIM, S.qdegree, npts, wts);
          RectRow<TDIM>::accumulate(R, S.kernel, S.scale, g, ia, ik, npts, pts, wts, fabs(g.detJ), acc);
        }
        if (mark & S.rule_bit)
        {
          for (int64_t e = first_rule(S.rule_keys, S.rule_first, S.rule_mask, (int32_t)c); e < dev_len(S.nr) && S.parent_map[e] == c; ++e)
          {
            const int32_t q0 = S.offsets[e], q1 = S.offsets[e + 1];
            RectRow<TDIM>::accumulate(R, S.kernel, S.scale, g, ia, ik, q1 - q0, S.points + (int64_t)q0 * TDIM, S.weights + q0, 1.0, acc);
          }
        }
      }
    }
    // columns of the trial dofs of the cell: bs1 consecutive entries from the position of the first
    int pos[MAXND];
#pragma unroll
    for (int j = 0; j < MAXND; ++j)
    {
      pos[j] = -1;
      if (mark && j < R.nd1)
      {
        const int32_t col0 = R.bs1 * R.dofmap1[c * R.nd1 + j];
        int lo = 0, hi = len;
        while (lo < hi)
        {
          const int mid = (lo + hi) >> 1;
          if (s_col[grp][mid] < col0) lo = mid + 1; else hi = mid;
        }
        if (lo < len && s_col[grp][lo] == col0) pos[j] = lo; else *A.error = 1;
        if (pos[j] >= 0)
        {
#pragma unroll
          for (int b = 0; b < MAXBS; ++b)
            if (b < R.bs1 && (row_bc || (A.bc1 && A.bc1[col0 + b]))) acc[j][b] = 0.0;
        }
      }
    }
    if constexpr (ORDERED)
    {
      for (int turn = 0; turn < G; ++turn)
      {
        if (gl == turn)
        {
#pragma unroll
          for (int j = 0; j < MAXND; ++j)
#pragma unroll
            for (int b = 0; b < MAXBS; ++b)
              if (pos[j] >= 0 && b < R.bs1) s_val[grp][pos[j] + b] += acc[j][b];
        }
        __syncthreads();
      }
    }
    else
    {
#pragma unroll
      for (int j = 0; j < MAXND; ++j)
#pragma unroll
        for (int b = 0; b < MAXBS; ++b)
          if (pos[j] >= 0 && b < R.bs1) atomicAdd(&s_val[grp][pos[j] + b], acc[j][b]);
    }
  }
  __syncthreads();
  for (int k = gl; k < len; k += G) A.values[rb + k] += s_val[grp][k]; // (one writer per row)
}

bool assemble_rect_rows(cfx_form_s* a, cfx_pattern_s* P, const int8_t* bc0, const int8_t* bc1, double* values, int* error)
{
  const char* rg = getenv("CFX_RECT_GATHER");
  if (rg && rg[0] == '0') return false;
  cfx_space_s* V0 = a->V;
  cfx_space_s* V1 = a->V1;
  for (const auto& I : a->integrals)
    if (I.type != CFX_CELL) return false;
  if ((int64_t)P->max_row_len * V1->bs > 256) return false;
  cfx_row_plan& plan = row_plan(a);
  if (!plan.usable || plan.n_cell_slots > 4) return false;
  if (plan.n_active_rows.cap() == 0) return true;
  const Adjacency& adj = V0->dof_cells();
  Rect2Args A{};
  A.n_active = plan.n_active_rows; A.active_rows = plan.active_rows.p;
  A.x = V0->mesh->x.p; A.conn = V0->mesh->conn.p; A.dofmap0 = V0->dofmap.p;
  A.d2c_off = adj.offsets.p; A.d2c = adj.cells.p; A.cellmark = plan.cellmark.p;
  A.bc0 = bc0; A.bc1 = bc1; A.indptr = P->indptr.p; A.indices = P->indices.p; A.values = values; A.error = error;
  A.R = RectArgs{V1->dofmap.p, V0->degree, V0->bs, V0->ndofs_cell, V1->degree, V1->bs, V1->ndofs_cell};
  A.n_slots = plan.n_cell_slots;
  for (int s = 0; s < plan.n_cell_slots; ++s)
  {
    const cfx_integral_dev& I = a->integrals[plan.cell_slot_integral[s]];
    Rect2Slot& S = A.slot[s];
    S.kernel = I.kernel; S.qdegree = I.qdegree; S.scale = I.params[0];
    S.std_bit = I.n_entities.cap() > 0 ? (uint8_t)(1u << s) : (uint8_t)0;
    S.rule_bit = 0;
    if (I.rules && I.rules->nr.cap() > 0)
    {
      S.rule_bit = (uint8_t)(16u << s);
      S.nr = rule_bound(I.rules);
      S.offsets = I.rules->offsets.p; S.parent_map = I.rules->parent_map.p;
      S.points = I.rules->points.p; S.weights = I.rules->weights.p;
      S.rule_keys = plan.rule_keys[s].p; S.rule_first = plan.rule_first[s].p; S.rule_mask = plan.rule_mask[s];
    }
  }
  const int64_t nrows_cap = A.n_active.cap * V0->bs;
  const dim3 grid = row_grid((nrows_cap + 3) / 4);
  const bool det = deterministic();
  if (V0->mesh->tdim == 2)
  {
    if (det) launch("assemble_rows2", assemble_rows2_kernel<2, true>, grid, dim3(kWave), 0, A);
    else launch("assemble_rows2", assemble_rows2_kernel<2, false>, grid, dim3(kWave), 0, A);
  }
  else
  {
    if (det) launch("assemble_rows2", assemble_rows2_kernel<3, true>, grid, dim3(kWave), 0, A);
    else launch("assemble_rows2", assemble_rows2_kernel<3, false>, grid, dim3(kWave), 0, A);
  }
  return true;
}

bool assemble_matrix_rows(cfx_form_s* a, cfx_pattern_s* P, const int8_t* bc0, const int8_t* bc1, double* values, bool fresh)
{
  cfx_row_plan& plan = row_plan(a);
  cfx_space_s* V = a->V;
  if (!plan.usable || V->degree > 2 || P->max_row_len > 512) return false;
  int err = 0;
  if (V->bs != 1)
  {
    // block spaces: bs = gdim, cell integrals only.  The local tensors of the uncut cells are
    // staged (30 x 30 doubles per P2 elasticity cell): recomputing a row per (cell, row) item costs
    // 30x the arithmetic (measured on config 5's rank share, 3.7 M cells: 181 ms inline, 149 ms
    // entity-parallel atomics, 66 ms staged + gathered).  Needs the staging to fit in HBM.
    {
      const int nloc = V->ndofs_cell * V->bs;
      size_t need = 0, free_b = 0, total_b = 0;
      for (const auto& I : a->integrals) // (closed-form uncut cells stage nothing)
        need += (size_t)((p2_elasticity_closed(a, I) ? 0 : I.n_entities.cap()) + (I.rules ? I.rules->nr.cap() : 0)) * nloc * nloc * sizeof(double);
      CFX_HIP(hipMemGetInfo(&free_b, &total_b));
      const char* bg = getenv("CFX_BLOCK_GATHER");
      if ((bg && bg[0] == '0') || need > free_b / 2) return false;
    }
    if (V->bs != V->mesh->tdim || P->max_row_len > 256) return false;
    for (const auto& I : a->integrals) // facet items: the ghost-penalty gradient jump (extension pairs keep the entity path)
      if (I.type == CFX_INTERIOR_FACET && I.kernel != CFX_K_GHOST_GRADJUMP) return false;
    if (V->mesh->tdim == 2)
      err = V->degree == 1 ? run_matrix_block<2, 1, 2>(a, P, bc0, bc1, values, fresh) : run_matrix_block<2, 2, 2>(a, P, bc0, bc1, values, fresh);
    else
      err = V->degree == 1 ? run_matrix_block<3, 1, 3>(a, P, bc0, bc1, values, fresh) : run_matrix_block<3, 2, 3>(a, P, bc0, bc1, values, fresh);
  }
  else if (V->degree == 1)
    err = V->mesh->tdim == 2 ? run_matrix<2, 1>(a, P, bc0, bc1, values, fresh) : run_matrix<3, 1>(a, P, bc0, bc1, values, fresh);
  else if (!kRowsInline<2> && [&]() {
             // degree 2 stages 100 doubles per uncut cell (38 GB at config 4): must fit next to the matrix
             size_t need = 0, free_b = 0, total_b = 0;
             const int nloc = V->ndofs_cell;
             for (const auto& I : a->integrals)
               if (I.type == CFX_CELL) need += (size_t)(I.n_entities.cap() + (I.rules ? I.rules->nr.cap() : 0)) * nloc * nloc * sizeof(double);
             CFX_HIP(hipMemGetInfo(&free_b, &total_b));
             return need > free_b / 2;
           }())
    return false;
  else
    err = V->mesh->tdim == 2 ? run_matrix<2, 2>(a, P, bc0, bc1, values, fresh) : run_matrix<3, 2>(a, P, bc0, bc1, values, fresh);
  raise_gather_error(err);
  return true;
}

// cfx_form_prepare: everything the gather assembly derives from the form's entity lists, built now
void prepare_form_tables(cfx_form_s* a)
{
  cfx_row_plan& plan = row_plan(a);
  cfx_space_s* V = a->V;
  // space-level tables of every degree and block size: the incidence lists, the neighbour lists and (degree 2) the
  // slot records -- built here so that an overlap section finds them in place (a table that is still built lazily
  // inside a section is published to the other lane, cfx::publish_across_lanes)
  (void)V->dof_cells();
  const Stencil& stn = space_stencil(V);
  if (V->degree == 2) (void)space_stencil_slotn(V);
  if (plan.nfacets.cap() > 0) (void)V->mesh->cell_neighbours();
  if (a->rank == 1 && plan.usable && V->bs == 1)
  {
    int slot = -1, count = 0;
    for (int s = 0; s < plan.n_cell_slots; ++s)
      if (a->integrals[plan.cell_slot_integral[s]].n_entities.cap() > 0) { slot = s; ++count; }
    uint8_t marks = count == 1 ? (uint8_t)(1u << slot) : (uint8_t)0;
    for (int s = 0; s < plan.n_cell_slots; ++s)
    {
      const cfx_integral_dev& I = a->integrals[plan.cell_slot_integral[s]];
      if (I.rules && I.rules->nr.cap() > 0) marks |= (uint8_t)(16u << s);
    }
    if (count <= 1 && marks != 0)
    {
      const BlockChoice bc = V->degree == 1 ? vec_block_choice<1>(a, plan, slot, count) : vec_block_choice<2>(a, plan, slot, count);
      if (bc.use && vec_block_plan(a, marks, bc.merged)) return;
    }
  }
  if (!plan.usable || V->degree != 1 || V->bs != 1) return;
  if (!stn.usable) return;
  (void)space_stencil_tiles(V);
  plain_row_masks(a);
  if (a->rank == 1)
  {
    int slot = -1, count = 0;
    for (int s = 0; s < plan.n_cell_slots; ++s)
      if (a->integrals[plan.cell_slot_integral[s]].n_entities.cap() > 0) { slot = s; ++count; }
    if (count == 1) (void)plain_vec_offsets(a, (uint8_t)(1u << slot));
  }
}

bool assemble_vector_rows(cfx_form_s* L, double* b)
{
  cfx_row_plan& plan = row_plan(L);
  cfx_space_s* V = L->V;
  for (const auto& I : L->integrals)
    if (I.type == CFX_INTERIOR_FACET) return false;
  if (!plan.usable || V->degree > 2 || V->bs != 1) return false;
  if (V->degree == 1) { if (V->mesh->tdim == 2) run_vector<2, 1>(L, b); else run_vector<3, 1>(L, b); }
  else { if (V->mesh->tdim == 2) run_vector<2, 2>(L, b); else run_vector<3, 2>(L, b); }
  return true;
}

} // namespace cfx
