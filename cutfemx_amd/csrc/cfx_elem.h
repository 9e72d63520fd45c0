// cutfemx_amd: element mathematics shared by the assembly kernels -- Lagrange
// tabulation, reference rules, analytic fields and the per-entity "local row"
// of every supported integrand.  These device functions stand in for the
// runintgen/FFCx generated tabulate_tensor kernels (third party):
//   void k(T* A, const T* w, const T* c, const U* coordinate_dofs,
//          const int* entity_local_index, const uint8_t* perm, void* custom_data)
// (cpp/dolfinx_custom_data/fem/Form.h:59-75).  A cut entity integrates over its
// runtime rule slice (points = parent reference coords, weights = physical
// measure, no detJ factor); an uncut entity uses the reference rule of degree
// qdegree times |detJ|.
#pragma once

#include "cfx_device.h"

#define CFX_QUAD_TABLE_QUALIFIER static __device__ const
#include "cfx_quadrature_tables.h"
#undef CFX_QUAD_TABLE_QUALIFIER

namespace cfx
{

constexpr double kPi = 3.14159265358979323846;

template <int TDIM, int DEG>
struct Elem
{
  static constexpr int ND = DEG == 1 ? TDIM + 1 : (TDIM == 2 ? 6 : 10);
  static constexpr int NS = DEG == 1 ? TDIM : (TDIM == 2 ? 3 : 6); // dofs on a facet: shared by its two cells (continuous space)
  static constexpr int WF = 2 * ND - NS;                           // macro dofs of a facet: cell 0's, then cell 1's others
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

// The values alone, and the values with the derivative along a reference-space direction k (dk_i = sum_t dN_i/dX_t
// k_t): what a linear form needs (N for f v, N and n . grad N = (K n) . dN/dX for the Nitsche datum) -- 2 ND numbers
// instead of the ND (1 + TDIM) of tabulate(): the degree-2 rule kernel drops from 180 registers to under 128
template <int TDIM, int DEG>
__device__ __forceinline__ void tabulate_values(const double* X, double* N)
{
  double lam[TDIM + 1];
  lam[0] = 1.0;
#pragma unroll
  for (int t = 0; t < TDIM; ++t) { lam[0] -= X[t]; lam[t + 1] = X[t]; }
  if constexpr (DEG == 1)
  {
#pragma unroll
    for (int i = 0; i <= TDIM; ++i) N[i] = lam[i];
  }
  else
  {
#pragma unroll
    for (int i = 0; i <= TDIM; ++i) N[i] = lam[i] * (2.0 * lam[i] - 1.0);
    constexpr int NE = TDIM == 2 ? 3 : 6;
    constexpr int ea2[3] = {1, 0, 0}, eb2[3] = {2, 2, 1};
    constexpr int ea3[6] = {2, 1, 1, 0, 0, 0}, eb3[6] = {3, 3, 2, 3, 2, 1};
#pragma unroll
    for (int e = 0; e < NE; ++e)
    {
      const int a = TDIM == 2 ? ea2[e % 3] : ea3[e], b = TDIM == 2 ? eb2[e % 3] : eb3[e];
      N[TDIM + 1 + e] = 4.0 * lam[a] * lam[b];
    }
  }
}

template <int TDIM, int DEG>
__device__ __forceinline__ void tabulate_dot(const double* X, const double* k, double* N, double* dk)
{
  double lam[TDIM + 1], gk[TDIM + 1]; // gk_i = d lam_i / dX . k
  lam[0] = 1.0;
  gk[0] = 0.0;
#pragma unroll
  for (int t = 0; t < TDIM; ++t) { lam[0] -= X[t]; lam[t + 1] = X[t]; gk[0] -= k[t]; gk[t + 1] = k[t]; }
  if constexpr (DEG == 1)
  {
#pragma unroll
    for (int i = 0; i <= TDIM; ++i) { N[i] = lam[i]; dk[i] = gk[i]; }
  }
  else
  {
#pragma unroll
    for (int i = 0; i <= TDIM; ++i)
    {
      N[i] = lam[i] * (2.0 * lam[i] - 1.0);
      dk[i] = (4.0 * lam[i] - 1.0) * gk[i];
    }
    constexpr int NE = TDIM == 2 ? 3 : 6;
    constexpr int ea2[3] = {1, 0, 0}, eb2[3] = {2, 2, 1};
    constexpr int ea3[6] = {2, 1, 1, 0, 0, 0}, eb3[6] = {3, 3, 2, 3, 2, 1};
#pragma unroll
    for (int e = 0; e < NE; ++e)
    {
      const int a = TDIM == 2 ? ea2[e % 3] : ea3[e], b = TDIM == 2 ? eb2[e % 3] : eb3[e];
      N[TDIM + 1 + e] = 4.0 * lam[a] * lam[b];
      dk[TDIM + 1 + e] = 4.0 * (lam[a] * gk[b] + gk[a] * lam[b]);
    }
  }
}

// Row `lr` of the degree-2 Lagrange stiffness tensor of an affine simplex in closed form.  grad(phi_p) is linear
// in the barycentric coordinates -- vertex i: (4 lam_i - 1) grad(lam_i); edge (a, b): 4 (lam_b grad(lam_a) +
// lam_a grad(lam_b)) -- so every entry is a combination of the P1 stiffness entries S_kl = |K| grad(lam_k) .
// grad(lam_l) with the moments m_xy = int lam_x lam_y / |K| = (1 + [x = y]) n! / (n + 2)!:
//   vertex i, vertex j        S_ij (16 m_ij - 8 / (n + 1) + 1)
//   vertex i, edge (c, d)     4 (S_ic s_id + S_id s_ic),  s_xy = 4 m_xy - 1 / (n + 1)
//   edge (a, b), edge (c, d)  16 (S_ac m_bd + S_ad m_bc + S_bc m_ad + S_bd m_ac)
// Same numbers as the degree-2 quadrature (exact for this integrand) at ~1/6 of the instructions and a third of
// the registers: the gather kernels can form the row per (row, cell) item instead of staging 100 doubles per cell.
// Dof order as tabulate(): vertices, then edges tri (1,2),(0,2),(0,1); tet (2,3),(1,3),(1,2),(0,3),(0,2),(0,1).
template <int TDIM>
__device__ __forceinline__ void p2_stiffness_row(const Geo<TDIM>& g, int lr, double scale, double* acc)
{
  constexpr int NV = TDIM + 1, NE = TDIM == 2 ? 3 : 6;
  constexpr int ea2[3] = {1, 0, 0}, eb2[3] = {2, 2, 1};
  constexpr int ea3[6] = {2, 1, 1, 0, 0, 0}, eb3[6] = {3, 3, 2, 3, 2, 1};
  constexpr double m_d = TDIM == 2 ? 1.0 / 12.0 : 1.0 / 20.0, m_s = 2.0 * m_d; // m_xy off / on the diagonal
  constexpr double inv = 1.0 / (TDIM + 1);
  constexpr double vv_d = 16.0 * m_d - 8.0 * inv + 1.0, vv_s = 16.0 * m_s - 8.0 * inv + 1.0;
  constexpr double s_d = 4.0 * m_d - inv, s_s = 4.0 * m_s - inv;
  // rows a and b of S for the dof: a vertex uses row a only (b = a)
  int a = lr, b = lr;
#pragma unroll
  for (int e = 0; e < NE; ++e)
  {
    const int ea = TDIM == 2 ? ea2[e % 3] : ea3[e], eb = TDIM == 2 ? eb2[e % 3] : eb3[e];
    a = (lr == NV + e) ? ea : a;
    b = (lr == NV + e) ? eb : b;
  }
  // gradients of the barycentric coordinates, then the two rows of S
  double G[NV][TDIM];
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    double s0 = 0.0;
#pragma unroll
    for (int t = 0; t < TDIM; ++t) { G[t + 1][d] = g.K[t][d]; s0 -= g.K[t][d]; }
    G[0][d] = s0;
  }
  const double vol = scale * fabs(g.detJ) * (TDIM == 2 ? 0.5 : 1.0 / 6.0);
  double Ga[TDIM], Gb[TDIM];
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    double va = 0.0, vb = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) { va = (a == k) ? G[k][d] : va; vb = (b == k) ? G[k][d] : vb; }
    Ga[d] = va * vol; Gb[d] = vb * vol;
  }
  double Sa[NV], Sb[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k)
  {
    double va = 0.0, vb = 0.0;
#pragma unroll
    for (int d = 0; d < TDIM; ++d) { va += Ga[d] * G[k][d]; vb += Gb[d] * G[k][d]; }
    Sa[k] = va; Sb[k] = vb;
  }
  if (lr < NV)
  {
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] += Sa[j] * ((j == a) ? vv_s : vv_d);
#pragma unroll
    for (int e = 0; e < NE; ++e)
    {
      const int c = TDIM == 2 ? ea2[e % 3] : ea3[e], d = TDIM == 2 ? eb2[e % 3] : eb3[e];
      acc[NV + e] += 4.0 * (Sa[c] * ((a == d) ? s_s : s_d) + Sa[d] * ((a == c) ? s_s : s_d));
    }
  }
  else
  {
    // edge (a, b) against vertex j: 4 (S_ja s_jb + S_jb s_ja)
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] += 4.0 * (Sa[j] * ((j == b) ? s_s : s_d) + Sb[j] * ((j == a) ? s_s : s_d));
#pragma unroll
    for (int e = 0; e < NE; ++e)
    {
      const int c = TDIM == 2 ? ea2[e % 3] : ea3[e], d = TDIM == 2 ? eb2[e % 3] : eb3[e];
      acc[NV + e] += 16.0 * (Sa[c] * ((b == d) ? m_s : m_d) + Sa[d] * ((b == c) ? m_s : m_d)
                             + Sb[c] * ((a == d) ? m_s : m_d) + Sb[d] * ((a == c) ? m_s : m_d));
    }
  }
}

// The same row over a CUT part of the cell: the integrals of the barycentric monomials over the cut part (a runtime
// rule) take the place of the full-cell moments -- mom = (m0; m1_x, x < NV; m2_xy, x <= y row by row), 15 numbers in
// 3-D, 10 in 2-D, with m0 = sum_q w_q, m1_x = sum_q w_q lam_x(q), m2_xy = sum_q w_q lam_x(q) lam_y(q) (the weights are
// physical measures).  S_kl = grad(lam_k) . grad(lam_l) carries no volume factor here.
//   vertex i, vertex j        S_ij (16 m2_ij - 4 m1_i - 4 m1_j + m0)
//   vertex i, edge (c, d)     4 (S_ic (4 m2_id - m1_d) + S_id (4 m2_ic - m1_c))
//   edge (a, b), edge (c, d)  16 (S_ac m2_bd + S_ad m2_bc + S_bc m2_ad + S_bd m2_ac)
template <int TDIM>
__device__ __forceinline__ void p2_stiffness_row_moments(const Geo<TDIM>& g, int lr, const double* mom, double* acc)
{
  constexpr int NV = TDIM + 1, NE = TDIM == 2 ? 3 : 6;
  constexpr int ea2[3] = {1, 0, 0}, eb2[3] = {2, 2, 1};
  constexpr int ea3[6] = {2, 1, 1, 0, 0, 0}, eb3[6] = {3, 3, 2, 3, 2, 1};
  int a = lr, b = lr;
#pragma unroll
  for (int e = 0; e < NE; ++e)
  {
    const int ea = TDIM == 2 ? ea2[e % 3] : ea3[e], eb = TDIM == 2 ? eb2[e % 3] : eb3[e];
    a = (lr == NV + e) ? ea : a;
    b = (lr == NV + e) ? eb : b;
  }
  double G[NV][TDIM];
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    double s0 = 0.0;
#pragma unroll
    for (int t = 0; t < TDIM; ++t) { G[t + 1][d] = g.K[t][d]; s0 -= g.K[t][d]; }
    G[0][d] = s0;
  }
  double Ga[TDIM], Gb[TDIM];
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    double va = 0.0, vb = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) { va = (a == k) ? G[k][d] : va; vb = (b == k) ? G[k][d] : vb; }
    Ga[d] = va; Gb[d] = vb;
  }
  double Sa[NV], Sb[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k)
  {
    double va = 0.0, vb = 0.0;
#pragma unroll
    for (int d = 0; d < TDIM; ++d) { va += Ga[d] * G[k][d]; vb += Gb[d] * G[k][d]; }
    Sa[k] = va; Sb[k] = vb;
  }
  // rows a and b of the (symmetric) second-moment matrix, and the first moments of a and b
  const double m0 = mom[0];
  double Ma[NV], Mb[NV], m1a = 0.0, m1b = 0.0;
#pragma unroll
  for (int y = 0; y < NV; ++y) { Ma[y] = 0.0; Mb[y] = 0.0; }
#pragma unroll
  for (int x = 0; x < NV; ++x)
  {
    m1a = (a == x) ? mom[1 + x] : m1a;
    m1b = (b == x) ? mom[1 + x] : m1b;
#pragma unroll
    for (int y = 0; y < NV; ++y)
    {
      // packed index of (min, max): rows of the upper triangle
      const int lo = x < y ? x : y, hi = x < y ? y : x;
      const int idx = 1 + NV + lo * NV - lo * (lo - 1) / 2 + (hi - lo);
      Ma[y] = (a == x) ? mom[idx] : Ma[y];
      Mb[y] = (b == x) ? mom[idx] : Mb[y];
    }
  }
  if (lr < NV)
  {
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] += Sa[j] * (16.0 * Ma[j] - 4.0 * m1a - 4.0 * mom[1 + j] + m0);
#pragma unroll
    for (int e = 0; e < NE; ++e)
    {
      const int c = TDIM == 2 ? ea2[e % 3] : ea3[e], d = TDIM == 2 ? eb2[e % 3] : eb3[e];
      acc[NV + e] += 4.0 * (Sa[c] * (4.0 * Ma[d] - mom[1 + d]) + Sa[d] * (4.0 * Ma[c] - mom[1 + c]));
    }
  }
  else
  {
    // edge (a, b) against vertex j: 4 (S_ja (4 m2_jb - m1_b) + S_jb (4 m2_ja - m1_a))
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] += 4.0 * (Sa[j] * (4.0 * Mb[j] - m1b) + Sb[j] * (4.0 * Ma[j] - m1a));
#pragma unroll
    for (int e = 0; e < NE; ++e)
    {
      const int c = TDIM == 2 ? ea2[e % 3] : ea3[e], d = TDIM == 2 ? eb2[e % 3] : eb3[e];
      acc[NV + e] += 16.0 * (Sa[c] * Mb[d] + Sa[d] * Mb[c] + Sb[c] * Ma[d] + Sb[d] * Ma[c]);
    }
  }
}

// Vector-valued degree 2 on an affine simplex: the gradient Gram blocks H_ij[a][b] = int d_a N_i d_b N_j of row dof
// `lr` against every column dof j, in closed form.  With g_k = grad(lam_k) (constant on the cell),
//   d N_i = sum_m (dN_i / d lam_m) g_m,  vertex i: (4 lam_i - 1) g_i;  edge (a, b): 4 (lam_b g_a + lam_a g_b),
// every block is |K| (g_a (x) ua_j + g_b (x) ub_j) with ua_j, ub_j fixed combinations of the g_k whose coefficients are
// the same barycentric moments as in p2_stiffness_row() -- that function is the trace of this one:
//   row vertex a:        col vertex j: ua = vv_aj g_j;                 col edge (c, d): ua = 4 (s_ad g_c + s_ac g_d)
//   row edge (a, b):     col vertex j: ua = 4 s_jb g_j, ub = 4 s_ja g_j;
//                        col edge (c, d): ua = 16 (m_bd g_c + m_bc g_d), ub = 16 (m_ad g_c + m_ac g_d)
// (m_xy = int lam_x lam_y / |K|, s_xy = 4 m_xy - 1 / (n + 1), vv_xy = 16 m_xy - 8 / (n + 1) + 1).  Exact for the affine
// cell: the same numbers as a quadrature rule of degree >= 2.  sink(j, H) is called for j = 0 .. ND - 1 with j a
// compile-time constant in the unrolled loop.  The elasticity block (python/demo/demo_elasticity.py:167-238) is
// Ae[(i, a), (j, b)] = lambda H[a][b] + mu H[b][a] + mu delta_ab tr(H); 30 x 30 staged doubles per uncut cell become
// ~50 flops per block.
template <int TDIM, typename Sink>
__device__ __forceinline__ void p2_gradient_gram_row(const Geo<TDIM>& g, int lr, Sink&& sink)
{
  constexpr int NV = TDIM + 1, NE = TDIM == 2 ? 3 : 6;
  constexpr int ea2[3] = {1, 0, 0}, eb2[3] = {2, 2, 1};
  constexpr int ea3[6] = {2, 1, 1, 0, 0, 0}, eb3[6] = {3, 3, 2, 3, 2, 1};
  constexpr double m_d = TDIM == 2 ? 1.0 / 12.0 : 1.0 / 20.0, m_s = 2.0 * m_d;
  constexpr double inv = 1.0 / (TDIM + 1);
  constexpr double vv_d = 16.0 * m_d - 8.0 * inv + 1.0, vv_s = 16.0 * m_s - 8.0 * inv + 1.0;
  constexpr double s_d = 4.0 * m_d - inv, s_s = 4.0 * m_s - inv;
  const bool vertex = lr < NV;
  int a = lr, b = lr;
#pragma unroll
  for (int e = 0; e < NE; ++e)
  {
    const int ea = TDIM == 2 ? ea2[e % 3] : ea3[e], eb = TDIM == 2 ? eb2[e % 3] : eb3[e];
    a = (lr == NV + e) ? ea : a;
    b = (lr == NV + e) ? eb : b;
  }
  double G[NV][TDIM];
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    double s0 = 0.0;
#pragma unroll
    for (int t = 0; t < TDIM; ++t) { G[t + 1][d] = g.K[t][d]; s0 -= g.K[t][d]; }
    G[0][d] = s0;
  }
  const double vol = fabs(g.detJ) * (TDIM == 2 ? 0.5 : 1.0 / 6.0);
  double Ga[TDIM], Gb[TDIM];
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    double va = 0.0, vb = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) { va = (a == k) ? G[k][d] : va; vb = (b == k) ? G[k][d] : vb; }
    Ga[d] = va * vol;
    Gb[d] = vertex ? 0.0 : vb * vol; // a vertex row has one gradient: its ub terms vanish
  }
#pragma unroll
  for (int j = 0; j < NV + NE; ++j)
  {
    double ua[TDIM], ub[TDIM];
    if (j < NV)
    {
      const double ka = vertex ? ((j == a) ? vv_s : vv_d) : 4.0 * ((j == b) ? s_s : s_d);
      const double kb = 4.0 * ((j == a) ? s_s : s_d);
#pragma unroll
      for (int d = 0; d < TDIM; ++d) { ua[d] = ka * G[j < NV ? j : 0][d]; ub[d] = kb * G[j < NV ? j : 0][d]; }
    }
    else
    {
      const int e = j - NV;
      const int c = TDIM == 2 ? ea2[(e >= 0 ? e : 0) % 3] : ea3[(e >= 0 && e < 6) ? e : 0];
      const int dd = TDIM == 2 ? eb2[(e >= 0 ? e : 0) % 3] : eb3[(e >= 0 && e < 6) ? e : 0];
      // coefficients of g_c and g_dd in ua / ub
      const double kac = vertex ? 4.0 * ((a == dd) ? s_s : s_d) : 16.0 * ((b == dd) ? m_s : m_d);
      const double kad = vertex ? 4.0 * ((a == c) ? s_s : s_d) : 16.0 * ((b == c) ? m_s : m_d);
      const double kbc = 16.0 * ((a == dd) ? m_s : m_d), kbd = 16.0 * ((a == c) ? m_s : m_d);
#pragma unroll
      for (int d = 0; d < TDIM; ++d)
      {
        ua[d] = kac * G[c][d] + kad * G[dd][d];
        ub[d] = kbc * G[c][d] + kbd * G[dd][d];
      }
    }
    double H[TDIM][TDIM];
#pragma unroll
    for (int p = 0; p < TDIM; ++p)
#pragma unroll
      for (int q = 0; q < TDIM; ++q) H[p][q] = Ga[p] * ua[q] + Gb[p] * ub[q];
    sink(j, H);
  }
}

// component row kc of dof lr of the degree-2 elasticity tensor of an affine simplex: acc[j * TDIM + b], closed form
template <int TDIM>
__device__ __forceinline__ void p2_elasticity_row(const Geo<TDIM>& g, int lr, int kc, double lmbda, double mu, double* acc)
{
  p2_gradient_gram_row<TDIM>(g, lr, [&](int j, const double (&H)[TDIM][TDIM])
  {
    double tr = 0.0;
#pragma unroll
    for (int d = 0; d < TDIM; ++d) tr += H[d][d];
#pragma unroll
    for (int b = 0; b < TDIM; ++b)
    {
      double hab = 0.0, hba = 0.0; // H[kc][b], H[b][kc] through select chains (kc is a runtime index)
#pragma unroll
      for (int p = 0; p < TDIM; ++p) { hab = (p == kc) ? H[p][b] : hab; hba = (p == kc) ? H[b][p] : hba; }
      acc[j * TDIM + b] += lmbda * hab + mu * hba + ((b == kc) ? mu * tr : 0.0);
    }
  });
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

// sin(pi x): x = k + r, |r| <= 1/2, sign from the parity of k.  Tighter than libm's sin(pi*x),
// whose argument already carries the rounding of pi, at ~1/4 of the instructions of ocml's sinpi:
// the source-term kernels are FP64-VALU bound on exactly this.
// sin(pi r) on |r| <= 1/2: r times a degree-8 polynomial in r^2 (Chebyshev interpolant computed with 60 digits:
// approximation error 3e-19; 2.5e-16 measured in double over [-3, 3])
__device__ __forceinline__ double cfx_sinpi_reduced(double r)
{
  const double r2 = r * r;
  double p = 0x1.9d462020fcc78p-21;
  p = fma(p, r2, -0x1.6f7acdb8f6580p-16);
  p = fma(p, r2, 0x1.e8f3675ee37ddp-12);
  p = fma(p, r2, -0x1.e3074dfaf87afp-8);
  p = fma(p, r2, 0x1.5078348551854p-4);
  p = fma(p, r2, -0x1.32d2cce627c86p-1);
  p = fma(p, r2, 0x1.466bc6775aa7dp+1);
  p = fma(p, r2, -0x1.4abbce625be52p+2);
  p = fma(p, r2, 0x1.921fb54442d18p+1);
  return p * r;
}

__device__ __forceinline__ double cfx_sinpi(double x)
{
  const double k = rint(x);
  const double s = cfx_sinpi_reduced(x - k);
  // (-1)^k through the sign bit; k = rint(x) converts exactly for |x| < 2^31 (beyond that the conversion
  // saturates: coordinates of that size have no fractional digits left that a mesh could use)
  const int odd = ((int)k) & 1;
  return __hiloint2double(__double2hiint(s) ^ (odd << 31), __double2loint(s));
}

// sin(pi x) and cos(pi x) from one reduction; cos(pi r) = 1 - 2 sin^2(pi r / 2) (absolute error ~2e-16)
__device__ __forceinline__ void cfx_sincospi(double x, double& s, double& c)
{
  const double k = rint(x);
  const double r = x - k;
  const double s0 = cfx_sinpi_reduced(r), sh = cfx_sinpi_reduced(0.5 * r);
  const double c0 = fma(-2.0 * sh, sh, 1.0);
  const int flip = (((int)k) & 1) << 31;
  s = __hiloint2double(__double2hiint(s0) ^ flip, __double2loint(s0));
  c = __hiloint2double(__double2hiint(c0) ^ flip, __double2loint(c0));
}

template <int GDIM>
__device__ __forceinline__ double field_eval(int id, const double* x)
{
  if (id == CFX_F_ONE) return 1.0;
  double p = 1.0;
#pragma unroll
  for (int d = 0; d < GDIM; ++d) p *= cfx_sinpi(x[d]);
  if (id == CFX_F_SINPROD) return p;
  return (double)GDIM * kPi * kPi * p;
}

// ---------------------------------------------------------------------------
// Bilinear forms whose test and trial spaces differ (cfx_form_create2; assemble_matrix_impl.h:68-189 with dofmap0 /
// bs0 != dofmap1 / bs1): row (ia, ik) of the [(nd0 bs0) x (nd1 bs1)] element tensor of one cell integral over one
// rule, accumulated into acc[j][b] (fixed strides: the sizes are run-time values -- one body serves every pair of
// Lagrange spaces).  Shared by the entity-parallel kernel (cfx_fem.hip) and the row gather (cfx_gather.hip).
// ---------------------------------------------------------------------------
struct RectArgs
{
  const int32_t* dofmap1;
  int deg0, bs0, nd0, deg1, bs1, nd1;
};

template <int TDIM>
struct RectRow
{
  static constexpr int MAXND = TDIM == 2 ? 6 : 10, MAXBS = TDIM;
  static __device__ __forceinline__ void accumulate(const RectArgs& R, int kernel, double scale, const Geo<TDIM>& g, int ia,
                                                    int ik, int npts, const double* __restrict__ pts,
                                                    const double* __restrict__ wts, double wscale,
                                                    double (&acc)[MAXND][MAXBS])
  {
    for (int q = 0; q < npts; ++q)
    {
      double X[TDIM];
#pragma unroll
      for (int t = 0; t < TDIM; ++t) X[t] = pts[(int64_t)q * TDIM + t];
      const double w = wts[q] * wscale;
      double N[MAXND], dN[MAXND][TDIM];
#pragma unroll
      for (int j = 0; j < MAXND; ++j)
      {
        N[j] = 0.0;
#pragma unroll
        for (int t = 0; t < TDIM; ++t) dN[j][t] = 0.0;
      }
      // the row's basis function of the test element: value and physical gradient
      if (R.deg0 == 1) tabulate<TDIM, 1>(X, N, dN); else tabulate<TDIM, 2>(X, N, dN);
      double Ni = 0.0, Gi[TDIM];
#pragma unroll
      for (int d = 0; d < TDIM; ++d) Gi[d] = 0.0;
#pragma unroll
      for (int j = 0; j < MAXND; ++j)
        if (j == ia)
        {
          Ni = N[j];
#pragma unroll
          for (int d = 0; d < TDIM; ++d)
          {
            double v = 0.0;
#pragma unroll
            for (int t = 0; t < TDIM; ++t) v += g.K[t][d] * dN[j][t];
            Gi[d] = v;
          }
        }
      double Gia = 0.0; // component ik of the row gradient (DIV_TEST)
#pragma unroll
      for (int d = 0; d < TDIM; ++d) Gia = (d == ik) ? Gi[d] : Gia;
      // the trial element
      if (R.deg1 == 1) tabulate<TDIM, 1>(X, N, dN); else tabulate<TDIM, 2>(X, N, dN);
#pragma unroll
      for (int j = 0; j < MAXND; ++j)
      {
        if (j >= R.nd1) continue;
        double Gj[TDIM];
#pragma unroll
        for (int d = 0; d < TDIM; ++d)
        {
          double v = 0.0;
#pragma unroll
          for (int t = 0; t < TDIM; ++t) v += g.K[t][d] * dN[j][t];
          Gj[d] = v;
        }
        switch (kernel)
        {
        case CFX_K_MASS:
#pragma unroll
          for (int b = 0; b < MAXBS; ++b) acc[j][b] += (b == ik) ? w * Ni * N[j] : 0.0;
          break;
        case CFX_K_STIFFNESS:
        {
          double sgg = 0.0;
#pragma unroll
          for (int d = 0; d < TDIM; ++d) sgg += Gi[d] * Gj[d];
#pragma unroll
          for (int b = 0; b < MAXBS; ++b) acc[j][b] += (b == ik) ? w * sgg : 0.0;
          break;
        }
        case CFX_K_DIV_TEST: // v = N0_i e_ik, p = N1_j
          acc[j][0] += w * scale * Gia * N[j];
          break;
        case CFX_K_DIV_TRIAL: // q = N0_i, u = N1_j e_b
#pragma unroll
          for (int b = 0; b < MAXBS; ++b) acc[j][b] += w * scale * Ni * Gj[b];
          break;
        default: break;
        }
      }
    }
  }
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
// Row (ia, ik) of a CELL integral's element tensor, accumulated into acc:
// RANK 2: acc[j*BS + b] over all trial dofs; RANK 1: acc[0].
// `g` must hold the vertex coordinates and K/detJ (jacobian() already called).
// ---------------------------------------------------------------------------
template <int TDIM, int DEG, int BS, int RANK>
__device__ __forceinline__ void cell_local_row(int kernel, const double* __restrict__ params, int point_stride,
                                               const Geo<TDIM>& g, double h, int npts,
                                               const double* __restrict__ pts, const double* __restrict__ wts,
                                               double wscale, const double* __restrict__ pdata, int ia, int ik,
                                               double* acc, const double* cw = nullptr)
{
  // cw: the cell's ND packed coefficient dof values (pack_coefficients, pack_form.h:32-170).  RANK 1: the source
  // field f = sum_j N_j cw[j] when the field id is CFX_F_COEFFICIENT (vector spaces: component ik of a vector-valued
  // Function); RANK 2: a scalar coefficient kappa = sum_j N_j cw[j] that multiplies the integrand
  constexpr int ND = Elem<TDIM, DEG>::ND;
  for (int q = 0; q < npts; ++q)
  {
    double X[TDIM];
#pragma unroll
    for (int t = 0; t < TDIM; ++t) X[t] = pts[(int64_t)q * TDIM + t];
    double w = wts[q] * wscale;
    double N[ND], dN[ND][TDIM], G[ND][TDIM];
    tabulate<TDIM, DEG>(X, N, dN);
    if (RANK == 2 && cw != nullptr)
    {
      double kappa = 0.0;
#pragma unroll
      for (int j = 0; j < ND; ++j) kappa += N[j] * cw[j];
      w *= kappa;
    }
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
      switch (kernel)
      {
      case CFX_K_MASS:
#pragma unroll
        for (int j = 0; j < ND; ++j)
#pragma unroll
          for (int b = 0; b < BS; ++b) acc[j * BS + b] += (b == ik) ? w * Ni * N[j] : 0.0;
        break;
      case CFX_K_STIFFNESS:
#pragma unroll
        for (int j = 0; j < ND; ++j)
        {
          double s = 0.0;
#pragma unroll
          for (int d = 0; d < TDIM; ++d) s += Gi[d] * G[j][d];
#pragma unroll
          for (int b = 0; b < BS; ++b) acc[j * BS + b] += (b == ik) ? w * s : 0.0;
        }
        break;
      case CFX_K_NITSCHE:
        if constexpr (BS == 1)
        {
          const double* nrm = pdata + (int64_t)q * point_stride;
          const double gam = params[0] / h;
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
          // sigma(u):eps(v), sigma = 2 mu eps + lambda tr(eps) I  (python/demo/demo_elasticity.py:167-238)
          const double E = params[0], nu = params[1];
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
      if (kernel == CFX_L_SOURCE)
      {
        double f;
        if ((int)params[0] == CFX_F_COEFFICIENT)
        {
          f = 0.0;
#pragma unroll
          for (int j = 0; j < ND; ++j) f += N[j] * cw[j];
          f *= params[1];
        }
        else
          f = params[1] * field_eval<TDIM>((int)params[0], xq);
        acc[0] += w * f * Ni;
      }
      else if (kernel == CFX_L_NITSCHE_RHS)
      {
        const double* nrm = pdata + (int64_t)q * point_stride;
        const double gam = params[0] / h;
        const double gv = params[2] * field_eval<TDIM>((int)params[1], xq);
        double dni = 0.0;
#pragma unroll
        for (int d = 0; d < TDIM; ++d) dni += Gi[d] * nrm[d];
        acc[0] += w * (-dni * gv + gam * gv * Ni);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Row (ia, ik) of an INTERIOR-FACET integral's macro element tensor.
// Macro element = [cell0 dofs, cell1 dofs]; Ae block layout [[00,01],[10,11]]
// (assemble_matrix_impl.h:537-542); ia in [0, 2 ND).  The facet quadrature
// points are pushed to physical space from cell0's facet lf0 and pulled back
// to both reference cells.  acc[j*BS + b], j in [0, 2 ND).
// ---------------------------------------------------------------------------
// KC: kernel class compiled in -- 0: ghost-penalty gradient jump + extension penalty; 1: the DG skeleton terms
// (value jump, symmetric interior penalty), which keep both cells' basis VALUES live (+55 VGPRs for P1 tets)
template <int TDIM, int DEG, int BS, int KC = 0>
__device__ __forceinline__ void facet_local_row(int kernel, const double* __restrict__ params, int qdegree,
                                                const Geo<TDIM>& g0, const Geo<TDIM>& g1, int lf0, int ia, int ik,
                                                double* acc, int npts = 0, const double* __restrict__ pts = nullptr,
                                                const double* __restrict__ wts = nullptr,
                                                const double (*xhost)[TDIM] = nullptr)
{
  // pts != nullptr: a facet-hosted runtime rule (8f-4) -- npts points on the reference simplex spanned by the
  // host vertices xhost, physical-measure weights; else the reference facet rule of degree qdegree
  constexpr int ND = Elem<TDIM, DEG>::ND;
  if (KC == 0 && kernel == CFX_K_EXTENSION_L2)
  {
    // pair (bad = cell0, root = cell1): full-cell rule of the bad cell, the root's basis evaluated
    // at the pulled-back points (its polynomial extension), macro basis M = [N_bad, -N_root]
    int nref;
    const double* wref;
    const double* pref = ref_rule(TDIM, qdegree, nref, wref);
    const double scale = params[0] * fabs(g0.detJ);
    for (int q = 0; q < nref; ++q)
    {
      double X0[TDIM], X1[TDIM], xq[TDIM], l0 = 1.0;
#pragma unroll
      for (int t = 0; t < TDIM; ++t) { X0[t] = pref[q * TDIM + t]; l0 -= X0[t]; }
#pragma unroll
      for (int d = 0; d < TDIM; ++d)
      {
        double v = l0 * g0.x[0][d];
#pragma unroll
        for (int t = 0; t < TDIM; ++t) v += X0[t] * g0.x[t + 1][d];
        xq[d] = v;
      }
#pragma unroll
      for (int t = 0; t < TDIM; ++t)
      {
        double v = 0.0;
#pragma unroll
        for (int d = 0; d < TDIM; ++d) v += g1.K[t][d] * (xq[d] - g1.x[0][d]);
        X1[t] = v;
      }
      double N0[ND], N1[ND], dN[ND][TDIM];
      tabulate<TDIM, DEG>(X0, N0, dN);
      tabulate<TDIM, DEG>(X1, N1, dN);
      const double mi = ia < ND ? N0[ia] : -N1[ia - ND];
      const double w = wref[q] * scale * mi;
#pragma unroll
      for (int j = 0; j < ND; ++j)
      {
        acc[j * BS + ik] += w * N0[j];
        acc[(ND + j) * BS + ik] -= w * N1[j];
      }
    }
    return;
  }
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

  int nref;
  const double* wref;
  const double* pref = ref_rule(TDIM - 1, qdegree, nref, wref);
  if (pts)
  {
    nref = npts; pref = pts; wref = wts; scale = 1.0;
#pragma unroll
    for (int j = 0; j < TDIM; ++j)
#pragma unroll
      for (int d = 0; d < TDIM; ++d) xf[j][d] = xhost[j][d];
  }
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
    // gamma h_avg^(1 + params[1]) (GHOST_GRADJUMP; the other kernels of this function read params[0] alone)
    const double w = wref[q] * scale * params[0] * havg * ((KC == 0 && params[1] != 0.0) ? pow(havg, params[1]) : 1.0);
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
    if (KC == 0 && kernel == CFX_K_GHOST_GRADJUMP)
    {
#pragma unroll
      for (int j = 0; j < 2 * ND; ++j)
#pragma unroll
        for (int b = 0; b < BS; ++b) acc[j * BS + b] += (b == ik) ? w * ji * jn[j] : 0.0;
    }
    else if (KC == 1 && kernel == CFX_K_SIP)
    {
      // symmetric interior penalty (python/demo/demo_dg_poisson.py:262-265):
      // -{dn u}[v] - {dn v}[u] + sigma / h_avg [u][v], {dn w} = (grad w+ + grad w-) . n+ / 2, [w] = w+ - w-
      const double wq = wref[q] * scale;
      const double pen = params[0] / havg;
      double vi = 0.0, ai = 0.0;
#pragma unroll
      for (int j = 0; j < ND; ++j)
      {
        vi = (j == ia) ? N0[j] : vi; vi = (ND + j == ia) ? -N1[j] : vi;
        ai = (j == ia) ? 0.5 * jn[j] : ai; ai = (ND + j == ia) ? -0.5 * jn[ND + j] : ai;
      }
#pragma unroll
      for (int j = 0; j < ND; ++j)
#pragma unroll
        for (int b = 0; b < BS; ++b)
        {
          acc[j * BS + b] += (b == ik) ? wq * (-0.5 * jn[j] * vi - ai * N0[j] + pen * vi * N0[j]) : 0.0;
          acc[(ND + j) * BS + b] += (b == ik) ? wq * (0.5 * jn[ND + j] * vi + ai * N1[j] - pen * vi * N1[j]) : 0.0;
        }
    }
    else if (KC == 1 && kernel == CFX_K_JUMP)
    {
      // gamma / h_avg [u][v]: the value jump of the macro basis is [N0, -N1]
      const double wj = wref[q] * scale * params[0] / havg;
      double vi = 0.0;
#pragma unroll
      for (int j = 0; j < ND; ++j) { vi = (j == ia) ? N0[j] : vi; vi = (ND + j == ia) ? -N1[j] : vi; }
#pragma unroll
      for (int j = 0; j < ND; ++j)
#pragma unroll
        for (int b = 0; b < BS; ++b)
        {
          acc[j * BS + b] += (b == ik) ? wj * vi * N0[j] : 0.0;
          acc[(ND + j) * BS + b] -= (b == ik) ? wj * vi * N1[j] : 0.0;
        }
    }
  }
}

// ---------------------------------------------------------------------------
// Row ia in [0, 2 nd0) of an interior-facet integral between two SCALAR spaces (test: R.deg0 / nd0, trial: R.deg1 /
// nd1): macro test dofs = [cell0, cell1] of the test dofmap, macro trial dofs = [cell0, cell1] of the trial dofmap
// (assemble_matrix_impl.h:462-606 builds dmapjoint0 / dmapjoint1 from dofmap0 and dofmap1 separately), acc[j], j in
// [0, 2 nd1).  gamma h_avg^(1 + p) [dn v][dn u] (CFX_K_GHOST_GRADJUMP) and gamma / h_avg [v][u] (CFX_K_JUMP) over the
// standard facet rule.  Run-time degrees, fixed strides: one body for every pair of Lagrange spaces.
// ---------------------------------------------------------------------------
template <int TDIM>
__device__ __forceinline__ void facet_local_row2(const RectArgs& R, int kernel, const double* __restrict__ params, int qdegree,
                                                 const Geo<TDIM>& g0, const Geo<TDIM>& g1, int lf0, int ia,
                                                 double (&acc)[2 * RectRow<TDIM>::MAXND])
{
  constexpr int MAXND = RectRow<TDIM>::MAXND;
  const double havg = 0.5 * (cell_diameter<TDIM>(g0) + cell_diameter<TDIM>(g1));
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
  const bool grad = kernel == CFX_K_GHOST_GRADJUMP;
  int nref;
  const double* wref;
  const double* pref = ref_rule(TDIM - 1, qdegree, nref, wref);
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
    // the jump of macro basis function m of a space of degree `deg`: [N0, -N1] or the normal derivatives
    double N[MAXND], dN[MAXND][TDIM];
    auto side_jumps = [&](int deg, const double* X, const Geo<TDIM>& g, double sign, double* out)
    {
#pragma unroll
      for (int j = 0; j < MAXND; ++j)
      {
        N[j] = 0.0;
#pragma unroll
        for (int t = 0; t < TDIM; ++t) dN[j][t] = 0.0;
      }
      if (deg == 1) tabulate<TDIM, 1>(X, N, dN); else tabulate<TDIM, 2>(X, N, dN);
#pragma unroll
      for (int j = 0; j < MAXND; ++j)
      {
        double a = 0.0;
#pragma unroll
        for (int d = 0; d < TDIM; ++d)
#pragma unroll
          for (int t = 0; t < TDIM; ++t) a += g.K[t][d] * dN[j][t] * nrm[d];
        out[j] = sign * (grad ? a : N[j]);
      }
    };
    double jt0[MAXND], jt1[MAXND], ju0[MAXND], ju1[MAXND];
    side_jumps(R.deg0, X0, g0, 1.0, jt0);
    side_jumps(R.deg0, X1, g1, -1.0, jt1);
    side_jumps(R.deg1, X0, g0, 1.0, ju0);
    side_jumps(R.deg1, X1, g1, -1.0, ju1);
    double ji = 0.0;
#pragma unroll
    for (int j = 0; j < MAXND; ++j)
    {
      ji = (j == ia && j < R.nd0) ? jt0[j] : ji; // (ia in [nd0, 2 nd0) is cell 1's row ia - nd0, not cell 0's row ia)
      ji = (R.nd0 + j == ia && j < R.nd0) ? jt1[j] : ji;
    }
    const double w = grad ? wref[q] * scale * params[0] * havg * (params[1] != 0.0 ? pow(havg, params[1]) : 1.0)
                          : wref[q] * scale * params[0] / havg;
#pragma unroll
    for (int j = 0; j < MAXND; ++j)
    {
      acc[j] += (j < R.nd1) ? w * ji * ju0[j] : 0.0;
      acc[MAXND + j] += (j < R.nd1) ? w * ji * ju1[j] : 0.0;
    }
  }
}

} // namespace cfx
