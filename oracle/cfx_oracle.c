/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see cfx_oracle.h for the parity
 * status).  Serial, scalar, written for clarity: this is the checker, never
 * the product path.  All "ref:" citations are relative to /root/reference.
 */
#include "cfx_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#include "cfx_quadrature_tables.h"

#define MAXV 4      /* vertices of a simplex cell */
#define MAXND 10    /* scalar dofs per cell (P2 tet) */
#define MAXBS 3
#define MAXLOC (2 * MAXND * MAXBS)

void orc_free(void* p) { free(p); }

void orc_rules_free(orc_rules* r)
{
  if (!r) return;
  free(r->points); free(r->weights); free(r->offsets); free(r->parent_map);
  memset(r, 0, sizeof(*r));
}

/* ------------------------------------------------------------------------ */
/* Synthetic mesh: box [0,1]^d, N^d cubes.  Vertex id ix+(N+1)(iy+(N+1)iz). */
/* Hex -> 6 tets {0,1,3,7},{0,1,5,7},{0,2,3,7},{0,2,6,7},{0,4,5,7},{0,4,6,7} */
/* quad -> {0,1,3},{0,3,2}; local corner i = bx + 2 by + 4 bz                */
/* ref: cpp/cutfemx/distance/fast_iterative.h:93-94,103-108                  */
/* ------------------------------------------------------------------------ */
static const int kuhn_tet[6][4] = {{0, 1, 3, 7}, {0, 1, 5, 7}, {0, 2, 3, 7},
                                   {0, 2, 6, 7}, {0, 4, 5, 7}, {0, 4, 6, 7}};
static const int kuhn_tri[2][3] = {{0, 1, 3}, {0, 3, 2}};

void orc_mesh_box(int tdim, int N, double* x, int32_t* conn)
{
  const int64_t n1 = N + 1;
  if (tdim == 2)
  {
    for (int64_t iy = 0; iy <= N; ++iy)
      for (int64_t ix = 0; ix <= N; ++ix)
      {
        int64_t v = ix + n1 * iy;
        x[3 * v + 0] = (double)ix / (double)N;
        x[3 * v + 1] = (double)iy / (double)N;
        x[3 * v + 2] = 0.0;
      }
    for (int64_t iy = 0; iy < N; ++iy)
      for (int64_t ix = 0; ix < N; ++ix)
      {
        int64_t q = ix + (int64_t)N * iy;
        int64_t corner[4];
        for (int i = 0; i < 4; ++i)
          corner[i] = (ix + (i & 1)) + n1 * (iy + ((i >> 1) & 1));
        for (int k = 0; k < 2; ++k)
          for (int j = 0; j < 3; ++j)
            conn[(2 * q + k) * 3 + j] = (int32_t)corner[kuhn_tri[k][j]];
      }
    return;
  }
  for (int64_t iz = 0; iz <= N; ++iz)
    for (int64_t iy = 0; iy <= N; ++iy)
      for (int64_t ix = 0; ix <= N; ++ix)
      {
        int64_t v = ix + n1 * (iy + n1 * iz);
        x[3 * v + 0] = (double)ix / (double)N;
        x[3 * v + 1] = (double)iy / (double)N;
        x[3 * v + 2] = (double)iz / (double)N;
      }
  for (int64_t iz = 0; iz < N; ++iz)
    for (int64_t iy = 0; iy < N; ++iy)
      for (int64_t ix = 0; ix < N; ++ix)
      {
        int64_t h = ix + (int64_t)N * (iy + (int64_t)N * iz);
        int64_t corner[8];
        for (int i = 0; i < 8; ++i)
          corner[i] = (ix + (i & 1)) + n1 * ((iy + ((i >> 1) & 1)) + n1 * (iz + ((i >> 2) & 1)));
        for (int k = 0; k < 6; ++k)
          for (int j = 0; j < 4; ++j)
            conn[(6 * h + k) * 4 + j] = (int32_t)corner[kuhn_tet[k][j]];
      }
}

/* ------------------------------------------------------------------------ */
/* a1 classification.  inside iff all dof values < 0, outside iff all > 0,   */
/* else intersected (a zero value => intersected).  Strict, no epsilon.      */
/* ref: cpp/cutfemx/cut/cut.cpp:292-321, docs/user-guide/level-sets.md:84-88 */
/* ------------------------------------------------------------------------ */
void orc_classify(int64_t ncells, int ndofs_cell, const int32_t* ls_dofmap,
                  const double* ls_values, int8_t* domain)
{
  for (int64_t c = 0; c < ncells; ++c)
  {
    int all_neg = 1, all_pos = 1;
    for (int i = 0; i < ndofs_cell; ++i)
    {
      const double v = ls_values[ls_dofmap[c * ndofs_cell + i]];
      all_neg = all_neg && (v < 0.0);
      all_pos = all_pos && (v > 0.0);
    }
    domain[c] = all_neg ? ORC_INSIDE : (all_pos ? ORC_OUTSIDE : ORC_INTERSECTED);
  }
}

/* ------------------------------------------------------------------------ */
/* a4 selector: DNF "phi<0 and phi1>0 or phi=0".                             */
/* relation -> domain set, ref: cut.cpp:323-342; scan ref: cut.cpp:887-921   */
/* ------------------------------------------------------------------------ */
typedef struct { int ls; int mask; int term; } sel_clause; /* mask bit0 inside, bit1 cut, bit2 outside */

static int parse_selector(const char* s, int nls, sel_clause* out, int cap)
{
  int n = 0, term = 0;
  const char* p = s;
  for (;;)
  {
    while (*p && isspace((unsigned char)*p)) ++p;
    if (!*p) return -1;
    /* name */
    const char* b = p;
    while (*p && (isalnum((unsigned char)*p) || *p == '_')) ++p;
    if (p == b) return -1;
    int ls = -1;
    size_t len = (size_t)(p - b);
    if (len >= 3 && strncmp(b, "phi", 3) == 0)
    {
      if (len == 3) ls = 0;
      else
      {
        ls = 0;
        for (size_t i = 3; i < len; ++i)
        {
          if (!isdigit((unsigned char)b[i])) return -1;
          ls = 10 * ls + (b[i] - '0');
        }
      }
    }
    if (ls < 0 || ls >= nls) return -1;
    while (*p && isspace((unsigned char)*p)) ++p;
    int mask;
    if (p[0] == '<' && p[1] == '=') { mask = 1 | 2; p += 2; }
    else if (p[0] == '>' && p[1] == '=') { mask = 4 | 2; p += 2; }
    else if (p[0] == '=' && p[1] == '=') { mask = 2; p += 2; }
    else if (p[0] == '<') { mask = 1; p += 1; }
    else if (p[0] == '>') { mask = 4; p += 1; }
    else if (p[0] == '=') { mask = 2; p += 1; }
    else return -1;
    while (*p && isspace((unsigned char)*p)) ++p;
    /* right-hand side must be zero */
    char* e;
    double rhs = strtod(p, &e);
    if (e == p || rhs != 0.0) return -1;
    p = e;
    if (n >= cap) return -1;
    out[n].ls = ls; out[n].mask = mask; out[n].term = term; ++n;
    while (*p && isspace((unsigned char)*p)) ++p;
    if (!*p) break;
    if (strncmp(p, "and", 3) == 0) p += 3;
    else if (strncmp(p, "&&", 2) == 0) p += 2;
    else if (*p == '&') p += 1;
    else if (strncmp(p, "or", 2) == 0) { p += 2; ++term; }
    else if (strncmp(p, "||", 2) == 0) { p += 2; ++term; }
    else if (*p == '|') { p += 1; ++term; }
    else return -1;
  }
  return n;
}

static int clause_match(const sel_clause* cl, int ncl, int nls, int64_t ncells,
                        const int8_t* domain, int64_t cell)
{
  (void)nls;
  int nterm = cl[ncl - 1].term + 1;
  for (int t = 0; t < nterm; ++t)
  {
    int ok = 1, any = 0;
    for (int k = 0; k < ncl; ++k)
    {
      if (cl[k].term != t) continue;
      any = 1;
      int d = domain[(int64_t)cl[k].ls * ncells + cell]; /* -1,0,1 */
      if (!((cl[k].mask >> (d + 1)) & 1)) { ok = 0; break; }
    }
    if (any && ok) return 1;
  }
  return 0;
}

int64_t orc_locate_entities(int64_t ncells, int nls, const int8_t* domain,
                            const char* selector, int32_t* out)
{
  sel_clause cl[32];
  int ncl = parse_selector(selector, nls, cl, 32);
  if (ncl <= 0) return -1;
  int64_t n = 0;
  for (int64_t c = 0; c < ncells; ++c)
    if (clause_match(cl, ncl, nls, ncells, domain, c)) out[n++] = (int32_t)c;
  return n;
}

/* ------------------------------------------------------------------------ */
/* geometry helpers (affine simplices, gdim == tdim)                         */
/* ------------------------------------------------------------------------ */
static void cell_coords(const orc_mesh* m, int64_t c, double xc[MAXV][3])
{
  const int nv = m->tdim + 1;
  for (int i = 0; i < nv; ++i)
  {
    const int64_t v = m->conn[c * nv + i];
    xc[i][0] = m->x[3 * v + 0];
    xc[i][1] = m->x[3 * v + 1];
    xc[i][2] = m->x[3 * v + 2];
  }
}

/* J[d][t] = x_{t+1}[d]-x_0[d]; K = J^{-1} (K[t][d]); returns det J */
static double jacobian(int tdim, double xc[MAXV][3], double J[3][3], double K[3][3])
{
  for (int d = 0; d < tdim; ++d)
    for (int t = 0; t < tdim; ++t)
      J[d][t] = xc[t + 1][d] - xc[0][d];
  if (tdim == 2)
  {
    double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    K[0][0] = J[1][1] / det;  K[0][1] = -J[0][1] / det;
    K[1][0] = -J[1][0] / det; K[1][1] = J[0][0] / det;
    return det;
  }
  double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
  double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
  K[0][0] = c00 / det;
  K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
  K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
  K[1][0] = c01 / det;
  K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
  K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
  K[2][0] = c02 / det;
  K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
  K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
  return det;
}

/* UFL CellDiameter: max distance between two vertices */
static double cell_diameter(int tdim, double xc[MAXV][3])
{
  double h2 = 0.0;
  for (int i = 0; i <= tdim; ++i)
    for (int j = i + 1; j <= tdim; ++j)
    {
      double d2 = 0.0;
      for (int d = 0; d < tdim; ++d)
        d2 += (xc[i][d] - xc[j][d]) * (xc[i][d] - xc[j][d]);
      if (d2 > h2) h2 = d2;
    }
  return sqrt(h2);
}

/* Lagrange tabulation on the reference simplex.  dof order = Basix: vertices,
   then edges (tri: (1,2),(0,2),(0,1); tet: (2,3),(1,3),(1,2),(0,3),(0,2),(0,1)) */
static const int tri_edges[3][2] = {{1, 2}, {0, 2}, {0, 1}};
static const int tet_edges[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};

static void tabulate(int tdim, int degree, const double* X, double* N, double dN[][3])
{
  double lam[4], dlam[4][3];
  lam[0] = 1.0;
  for (int t = 0; t < tdim; ++t) { lam[0] -= X[t]; lam[t + 1] = X[t]; }
  for (int i = 0; i <= tdim; ++i)
    for (int t = 0; t < tdim; ++t)
      dlam[i][t] = (i == 0) ? -1.0 : ((i - 1 == t) ? 1.0 : 0.0);
  if (degree == 1)
  {
    for (int i = 0; i <= tdim; ++i)
    {
      N[i] = lam[i];
      for (int t = 0; t < tdim; ++t) dN[i][t] = dlam[i][t];
    }
    return;
  }
  for (int i = 0; i <= tdim; ++i)
  {
    N[i] = lam[i] * (2.0 * lam[i] - 1.0);
    for (int t = 0; t < tdim; ++t) dN[i][t] = (4.0 * lam[i] - 1.0) * dlam[i][t];
  }
  const int ne = tdim == 2 ? 3 : 6;
  for (int e = 0; e < ne; ++e)
  {
    const int a = tdim == 2 ? tri_edges[e][0] : tet_edges[e][0];
    const int b = tdim == 2 ? tri_edges[e][1] : tet_edges[e][1];
    N[tdim + 1 + e] = 4.0 * lam[a] * lam[b];
    for (int t = 0; t < tdim; ++t)
      dN[tdim + 1 + e][t] = 4.0 * (lam[a] * dlam[b][t] + dlam[a][t] * lam[b]);
  }
}

static void ref_rule(int dim, int degree, int* n, const double** pts, const double** wts)
{
  if (degree < 0) degree = 0;
  if (degree > CFX_QUAD_MAX_DEGREE) degree = CFX_QUAD_MAX_DEGREE;
  if (dim == 1)
  {
    *n = cfx_quad_offset_1d[degree + 1] - cfx_quad_offset_1d[degree];
    *pts = cfx_quad_points_1d + 1 * cfx_quad_offset_1d[degree];
    *wts = cfx_quad_weights_1d + cfx_quad_offset_1d[degree];
  }
  else if (dim == 2)
  {
    *n = cfx_quad_offset_2d[degree + 1] - cfx_quad_offset_2d[degree];
    *pts = cfx_quad_points_2d + 2 * cfx_quad_offset_2d[degree];
    *wts = cfx_quad_weights_2d + cfx_quad_offset_2d[degree];
  }
  else
  {
    *n = cfx_quad_offset_3d[degree + 1] - cfx_quad_offset_3d[degree];
    *pts = cfx_quad_points_3d + 3 * cfx_quad_offset_3d[degree];
    *wts = cfx_quad_weights_3d + cfx_quad_offset_3d[degree];
  }
}

/* ------------------------------------------------------------------------ */
/* a2 sub-triangulation of one cut simplex with a P1 level set.              */
/* Vertex v is "negative" iff phi_v < 0 (zeros side with the positive part). */
/* Local point ids: 0..tdim = parent vertices, then edge cut points.         */
/* Emits sub-simplices of the negative part, the positive part and the       */
/* interface in PARENT REFERENCE coordinates.                                */
/* Semantics: SURVEY 8a-a2 (CutCells, third party: topology unpinned).       */
/* ------------------------------------------------------------------------ */
typedef struct {
  int npts;
  double P[8][3];        /* parent-reference coords of the local points */
  int n_in, n_out, n_if; /* simplices in each part */
  int in[3][4], out[3][4], iface[2][3];
} subtri;

static void ref_vertex(int tdim, int v, double* X)
{
  for (int t = 0; t < tdim; ++t) X[t] = (v == t + 1) ? 1.0 : 0.0;
}

static int cut_point(subtri* s, int tdim, int a, int b, const double* phi)
{
  /* a negative, b non-negative: t = phi_a / (phi_a - phi_b) in [0,1] */
  const double t = phi[a] / (phi[a] - phi[b]);
  double Xa[3], Xb[3];
  ref_vertex(tdim, a, Xa);
  ref_vertex(tdim, b, Xb);
  for (int d = 0; d < tdim; ++d) s->P[s->npts][d] = Xa[d] + t * (Xb[d] - Xa[d]);
  return s->npts++;
}

static void set_simplex(int* dst, int n, int a, int b, int c, int d)
{
  dst[0] = a; dst[1] = b; dst[2] = c;
  if (n == 4) dst[3] = d;
}

/* prism A=(a0,a1,a2), B=(b0,b1,b2) with edges ai-bi -> 3 tets */
static void prism(int out[3][4], int a0, int a1, int a2, int b0, int b1, int b2)
{
  set_simplex(out[0], 4, a0, a1, a2, b2);
  set_simplex(out[1], 4, a0, a1, b1, b2);
  set_simplex(out[2], 4, a0, b0, b1, b2);
}

static void subtriangulate(int tdim, const double* phi, subtri* s)
{
  memset(s, 0, sizeof(*s));
  const int nv = tdim + 1;
  int neg[4], pos[4], nn = 0, np = 0;
  for (int v = 0; v < nv; ++v)
  {
    ref_vertex(tdim, v, s->P[v]);
    if (phi[v] < 0.0) neg[nn++] = v; else pos[np++] = v;
  }
  s->npts = nv;
  if (tdim == 1)
  {
    /* segment host of a facet-hosted cut in 2-D (8f-4): a = negative end, b = the other end */
    if (nn == 0) { s->n_out = 1; s->out[0][0] = 0; s->out[0][1] = 1; return; }
    if (nn == 2) { s->n_in = 1; s->in[0][0] = 0; s->in[0][1] = 1; return; }
    const int a = neg[0], b = pos[0];
    const int q = cut_point(s, 1, a, b, phi);
    s->n_in = 1; s->in[0][0] = a; s->in[0][1] = q;
    s->n_out = 1; s->out[0][0] = q; s->out[0][1] = b;
    s->n_if = 1; s->iface[0][0] = q;
    return;
  }
  if (tdim == 2)
  {
    if (nn == 0) { s->n_out = 1; set_simplex(s->out[0], 3, 0, 1, 2, 0); return; }
    if (nn == 3) { s->n_in = 1; set_simplex(s->in[0], 3, 0, 1, 2, 0); return; }
    if (nn == 1)
    {
      int a = neg[0], b0 = pos[0], b1 = pos[1];
      int q0 = cut_point(s, 2, a, b0, phi), q1 = cut_point(s, 2, a, b1, phi);
      s->n_in = 1; set_simplex(s->in[0], 3, a, q0, q1, 0);
      s->n_out = 2;
      set_simplex(s->out[0], 3, q0, b0, b1, 0);
      set_simplex(s->out[1], 3, q0, b1, q1, 0);
      s->n_if = 1; s->iface[0][0] = q0; s->iface[0][1] = q1;
    }
    else
    {
      int a0 = neg[0], a1 = neg[1], b = pos[0];
      int q0 = cut_point(s, 2, a0, b, phi), q1 = cut_point(s, 2, a1, b, phi);
      s->n_in = 2;
      set_simplex(s->in[0], 3, a0, a1, q1, 0);
      set_simplex(s->in[1], 3, a0, q1, q0, 0);
      s->n_out = 1; set_simplex(s->out[0], 3, b, q0, q1, 0);
      s->n_if = 1; s->iface[0][0] = q0; s->iface[0][1] = q1;
    }
    return;
  }
  if (nn == 0) { s->n_out = 1; set_simplex(s->out[0], 4, 0, 1, 2, 3); return; }
  if (nn == 4) { s->n_in = 1; set_simplex(s->in[0], 4, 0, 1, 2, 3); return; }
  if (nn == 1)
  {
    int a = neg[0];
    int q0 = cut_point(s, 3, a, pos[0], phi);
    int q1 = cut_point(s, 3, a, pos[1], phi);
    int q2 = cut_point(s, 3, a, pos[2], phi);
    s->n_in = 1; set_simplex(s->in[0], 4, a, q0, q1, q2);
    s->n_out = 3; prism(s->out, q0, q1, q2, pos[0], pos[1], pos[2]);
    s->n_if = 1; s->iface[0][0] = q0; s->iface[0][1] = q1; s->iface[0][2] = q2;
  }
  else if (nn == 3)
  {
    int b = pos[0];
    int q0 = cut_point(s, 3, neg[0], b, phi);
    int q1 = cut_point(s, 3, neg[1], b, phi);
    int q2 = cut_point(s, 3, neg[2], b, phi);
    s->n_in = 3; prism(s->in, q0, q1, q2, neg[0], neg[1], neg[2]);
    s->n_out = 1; set_simplex(s->out[0], 4, b, q0, q1, q2);
    s->n_if = 1; s->iface[0][0] = q0; s->iface[0][1] = q1; s->iface[0][2] = q2;
  }
  else
  {
    int a0 = neg[0], a1 = neg[1], b0 = pos[0], b1 = pos[1];
    int q00 = cut_point(s, 3, a0, b0, phi);
    int q01 = cut_point(s, 3, a0, b1, phi);
    int q10 = cut_point(s, 3, a1, b0, phi);
    int q11 = cut_point(s, 3, a1, b1, phi);
    s->n_in = 3; prism(s->in, a0, q00, q01, a1, q10, q11);
    s->n_out = 3; prism(s->out, b0, q00, q10, b1, q01, q11);
    s->n_if = 2;
    s->iface[0][0] = q00; s->iface[0][1] = q01; s->iface[0][2] = q11;
    s->iface[1][0] = q00; s->iface[1][1] = q11; s->iface[1][2] = q10;
  }
}

/* ------------------------------------------------------------------------ */
/* a3 runtime quadrature.  points = parent reference coords, weights =       */
/* physical measure (w_ref * |det sub->parent| * |det parent->phys|; surface */
/* measure for the interface).  Only cut entities are emitted; one rule per  */
/* cut cell for volume parts, one rule per interface sub-facet for "phi=0".  */
/* ref: cpp/cutfemx/cut/cut.cpp:1311-1335, runtime_quadrature.h:223-231,     */
/* python/tests/quadrature_utils.py:40-61, docs/user-guide/quadrature.md     */
/* ------------------------------------------------------------------------ */
static double det_sub(int tdim, double V[4][3])
{
  if (tdim == 1) return V[1][0] - V[0][0];
  if (tdim == 2)
    return (V[1][0] - V[0][0]) * (V[2][1] - V[0][1]) - (V[1][1] - V[0][1]) * (V[2][0] - V[0][0]);
  double a[3], b[3], c[3];
  for (int d = 0; d < 3; ++d)
  {
    a[d] = V[1][d] - V[0][d]; b[d] = V[2][d] - V[0][d]; c[d] = V[3][d] - V[0][d];
  }
  return a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0])
         + a[2] * (b[0] * c[1] - b[1] * c[0]);
}

int orc_runtime_quadrature(const orc_mesh* mesh, const int32_t* ls_dofmap,
                           const double* ls_values, const int8_t* domain,
                           const char* selector, int order, orc_rules* out)
{
  sel_clause cl[4];
  int ncl = parse_selector(selector, 1, cl, 4);
  if (ncl != 1) return -1;
  /* which parts: bit0 negative volume, bit2 positive volume, bit1 interface */
  const int mask = cl[0].mask;
  const int want_in = mask & 1, want_out = (mask & 4) != 0;
  const int want_if = (mask == 2);
  if (want_in && want_out) return -1;
  const int tdim = mesh->tdim, nv = tdim + 1;
  int nref; const double *pref, *wref;
  ref_rule(want_if ? tdim - 1 : tdim, order, &nref, &pref, &wref);

  /* pass 1: count */
  int64_t nr = 0, nq = 0;
  for (int pass = 0; pass < 2; ++pass)
  {
    if (pass == 1)
    {
      memset(out, 0, sizeof(*out));
      out->tdim = tdim; out->nq = nq; out->nr = nr;
      out->points = (double*)malloc(sizeof(double) * (size_t)(nq * tdim + 1));
      out->weights = (double*)malloc(sizeof(double) * (size_t)(nq + 1));
      out->offsets = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nr + 1));
      out->parent_map = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nr + 1));
      out->offsets[0] = 0;
      nr = 0; nq = 0;
    }
    for (int64_t c = 0; c < mesh->ncells; ++c)
    {
      if (domain[c] != ORC_INTERSECTED) continue;
      double phi[4];
      for (int i = 0; i < nv; ++i) phi[i] = ls_values[ls_dofmap[c * nv + i]];
      subtri s;
      subtriangulate(tdim, phi, &s);
      if (want_if)
      {
        if (pass == 0) { nr += s.n_if; nq += (int64_t)s.n_if * nref; continue; }
        double xc[MAXV][3], J[3][3], K[3][3];
        cell_coords(mesh, c, xc);
        jacobian(tdim, xc, J, K);
        for (int f = 0; f < s.n_if; ++f)
        {
          double V[3][3], xp[3][3];
          for (int k = 0; k < tdim; ++k)
            for (int d = 0; d < tdim; ++d) V[k][d] = s.P[s.iface[f][k]][d];
          for (int k = 0; k < tdim; ++k)
            for (int d = 0; d < tdim; ++d)
            {
              xp[k][d] = xc[0][d];
              for (int t = 0; t < tdim; ++t) xp[k][d] += J[d][t] * V[k][t];
            }
          double scale;
          if (tdim == 2)
          {
            double dx = xp[1][0] - xp[0][0], dy = xp[1][1] - xp[0][1];
            scale = sqrt(dx * dx + dy * dy);
          }
          else
          {
            double a[3], b[3];
            for (int d = 0; d < 3; ++d) { a[d] = xp[1][d] - xp[0][d]; b[d] = xp[2][d] - xp[0][d]; }
            double cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2],
                   cz = a[0] * b[1] - a[1] * b[0];
            scale = sqrt(cx * cx + cy * cy + cz * cz);
          }
          for (int q = 0; q < nref; ++q)
          {
            const double* xi = pref + (tdim - 1) * q;
            double l0 = 1.0;
            for (int t = 0; t < tdim - 1; ++t) l0 -= xi[t];
            for (int d = 0; d < tdim; ++d)
            {
              double v = l0 * V[0][d];
              for (int t = 0; t < tdim - 1; ++t) v += xi[t] * V[t + 1][d];
              out->points[(nq + q) * tdim + d] = v;
            }
            out->weights[nq + q] = wref[q] * scale;
          }
          nq += nref;
          out->parent_map[nr] = (int32_t)c;
          out->offsets[nr + 1] = (int32_t)nq;
          ++nr;
        }
        continue;
      }
      const int ns = want_in ? s.n_in : s.n_out;
      if (ns == 0) continue; /* empty part: no rule (keeps active domain clean) */
      if (pass == 0) { nr += 1; nq += (int64_t)ns * nref; continue; }
      double xc[MAXV][3], J[3][3], K[3][3];
      cell_coords(mesh, c, xc);
      const double detJ = fabs(jacobian(tdim, xc, J, K));
      for (int k = 0; k < ns; ++k)
      {
        const int* sx = want_in ? s.in[k] : s.out[k];
        double V[4][3];
        for (int i = 0; i < nv; ++i)
          for (int d = 0; d < tdim; ++d) V[i][d] = s.P[sx[i]][d];
        const double scale = fabs(det_sub(tdim, V)) * detJ;
        for (int q = 0; q < nref; ++q)
        {
          const double* xi = pref + tdim * q;
          double l0 = 1.0;
          for (int t = 0; t < tdim; ++t) l0 -= xi[t];
          for (int d = 0; d < tdim; ++d)
          {
            double v = l0 * V[0][d];
            for (int t = 0; t < tdim; ++t) v += xi[t] * V[t + 1][d];
            out->points[(nq + q) * tdim + d] = v;
          }
          out->weights[nq + q] = wref[q] * scale;
        }
        nq += nref;
      }
      out->parent_map[nr] = (int32_t)c;
      out->offsets[nr + 1] = (int32_t)nq;
      ++nr;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* a3 with several level sets: runtime_quadrature(cut([phi, phi1, ...]), "phi<0 and phi1>0", k)             */
/* (cpp/cutfemx/cut/cut.h:122-181, docs/user-guide/element-classification.md:145-160).  One conjunction of  */
/* clauses; each P1 level set is planar inside a cell, so the region of the cell is the parent simplex      */
/* clipped by one half-space per clause, one after the other, in the order the clauses are written: every   */
/* simplex of the current list is sub-triangulated with the single-level-set tables above (phi_k            */
/* interpolated at its vertices) and its negative ("<") or positive (">") part kept.  With an "=0" clause   */
/* the list starts from that level set's interface sub-facets and the other clauses clip them (triangles    */
/* in 3-D, segments in 2-D).  Rule cells: no clause entirely on the wrong side, at least one level set      */
/* intersected; one rule per cell (volume) / per interface sub-facet ("=0"), none where the clipped part is */
/* empty.  Inclusive relations behave like the strict ones, as for one level set                            */
/* (python/tests/test_cut_api.py::test_cut_api_runtime_quadrature_accepts_inclusive_selector).              */
/* CutCells' own multi-level-set sub-triangulation is third party and absent: parity unpinned, like a2.     */
/* ------------------------------------------------------------------------ */
static double host_measure_fwd(int tdim, double xv[3][3]); /* = host_measure below */
#define ORC_MAX_CLIP 3   /* clipping clauses per conjunction */
#define ORC_MAX_SIMP 27  /* 3^ORC_MAX_CLIP sub-simplices */
typedef struct { double V[4][3]; } simp;

static double phi_at(int tdim, const double* phi, const double* X)
{
  double v = phi[0], l0 = 1.0;
  for (int t = 0; t < tdim; ++t) l0 -= X[t];
  v = l0 * phi[0];
  for (int t = 0; t < tdim; ++t) v += X[t] * phi[t + 1];
  return v;
}

/* the phi < 0 part (keep_out = 0) or the phi > 0 part (keep_out = 1: the "out" simplices of the same tables, so
   that a ">" clause triangulates exactly like the single-level-set "phi>0" rules) of every simplex of the list
   (dimension dim, vertices in parent reference coordinates) */
static int clip_list(int dim, int tdim, const simp* in, int n, const double* phi, int keep_out, simp* out)
{
  int m = 0;
  for (int i = 0; i < n; ++i)
  {
    double psi[4];
    int nneg = 0;
    for (int v = 0; v <= dim; ++v) { psi[v] = phi_at(tdim, phi, in[i].V[v]); nneg += psi[v] < 0.0; }
    if (nneg == (keep_out ? 0 : dim + 1)) { out[m++] = in[i]; continue; } /* wholly on the kept side: unchanged */
    subtri s;
    subtriangulate(dim, psi, &s);
    const int ns = keep_out ? s.n_out : s.n_in;
    for (int k = 0; k < ns; ++k)
    {
      const int* sx = keep_out ? s.out[k] : s.in[k];
      for (int v = 0; v <= dim; ++v)
      {
        const double* L = s.P[sx[v]]; /* the sub-simplex vertex in the coordinates of in[i] */
        for (int d = 0; d < tdim; ++d)
        {
          double x = in[i].V[0][d];
          for (int t = 0; t < dim; ++t) x += L[t] * (in[i].V[t + 1][d] - in[i].V[0][d]);
          out[m].V[v][d] = x;
        }
      }
      ++m;
    }
  }
  return m;
}

int orc_runtime_quadrature_multi(const orc_mesh* mesh, int nls, const int32_t* ls_dofmap,
                                 const double* const* ls_values, const int8_t* domain,
                                 const char* selector, int order, orc_rules* out)
{
  sel_clause cl[8];
  const int ncl = parse_selector(selector, nls, cl, 8);
  if (ncl < 1 || cl[ncl - 1].term != 0) return -1; /* one conjunction */
  int eq = -1, nclip = 0;
  for (int k = 0; k < ncl; ++k)
  {
    if (cl[k].mask == 2) { if (eq >= 0) return -1; eq = k; }
    else if (cl[k].mask == 7 || cl[k].mask == 5) return -1;
    else ++nclip;
  }
  if (nclip > ORC_MAX_CLIP) return -1;
  const int tdim = mesh->tdim, nv = tdim + 1, dim = eq >= 0 ? tdim - 1 : tdim;
  const int64_t nc = mesh->ncells;
  int nref; const double *pref, *wref;
  ref_rule(dim, order, &nref, &pref, &wref);
  int64_t nr = 0, nq = 0;
  for (int pass = 0; pass < 2; ++pass)
  {
    if (pass == 1)
    {
      memset(out, 0, sizeof(*out));
      out->tdim = tdim; out->nq = nq; out->nr = nr;
      out->points = (double*)malloc(sizeof(double) * (size_t)(nq * tdim + 1));
      out->weights = (double*)malloc(sizeof(double) * (size_t)(nq + 1));
      out->offsets = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nr + 1));
      out->parent_map = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nr + 1));
      out->offsets[0] = 0;
      nr = 0; nq = 0;
    }
    for (int64_t c = 0; c < nc; ++c)
    {
      int ok = 1, any_cut = 0;
      for (int k = 0; k < ncl; ++k)
      {
        const int d = domain[(int64_t)cl[k].ls * nc + c];
        if (d == ORC_INTERSECTED) any_cut = 1;
        else if (cl[k].mask == 2 || !((cl[k].mask >> (d + 1)) & 1)) ok = 0;
      }
      if (!ok || !any_cut) continue;
      double phi[8][4];
      for (int k = 0; k < ncl; ++k)
        for (int i = 0; i < nv; ++i) phi[k][i] = ls_values[cl[k].ls][ls_dofmap[c * nv + i]];
      /* base list: the parent simplex, or the interface sub-facets of the "=0" level set */
      simp base[2], a[ORC_MAX_SIMP], b[ORC_MAX_SIMP];
      int nbase = 1;
      if (eq < 0)
        for (int v = 0; v < nv; ++v) ref_vertex(tdim, v, base[0].V[v]);
      else
      {
        subtri s;
        subtriangulate(tdim, phi[eq], &s);
        nbase = s.n_if;
        for (int f = 0; f < nbase; ++f)
          for (int v = 0; v < tdim; ++v)
            for (int d = 0; d < tdim; ++d) base[f].V[v][d] = s.P[s.iface[f][v]][d];
      }
      double xc[MAXV][3], J[3][3], K[3][3];
      cell_coords(mesh, c, xc);
      const double detJ = fabs(jacobian(tdim, xc, J, K));
      /* volume rules: one rule for the whole cell; interface rules: one per base sub-facet */
      const int ngroups = eq < 0 ? 1 : nbase;
      for (int gidx = 0; gidx < ngroups; ++gidx)
      {
        int n = 1;
        a[0] = base[eq < 0 ? 0 : gidx];
        simp *cur = a, *nxt = b;
        for (int k = 0; k < ncl && n > 0; ++k)
        {
          if (k == eq) continue;
          n = clip_list(dim, tdim, cur, n, phi[k], (cl[k].mask & 1) ? 0 : 1, nxt);
          simp* t = cur; cur = nxt; nxt = t;
        }
        if (n == 0) continue;
        if (pass == 0) { nr += 1; nq += (int64_t)n * nref; continue; }
        for (int i = 0; i < n; ++i)
        {
          double scale;
          if (dim == tdim) scale = fabs(det_sub(tdim, cur[i].V)) * detJ;
          else
          {
            double xp[3][3];
            for (int v = 0; v < tdim; ++v)
              for (int d = 0; d < tdim; ++d)
              {
                xp[v][d] = xc[0][d];
                for (int t = 0; t < tdim; ++t) xp[v][d] += J[d][t] * cur[i].V[v][t];
              }
            scale = host_measure_fwd(tdim, xp);
          }
          for (int q = 0; q < nref; ++q)
          {
            const double* xi = pref + dim * q;
            double l0 = 1.0;
            for (int t = 0; t < dim; ++t) l0 -= xi[t];
            for (int d = 0; d < tdim; ++d)
            {
              double v = l0 * cur[i].V[0][d];
              for (int t = 0; t < dim; ++t) v += xi[t] * cur[i].V[t + 1][d];
              out->points[(nq + q) * tdim + d] = v;
            }
            out->weights[nq + q] = wref[q] * scale;
          }
          nq += nref;
        }
        out->parent_map[nr] = (int32_t)c;
        out->offsets[nr + 1] = (int32_t)nq;
        ++nr;
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* 8f-4 facet hosts: cut(level_set, facets, tdim-1) (cut.cpp:540-591, 788-830, */
/* 1022-1063).  Host i is the facet spanned by the mesh vertices verts[i*tdim..] */
/* with level-set dofs ls[i*tdim..]; its rule lives on the (tdim-1)-simplex of   */
/* those vertices, weights carry the physical measure, parent_map = ids[i]       */
/* (host_parent_index, cut.cpp:352-359).  whole != 0: whole-host rules over the  */
/* hosts whose domain code is in `mask` (the standard facets of a mixed          */
/* measure); else one rule per INTERSECTED host over its phi<0 / phi>0 part.     */
/* rule_host[r] = host index of rule r (caller frees).                           */
/* ------------------------------------------------------------------------ */
static double host_measure(int tdim, double xv[3][3]);
static double host_measure_fwd(int tdim, double xv[3][3]) { return host_measure(tdim, xv); }
static double host_measure(int tdim, double xv[3][3])
{
  if (tdim == 2)
  {
    const double dx = xv[1][0] - xv[0][0], dy = xv[1][1] - xv[0][1];
    return sqrt(dx * dx + dy * dy);
  }
  double a[3], b[3];
  for (int d = 0; d < 3; ++d) { a[d] = xv[1][d] - xv[0][d]; b[d] = xv[2][d] - xv[0][d]; }
  const double cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
  return sqrt(cx * cx + cy * cy + cz * cz);
}

int orc_facet_runtime_quadrature(const orc_mesh* mesh, int64_t n, const int32_t* verts, const int32_t* ls,
                                 const int32_t* ids, const double* ls_values, const int8_t* domain,
                                 const char* selector, int order, int whole, orc_rules* out, int32_t** rule_host)
{
  sel_clause cl[4];
  int mask = 7;
  if (selector)
  {
    if (parse_selector(selector, 1, cl, 4) != 1) return -1;
    mask = cl[0].mask;
  }
  const int want_in = mask & 1;
  const int want_if = !whole && mask == 2; /* phi = 0 on the host: a point (segment) / a straight segment (triangle) */
  if (!whole && !want_if && ((mask & 1) && (mask & 4))) return -1;
  const int tdim = mesh->tdim, hd = tdim - 1, nv = hd + 1;
  int nref; const double *pref, *wref;
  static const double one = 1.0, zero = 0.0;
  if (want_if && hd == 1) { nref = 1; pref = &zero; wref = &one; }
  else ref_rule(want_if ? 1 : hd, order, &nref, &pref, &wref);
  int64_t nr = 0, nq = 0;
  for (int pass = 0; pass < 2; ++pass)
  {
    if (pass == 1)
    {
      memset(out, 0, sizeof(*out));
      out->tdim = hd; out->nq = nq; out->nr = nr;
      out->points = (double*)malloc(sizeof(double) * (size_t)(nq * hd + 1));
      out->weights = (double*)malloc(sizeof(double) * (size_t)(nq + 1));
      out->offsets = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nr + 1));
      out->parent_map = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nr + 1));
      *rule_host = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nr + 1));
      out->offsets[0] = 0;
      nr = 0; nq = 0;
    }
    for (int64_t h = 0; h < n; ++h)
    {
      subtri s;
      int ns;
      if (whole)
      {
        if (!((mask >> (domain[h] + 1)) & 1)) continue;
        double none[3] = {-1.0, -1.0, -1.0};
        subtriangulate(hd, none, &s); /* the whole host as one "inside" simplex */
        ns = 1;
      }
      else
      {
        if (domain[h] != ORC_INTERSECTED) continue;
        double phi[3];
        for (int i = 0; i < nv; ++i) phi[i] = ls_values[ls[h * tdim + i]];
        subtriangulate(hd, phi, &s);
        ns = want_if ? s.n_if : (want_in ? s.n_in : s.n_out);
        if (ns == 0) continue;
      }
      if (pass == 0) { nr += 1; nq += (int64_t)ns * nref; continue; }
      double xv[3][3];
      for (int i = 0; i < nv; ++i)
        for (int d = 0; d < 3; ++d) xv[i][d] = mesh->x[3 * (int64_t)verts[h * tdim + i] + d];
      if (want_if)
      {
        if (hd == 1)
        {
          out->points[nq] = s.P[s.iface[0][0]][0];
          out->weights[nq] = 1.0;
        }
        else
        {
          double xp[2][3];
          for (int j = 0; j < 2; ++j)
          {
            const double* V = s.P[s.iface[0][j]];
            const double l0 = 1.0 - V[0] - V[1];
            for (int d = 0; d < 3; ++d) xp[j][d] = l0 * xv[0][d] + V[0] * xv[1][d] + V[1] * xv[2][d];
          }
          double len = 0.0;
          for (int d = 0; d < tdim; ++d) len += (xp[1][d] - xp[0][d]) * (xp[1][d] - xp[0][d]);
          len = sqrt(len);
          const double *V0 = s.P[s.iface[0][0]], *V1 = s.P[s.iface[0][1]];
          for (int q = 0; q < nref; ++q)
          {
            for (int d = 0; d < hd; ++d) out->points[(nq + q) * hd + d] = V0[d] + pref[q] * (V1[d] - V0[d]);
            out->weights[nq + q] = wref[q] * len;
          }
        }
        nq += nref;
        out->parent_map[nr] = ids ? ids[h] : (int32_t)h;
        (*rule_host)[nr] = (int32_t)h;
        out->offsets[nr + 1] = (int32_t)nq;
        ++nr;
        continue;
      }
      const double measure = host_measure(tdim, xv);
      for (int k = 0; k < ns; ++k)
      {
        const int* sx = (whole || want_in) ? s.in[k] : s.out[k];
        double V[4][3];
        for (int i = 0; i < nv; ++i)
          for (int d = 0; d < hd; ++d) V[i][d] = s.P[sx[i]][d];
        const double scale = fabs(det_sub(hd, V)) * measure;
        for (int q = 0; q < nref; ++q)
        {
          const double* xi = pref + hd * q;
          double l0 = 1.0;
          for (int t = 0; t < hd; ++t) l0 -= xi[t];
          for (int d = 0; d < hd; ++d)
          {
            double v = l0 * V[0][d];
            for (int t = 0; t < hd; ++t) v += xi[t] * V[t + 1][d];
            out->points[(nq + q) * hd + d] = v;
          }
          out->weights[nq + q] = wref[q] * scale;
        }
        nq += nref;
      }
      out->parent_map[nr] = ids ? ids[h] : (int32_t)h;
      (*rule_host)[nr] = (int32_t)h;
      out->offsets[nr + 1] = (int32_t)nq;
      ++nr;
    }
  }
  return 0;
}

/* full-cell rules: reference points + weights*|detJ|                        */
/* ref: python/tests/quadrature_utils.py:12-70                               */
int orc_full_cell_rules(const orc_mesh* mesh, const int32_t* cells, int64_t n,
                        int order, orc_rules* out)
{
  const int tdim = mesh->tdim;
  int nref; const double *pref, *wref;
  ref_rule(tdim, order, &nref, &pref, &wref);
  memset(out, 0, sizeof(*out));
  out->tdim = tdim; out->nr = n; out->nq = n * nref;
  out->points = (double*)malloc(sizeof(double) * (size_t)(out->nq * tdim + 1));
  out->weights = (double*)malloc(sizeof(double) * (size_t)(out->nq + 1));
  out->offsets = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
  out->parent_map = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
  out->offsets[0] = 0;
  for (int64_t i = 0; i < n; ++i)
  {
    double xc[MAXV][3], J[3][3], K[3][3];
    cell_coords(mesh, cells[i], xc);
    const double detJ = fabs(jacobian(tdim, xc, J, K));
    for (int q = 0; q < nref; ++q)
    {
      for (int d = 0; d < tdim; ++d) out->points[(i * nref + q) * tdim + d] = pref[q * tdim + d];
      out->weights[i * nref + q] = wref[q] * detJ;
    }
    out->parent_map[i] = cells[i];
    out->offsets[i + 1] = (int32_t)((i + 1) * nref);
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* a12 per-point level-set evaluators (P1 level set over the P1 geometry)    */
/* n = sign * K^T grad_ref(phi) / max(||.||, 1e-14)                          */
/* ref: cpp/cutfemx/level_set/normal.h:39-187 (floor :176-177), value.h      */
/* ------------------------------------------------------------------------ */
void orc_evaluate_normals(const orc_mesh* mesh, const int32_t* ls_dofmap,
                          const double* ls_values, const orc_rules* rules,
                          double sign, double* out)
{
  const int tdim = mesh->tdim, gdim = mesh->gdim, nv = tdim + 1;
  for (int64_t r = 0; r < rules->nr; ++r)
  {
    const int64_t c = rules->parent_map[r];
    double xc[MAXV][3], J[3][3], K[3][3], phi[4];
    cell_coords(mesh, c, xc);
    jacobian(tdim, xc, J, K);
    for (int i = 0; i < nv; ++i) phi[i] = ls_values[ls_dofmap[c * nv + i]];
    for (int32_t q = rules->offsets[r]; q < rules->offsets[r + 1]; ++q)
    {
      double N[MAXND], dN[MAXND][3], gref[3] = {0, 0, 0}, g[3] = {0, 0, 0};
      tabulate(tdim, 1, rules->points + (int64_t)q * tdim, N, dN);
      for (int t = 0; t < tdim; ++t)
        for (int j = 0; j < nv; ++j) gref[t] += dN[j][t] * phi[j];
      for (int i = 0; i < gdim; ++i)
        for (int t = 0; t < tdim; ++t) g[i] += K[t][i] * gref[t];
      double norm = 0.0;
      for (int i = 0; i < gdim; ++i) norm += g[i] * g[i];
      norm = sqrt(norm);
      if (norm < 1.0e-14) norm = 1.0e-14;
      for (int i = 0; i < gdim; ++i) out[(int64_t)q * gdim + i] = sign * g[i] / norm;
    }
  }
}

void orc_evaluate_values(const orc_mesh* mesh, const int32_t* ls_dofmap,
                         const double* ls_values, const orc_rules* rules,
                         double* out)
{
  const int tdim = mesh->tdim, nv = tdim + 1;
  for (int64_t r = 0; r < rules->nr; ++r)
  {
    const int64_t c = rules->parent_map[r];
    for (int32_t q = rules->offsets[r]; q < rules->offsets[r + 1]; ++q)
    {
      double N[MAXND], dN[MAXND][3], v = 0.0;
      tabulate(tdim, 1, rules->points + (int64_t)q * tdim, N, dN);
      for (int j = 0; j < nv; ++j) v += N[j] * ls_values[ls_dofmap[c * nv + j]];
      out[q] = v;
    }
  }
}

/* physical points, row-major (nq, gdim); ref: runtime_quadrature.h:102-221 */
void orc_physical_points(const orc_mesh* mesh, const orc_rules* rules, double* out)
{
  const int tdim = mesh->tdim, gdim = mesh->gdim;
  for (int64_t r = 0; r < rules->nr; ++r)
  {
    double xc[MAXV][3];
    cell_coords(mesh, rules->parent_map[r], xc);
    for (int32_t q = rules->offsets[r]; q < rules->offsets[r + 1]; ++q)
    {
      const double* X = rules->points + (int64_t)q * tdim;
      double l0 = 1.0;
      for (int t = 0; t < tdim; ++t) l0 -= X[t];
      for (int d = 0; d < gdim; ++d)
      {
        double v = l0 * xc[0][d];
        for (int t = 0; t < tdim; ++t) v += X[t] * xc[t + 1][d];
        out[(int64_t)q * gdim + d] = v;
      }
    }
  }
}

/* ------------------------------------------------------------------------ */
/* facet topology: facet lf of a simplex = vertices except lf (Basix)        */
/* ------------------------------------------------------------------------ */
typedef struct { int32_t v[3]; int32_t cell; int32_t lf; } facet_key;

static int facet_cmp(const void* a, const void* b)
{
  const facet_key* x = (const facet_key*)a; const facet_key* y = (const facet_key*)b;
  for (int i = 0; i < 3; ++i)
    if (x->v[i] != y->v[i]) return x->v[i] < y->v[i] ? -1 : 1;
  if (x->cell != y->cell) return x->cell < y->cell ? -1 : 1;
  return 0;
}

static int row4_cmp(const void* a, const void* b)
{
  const int32_t* x = (const int32_t*)a; const int32_t* y = (const int32_t*)b;
  for (int i = 0; i < 4; ++i)
    if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
  return 0;
}

/* all facets of the cells with flag[c]!=0, sorted so equal facets are adjacent */
static facet_key* build_facets(const orc_mesh* m, const uint8_t* flag, int64_t* nout)
{
  const int nv = m->tdim + 1;
  int64_t n = 0;
  for (int64_t c = 0; c < m->ncells; ++c) if (!flag || flag[c]) n += nv;
  facet_key* f = (facet_key*)malloc(sizeof(facet_key) * (size_t)(n + 1));
  int64_t k = 0;
  for (int64_t c = 0; c < m->ncells; ++c)
  {
    if (flag && !flag[c]) continue;
    for (int lf = 0; lf < nv; ++lf)
    {
      int32_t v[3] = {-1, -1, -1};
      int j = 0;
      for (int i = 0; i < nv; ++i) if (i != lf) v[j++] = m->conn[c * nv + i];
      /* sort the (<=3) vertices */
      for (int a = 0; a < j; ++a)
        for (int b = a + 1; b < j; ++b)
          if (v[b] < v[a]) { int32_t t = v[a]; v[a] = v[b]; v[b] = t; }
      f[k].v[0] = v[0]; f[k].v[1] = v[1]; f[k].v[2] = v[2];
      f[k].cell = (int32_t)c; f[k].lf = lf; ++k;
    }
  }
  qsort(f, (size_t)n, sizeof(facet_key), facet_cmp);
  *nout = n;
  return f;
}

/* interior facets whose two cells are both selected.
   ref: cpp/cutfemx/cut/cut.cpp:926-994 (+ rows of wrappers/cut.cpp:54-115) */
int64_t orc_interior_facets_for_cells(const orc_mesh* mesh, const int32_t* cells,
                                      int64_t ncells_sel, int32_t** rows_out)
{
  uint8_t* sel = (uint8_t*)calloc((size_t)mesh->ncells + 1, 1);
  for (int64_t i = 0; i < ncells_sel; ++i) sel[cells[i]] = 1;
  int64_t nf; facet_key* f = build_facets(mesh, NULL, &nf);
  int32_t* rows = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)(nf / 2 + 1));
  int64_t n = 0;
  for (int64_t i = 0; i + 1 < nf; ++i)
  {
    if (f[i].v[0] == f[i + 1].v[0] && f[i].v[1] == f[i + 1].v[1] && f[i].v[2] == f[i + 1].v[2])
    {
      if (sel[f[i].cell] && sel[f[i + 1].cell])
      {
        rows[4 * n + 0] = f[i].cell; rows[4 * n + 1] = f[i].lf;
        rows[4 * n + 2] = f[i + 1].cell; rows[4 * n + 3] = f[i + 1].lf; ++n;
      }
      ++i;
    }
  }
  qsort(rows, (size_t)n, 4 * sizeof(int32_t), row4_cmp);
  free(f); free(sel);
  *rows_out = rows;
  return n;
}

/* ghost-penalty band: interior facets of cut cells whose two cells both lie
   in cut U selected.  Order: by (smallest cut cell of the pair, its local
   facet) -- the reference orders by DOLFINx facet id, a numbering that does
   not exist outside DOLFINx.  ref: python/cutfemx/cut.py:340-380 */
typedef struct { int32_t e, elf, row[4]; } ghost_rec;

static int ghost_cmp(const void* a, const void* b)
{
  const ghost_rec* x = (const ghost_rec*)a; const ghost_rec* y = (const ghost_rec*)b;
  if (x->e != y->e) return x->e < y->e ? -1 : 1;
  if (x->elf != y->elf) return x->elf < y->elf ? -1 : 1;
  return 0;
}

int64_t orc_ghost_penalty_facets(const orc_mesh* mesh, const int8_t* domain,
                                 const char* selector, int32_t** rows_out)
{
  const int64_t nc = mesh->ncells;
  int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nc + 1));
  uint8_t* active = (uint8_t*)calloc((size_t)nc + 1, 1);
  int64_t ns = orc_locate_entities(nc, 1, domain, selector, tmp);
  if (ns < 0) { free(tmp); free(active); return -1; }
  for (int64_t i = 0; i < ns; ++i) active[tmp[i]] = 1;
  for (int64_t c = 0; c < nc; ++c) if (domain[c] == ORC_INTERSECTED) active[c] = 1;
  int64_t nf; facet_key* f = build_facets(mesh, active, &nf);
  ghost_rec* rec = (ghost_rec*)malloc(sizeof(ghost_rec) * (size_t)(nf / 2 + 1));
  int64_t n = 0;
  for (int64_t i = 0; i + 1 < nf; ++i)
  {
    if (f[i].v[0] == f[i + 1].v[0] && f[i].v[1] == f[i + 1].v[1] && f[i].v[2] == f[i + 1].v[2])
    {
      /* f[i].cell < f[i+1].cell by the sort */
      const int cut0 = domain[f[i].cell] == ORC_INTERSECTED;
      const int cut1 = domain[f[i + 1].cell] == ORC_INTERSECTED;
      if (cut0 || cut1)
      {
        rec[n].row[0] = f[i].cell; rec[n].row[1] = f[i].lf;
        rec[n].row[2] = f[i + 1].cell; rec[n].row[3] = f[i + 1].lf;
        if (cut0) { rec[n].e = f[i].cell; rec[n].elf = f[i].lf; }
        else { rec[n].e = f[i + 1].cell; rec[n].elf = f[i + 1].lf; }
        ++n;
      }
      ++i;
    }
  }
  qsort(rec, (size_t)n, sizeof(ghost_rec), ghost_cmp);
  int32_t* rows = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)(n + 1));
  for (int64_t i = 0; i < n; ++i) memcpy(rows + 4 * i, rec[i].row, 4 * sizeof(int32_t));
  free(rec); free(f); free(active); free(tmp);
  *rows_out = rows;
  return n;
}

/* ------------------------------------------------------------------------ */
/* analytic fields                                                           */
/* ------------------------------------------------------------------------ */
static double field_eval(int id, int gdim, const double* x)
{
  const double pi = 3.14159265358979323846;
  if (id == ORC_F_ONE) return 1.0;
  double p = 1.0;
  for (int d = 0; d < gdim; ++d) p *= sin(pi * x[d]);
  if (id == ORC_F_SINPROD) return p;
  return (double)gdim * pi * pi * p;
}

/* ------------------------------------------------------------------------ */
/* a6 element kernels.  One call = one entity, output Ae zero-initialised by */
/* the caller.  A cut entity integrates over its runtime rule (weights are   */
/* physical, no detJ factor); an uncut entity uses the reference rule of     */
/* degree qdegree times |detJ|.                                              */
/* ref: SURVEY 8a-a6; kernel ABI cpp/dolfinx_custom_data/fem/Form.h:59-75;   */
/* weak forms python/demo/demo_poisson.py:183-201                            */
/* ------------------------------------------------------------------------ */
static void cell_kernel(const orc_mesh* m, const orc_space* V, const orc_integral* I,
                        int64_t cell, int npts, const double* pts, const double* wts,
                        double wscale, const double* pdata, double* Ae)
{
  const int tdim = m->tdim, gdim = m->gdim;
  const int nd = V->ndofs_cell, bs = V->bs, nloc = nd * bs;
  double xc[MAXV][3], J[3][3], K[3][3];
  cell_coords(m, cell, xc);
  jacobian(tdim, xc, J, K);
  const double h = cell_diameter(tdim, xc);
  for (int q = 0; q < npts; ++q)
  {
    const double* X = pts + (int64_t)q * tdim;
    const double w = wts[q] * wscale;
    double N[MAXND], dN[MAXND][3], G[MAXND][3];
    tabulate(tdim, V->degree, X, N, dN);
    for (int i = 0; i < nd; ++i)
      for (int d = 0; d < gdim; ++d)
      {
        G[i][d] = 0.0;
        for (int t = 0; t < tdim; ++t) G[i][d] += K[t][d] * dN[i][t];
      }
    double xq[3] = {0, 0, 0};
    {
      double l0 = 1.0;
      for (int t = 0; t < tdim; ++t) l0 -= X[t];
      for (int d = 0; d < gdim; ++d)
      {
        xq[d] = l0 * xc[0][d];
        for (int t = 0; t < tdim; ++t) xq[d] += X[t] * xc[t + 1][d];
      }
    }
    /* a scalar coefficient of the form's element multiplies a bilinear integrand (pack_form.h:32-170) */
    double kw = w;
    if (I->kernel < 100 && I->coefficient)
    {
      double kappa = 0.0;
      for (int j = 0; j < nd; ++j) kappa += N[j] * I->coefficient[V->dofmap[cell * nd + j]];
      kw = w * kappa;
    }
    switch (I->kernel)
    {
    case ORC_K_MASS:
      for (int i = 0; i < nd; ++i)
        for (int j = 0; j < nd; ++j)
          for (int k = 0; k < bs; ++k)
            Ae[(i * bs + k) * nloc + j * bs + k] += kw * N[i] * N[j];
      break;
    case ORC_K_STIFFNESS:
      for (int i = 0; i < nd; ++i)
        for (int j = 0; j < nd; ++j)
        {
          double s = 0.0;
          for (int d = 0; d < gdim; ++d) s += G[i][d] * G[j][d];
          for (int k = 0; k < bs; ++k) Ae[(i * bs + k) * nloc + j * bs + k] += kw * s;
        }
      break;
    case ORC_K_NITSCHE:
    {
      const double* n = pdata + (int64_t)q * I->point_stride;
      const double gam = I->params[0] / h;
      double dn[MAXND];
      for (int i = 0; i < nd; ++i)
      {
        dn[i] = 0.0;
        for (int d = 0; d < gdim; ++d) dn[i] += G[i][d] * n[d];
      }
      for (int i = 0; i < nd; ++i)   /* test v = N_i */
        for (int j = 0; j < nd; ++j) /* trial u = N_j */
          Ae[i * nloc + j] += w * (-dn[j] * N[i] - dn[i] * N[j] + gam * N[j] * N[i]);
      break;
    }
    case ORC_K_ELASTICITY:
    {
      /* sigma(u):eps(v), sigma = 2 mu eps + lambda tr(eps) I
         ref: python/demo/demo_elasticity.py:167-238 */
      const double E = I->params[0], nu = I->params[1];
      const double mu = E / (2.0 * (1.0 + nu));
      const double lmbda = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu));
      for (int i = 0; i < nd; ++i)
        for (int a = 0; a < gdim; ++a)
          for (int j = 0; j < nd; ++j)
            for (int b = 0; b < gdim; ++b)
            {
              /* v = N_i e_a, u = N_j e_b */
              double gg = 0.0;
              for (int d = 0; d < gdim; ++d) gg += G[i][d] * G[j][d];
              double val = mu * ((a == b ? gg : 0.0) + G[i][b] * G[j][a])
                           + lmbda * G[i][a] * G[j][b];
              Ae[(i * bs + a) * nloc + j * bs + b] += kw * val;
            }
      break;
    }
    case ORC_L_SOURCE:
    {
      if (bs > 1) /* vector space: a vector-valued Function f, be[(i,a)] = int f_a N_i */
      {
        for (int a = 0; a < bs; ++a)
        {
          double f = 0.0;
          for (int j = 0; j < nd; ++j) f += N[j] * I->coefficient[(int64_t)V->dofmap[cell * nd + j] * bs + a];
          f *= I->params[1];
          for (int i = 0; i < nd; ++i) Ae[i * bs + a] += w * f * N[i];
        }
        break;
      }
      double f;
      if ((int)I->params[0] == ORC_F_COEFFICIENT)
      {
        f = 0.0;
        for (int j = 0; j < nd; ++j) f += N[j] * I->coefficient[V->dofmap[cell * nd + j]];
        f *= I->params[1];
      }
      else
        f = I->params[1] * field_eval((int)I->params[0], gdim, xq);
      for (int i = 0; i < nd; ++i) Ae[i] += w * f * N[i];
      break;
    }
    case ORC_L_NITSCHE_RHS:
    {
      const double* n = pdata + (int64_t)q * I->point_stride;
      const double gam = I->params[0] / h;
      const double g = I->params[2] * field_eval((int)I->params[1], gdim, xq);
      for (int i = 0; i < nd; ++i)
      {
        double dn = 0.0;
        for (int d = 0; d < gdim; ++d) dn += G[i][d] * n[d];
        Ae[i] += w * (-dn * g + gam * g * N[i]);
      }
      break;
    }
    default: break;
    }
  }
}

/* interior-facet kernel: gamma_g h_avg [grad u . n][grad v . n] on facet    */
/* Ae is (2 nd)^2 with block layout [[00,01],[10,11]]                        */
/* ref: assemble_matrix_impl.h:537-542; python/demo/demo_poisson.py:189-198 */
static void facet_kernel(const orc_mesh* m, const orc_space* V, const orc_integral* I,
                         const int32_t* row, int64_t idx, double* Ae)
{
  const int tdim = m->tdim, gdim = m->gdim, nv = tdim + 1;
  const int nd = V->ndofs_cell, bs = V->bs, nloc = 2 * nd * bs;
  const int64_t c0 = row[0], c1 = row[2];
  const int lf0 = row[1];
  double x0[MAXV][3], x1[MAXV][3], J0[3][3], K0[3][3], J1[3][3], K1[3][3];
  cell_coords(m, c0, x0); cell_coords(m, c1, x1);
  const double det0 = jacobian(tdim, x0, J0, K0); jacobian(tdim, x1, J1, K1);
  if (I->kernel == ORC_K_EXTENSION_L2)
  {
    /* extension penalty pair block (extension_penalty.cpp:191-369, :395-470): full-cell rule of
       the bad cell c0; the root cell's basis is evaluated at the pulled-back points (outside
       its reference simplex: the polynomial extension); macro basis [N_bad, -N_root] */
    int nref; const double *pref, *wref;
    ref_rule(tdim, I->qdegree, &nref, &pref, &wref);
    for (int q = 0; q < nref; ++q)
    {
      const double* X0 = pref + tdim * q;
      double l0 = 1.0, xq[3] = {0, 0, 0}, X1[3] = {0, 0, 0};
      for (int t = 0; t < tdim; ++t) l0 -= X0[t];
      for (int d = 0; d < gdim; ++d)
      {
        xq[d] = l0 * x0[0][d];
        for (int t = 0; t < tdim; ++t) xq[d] += X0[t] * x0[t + 1][d];
      }
      for (int t = 0; t < tdim; ++t)
        for (int d = 0; d < gdim; ++d) X1[t] += K1[t][d] * (xq[d] - x1[0][d]);
      double N0[MAXND], dN0[MAXND][3], N1[MAXND], dN1[MAXND][3], M[2 * MAXND];
      tabulate(tdim, V->degree, X0, N0, dN0);
      tabulate(tdim, V->degree, X1, N1, dN1);
      for (int i = 0; i < nd; ++i) { M[i] = N0[i]; M[nd + i] = -N1[i]; }
      const double w = wref[q] * fabs(det0) * I->params[0];
      for (int i = 0; i < 2 * nd; ++i)
        for (int j = 0; j < 2 * nd; ++j)
          for (int kk = 0; kk < bs; ++kk)
            Ae[(i * bs + kk) * nloc + j * bs + kk] += w * M[i] * M[j];
    }
    return;
  }
  const double havg = 0.5 * (cell_diameter(tdim, x0) + cell_diameter(tdim, x1));
  /* outward normal of cell0 on facet lf0: -grad(lambda_lf0)/|.| */
  double n[3] = {0, 0, 0};
  {
    double dl[3];
    for (int t = 0; t < tdim; ++t) dl[t] = (lf0 == 0) ? -1.0 : ((lf0 - 1 == t) ? 1.0 : 0.0);
    double nn = 0.0;
    for (int d = 0; d < gdim; ++d)
    {
      for (int t = 0; t < tdim; ++t) n[d] -= K0[t][d] * dl[t];
      nn += n[d] * n[d];
    }
    nn = sqrt(nn);
    for (int d = 0; d < gdim; ++d) n[d] /= nn;
  }
  /* physical facet vertices (from cell0) */
  double xf[3][3]; int k = 0;
  for (int i = 0; i < nv; ++i)
    if (i != lf0) { for (int d = 0; d < 3; ++d) xf[k][d] = x0[i][d]; ++k; }
  double scale;
  if (tdim == 2)
  {
    double dx = xf[1][0] - xf[0][0], dy = xf[1][1] - xf[0][1];
    scale = sqrt(dx * dx + dy * dy);
  }
  else
  {
    double a[3], b[3];
    for (int d = 0; d < 3; ++d) { a[d] = xf[1][d] - xf[0][d]; b[d] = xf[2][d] - xf[0][d]; }
    double cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2],
           cz = a[0] * b[1] - a[1] * b[0];
    scale = sqrt(cx * cx + cy * cy + cz * cz);
  }
  int nref; const double *pref, *wref;
  ref_rule(tdim - 1, I->qdegree, &nref, &pref, &wref);
  if (I->host_verts && idx >= I->n_std)
  {
    /* facet-hosted runtime rule (8f-4): points on the simplex of the host vertices, physical weights */
    const orc_rules* R = I->rules;
    const int64_t r = idx - I->n_std;
    nref = R->offsets[r + 1] - R->offsets[r];
    pref = R->points + (int64_t)R->offsets[r] * (tdim - 1);
    wref = R->weights + R->offsets[r];
    scale = 1.0;
    for (int j = 0; j < tdim; ++j)
      for (int d = 0; d < 3; ++d) xf[j][d] = m->x[3 * (int64_t)I->host_verts[r * tdim + j] + d];
  }
  for (int q = 0; q < nref; ++q)
  {
    const double* xi = pref + (tdim - 1) * q;
    double l0 = 1.0, xq[3] = {0, 0, 0};
    for (int t = 0; t < tdim - 1; ++t) l0 -= xi[t];
    for (int d = 0; d < gdim; ++d)
    {
      xq[d] = l0 * xf[0][d];
      for (int t = 0; t < tdim - 1; ++t) xq[d] += xi[t] * xf[t + 1][d];
    }
    double X0[3], X1[3];
    for (int t = 0; t < tdim; ++t)
    {
      X0[t] = 0.0; X1[t] = 0.0;
      for (int d = 0; d < gdim; ++d)
      {
        X0[t] += K0[t][d] * (xq[d] - x0[0][d]);
        X1[t] += K1[t][d] * (xq[d] - x1[0][d]);
      }
    }
    double N0[MAXND], dN0[MAXND][3], N1[MAXND], dN1[MAXND][3];
    tabulate(tdim, V->degree, X0, N0, dN0);
    tabulate(tdim, V->degree, X1, N1, dN1);
    /* gamma h_avg^(1 + params[1]): params[1] = 0 is the velocity-type term avg(h) [dn u][dn v], 2 the pressure-type
       term avg(h)^3 [dn p][dn q] of python/tests/test_assembly_stokes.py:123-131 */
    const double w = wref[q] * scale * I->params[0] * havg * (I->params[1] != 0.0 ? pow(havg, I->params[1]) : 1.0);
    if (I->kernel == ORC_K_GHOST_GRADJUMP)
    {
      double jn[2 * MAXND]; /* normal-derivative jump of each macro basis fn */
      for (int i = 0; i < nd; ++i)
      {
        double a = 0.0, b = 0.0;
        for (int d = 0; d < gdim; ++d)
          for (int t = 0; t < tdim; ++t)
          {
            a += K0[t][d] * dN0[i][t] * n[d];
            b += K1[t][d] * dN1[i][t] * n[d];
          }
        jn[i] = a; jn[nd + i] = -b;
      }
      for (int i = 0; i < 2 * nd; ++i)
        for (int j = 0; j < 2 * nd; ++j)
          for (int kk = 0; kk < bs; ++kk)
            Ae[(i * bs + kk) * nloc + j * bs + kk] += w * jn[i] * jn[j];
    }
    else if (I->kernel == ORC_K_SIP)
    {
      /* symmetric interior penalty, python/demo/demo_dg_poisson.py:262-265:
         -<avg(grad u), jump(v, n)> - <avg(grad v), jump(u, n)> + sigma / h_avg <jump(u, n), jump(v, n)>
         with jump(w, n) = w+ n+ + w- n- = (w+ - w-) n+ and avg(grad w) = (grad w+ + grad w-) / 2 */
      double jv[2 * MAXND], an[2 * MAXND];
      for (int i = 0; i < nd; ++i)
      {
        double a = 0.0, b = 0.0;
        for (int d = 0; d < gdim; ++d)
          for (int t = 0; t < tdim; ++t)
          {
            a += K0[t][d] * dN0[i][t] * n[d];
            b += K1[t][d] * dN1[i][t] * n[d];
          }
        jv[i] = N0[i]; jv[nd + i] = -N1[i];
        an[i] = 0.5 * a; an[nd + i] = 0.5 * b;
      }
      const double wq = wref[q] * scale, pen = I->params[0] / havg;
      for (int i = 0; i < 2 * nd; ++i)
        for (int j = 0; j < 2 * nd; ++j)
          for (int kk = 0; kk < bs; ++kk)
            Ae[(i * bs + kk) * nloc + j * bs + kk] += wq * (-an[j] * jv[i] - an[i] * jv[j] + pen * jv[i] * jv[j]);
    }
    else if (I->kernel == ORC_K_JUMP)
    {
      /* gamma / h_avg [u][v] */
      double jv[2 * MAXND];
      for (int i = 0; i < nd; ++i) { jv[i] = N0[i]; jv[nd + i] = -N1[i]; }
      const double wj = wref[q] * scale * I->params[0] / havg;
      for (int i = 0; i < 2 * nd; ++i)
        for (int j = 0; j < 2 * nd; ++j)
          for (int kk = 0; kk < bs; ++kk)
            Ae[(i * bs + kk) * nloc + j * bs + kk] += wj * jv[i] * jv[j];
    }
  }
}

static void standard_rule(const orc_mesh* m, const orc_integral* I, int* n,
                          const double** p, const double** w)
{
  ref_rule(m->tdim, I->qdegree, n, p, w);
}

int orc_tabulate_entity(const orc_mesh* mesh, const orc_space* V,
                        const orc_integral* I, int64_t idx, int use_rule, double* Ae)
{
  if (I->type == ORC_INTERIOR_FACET)
  {
    facet_kernel(mesh, V, I, I->entities + 4 * idx, idx, Ae);
    if (I->kernel == ORC_K_EXTENSION_L2 && I->point_data)
    {
      /* cellwise (DG0) beta: beta_cell_values[bad_cell] gathered per pair by the caller */
      const int nloc = 2 * V->ndofs_cell * V->bs;
      for (int i = 0; i < nloc * nloc; ++i) Ae[i] *= I->point_data[idx];
    }
    return 0;
  }
  if (use_rule)
  {
    const orc_rules* R = I->rules;
    const int32_t q0 = R->offsets[idx], q1 = R->offsets[idx + 1];
    cell_kernel(mesh, V, I, R->parent_map[idx], q1 - q0, R->points + (int64_t)q0 * R->tdim,
                R->weights + q0, 1.0,
                I->point_data ? I->point_data + (int64_t)q0 * I->point_stride : NULL, Ae);
    return 0;
  }
  int n; const double *p, *w;
  standard_rule(mesh, I, &n, &p, &w);
  double xc[MAXV][3], J[3][3], K[3][3];
  cell_coords(mesh, I->entities[idx], xc);
  const double detJ = fabs(jacobian(mesh->tdim, xc, J, K));
  cell_kernel(mesh, V, I, I->entities[idx], n, p, w, detJ, NULL, Ae);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* entity dof lists                                                          */
/* ------------------------------------------------------------------------ */
static int entity_dofs(const orc_space* V, const orc_integral* I, int64_t idx,
                       int use_rule, int32_t* dofs)
{
  const int nd = V->ndofs_cell, bs = V->bs;
  int n = 0;
  if (I->type == ORC_INTERIOR_FACET)
  {
    const int32_t* row = I->entities + 4 * idx;
    for (int s = 0; s < 2; ++s)
      for (int i = 0; i < nd; ++i)
        for (int k = 0; k < bs; ++k)
          dofs[n++] = bs * V->dofmap[(int64_t)row[2 * s] * nd + i] + k;
    return n;
  }
  const int64_t c = use_rule ? I->rules->parent_map[idx] : I->entities[idx];
  for (int i = 0; i < nd; ++i)
    for (int k = 0; k < bs; ++k) dofs[n++] = bs * V->dofmap[c * nd + i] + k;
  return n;
}

/* ------------------------------------------------------------------------ */
/* a9 sparsity: union over integrals of rows(entity) x cols(entity), plus a  */
/* diagonal entry for every row.  ref: assembler.h:442-529, :538-560, :567   */
/* ------------------------------------------------------------------------ */
static int i32_cmp(const void* a, const void* b)
{
  int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return x < y ? -1 : (x > y);
}

int orc_create_sparsity(const orc_mesh* mesh, const orc_space* V,
                        const orc_integral* integrals, int n_integrals,
                        int64_t** indptr_out, int32_t** indices_out)
{
  (void)mesh;
  const int64_t nrows = V->ndofs * V->bs;
  int64_t* cnt = (int64_t*)calloc((size_t)nrows + 1, sizeof(int64_t));
  int32_t dofs[MAXLOC];
  for (int pass = 0; pass < 2; ++pass)
  {
    int64_t* ptr = NULL; int32_t* cols = NULL; int64_t* fill = NULL;
    if (pass == 1)
    {
      ptr = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nrows + 1));
      ptr[0] = 0;
      /* the all-rows diagonal is a bs x bs BLOCK: DOLFINx patterns are in block-index space
         (insert_diagonal on the blocked index map, assembler.h:538-560) */
      for (int64_t r = 0; r < nrows; ++r) ptr[r + 1] = ptr[r] + cnt[r] + V->bs;
      cols = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ptr[nrows] + 1));
      fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nrows + 1));
      for (int64_t r = 0; r < nrows; ++r)
      {
        const int64_t dof = r / V->bs;
        for (int b = 0; b < V->bs; ++b) cols[ptr[r] + b] = (int32_t)(dof * V->bs + b);
        fill[r] = ptr[r] + V->bs;
      }
    }
    for (int ii = 0; ii < n_integrals; ++ii)
    {
      const orc_integral* I = &integrals[ii];
      for (int part = 0; part < 2; ++part)
      {
        const int64_t ne = part == 0 ? I->n_entities : ((I->rules && I->type == ORC_CELL) ? I->rules->nr : 0);
        for (int64_t e = 0; e < ne; ++e)
        {
          const int n = entity_dofs(V, I, e, part, dofs);
          for (int i = 0; i < n; ++i)
          {
            if (pass == 0) cnt[dofs[i]] += n;
            else
              for (int j = 0; j < n; ++j) cols[fill[dofs[i]]++] = dofs[j];
          }
        }
      }
    }
    if (pass == 1)
    {
      /* sort + unique each row, then compact */
      int64_t* indptr = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nrows + 1));
      int64_t nnz = 0;
      indptr[0] = 0;
      for (int64_t r = 0; r < nrows; ++r)
      {
        int32_t* row = cols + ptr[r];
        int64_t len = fill[r] - ptr[r];
        qsort(row, (size_t)len, sizeof(int32_t), i32_cmp);
        int64_t u = 0;
        for (int64_t k = 0; k < len; ++k)
          if (k == 0 || row[k] != row[k - 1]) cols[nnz + u++] = row[k];
        nnz += u;
        indptr[r + 1] = nnz;
      }
      int32_t* indices = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
      memcpy(indices, cols, sizeof(int32_t) * (size_t)nnz);
      free(cols); free(ptr); free(fill);
      *indptr_out = indptr; *indices_out = indices;
    }
  }
  free(cnt);
  return 0;
}

/* DOLFINx MatrixCSR::add behaviour: per row, search the sorted columns.     */
static int mat_add(const int64_t* indptr, const int32_t* indices, double* values,
                   int n, const int32_t* rows, const int32_t* cols, const double* Ae)
{
  for (int i = 0; i < n; ++i)
  {
    const int64_t b = indptr[rows[i]], e = indptr[rows[i] + 1];
    for (int j = 0; j < n; ++j)
    {
      int64_t lo = b, hi = e;
      while (lo < hi)
      {
        int64_t mid = (lo + hi) / 2;
        if (indices[mid] < cols[j]) lo = mid + 1; else hi = mid;
      }
      if (lo == e || indices[lo] != cols[j]) return -1;
      values[lo] += Ae[i * n + j];
    }
  }
  return 0;
}

/* a5/a7 matrix loops.  Standard entities first, then the runtime (cut)      */
/* entities of the same integral.                                            */
/* ref: assemble_matrix_impl.h:103-188 (cells), :462-606 (interior facets)   */
int orc_assemble_matrix(const orc_mesh* mesh, const orc_space* V,
                        const orc_integral* integrals, int n_integrals,
                        const int8_t* bc0, const int8_t* bc1,
                        const int64_t* indptr, const int32_t* indices, double* values)
{
  double Ae[MAXLOC * MAXLOC];
  int32_t dofs[MAXLOC];
  for (int ii = 0; ii < n_integrals; ++ii)
  {
    const orc_integral* I = &integrals[ii];
    for (int part = 0; part < 2; ++part)
    {
      const int64_t ne = part == 0 ? I->n_entities : ((I->rules && I->type == ORC_CELL) ? I->rules->nr : 0);
      for (int64_t e = 0; e < ne; ++e)
      {
        const int n = entity_dofs(V, I, e, part, dofs);
        memset(Ae, 0, sizeof(double) * (size_t)(n * n));
        orc_tabulate_entity(mesh, V, I, e, part, Ae);
        if (bc0)
          for (int i = 0; i < n; ++i)
            if (bc0[dofs[i]]) for (int j = 0; j < n; ++j) Ae[i * n + j] = 0.0;
        if (bc1)
          for (int j = 0; j < n; ++j)
            if (bc1[dofs[j]]) for (int i = 0; i < n; ++i) Ae[i * n + j] = 0.0;
        if (mat_add(indptr, indices, values, n, dofs, dofs, Ae) != 0) return -1;
      }
    }
  }
  return 0;
}

/* a8 vector loops.  ref: assemble_vector_impl.h:62-122 */
int orc_assemble_vector(const orc_mesh* mesh, const orc_space* V,
                        const orc_integral* integrals, int n_integrals, double* b)
{
  double be[MAXLOC];
  int32_t dofs[MAXLOC];
  for (int ii = 0; ii < n_integrals; ++ii)
  {
    const orc_integral* I = &integrals[ii];
    for (int part = 0; part < 2; ++part)
    {
      const int64_t ne = part == 0 ? I->n_entities : ((I->rules && I->type == ORC_CELL) ? I->rules->nr : 0);
      for (int64_t e = 0; e < ne; ++e)
      {
        const int n = entity_dofs(V, I, e, part, dofs);
        memset(be, 0, sizeof(double) * (size_t)n);
        orc_tabulate_entity(mesh, V, I, e, part, be);
        for (int i = 0; i < n; ++i) b[dofs[i]] += be[i];
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* 8f-3 cell aggregation.  ref: cell_aggregation.cpp:143-270                  */
/* ------------------------------------------------------------------------ */
int64_t orc_cell_aggregation(const orc_mesh* mesh, const int8_t* domain, int relation,
                             const double* fraction, double threshold, int policy,
                             int max_iterations, int32_t* root_cell, int32_t* aggregate_id,
                             int32_t* depth)
{
  const int64_t nc = mesh->ncells;
  const int nv = mesh->tdim + 1;
  /* facet neighbours of every cell, ascending (cell_neighbors, :73-110) */
  int32_t* nb = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nc * nv + 1));
  for (int64_t i = 0; i < nc * nv; ++i) nb[i] = -1;
  {
    int64_t nf; facet_key* f = build_facets(mesh, NULL, &nf);
    for (int64_t i = 0; i + 1 < nf; ++i)
      if (f[i].v[0] == f[i + 1].v[0] && f[i].v[1] == f[i + 1].v[1] && f[i].v[2] == f[i + 1].v[2])
      {
        const int32_t a = f[i].cell, b = f[i + 1].cell;
        for (int k = 0; k < nv; ++k) if (nb[a * nv + k] < 0) { nb[a * nv + k] = b; break; }
        for (int k = 0; k < nv; ++k) if (nb[b * nv + k] < 0) { nb[b * nv + k] = a; break; }
        ++i;
      }
    free(f);
    for (int64_t c = 0; c < nc; ++c)   /* sort ascending, -1 (none) last */
      for (int a = 0; a < nv; ++a)
        for (int b = a + 1; b < nv; ++b)
        {
          int32_t* p = nb + c * nv;
          if (p[b] >= 0 && (p[a] < 0 || p[b] < p[a])) { int32_t t = p[a]; p[a] = p[b]; p[b] = t; }
        }
  }
  const int8_t sel = (int8_t)(relation < 0 ? -1 : 1); /* classification code of the strict selector */
  int32_t next = 0;
  for (int64_t c = 0; c < nc; ++c)
  {
    root_cell[c] = aggregate_id[c] = depth[c] = -1;
    const int interior = domain[c] == sel;
    const int well_cut = domain[c] == 0 && policy == 1 && fraction[c] >= threshold;
    if (interior || well_cut) { root_cell[c] = (int32_t)c; aggregate_id[c] = next++; depth[c] = 0; }
  }
  const int64_t limit = max_iterations < 0 ? nc : max_iterations;
  for (int64_t it = 0; it < limit; ++it)
  {
    int64_t mapped = 0;
    for (int64_t c = 0; c < nc; ++c)
    {
      if (domain[c] != 0 || root_cell[c] >= 0) continue;       /* ill-posed = unrooted cut cells */
      for (int k = 0; k < nv; ++k)
      {
        const int32_t o = nb[c * nv + k];
        if (o < 0) break;
        if (!(domain[o] == sel || domain[o] == 0)) continue;      /* active cells only */
        if (root_cell[o] < 0) continue;
        root_cell[c] = root_cell[o]; aggregate_id[c] = aggregate_id[o]; depth[c] = depth[o] + 1;
        ++mapped;
        break;
      }
    }
    if (mapped == 0) break;
  }
  int64_t rootless = 0;
  for (int64_t c = 0; c < nc; ++c) if (domain[c] == 0 && root_cell[c] < 0) ++rootless;
  free(nb);
  return rootless;
}

/* Dirichlet lifting: b -= alpha Ae (g - x0) over the marked columns, only on      */
/* entities that have one.  ref: assemble_vector_impl.h:383-436 (lifting_fn),     */
/* assemble_matrix_impl.h LiftingMode                                            */
int orc_apply_lifting(const orc_mesh* mesh, const orc_space* V,
                      const orc_integral* integrals, int n_integrals,
                      const int8_t* markers, const double* g, const double* x0,
                      double alpha, double* b)
{
  double Ae[MAXLOC * MAXLOC];
  int32_t dofs[MAXLOC];
  for (int ii = 0; ii < n_integrals; ++ii)
  {
    const orc_integral* I = &integrals[ii];
    for (int part = 0; part < 2; ++part)
    {
      const int64_t ne = part == 0 ? I->n_entities : ((I->rules && I->type == ORC_CELL) ? I->rules->nr : 0);
      for (int64_t e = 0; e < ne; ++e)
      {
        const int n = entity_dofs(V, I, e, part, dofs);
        int any = 0;
        for (int j = 0; j < n; ++j) any |= markers[dofs[j]] != 0;
        if (!any) continue;
        memset(Ae, 0, sizeof(double) * (size_t)(n * n));
        orc_tabulate_entity(mesh, V, I, e, part, Ae);
        for (int j = 0; j < n; ++j)
          if (markers[dofs[j]])
          {
            const double d = alpha * (g[dofs[j]] - (x0 ? x0[dofs[j]] : 0.0));
            for (int i = 0; i < n; ++i) b[dofs[i]] -= Ae[i * n + j] * d;
          }
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* a11 active domain / deactivation                                          */
/* ref: cpp/cutfemx/fem/deactivate.h:103-162 (cells), :164-183 (indicator),  */
/* :47-64 (inactive dofs), :402-418 (diag=1, rhs=0)                          */
/* ------------------------------------------------------------------------ */
int64_t orc_active_cells(const orc_integral* integrals, int n_integrals,
                         int64_t ncells, int32_t* out)
{
  uint8_t* flag = (uint8_t*)calloc((size_t)ncells + 1, 1);
  for (int ii = 0; ii < n_integrals; ++ii)
  {
    const orc_integral* I = &integrals[ii];
    if (I->type == ORC_INTERIOR_FACET)
      for (int64_t e = 0; e < I->n_entities; ++e)
      {
        flag[I->entities[4 * e]] = 1; flag[I->entities[4 * e + 2]] = 1;
      }
    else
      for (int64_t e = 0; e < I->n_entities; ++e) flag[I->entities[e]] = 1;
    if (I->rules && I->type == ORC_CELL)
      for (int64_t r = 0; r < I->rules->nr; ++r) flag[I->rules->parent_map[r]] = 1;
  }
  int64_t n = 0;
  for (int64_t c = 0; c < ncells; ++c) if (flag[c]) out[n++] = (int32_t)c;
  free(flag);
  return n;
}

int64_t orc_inactive_dofs(const orc_space* V, const int32_t* active_cells,
                          int64_t n_active, int32_t* out)
{
  const int64_t nrows = V->ndofs * V->bs;
  uint8_t* ind = (uint8_t*)calloc((size_t)nrows + 1, 1);
  for (int64_t i = 0; i < n_active; ++i)
    for (int j = 0; j < V->ndofs_cell; ++j)
      for (int k = 0; k < V->bs; ++k)
        ind[(int64_t)V->bs * V->dofmap[(int64_t)active_cells[i] * V->ndofs_cell + j] + k] = 1;
  int64_t n = 0;
  for (int64_t r = 0; r < nrows; ++r) if (!ind[r]) out[n++] = (int32_t)r;
  free(ind);
  return n;
}

void orc_deactivate(const int32_t* inactive, int64_t n, int bs_unused,
                    const int64_t* indptr, const int32_t* indices,
                    double* values, double* b, double diagonal, double rhs_value)
{
  (void)bs_unused;
  for (int64_t i = 0; i < n; ++i)
  {
    const int32_t r = inactive[i];
    for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k)
      if (indices[k] == r) values[k] = diagonal;
    if (b) b[r] = rhs_value;
  }
}

/* ------------------------------------------------------------------------ */
/* rectangular bilinear forms: test space V0 (rows), trial space V1 (cols).  */
/* ref: assemble_matrix_impl.h:68-189 (dofmap0 / bs0 and dofmap1 / bs1 are   */
/* separate arguments of the cell loop), assembler.h:442-529 (sparsity from  */
/* both dofmaps), :537-560 (no diagonal unless the index maps coincide).     */
/* ------------------------------------------------------------------------ */
static void cell_kernel2(const orc_mesh* m, const orc_space* V0, const orc_space* V1, const orc_integral* I,
                         int64_t cell, int npts, const double* pts, const double* wts, double wscale, double* Ae)
{
  const int tdim = m->tdim, gdim = m->gdim;
  const int nd0 = V0->ndofs_cell, bs0 = V0->bs, nd1 = V1->ndofs_cell, bs1 = V1->bs, n1 = nd1 * bs1;
  double xc[MAXV][3], J[3][3], K[3][3];
  cell_coords(m, cell, xc);
  jacobian(tdim, xc, J, K);
  for (int q = 0; q < npts; ++q)
  {
    const double* X = pts + (int64_t)q * tdim;
    const double w = wts[q] * wscale;
    double N0[MAXND], dN0[MAXND][3], G0[MAXND][3], N1[MAXND], dN1[MAXND][3], G1[MAXND][3];
    tabulate(tdim, V0->degree, X, N0, dN0);
    tabulate(tdim, V1->degree, X, N1, dN1);
    for (int d = 0; d < gdim; ++d)
    {
      for (int i = 0; i < nd0; ++i)
      {
        G0[i][d] = 0.0;
        for (int t = 0; t < tdim; ++t) G0[i][d] += K[t][d] * dN0[i][t];
      }
      for (int j = 0; j < nd1; ++j)
      {
        G1[j][d] = 0.0;
        for (int t = 0; t < tdim; ++t) G1[j][d] += K[t][d] * dN1[j][t];
      }
    }
    switch (I->kernel)
    {
    case ORC_K_MASS:
      for (int i = 0; i < nd0; ++i)
        for (int j = 0; j < nd1; ++j)
          for (int k = 0; k < bs0; ++k) Ae[(i * bs0 + k) * n1 + j * bs1 + k] += w * N0[i] * N1[j];
      break;
    case ORC_K_STIFFNESS:
      for (int i = 0; i < nd0; ++i)
        for (int j = 0; j < nd1; ++j)
        {
          double s = 0.0;
          for (int d = 0; d < gdim; ++d) s += G0[i][d] * G1[j][d];
          for (int k = 0; k < bs0; ++k) Ae[(i * bs0 + k) * n1 + j * bs1 + k] += w * s;
        }
      break;
    case ORC_K_DIV_TEST: /* v = N0_i e_a, p = N1_j */
      for (int i = 0; i < nd0; ++i)
        for (int a = 0; a < bs0; ++a)
          for (int j = 0; j < nd1; ++j) Ae[(i * bs0 + a) * n1 + j] += w * I->params[0] * G0[i][a] * N1[j];
      break;
    case ORC_K_DIV_TRIAL: /* q = N0_i, u = N1_j e_b */
      for (int i = 0; i < nd0; ++i)
        for (int j = 0; j < nd1; ++j)
          for (int b = 0; b < bs1; ++b) Ae[i * n1 + j * bs1 + b] += w * I->params[0] * N0[i] * G1[j][b];
      break;
    default: break;
    }
  }
}

/* interior-facet integral between two (scalar) spaces: macro test dofs = [cell0, cell1] of V0, macro trial dofs =
   [cell0, cell1] of V1 (assemble_matrix_impl.h:462-606: dmapjoint0 / dmapjoint1 are built from dofmap0 and dofmap1
   separately); Ae is (2 nd0) x (2 nd1).  gamma h_avg^(1 + p) [dn u][dn v] and gamma / h_avg [u][v], standard facet rule. */
static int facet_kernel2(const orc_mesh* m, const orc_space* V0, const orc_space* V1, const orc_integral* I,
                         const int32_t* row, double* Ae)
{
  const int tdim = m->tdim, gdim = m->gdim, nv = tdim + 1;
  const int nd0 = V0->ndofs_cell, nd1 = V1->ndofs_cell, n1 = 2 * nd1;
  if (V0->bs != 1 || V1->bs != 1) return -2;
  if (I->kernel != ORC_K_GHOST_GRADJUMP && I->kernel != ORC_K_JUMP) return -2;
  const int64_t c0 = row[0], c1 = row[2];
  const int lf0 = row[1];
  double x0[MAXV][3], x1[MAXV][3], J0[3][3], K0[3][3], J1[3][3], K1[3][3];
  cell_coords(m, c0, x0); cell_coords(m, c1, x1);
  jacobian(tdim, x0, J0, K0); jacobian(tdim, x1, J1, K1);
  const double havg = 0.5 * (cell_diameter(tdim, x0) + cell_diameter(tdim, x1));
  double n[3] = {0, 0, 0};
  {
    double dl[3];
    for (int t = 0; t < tdim; ++t) dl[t] = (lf0 == 0) ? -1.0 : ((lf0 - 1 == t) ? 1.0 : 0.0);
    double nn = 0.0;
    for (int d = 0; d < gdim; ++d)
    {
      for (int t = 0; t < tdim; ++t) n[d] -= K0[t][d] * dl[t];
      nn += n[d] * n[d];
    }
    nn = sqrt(nn);
    for (int d = 0; d < gdim; ++d) n[d] /= nn;
  }
  double xf[3][3]; int k = 0;
  for (int i = 0; i < nv; ++i)
    if (i != lf0) { for (int d = 0; d < 3; ++d) xf[k][d] = x0[i][d]; ++k; }
  double scale;
  if (tdim == 2)
  {
    double dx = xf[1][0] - xf[0][0], dy = xf[1][1] - xf[0][1];
    scale = sqrt(dx * dx + dy * dy);
  }
  else
  {
    double a[3], b[3];
    for (int d = 0; d < 3; ++d) { a[d] = xf[1][d] - xf[0][d]; b[d] = xf[2][d] - xf[0][d]; }
    double cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
    scale = sqrt(cx * cx + cy * cy + cz * cz);
  }
  int nref; const double *pref, *wref;
  ref_rule(tdim - 1, I->qdegree, &nref, &pref, &wref);
  for (int q = 0; q < nref; ++q)
  {
    const double* xi = pref + (tdim - 1) * q;
    double l0 = 1.0, xq[3] = {0, 0, 0};
    for (int t = 0; t < tdim - 1; ++t) l0 -= xi[t];
    for (int d = 0; d < gdim; ++d)
    {
      xq[d] = l0 * xf[0][d];
      for (int t = 0; t < tdim - 1; ++t) xq[d] += xi[t] * xf[t + 1][d];
    }
    double X0[3], X1[3];
    for (int t = 0; t < tdim; ++t)
    {
      X0[t] = 0.0; X1[t] = 0.0;
      for (int d = 0; d < gdim; ++d)
      {
        X0[t] += K0[t][d] * (xq[d] - x0[0][d]);
        X1[t] += K1[t][d] * (xq[d] - x1[0][d]);
      }
    }
    double jt[2 * MAXND], ju[2 * MAXND]; /* the jump of every macro test / trial basis function */
    for (int side = 0; side < 2; ++side)
    {
      const orc_space* V = side == 0 ? V0 : V1;
      const int nd = V->ndofs_cell;
      double* jj = side == 0 ? jt : ju;
      double N0[MAXND], dN0[MAXND][3], N1[MAXND], dN1[MAXND][3];
      tabulate(tdim, V->degree, X0, N0, dN0);
      tabulate(tdim, V->degree, X1, N1, dN1);
      for (int i = 0; i < nd; ++i)
      {
        if (I->kernel == ORC_K_GHOST_GRADJUMP)
        {
          double a = 0.0, b = 0.0;
          for (int d = 0; d < gdim; ++d)
            for (int t = 0; t < tdim; ++t)
            {
              a += K0[t][d] * dN0[i][t] * n[d];
              b += K1[t][d] * dN1[i][t] * n[d];
            }
          jj[i] = a; jj[nd + i] = -b;
        }
        else { jj[i] = N0[i]; jj[nd + i] = -N1[i]; }
      }
    }
    const double w = I->kernel == ORC_K_GHOST_GRADJUMP
                         ? wref[q] * scale * I->params[0] * havg * (I->params[1] != 0.0 ? pow(havg, I->params[1]) : 1.0)
                         : wref[q] * scale * I->params[0] / havg;
    for (int i = 0; i < 2 * nd0; ++i)
      for (int j = 0; j < 2 * nd1; ++j) Ae[i * n1 + j] += w * jt[i] * ju[j];
  }
  return 0;
}

int orc_tabulate_entity2(const orc_mesh* mesh, const orc_space* V0, const orc_space* V1, const orc_integral* I,
                         int64_t idx, int use_rule, double* Ae)
{
  if (I->type == ORC_INTERIOR_FACET) return facet_kernel2(mesh, V0, V1, I, I->entities + 4 * idx, Ae);
  if (I->type != ORC_CELL) return -1;
  if (use_rule)
  {
    const orc_rules* R = I->rules;
    const int32_t q0 = R->offsets[idx], q1 = R->offsets[idx + 1];
    cell_kernel2(mesh, V0, V1, I, R->parent_map[idx], q1 - q0, R->points + (int64_t)q0 * R->tdim, R->weights + q0, 1.0, Ae);
    return 0;
  }
  int n; const double *p, *w;
  standard_rule(mesh, I, &n, &p, &w);
  double xc[MAXV][3], J[3][3], K[3][3];
  cell_coords(mesh, I->entities[idx], xc);
  const double detJ = fabs(jacobian(mesh->tdim, xc, J, K));
  cell_kernel2(mesh, V0, V1, I, I->entities[idx], n, p, w, detJ, Ae);
  return 0;
}

static int cell_dofs(const orc_space* V, int64_t c, int32_t* dofs)
{
  int n = 0;
  for (int i = 0; i < V->ndofs_cell; ++i)
    for (int k = 0; k < V->bs; ++k) dofs[n++] = V->bs * V->dofmap[c * V->ndofs_cell + i] + k;
  return n;
}

int orc_create_sparsity2(const orc_mesh* mesh, const orc_space* V0, const orc_space* V1, const orc_integral* integrals,
                         int n_integrals, int64_t** indptr_out, int32_t** indices_out)
{
  (void)mesh;
  const int64_t nrows = V0->ndofs * V0->bs;
  int64_t* cnt = (int64_t*)calloc((size_t)nrows + 1, sizeof(int64_t));
  int32_t r[MAXLOC], c[MAXLOC];
  int64_t* ptr = NULL; int32_t* cols = NULL; int64_t* fill = NULL;
  for (int pass = 0; pass < 2; ++pass)
  {
    if (pass == 1)
    {
      ptr = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nrows + 1));
      ptr[0] = 0;
      for (int64_t k = 0; k < nrows; ++k) ptr[k + 1] = ptr[k] + cnt[k];
      cols = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ptr[nrows] + 1));
      fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nrows + 1));
      for (int64_t k = 0; k < nrows; ++k) fill[k] = ptr[k];
    }
    for (int ii = 0; ii < n_integrals; ++ii)
    {
      const orc_integral* I = &integrals[ii];
      if (I->type == ORC_INTERIOR_FACET)
      {
        /* macro rows (both cells, V0) x macro columns (both cells, V1): assembler.h:442-529 with two dofmaps */
        if (V0->bs != 1 || V1->bs != 1) { free(cnt); return -1; }
        for (int64_t e = 0; e < I->n_entities; ++e)
        {
          int32_t rr[2 * MAXLOC], cc[2 * MAXLOC];
          int n0 = 0, n1 = 0;
          for (int sd = 0; sd < 2; ++sd)
          {
            const int64_t cell = I->entities[4 * e + 2 * sd];
            n0 += cell_dofs(V0, cell, rr + n0);
            n1 += cell_dofs(V1, cell, cc + n1);
          }
          for (int i = 0; i < n0; ++i)
          {
            if (pass == 0) cnt[rr[i]] += n1;
            else
              for (int j = 0; j < n1; ++j) cols[fill[rr[i]]++] = cc[j];
          }
        }
        continue;
      }
      if (I->type != ORC_CELL) { free(cnt); return -1; }
      for (int part = 0; part < 2; ++part)
      {
        const int64_t ne = part == 0 ? I->n_entities : (I->rules ? I->rules->nr : 0);
        for (int64_t e = 0; e < ne; ++e)
        {
          const int64_t cell = part ? I->rules->parent_map[e] : I->entities[e];
          const int n0 = cell_dofs(V0, cell, r), n1 = cell_dofs(V1, cell, c);
          for (int i = 0; i < n0; ++i)
          {
            if (pass == 0) cnt[r[i]] += n1;
            else
              for (int j = 0; j < n1; ++j) cols[fill[r[i]]++] = c[j];
          }
        }
      }
    }
  }
  int64_t* indptr = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nrows + 1));
  int64_t nnz = 0;
  indptr[0] = 0;
  for (int64_t k = 0; k < nrows; ++k)
  {
    int32_t* row = cols + ptr[k];
    const int64_t len = fill[k] - ptr[k];
    qsort(row, (size_t)len, sizeof(int32_t), i32_cmp);
    int64_t u = 0;
    for (int64_t j = 0; j < len; ++j)
      if (j == 0 || row[j] != row[j - 1]) cols[nnz + u++] = row[j];
    nnz += u;
    indptr[k + 1] = nnz;
  }
  int32_t* indices = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
  memcpy(indices, cols, sizeof(int32_t) * (size_t)nnz);
  free(cols); free(ptr); free(fill); free(cnt);
  *indptr_out = indptr; *indices_out = indices;
  return 0;
}

int orc_assemble_matrix2(const orc_mesh* mesh, const orc_space* V0, const orc_space* V1, const orc_integral* integrals,
                         int n_integrals, const int8_t* bc0, const int8_t* bc1, const int64_t* indptr,
                         const int32_t* indices, double* values)
{
  double Ae[MAXLOC * MAXLOC];
  int32_t r[MAXLOC], c[MAXLOC];
  for (int ii = 0; ii < n_integrals; ++ii)
  {
    const orc_integral* I = &integrals[ii];
    if (I->type == ORC_INTERIOR_FACET)
    {
      for (int64_t e = 0; e < I->n_entities; ++e)
      {
        int32_t rr[2 * MAXLOC], cc[2 * MAXLOC];
        double Af[4 * MAXND * MAXND];
        int n0 = 0, n1 = 0;
        for (int sd = 0; sd < 2; ++sd)
        {
          const int64_t cell = I->entities[4 * e + 2 * sd];
          n0 += cell_dofs(V0, cell, rr + n0);
          n1 += cell_dofs(V1, cell, cc + n1);
        }
        memset(Af, 0, sizeof(double) * (size_t)(n0 * n1));
        if (orc_tabulate_entity2(mesh, V0, V1, I, e, 0, Af) != 0) return -2;
        for (int i = 0; i < n0; ++i)
          for (int j = 0; j < n1; ++j)
          {
            if ((bc0 && bc0[rr[i]]) || (bc1 && bc1[cc[j]])) continue;
            const int64_t b = indptr[rr[i]], en = indptr[rr[i] + 1];
            int64_t lo = b, hi = en;
            while (lo < hi)
            {
              const int64_t mid = (lo + hi) / 2;
              if (indices[mid] < cc[j]) lo = mid + 1; else hi = mid;
            }
            if (lo == en || indices[lo] != cc[j]) return -1;
            values[lo] += Af[i * n1 + j];
          }
      }
      continue;
    }
    for (int part = 0; part < 2; ++part)
    {
      const int64_t ne = part == 0 ? I->n_entities : (I->rules ? I->rules->nr : 0);
      for (int64_t e = 0; e < ne; ++e)
      {
        const int64_t cell = part ? I->rules->parent_map[e] : I->entities[e];
        const int n0 = cell_dofs(V0, cell, r), n1 = cell_dofs(V1, cell, c);
        memset(Ae, 0, sizeof(double) * (size_t)(n0 * n1));
        if (orc_tabulate_entity2(mesh, V0, V1, I, e, part, Ae) != 0) return -2;
        /* zero BC rows (test side) and columns (trial side): assemble_matrix_impl.h:151-185 */
        for (int i = 0; i < n0; ++i)
          for (int j = 0; j < n1; ++j)
            if ((bc0 && bc0[r[i]]) || (bc1 && bc1[c[j]])) Ae[i * n1 + j] = 0.0;
        for (int i = 0; i < n0; ++i)
        {
          const int64_t b = indptr[r[i]], en = indptr[r[i] + 1];
          for (int j = 0; j < n1; ++j)
          {
            int64_t lo = b, hi = en;
            while (lo < hi)
            {
              const int64_t mid = (lo + hi) / 2;
              if (indices[mid] < c[j]) lo = mid + 1; else hi = mid;
            }
            if (lo == en || indices[lo] != c[j]) return -1;
            values[lo] += Ae[i * n1 + j];
          }
        }
      }
    }
  }
  return 0;
}
