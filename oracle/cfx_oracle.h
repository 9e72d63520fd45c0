/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the CutFEMx hot path (level-set classification ->
 * sub-triangulation -> runtime quadrature -> local tensors -> CSR scatter ->
 * deactivation).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (cutfemx_amd/) never
 * does.
 *
 * PARITY STATUS (see DESIGN.md "Oracle"):
 *   - loop structure, argument order, Ae layouts, BC zeroing, sparsity with the
 *     all-rows diagonal, selector semantics, classification rule, normal
 *     evaluator, deactivation: restated from the reference files cited on each
 *     function (paths relative to /root/reference).
 *   - classification, selector algebra, rule-array contracts, area/perimeter,
 *     sum(b)==area, active-domain and runtime-vs-standard matrix parity are
 *     pinned by the reference's own test invariants (tests/test_oracle_*.py).
 *   - per-point quadrature coordinates/weights and the sub-triangulation
 *     topology live in CutCells (third party, >=0.4.0,<0.5.0, not vendored) and
 *     the reference holds no fixture for them: PARITY UNPINNED at that level.
 *     Integrals of polynomial integrands do not depend on that choice.
 */
#ifndef CFX_ORACLE_H
#define CFX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* domain codes (sign of the level set on the cell) */
enum { ORC_INSIDE = -1, ORC_INTERSECTED = 0, ORC_OUTSIDE = 1 };

/* integral types, order of python/cutfemx/fem.py:261-267 */
enum { ORC_CELL = 0, ORC_EXTERIOR_FACET = 1, ORC_INTERIOR_FACET = 2 };

/* integrand ids (stand in for the JIT kernel pointer) */
enum {
  ORC_K_MASS = 1,           /* u v                                         */
  ORC_K_STIFFNESS = 2,      /* grad u . grad v                             */
  ORC_K_NITSCHE = 3,        /* -dn(u) v - dn(v) u + gamma/h u v  (params[0]=gamma) */
  ORC_K_GHOST_GRADJUMP = 4, /* gamma_g h_avg [dn u][dn v]  (params[0]=gamma_g)     */
  ORC_K_ELASTICITY = 5,     /* sigma(u):eps(v), params[0]=E, params[1]=nu  */
  ORC_K_NITSCHE_ELASTICITY = 6,
  ORC_K_GHOST_GRADJUMP_VEC = 7,
  ORC_K_EXTENSION_L2 = 8,   /* beta (v|bad - E v|root)(u|bad - E u|root) over the bad cell; pairs (bad,0,root,0) as
                               interior-facet-type entities; params[0]=beta, point_data (stride 1) = per-pair factor */
  ORC_K_JUMP = 9,           /* interior facets: gamma / h_avg [u][v]  (params[0]=gamma)         */
  ORC_K_SIP = 10,           /* interior facets: -{dn u}[v] - {dn v}[u] + sigma/h_avg [u][v]  (params[0]=sigma) */
  /* rectangular blocks (test space != trial space): the off-diagonal blocks of Stokes and friends.
     ref: assemble_matrix_impl.h:68-189 takes dofmap0/bs0 and dofmap1/bs1 separately; invariants
     python/tests/test_assembly_stokes.py:34-95 */
  ORC_K_DIV_TEST = 20,      /* scale div(v) p : test vector (bs0 = gdim), trial scalar; params[0] = scale */
  ORC_K_DIV_TRIAL = 21,     /* scale q div(u) : test scalar, trial vector (bs1 = gdim); params[0] = scale  */
  ORC_L_SOURCE = 101,       /* f v, f = analytic id params[0], scale params[1] */
  ORC_L_NITSCHE_RHS = 102   /* -dn(v) g + gamma/h g v, gamma=params[0], g id params[1], scale params[2] */
};

/* analytic scalar fields evaluated at physical points */
enum {
  ORC_F_ONE = 0,           /* 1                                    */
  ORC_F_SINPROD = 1,       /* prod_i sin(pi x_i)                   */
  ORC_F_POISSON_RHS = 2,   /* gdim pi^2 prod_i sin(pi x_i)         */
  ORC_F_COEFFICIENT = 3    /* sum_j N_j w[dof_j], w = integral.coefficient (pack_form.h:32-170) */
};

typedef struct {
  int32_t tdim;        /* dimension of the reference points (parent cell) */
  int64_t nq;          /* total points */
  int64_t nr;          /* number of rules */
  double* points;      /* [nq*tdim] parent-reference coordinates */
  double* weights;     /* [nq] physical measure weights          */
  int32_t* offsets;    /* [nr+1]                                 */
  int32_t* parent_map; /* [nr] parent cell of each rule          */
} orc_rules;

typedef struct {
  int32_t type;          /* ORC_CELL / ORC_INTERIOR_FACET                      */
  int32_t kernel;        /* ORC_K_* or ORC_L_*                                  */
  int32_t qdegree;       /* standard-quadrature degree for uncut entities       */
  int32_t point_stride;  /* doubles per point in point_data                     */
  const int32_t* entities; /* cells: ids; interior facets: (c0,lf0,c1,lf1) rows */
  int64_t n_entities;
  const orc_rules* rules;  /* runtime rules for the cut entities, or NULL       */
  const double* point_data;/* per-point coefficients aligned with rules points  */
  double params[8];
  const double* coefficient; /* dof values of the coefficient Function, or NULL */
  /* interior-facet integrals with facet-hosted rules (8f-4): entities [n_std, n_entities) are the rows of the
     rules' facets, host_verts [nr*tdim] the mesh vertices spanning each rule's reference simplex; else NULL */
  int64_t n_std;
  const int32_t* host_verts;
} orc_integral;

typedef struct {
  int32_t tdim, gdim;
  int64_t nnodes, ncells;
  const double* x;        /* [nnodes*3], stride 3 (cut.cpp:528-529)          */
  const int32_t* conn;    /* [ncells*(tdim+1)] geometry dofmap, P1           */
} orc_mesh;

typedef struct {
  int32_t degree;         /* Lagrange degree 1 or 2                          */
  int32_t bs;             /* block size (1 scalar, gdim vector)              */
  int32_t ndofs_cell;     /* scalar dofs per cell                            */
  int64_t ndofs;          /* scalar dofs in the space                        */
  const int32_t* dofmap;  /* [ncells*ndofs_cell]                             */
} orc_space;

/* rectangular forms: test space V0 (rows), trial space V1 (columns); cell integrals with kernels ORC_K_MASS /
   ORC_K_STIFFNESS (bs0 == bs1, any degrees), ORC_K_DIV_TEST, ORC_K_DIV_TRIAL.  Ae is [(nd0 bs0) x (nd1 bs1)] row-major.
   The sparsity has NO all-rows diagonal (insert_deactivation_diagonal returns when the two index maps differ,
   assembler.h:537-560). */
int orc_tabulate_entity2(const orc_mesh* mesh, const orc_space* V0, const orc_space* V1, const orc_integral* I,
                         int64_t idx, int use_rule, double* Ae);
int orc_create_sparsity2(const orc_mesh* mesh, const orc_space* V0, const orc_space* V1, const orc_integral* integrals,
                         int n_integrals, int64_t** indptr_out, int32_t** indices_out);
int orc_assemble_matrix2(const orc_mesh* mesh, const orc_space* V0, const orc_space* V1, const orc_integral* integrals,
                         int n_integrals, const int8_t* bc0, const int8_t* bc1, const int64_t* indptr,
                         const int32_t* indices, double* values);

void orc_free(void* p);
void orc_rules_free(orc_rules* r);

/* synthetic inputs: box [0,1]^d, N^d cubes, Kuhn split */
void orc_mesh_box(int tdim, int N, double* x, int32_t* conn);

/* a1 */
void orc_classify(int64_t ncells, int ndofs_cell, const int32_t* ls_dofmap,
                  const double* ls_values, int8_t* domain);
/* a4: returns count, fills out (capacity ncells); -1 on selector error */
int64_t orc_locate_entities(int64_t ncells, int nls, const int8_t* domain,
                            const char* selector, int32_t* out);
/* a2+a3 */
int orc_runtime_quadrature(const orc_mesh* mesh, const int32_t* ls_dofmap,
                           const double* ls_values, const int8_t* domain,
                           const char* selector, int order, orc_rules* out);
/* several level sets: one conjunction of clauses, e.g. "phi<0 and phi1>0" or "phi=0 and phi1<0"
   (cut.h:122-181, element-classification.md:145-160); ls_values[k] = dof values of level set k,
   domain = [nls][ncells] classification codes */
int orc_runtime_quadrature_multi(const orc_mesh* mesh, int nls, const int32_t* ls_dofmap,
                                 const double* const* ls_values, const int8_t* domain,
                                 const char* selector, int order, orc_rules* out);
/* rule for a whole reference simplex of each listed cell (test helper that
   mirrors python/tests/quadrature_utils.py:12-70) */
int orc_facet_runtime_quadrature(const orc_mesh* mesh, int64_t n, const int32_t* verts, const int32_t* ls,
                                 const int32_t* ids, const double* ls_values, const int8_t* domain,
                                 const char* selector, int order, int whole, orc_rules* out, int32_t** rule_host);
int orc_full_cell_rules(const orc_mesh* mesh, const int32_t* cells, int64_t n,
                        int order, orc_rules* out);
/* a12 */
void orc_evaluate_normals(const orc_mesh* mesh, const int32_t* ls_dofmap,
                          const double* ls_values, const orc_rules* rules,
                          double sign, double* out);
void orc_evaluate_values(const orc_mesh* mesh, const int32_t* ls_dofmap,
                         const double* ls_values, const orc_rules* rules,
                         double* out);
void orc_physical_points(const orc_mesh* mesh, const orc_rules* rules, double* out);
/* a7 facet selection: rows (c0,lf0,c1,lf1), c0<c1, sorted; returns count */
int64_t orc_ghost_penalty_facets(const orc_mesh* mesh, const int8_t* domain,
                                 const char* selector, int32_t** rows_out);
int64_t orc_interior_facets_for_cells(const orc_mesh* mesh, const int32_t* cells,
                                      int64_t ncells_sel, int32_t** rows_out);
/* a9 */
int orc_create_sparsity(const orc_mesh* mesh, const orc_space* V,
                        const orc_integral* integrals, int n_integrals,
                        int64_t** indptr, int32_t** indices);
/* a5/a6/a7 */
int orc_assemble_matrix(const orc_mesh* mesh, const orc_space* V,
                        const orc_integral* integrals, int n_integrals,
                        const int8_t* bc0, const int8_t* bc1,
                        const int64_t* indptr, const int32_t* indices,
                        double* values);
/* a8 */
/* Cell aggregation: ill-posed cut cells inherit a root from a facet neighbour, in sequential
   sweeps over the ascending ill-posed list (an earlier cell of the same sweep already counts).
   ref: cpp/cutfemx/extensions/cell_aggregation.cpp:143-270.  `domain` is the classification of
   the selector's level set, `relation` -1 for "phi<0", +1 for "phi>0", `fraction[c]` the cut
   volume fraction of the selected part (cut cells).  policy 0 interior_only, 1 interior_or_well_cut.
   Outputs (size ncells): root_cell, aggregate_id, depth (-1 where unset).  Returns the number of
   rootless ill-posed cells. */
int64_t orc_cell_aggregation(const orc_mesh* mesh, const int8_t* domain, int relation,
                             const double* fraction, double threshold, int policy,
                             int max_iterations, int32_t* root_cell, int32_t* aggregate_id,
                             int32_t* depth);

int orc_assemble_vector(const orc_mesh* mesh, const orc_space* V,
                        const orc_integral* integrals, int n_integrals,
                        double* b);
/* b -= alpha Ae (g - x0) on entities with a Dirichlet column: assemble_vector_impl.h:383-436 */
int orc_apply_lifting(const orc_mesh* mesh, const orc_space* V,
                      const orc_integral* integrals, int n_integrals,
                      const int8_t* markers, const double* g, const double* x0,
                      double alpha, double* b);
/* local tensor of one entity (for local-entry parity tests) */
int orc_tabulate_entity(const orc_mesh* mesh, const orc_space* V,
                        const orc_integral* integral, int64_t entity_or_rule,
                        int use_rule, double* Ae);
/* a11 */
int64_t orc_active_cells(const orc_integral* integrals, int n_integrals,
                         int64_t ncells, int32_t* out);
int64_t orc_inactive_dofs(const orc_space* V, const int32_t* active_cells,
                          int64_t n_active, int32_t* out);
void orc_deactivate(const int32_t* inactive, int64_t n, int bs_unused,
                    const int64_t* indptr, const int32_t* indices,
                    double* values, double* b, double diagonal, double rhs_value);

#ifdef __cplusplus
}
#endif
#endif
