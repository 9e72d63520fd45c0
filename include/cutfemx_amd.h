/* cutfemx_amd -- C ABI of the MI355X-native cut-FEM quadrature-and-assembly
 * engine (libcutfemx_amd.so).
 *
 * Drop-in boundary for the CutFEMx hot path.  Every entry point
 * names the reference interface it replaces (paths relative to the CutFEMx
 * source tree).  Conventions:
 *   - plain C types only; handles are opaque pointers;
 *   - every function returns 0 on success, a negative CFX_ERR_* otherwise, and
 *     cfx_last_error() returns the thread-local message (the reference throws
 *     std::invalid_argument / runtime_error / out_of_range; the code says which);
 *   - INPUT array pointers may be host or device (HIP) pointers: the library
 *     detects the memory space and copies host data to HBM once;
 *   - OUTPUT arrays live in HBM and are owned by their handle; `*_view`
 *     functions expose the device pointers, cfx_copy() moves bytes to a host
 *     or device destination;
 *   - the caller keeps ownership of inputs; a handle that was given a DEVICE
 *     pointer aliases it (zero copy) and the caller must keep it alive;
 *   - calls on one device are issued on one HIP stream (cfx_set_stream) and
 *     are not thread-safe;
 *   - one process drives one GPU; several GPUs = several processes joined by a
 *     cfx_comm_t (cfx_dist_*, at the end of this header).
 * Scalar/geometry type: float64/float64 (north_star).  Of the other instantiations of
 * python/cutfemx/wrappers/fem.cpp:490-500 the boundary carries float32 containers (`*_f32`: widened on the way in,
 * fp64 arithmetic, rounded once on the way out), complex128 (`*_c128`) and complex64 (`*_c64`: interleaved float32
 * containers, the complex128 arithmetic, rounded once).
 */
#ifndef CUTFEMX_AMD_H
#define CUTFEMX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (exception classes of the reference) ------------------- */
#define CFX_OK 0
#define CFX_ERR_INVALID_ARGUMENT (-1) /* std::invalid_argument -> ValueError  */
#define CFX_ERR_RUNTIME (-2)          /* std::runtime_error   -> RuntimeError */
#define CFX_ERR_OUT_OF_RANGE (-3)     /* std::out_of_range    -> IndexError   */
#define CFX_ERR_HIP (-4)              /* HIP runtime failure / no device      */
#define CFX_ERR_STEP_VOID (-5)        /* inside cfx_step_begin / cfx_step_end: a count did not fit the capacity taken
                                         from the previous step, the results of the step are void -- call
                                         cfx_step_end() (it reports redo = 1) and repeat the step.  ANY call of
                                         a step may return it: an error raised while the step is void, and every
                                         size read back to the host after the count overflowed (the read-back
                                         carries the step's poison word: a total of lengths nobody wrote must not
                                         size anything)                                                         */

/* ---- classification codes: cutcells::cell::domain as used by
 *      cpp/cutfemx/cut/cut.cpp:292-321 ---------------------------------------- */
#define CFX_INSIDE (-1)     /* all level-set dof values < 0 */
#define CFX_INTERSECTED 0   /* anything else, incl. a value == 0 */
#define CFX_OUTSIDE 1       /* all level-set dof values > 0 */

/* ---- integral types, order of python/cutfemx/fem.py:261-267 --------------- */
#define CFX_CELL 0
#define CFX_EXTERIOR_FACET 1
#define CFX_INTERIOR_FACET 2

/* ---- integrand ids: replace the JIT kernel pointer of
 *      cpp/dolfinx_custom_data/fem/Form.h:59-75 (a GPU engine cannot call a
 *      CPU function pointer per entity) ---------------------- */
#define CFX_K_MASS 1            /* u v                                         */
#define CFX_K_STIFFNESS 2       /* grad u . grad v                             */
#define CFX_K_NITSCHE 3         /* -dn(u) v - dn(v) u + gamma/h u v; params[0]=gamma;
                                   point_data = unit normals (gdim per point)  */
#define CFX_K_GHOST_GRADJUMP 4  /* gamma_g h_avg^(1 + e) [dn u][dn v]; params[0]=gamma_g, params[1]=e (0: the usual
                                   ghost penalty; 2: the pressure term avg(h)^3 of test_assembly_stokes.py:123-131) */
#define CFX_K_ELASTICITY 5      /* sigma(u):eps(v); params[0]=E, params[1]=nu  */
/* extension penalty pair block, beta (v|bad - E v|root)(u|bad - E u|root) over the full bad cell
 * (cpp/cutfemx/extensions/extension_penalty.cpp:191-369): an interior-facet-TYPE integral whose
 * entity rows are (bad_cell, 0, root_cell, 0) from cfx_extension_pairs; params[0]=beta, qdegree =
 * quadrature degree on the bad cell; point_data (stride 1, one value per pair) = cellwise beta factor */
#define CFX_K_EXTENSION_L2 8
#define CFX_K_JUMP 9             /* interior facets: gamma / h_avg [u][v] (DG / skeleton value-jump penalty);
                                   params[0]=gamma */
#define CFX_K_SIP 10             /* interior facets: symmetric interior penalty of DG Poisson,
                                   -{dn u}[v] - {dn v}[u] + sigma / h_avg [u][v] (python/demo/demo_dg_poisson.py:262-265);
                                   params[0]=sigma */
/* rectangular blocks (test space != trial space, cfx_form_create2): the off-diagonal blocks of Stokes and friends.
 * assemble_matrix_impl.h:68-189 takes dofmap0 / bs0 and dofmap1 / bs1 separately; invariants
 * python/tests/test_assembly_stokes.py:34-95.  CFX_K_MASS / CFX_K_STIFFNESS are also accepted there (bs0 == bs1) */
#define CFX_K_DIV_TEST 20        /* scale div(v) p: test vector (bs = gdim), trial scalar; params[0] = scale */
#define CFX_K_DIV_TRIAL 21       /* scale q div(u): test scalar, trial vector (bs = gdim); params[0] = scale  */
#define CFX_L_SOURCE 101        /* f v; params[0]=field id, params[1]=scale    */
#define CFX_L_NITSCHE_RHS 102   /* -dn(v) g + gamma/h g v; params[0]=gamma,
                                   params[1]=field id of g, params[2]=scale    */
/* analytic fields evaluated at physical quadrature points */
#define CFX_F_ONE 0             /* 1 */
#define CFX_F_SINPROD 1         /* prod_i sin(pi x_i) */
#define CFX_F_POISSON_RHS 2     /* gdim pi^2 prod_i sin(pi x_i) */
#define CFX_F_COEFFICIENT 3     /* a Function of the form's space: sum_j N_j(X_q) w[dof_j], `coefficient` = its dof values
                                   (the packed coefficient of pack_form.h:32-170 evaluated by the kernel) */

/* ids >= CFX_K_USER_BASE: integrands registered at run time (cfx_integrand_register, below) */
#define CFX_K_USER_BASE 1000

typedef struct cfx_mesh_s* cfx_mesh_t;
typedef struct cfx_cut_s* cfx_cut_t;
typedef struct cfx_rules_s* cfx_rules_t;
typedef struct cfx_space_s* cfx_space_t;
typedef struct cfx_form_s* cfx_form_t;
typedef struct cfx_pattern_s* cfx_pattern_t;
typedef struct cfx_active_s* cfx_active_t;
typedef struct cfx_aggregation_s* cfx_aggregation_t;

/* Options of cutfemx.cut(); defaults of python/cutfemx/wrappers/cut.cpp:117-140 */
typedef struct
{
  int32_t cut_approximation_order;   /* 1 */
  int32_t max_refinement_iterations; /* 8  (unused for P1 level sets) */
  int32_t edge_max_depth;            /* 20 (unused for P1 level sets) */
  int32_t reserved;
} cfx_cut_options;

/* Device view of cutfemx::RuntimeQuadrature
 * (cpp/cutfemx/cut/runtime_quadrature.h:223-231, python/cutfemx/wrappers/cut.cpp:181-240) */
typedef struct
{
  int32_t tdim;              /* columns of points                              */
  int32_t gdim;
  int64_t nq;                /* total points = offsets[nr]                     */
  int64_t nr;                /* number of rules = len(parent_map)              */
  const double* points;      /* [nq*tdim] parent-reference coordinates (HBM)   */
  const double* weights;     /* [nq] physical-measure weights (HBM)            */
  const int32_t* offsets;    /* [nr+1] (HBM)                                   */
  const int32_t* parent_map; /* [nr] parent background cell (HBM); facet-hosted rules: the caller's facet id */
  /* facet-hosted rules (cfx_cut_create_facets): tdim above is the facet dimension */
  int32_t host_width;        /* 0: hosted by cells; 2 | 4: width of host_rows                   */
  int32_t reserved;
  const int32_t* host_rows;  /* [nr*host_width] integration row of each rule's facet (HBM), or NULL */
  const int32_t* host_verts; /* [nr*(tdim+1)] mesh vertices spanning each rule's reference simplex, or NULL */
} cfx_rules_view;

/* One integral of a form: the (kernel_ptr, entities, active_coeffs, custom_data)
 * tuple of python/cutfemx/fem.py:346-351 with the kernel selected by id.
 * `entities` are the standard (uncut) entities, `rules` the runtime (cut)
 * entities of the same measure -- the [inside_cells, rules] measure of
 * python/demo/demo_poisson.py:143-147.
 * Entity / rule / point_data arrays given as DEVICE pointers are aliased: they
 * must stay allocated and unchanged while any form created from them is alive
 * (forms of one space that reference the same arrays share derived tables). */
typedef struct
{
  int32_t type;            /* CFX_CELL | CFX_INTERIOR_FACET (exterior-facet terms are CFX_CELL integrals over
                              cfx_facet_rules_to_cells rules)                                          */
  int32_t kernel;          /* CFX_K_* (rank 2) or CFX_L_* (rank 1)             */
  int32_t qdegree;         /* standard quadrature degree for uncut entities    */
  int32_t point_stride;    /* doubles per point in point_data                  */
  const int32_t* entities; /* cells: ids; interior facets: (c0,lf0,c1,lf1)     */
  int64_t n_entities;
  cfx_rules_t rules;       /* or NULL.  CFX_CELL: cell-hosted rules; CFX_INTERIOR_FACET: facet-hosted rules
                              over interior rows (cfx_cut_create_facets, row_width 4) -- the cut facets of
                              a dS measure, `entities` being its standard facets                      */
  const double* point_data;/* per-point coefficients aligned with rules, or NULL */
  double params[8];
  const double* coefficient; /* packed coefficient of pack_form.h:32-170, given as the dof values of a Function:
                                rank 1, field id CFX_F_COEFFICIENT: the source f (ndofs values; on a vector
                                space ndofs * bs values of a vector-valued f); rank 2 (mass, stiffness,
                                elasticity): a scalar coefficient kappa of the form's element (ndofs values)
                                that multiplies the integrand; else NULL.  Constants travel in `params`  */
} cfx_integral;

typedef struct
{
  int64_t nrows;
  int64_t nnz;
  const int64_t* indptr;  /* [nrows+1] (HBM), DOLFINx MatrixCSR row_ptr type */
  const int32_t* indices; /* [nnz] sorted per row (HBM)                       */
  int64_t ncols;          /* = nrows for square forms; trial-space dofs x bs for cfx_form_create2 forms */
} cfx_pattern_view;

/* ---- runtime ------------------------------------------------------------- */
int cfx_init(int device);              /* select the HIP device; fails loudly without one.  One process drives one
                                          GPU: a later call naming another device is CFX_ERR_INVALID_ARGUMENT */
const char* cfx_last_error(void);
int cfx_set_stream(void* hip_stream);  /* all later launches go to this stream (the old one is synchronised first) */
int cfx_synchronize(void);
/* Two independent pieces of a step side by side (an extension: the reference is serial).  cfx_overlap_begin() opens
 * a section with a second HIP stream that starts behind everything queued so far; cfx_overlap_side(1) / (0) chooses the
 * lane the following calls are queued on; cfx_overlap_end() joins the lanes.  The caller guarantees that the two
 * lanes neither write the same arrays nor build the same derived tables: a form's row plan is built by the first
 * call that needs it, so call cfx_form_prepare() on forms that share entity lists before the section.  Host-side
 * size read-backs inside a call wait for that call's lane only: queue the lane with the fewest of them first.
 * Typical use: assemble_vector(L) next to create_sparsity(a) + assemble_matrix(a) (python/demo/demo_poisson.py:40-60
 * assembles them one after the other; they share inputs only).  Pays on small meshes, where single kernels leave
 * CUs idle; at 512^3 every kernel fills the chip and the two lanes only contend (measured: DESIGN.md 3). */
int cfx_overlap_begin(void);
int cfx_overlap_side(int side);
int cfx_overlap_end(void);
/* Sync-free steps of a moving-domain loop (python/demo/demo_moving_poisson.py:53-67: cut.update -> rules -> forms ->
 * create_matrix -> assemble, every time step).  The sizes of the data-dependent lists of such a step (located cells,
 * rule points, ghost facets, row classes, nnz ...) change little from one step to the next, so between
 * cfx_step_begin(key) and cfx_step_end() the library does not read them back where they are produced: buffers and
 * grids are sized by the same site's count in the previous step of the loop `key` (x 1.03125 + 256 by default; a count that grew
 * over the last two valid steps is first extrapolated by the same amount), the
 * exact lengths stay in HBM where the kernels read them, and cfx_step_end() fetches all of them -- and the error
 * words of the assembly calls -- in ONE read-back.  The first step of a key (no history) reads every size back as
 * outside a step.  *redo = 1: some count did not fit its capacity; every kernel after that point did nothing, the
 * results of the step are void and the caller repeats the same calls (the repeat reads sizes back and always fits).
 * While a step is open, counts returned through this ABI (cfx_locate_entities, cfx_rules_view_get, cfx_pattern_view_get,
 * cfx_active_view, ...) are CAPACITIES >= the true count when the true count is still in HBM; pass them back to the
 * library unchanged (it recognises its own lists by address) and query again after cfx_step_end() for the exact
 * values.  *published / *read_back (optional): sites of the step that stayed in HBM / were read back at once.
 * No reference counterpart (the reference is a CPU code: sizes are free there). */
int cfx_step_begin(const char* key);
int cfx_step_end(int* redo, int64_t* published, int64_t* read_back);
int cfx_step_resolve(void);                              /* inside a step: fetch every count published so far (one read-back), e.g.
                                                            before copying a list to the host; no-op outside a step */
int cfx_step_abort(void);                                /* leave a step after an error: resolves what is pending, drops the history */
int cfx_step_set_margin(double factor, int64_t slack);   /* capacity = previous count x factor + slack (tests force the redo path) */
int cfx_step_forget(const char* key);                    /* drop the history of a loop (NULL: of all loops) */
/* Length of an entity list the library handed out as (pointer, count) -- cfx_locate_entities, cfx_ghost_penalty_facets,
 * cfx_interior_facets_for_cells -- as the engine knows it NOW: the count returned inside a sync-free step is the list's
 * capacity; after the step the exact length.  Looked up by the list's identity; nothing is recomputed.  A pointer the
 * engine does not track (a list made outside a step, whose count was exact; the caller's own array) gives n_given back. */
int cfx_list_count(const void* list, int64_t n_given, int64_t* n);
int cfx_sync_count(int64_t* n);                          /* host round trips (size / error read-backs) since start-up */
int cfx_copy(void* dst, const void* src, size_t bytes); /* hipMemcpyDefault on the stream + sync */
int cfx_device_alloc(void** ptr, size_t bytes);
int cfx_device_free(void* ptr);
int cfx_device_cache_release(void); /* hand the library's cached (free) HBM blocks back to the driver */
/* HBM held by the library's block cache: bytes handed out, bytes cached (free), and the high-water mark of their sum
 * since start-up or the last call with reset_peak != 0 (mesh-static tables + temporaries of a step; caller-owned
 * arrays are not counted).  No reference counterpart: the measurement side of SURVEY 8d. */
int cfx_device_memory_stats(size_t* in_use, size_t* cached, size_t* peak, int reset_peak);
int cfx_device_memset(void* ptr, int byte, size_t bytes); /* la::MatrixCSR::set_value(0) / la::Vector zeroing, on the stream */
/* per-kernel HIP-event timing of the launches made by this library */
int cfx_profile_enable(int on);
int cfx_profile_reset(void);
int cfx_profile_count(void);
int cfx_profile_get(int i, const char** name, double* total_ms, int64_t* launches);
/* HIP events on the library stream (bench.py timing) */
int cfx_event_create(void** ev);
int cfx_event_record(void* ev);
int cfx_event_elapsed_ms(void* start, void* stop, double* ms); /* synchronises on stop */
int cfx_event_destroy(void* ev);

/* ---- mesh: cutcells::MeshView built by build_mesh_view,
 *      cpp/cutfemx/cut/cut.cpp:500-538 (x stride 3, int32 connectivity) ------ */
int cfx_mesh_create(int tdim, int gdim, int64_t nnodes, const double* x,
                    int64_t ncells, const int32_t* conn, int cell_stride,
                    cfx_mesh_t* out);
/* synthetic box mesh generated in HBM (Kuhn split of
 * cpp/cutfemx/distance/fast_iterative.h:93-108) */
int cfx_mesh_create_box(int tdim, int n, cfx_mesh_t* out);
/* hex layers z0 .. z0+nz-1 of the n^3 box mesh (one rank's slab + halo): local
 * vertex / cell ids are the global ids minus (n+1)^2 z0 / 6 n^2 z0 */
int cfx_mesh_create_slab(int n, int z0, int nz, cfx_mesh_t* out);
int cfx_mesh_info(cfx_mesh_t m, int* tdim, int* gdim, int64_t* nnodes, int64_t* ncells,
                  const double** x, const int32_t** conn);
int cfx_mesh_destroy(cfx_mesh_t m);

/* ---- cut: cutfemx::cut / update / locate_entities / runtime_quadrature,
 *      cpp/cutfemx/cut/cut.h:104-181 ------------------------------------------ */
int cfx_cut_options_default(cfx_cut_options* opt);
/* cut(): cut.cpp:743-786 + update :845-868.  ls_values[k] = dof values of level
 * set k; one shared P1 dofmap [ncells*ls_ndofs_cell]. */
int cfx_cut_create(cfx_mesh_t mesh, int n_level_sets, const int32_t* ls_dofmap,
                   int ls_ndofs_cell, int64_t ls_ndofs, const double* const* ls_values,
                   const cfx_cut_options* opt, cfx_cut_t* out);
/* update(): cut.cpp:845-868 -- re-classify from new values (same pointers if NULL) */
/* cut(level_set, entities, entity_dim = tdim) (cut.cpp:788-830, python/cutfemx/cut.py:186-249): keep only
 * the listed background cells as candidates; every other cell gets the classification code
 * CFX_NOT_CANDIDATE, which no selector matches, so located lists, rules, ghost facets and aggregations
 * see the subset only.  Survives cfx_cut_update. */
#define CFX_NOT_CANDIDATE 2
int cfx_cut_restrict(cfx_cut_t cut, const int32_t* cells, int64_t n);
int cfx_cut_update(cfx_cut_t cut, const double* const* ls_values);
int cfx_cut_info(cfx_cut_t cut, int* tdim, int* gdim, int64_t* num_local_cells,
                 int* n_level_sets);
/* ParentCellClassification::domain(ls, cell) for all cells: device int8 [ncells] */
int cfx_cut_domain(cfx_cut_t cut, int level_set, const int8_t** domain);
/* locate_entities(): cut.cpp:877-924.  *entities is owned by `cut` until the
 * next locate call with the same selector or cfx_cut_destroy. */
int cfx_locate_entities(cfx_cut_t cut, const char* selector,
                        const int32_t** entities, int64_t* n);
/* runtime_quadrature(): cut.cpp:1311-1335 (backend "straight" only) */
int cfx_runtime_quadrature(cfx_cut_t cut, const char* selector, int order,
                           const char* backend, cfx_rules_t* out);
/* runtime_quadratures(): cut.h:178-181, python/cutfemx/cut.py runtime_quadratures(cut_data, ls_parts, order) -- one
 * rule set per selector, out[n].  Pairs of plain selectors ("phi<0", "phi=0", "phi>0") of a single level set share
 * one pass over the cut cells (each cell's vertices, level-set values and cut points are staged once) and one size
 * read-back; any other selector takes the single call.  On error nothing is returned (out[] all NULL). */
int cfx_runtime_quadratures(cfx_cut_t cut, int n, const char* const* selectors, int order,
                            const char* backend, cfx_rules_t* out);
/* rules over whole cells (reference points, weights*|detJ|): the test helper
 * python/tests/quadrature_utils.py:12-70 */
int cfx_full_cell_rules(cfx_mesh_t mesh, const int32_t* cells, int64_t n, int order,
                        cfx_rules_t* out);
/* wrap caller-provided rule arrays (runintgen.QuadratureRules(kind="per_entity")) */
int cfx_rules_create(cfx_mesh_t mesh, int tdim, int64_t nq, int64_t nr, const double* points,
                     const double* weights, const int32_t* offsets,
                     const int32_t* parent_map, cfx_rules_t* out);
int cfx_rules_view_get(cfx_rules_t r, cfx_rules_view* view);
/* RuntimeQuadrature::physical_points(): runtime_quadrature.h:102-221; out [nq*gdim] */
int cfx_rules_physical_points(cfx_rules_t r, double* out);
int cfx_rules_destroy(cfx_rules_t r);
/* level_set::evaluate_normals / evaluate_values:
 * cpp/cutfemx/level_set/normal.h:39-187, value.h:34-119; out in HBM or host */
int cfx_evaluate_normals(cfx_cut_t cut, int level_set, cfx_rules_t rules, double sign,
                         double* out /* [nq*gdim] */);
int cfx_evaluate_values(cfx_cut_t cut, int level_set, cfx_rules_t rules,
                        double* out /* [nq] */);
/* ghost_penalty_facets(): python/cutfemx/cut.py:340-380 fused with
 * facet_integration_rows(): python/cutfemx/wrappers/cut.cpp:54-115.
 * rows = (cell0, local_facet0, cell1, local_facet1), cell0 < cell1. */
int cfx_ghost_penalty_facets(cfx_cut_t cut, const char* selector,
                             const int32_t** rows, int64_t* n);
/* interior_facets_for_cells(mesh, cells) (cut.cpp:926-994 + wrappers/cut.cpp:54-115): the interior facets
 * whose two cells both belong to `cells`, as (c0, lf0, c1, lf1) rows with c0 < c1, ascending.  *rows is a
 * device array the caller releases with cfx_device_free. */
int cfx_interior_facets_for_cells(cfx_mesh_t mesh, const int32_t* cells, int64_t n, int32_t** rows, int64_t* n_rows);

/* ---- facet hosts: cutfemx::cut(mesh, level_sets, entities, entity_dim = tdim - 1)
 *      (cpp/cutfemx/cut/cut.cpp:540-591 build_entity_mesh_view, :788-830, :1022-1063 build_entity_level_sets;
 *      tests python/tests/test_cut_api.py:171-187, 349-367, 424-496) -----------------------------------------
 * The hosts are n facets of the mesh.  A facet is handed over the way DOLFINx integrates over it: as its
 * integration row (cell, local facet) for exterior facets (row_width 2) or (cell0, local facet0, cell1,
 * local facet1) for interior facets (row_width 4) -- facet_integration_rows(), python/cutfemx/wrappers/cut.cpp:54-115.
 * facet_ids[n] (NULL: 0..n-1) are the caller's facet numbers: locate_entities and rules.parent_map answer with
 * them (host_parent_index, cut.cpp:352-359).  entity_geometry [n*tdim] (NULL: the vertices of cell0 other than
 * the one opposite the facet, ascending local index) is the host vertex order of entities_to_geometry()
 * (cut.cpp:567-569): the reference simplex of host i is spanned by its vertices in that order.
 * ls_dofmap is the CELL dofmap of the P1 level sets, as for cfx_cut_create.
 * On the returned handle: cfx_cut_info reports tdim - 1 and n; cfx_cut_domain / cfx_locate_entities /
 * cfx_cut_update work per host; cfx_runtime_quadrature(cut, "phi<0" | "phi>0", order) gives one rule per cut
 * host with points on the host's reference simplex (view.tdim = tdim - 1), physical-measure weights and
 * parent_map = facet ids; "phi=0" gives the cut of the facet itself (docs/user-guide/element-classification.md
 * :138-142: "a curve on those boundary facets"): the cut point of a segment host (weight 1) or a rule along the
 * straight cut segment of a triangle host (weights carry its length); cfx_rules_physical_points works on such rules.  Everything that needs cell hosts
 * (cfx_cut_restrict, normals, ghost facets, aggregation) returns CFX_ERR_INVALID_ARGUMENT. */
int cfx_cut_create_facets(cfx_mesh_t mesh, int64_t n, const int32_t* facet_ids, const int32_t* rows, int row_width,
                          const int32_t* entity_geometry, int n_level_sets, const int32_t* ls_dofmap, int ls_ndofs_cell,
                          int64_t ls_ndofs, const double* const* ls_values, const cfx_cut_options* opt, cfx_cut_t* out);
/* exterior_facet_indices() + facet_integration_rows(): the (cell, local facet) rows of the boundary facets,
 * ascending; *rows is a device array released with cfx_device_free. */
int cfx_exterior_facets(cfx_mesh_t mesh, int32_t** rows, int64_t* n_rows);
/* rules over whole hosts -- the standard facets of a mixed [facets, rules] measure (test_cut_api.py:527-560)
 * taken through the same path: the hosts matching `selector` (NULL: all), reference points of `order`,
 * weights * facet measure. */
int cfx_full_facet_rules(cfx_cut_t facet_cut, const char* selector, int order, cfx_rules_t* out);
/* facet_runtime_quadrature_payload / interior_facet_runtime_quadrature_payload
 * (python/cutfemx/_runintgen_adapter.py:605-680): the same points seen from cell `side` (0, or 1 for interior
 * rows) of each rule's facet -- cell-hosted rules (view.tdim = tdim, parent_map = that cell, ascending) that
 * any CFX_CELL integral takes, which is how exterior-facet terms (ds: mass, source, Nitsche with the facet
 * normal as point_data) are assembled. */
int cfx_facet_rules_to_cells(cfx_rules_t facet_rules, int side, cfx_rules_t* out);

/* ---- cell aggregation (extension stabilisation): cutfemx::extensions::create_cell_aggregation,
 *      cpp/cutfemx/extensions/cell_aggregation.{h,cpp}, python/cutfemx/extensions.py -------------
 * selector: strict single level set ("phi<0" / "phi>0").  Roots = interior cells (+ cut cells whose
 * selected volume fraction >= threshold when root_policy = 1, "interior_or_well_cut"; 0 =
 * "interior_only"); every other cut cell is ill-posed and inherits the root of a facet neighbour.
 * The reference's sequential sweeps (ascending cells, a cell rooted earlier in the same sweep
 * already counts, first rooted neighbour in ascending order wins) are reproduced exactly.
 * max_iterations < 0: unlimited.  Rootless ill-posed cells raise CFX_ERR_RUNTIME unless
 * allow_rootless. */
typedef struct
{
  int64_t ncells;
  const int32_t* root_cell;         /* [ncells], -1 where unset */
  const int32_t* aggregate_id;      /* [ncells] */
  const int32_t* propagation_depth; /* [ncells] */
  const double* cut_volume_fraction; /* [ncells], 0 off the cut cells */
  const int32_t *active_cells, *cut_cells, *interior_cells, *well_posed_cells, *ill_posed_cells, *rootless_cells;
  int64_t n_active, n_cut, n_interior, n_well_posed, n_ill_posed, n_rootless;
  const int32_t* pairs;             /* [n_pairs*4] (bad, 0, root, 0): extension_pairs(), entity rows of CFX_K_EXTENSION_L2 */
  int64_t n_pairs;
} cfx_aggregation_view;
int cfx_cell_aggregation_create(cfx_cut_t cut, const char* selector, double volume_fraction_threshold,
                                int root_policy, int max_iterations, int allow_rootless, cfx_aggregation_t* out);
int cfx_cell_aggregation_view_get(cfx_aggregation_t agg, cfx_aggregation_view* view);
int cfx_cell_aggregation_destroy(cfx_aggregation_t agg);
int cfx_cut_destroy(cfx_cut_t cut);

/* ---- function space: dolfinx DofMap as read by
 *      cpp/dolfinx_custom_data/fem/assemble_matrix_impl.h:97-116 ------------- */
int cfx_space_create(cfx_mesh_t mesh, int degree, int bs, int64_t ndofs,
                     const int32_t* dofmap, int ndofs_cell, cfx_space_t* out);
/* HBM held by the mesh-static tables that the first assembly on a space builds and every later step reuses
 * (the reference holds the same information inside dolfinx::mesh::Topology / fem::DofMap): bytes[0] dof -> cells
 * incidence, [1] row stencil of a P1 space (neighbour lists, slot4, diagpos, cpos), [2] its row tiles (tile vertex
 * unions, st_loc) and the cell blocks a linear form is summed over (slots, segments, dof -> partials lists), [3] the
 * mesh's cell -> cell table (built by the first ghost-penalty query). */
int cfx_space_static_bytes(cfx_space_t V, int64_t bytes[4]);
int cfx_space_destroy(cfx_space_t V);

/* ---- forms: dolfinx_custom_data::fem::Form, Form.h:119-178, built as in
 *      python/cutfemx/wrappers/fem.cpp:124-172 ------------------------------- */
int cfx_form_create(cfx_space_t V, int rank, int n_integrals,
                    const cfx_integral* integrals, cfx_form_t* out);
/* Bilinear form with DIFFERENT test and trial spaces on one mesh (Form::function_spaces() = {V_test, V_trial},
 * Form.h:119-178; the cell loop takes both dofmaps, assemble_matrix_impl.h:68-189, the sparsity both index maps and no
 * all-rows diagonal, assembler.h:442-560).  Cell integrals (standard entities and / or runtime rules) with kernels
 * CFX_K_DIV_TEST, CFX_K_DIV_TRIAL, CFX_K_MASS, CFX_K_STIFFNESS; the element tensor is [(nd0 bs0) x (nd1 bs1)] row-major
 * (cfx_tabulate_entity), bc0 marks test-space rows, bc1 trial-space columns, cfx_apply_lifting takes trial-space data
 * and a test-space vector.  cfx_active_domain / deactivation are for square systems and refuse such forms. */
int cfx_form_create2(cfx_space_t V_test, cfx_space_t V_trial, int n_integrals, const cfx_integral* integrals,
                     cfx_form_t* out);
int cfx_form_destroy(cfx_form_t a);
/* build the form's derived tables now (row plan: cell / row marks, row classes, rule maps, facet incidence, stencil
 * masks, the row-ordered staging layout of a linear form) instead of inside the first assembly call that needs them */
int cfx_form_prepare(cfx_form_t a);
/* create_sparsity_pattern(): assembler.h:567-592 (+ :442-529, :538-560) */
int cfx_create_sparsity(cfx_form_t a, cfx_pattern_t* out);
int cfx_pattern_view_get(cfx_pattern_t p, cfx_pattern_view* view);
/* Moving-domain loops: the pattern of a space whose rows come from hash sets (degree 2, vector-valued, DG spaces)
 * remembers the previous pattern of that space; a row whose incident cells kept their marks and facet sides since then
 * copies its columns instead of being hashed and ranked again (CFX_PATTERN_REUSE=0 switches this off; the result is the
 * same pattern bit for bit).  hashed_rows: rows of this pattern that needed a hash set; reused_rows: those of them
 * that were copied.  The reference rebuilds every pattern from scratch (cut.cpp:845-868 + assembler.h:567-592). */
int cfx_pattern_reuse_stats(cfx_pattern_t p, int64_t* hashed_rows, int64_t* reused_rows);
int cfx_pattern_destroy(cfx_pattern_t p);
/* assemble_matrix(): assembler.h:690-703 -> assemble_matrix_impl.h:629-810.
 * Accumulates into values[nnz] (HBM); bc0/bc1 int8 markers or NULL. */
int cfx_assemble_matrix(cfx_form_t a, cfx_pattern_t pattern, const int8_t* bc0,
                        const int8_t* bc1, double* values);
/* A.set_value(0) + assemble_matrix(A, a, bcs) in one call -- how python/cutfemx/fem.py:886-942 and
 * python/demo/demo_poisson.py:40-60 use a matrix: created (zeros) or cleared, then assembled once.  Same
 * result as cfx_device_memset(values, 0, 8 nnz) followed by cfx_assemble_matrix; the library schedules the
 * zeroing itself and stores the rows that have one writer instead of reading them back. */
int cfx_assemble_matrix_zeroed(cfx_form_t a, cfx_pattern_t pattern, const int8_t* bc0,
                               const int8_t* bc1, double* values);
/* assemble_vector(): assembler.h:252-262 -> assemble_vector_impl.h:573-767 */
int cfx_assemble_vector(cfx_form_t L, double* b);

/* Dirichlet lifting  b <- b - alpha A (g - x0)  restricted to the Dirichlet columns, formed entity by
 * entity without the assembled matrix: dolfinx_custom_data::fem::apply_lifting / lift_bc_impl
 * (cpp/dolfinx_custom_data/fem/assemble_vector_impl.h:383-436, python/cutfemx/fem.py:604-632).
 * bc_markers / bc_values / x0 (nullable) have one entry per dof (ndofs * bs); only entities with a
 * marked column are tabulated (LiftingMode).  All integrals of the form take part: uncut cells,
 * runtime-rule cells and interior facets. */
int cfx_apply_lifting(cfx_form_t a, const int8_t* bc_markers, const double* bc_values, const double* x0,
                      double alpha, double* b);
/* b[i] = alpha (g[i] - x0[i]) on the marked dofs: dolfinx::fem::DirichletBC::set as called by
 * python/demo/demo_elasticity.py:87-93 after the lifting. */
int cfx_set_bc(int64_t n, const int8_t* bc_markers, const double* bc_values, const double* x0, double alpha,
               double* b);
/* local tensor of one entity (parity tests of local entries) */
/* zero_rows(A, tol) (python/cutfemx/fem.py:777-782): rows whose assembled entries are all <= tol in
 * magnitude, ascending; *rows is released with cfx_device_free. */
int cfx_zero_rows(cfx_pattern_t P, const double* values, double tol, int32_t** rows, int64_t* n_rows);
/* One CSR matrix from an nbr x nbc block system (the blocks of cfx_form_create2 forms): row r of block row i becomes
 * row (nrows[0] + ... + nrows[i-1]) + r and holds the entries of A[i][0], A[i][1], ... with the columns of block
 * column j shifted by ncols[0] + ... + ncols[j-1] -- the monolithic matrix that the reference assembles on a mixed
 * element (python/tests/test_assembly_stokes.py:34-95: ONE matrix on mixed_element([P2 vector, P1])), in block-ordered
 * dof numbering (all dofs of space 0, then of space 1 ...; DOLFINx interleaves them cell by cell: the two differ by that
 * permutation of rows and columns).  indptr / indices / values: nbr * nbc pointers, row-major, host or device; a NULL
 * indptr marks an empty block; values may be NULL (pattern only: out_values NULL as well) and a NULL values[k] gives
 * zeros.  Columns of a merged row ascend.  The outputs are device arrays released with cfx_device_free; at most 8
 * block columns. */
int cfx_csr_block_merge(int nbr, int nbc, const int64_t* const* indptr, const int32_t* const* indices,
                        const double* const* values, const int64_t* nrows, const int64_t* ncols, int64_t** out_indptr,
                        int32_t** out_indices, double** out_values, int64_t* out_nnz);
/* The same matrix in another dof numbering: row r becomes row row_perm[r], column c becomes column col_perm[c], the
 * columns of every row ascend again.  With cfx_csr_block_merge this gives the monolithic matrix of a mixed element in
 * the numbering the CALLER holds -- DOLFINx numbers the dofs of mixed_element([P2 vector, P1]) cell by cell, interleaved
 * (python/tests/test_assembly_stokes.py:34-95): row_perm[block-ordered dof] = mixed dof, built from the caller's mixed
 * dofmap (cutfemx_amd.fem.MixedSpace).  Both maps must be permutations (checked); arrays host or device; values / out_values
 * may be NULL (pattern only); rows of at most 2048 entries.  Outputs are device arrays released with cfx_device_free. */
int cfx_csr_permute(int64_t nrows, int64_t ncols, const int64_t* indptr, const int32_t* indices, const double* values,
                    const int32_t* row_perm, const int32_t* col_perm, int64_t** out_indptr, int32_t** out_indices,
                    double** out_values);
int cfx_tabulate_entity(cfx_form_t a, int integral, int64_t index, int use_rule, double* Ae);

/* ---- user-supplied integrands: the runtime-generated kernel of a form
 *      (cpp/dolfinx_custom_data/fem/Form.h:59-75: std::function tabulate_tensor per integral,
 *      python/cutfemx/_runintgen_adapter.py:181-217: runintgen / FFCx generate and JIT-compile it).  A GPU engine cannot
 *      call a CPU function pointer per entity; it compiles the integrand's SOURCE for gfx950 with hipRTC instead.
 *      `source` defines a device function `name` with the argument list of a UFCx tabulate_tensor kernel extended by
 *      the rule, as runintgen passes it through custom_data:
 *
 *        __device__ void name(double* A,                     local tensor, zero on entry: [ND][ND] row-major (rank 2) or [ND]
 *                             const double* w,               packed coefficient (the ND cell dofs of `coefficient`) or NULL
 *                             const double* c,               constants: cfx_integral.params[8]
 *                             const double* coordinate_dofs, vertex coordinates of the cell, [(TDIM+1)][3]
 *                             int nq, const double* points,  rule of this entity: [nq][TDIM] parent-reference coordinates
 *                             const double* weights,         [nq] physical-measure weights (standard entities: reference
 *                                                            weights x |det J| of the cell; runtime rules: as stored)
 *                             const double* point_data);     [nq][point_stride] per-point data of the rules, or NULL
 *
 *      CFX_TDIM and CFX_ND (dofs per cell) are defined at compile time; helpers in scope: cfx_tabulate(X, N, dN),
 *      cfx_tabulate_p1 / _p2, cfx_inverse_jacobian(coordinate_dofs, K) -> det J, cfx_cell_diameter(coordinate_dofs).
 *      The id returned in *kernel_id goes into cfx_integral.kernel (cell integrals of Lagrange spaces of degree 1 or 2,
 *      scalar or vector-valued, standard entities and / or runtime rules).  Stage 1 (one thread per entity) stages the local tensors in HBM; the
 *      row gather -- or the entity-parallel scatter -- then assembles them like those of the built-in integrands that
 *      are not formed in line.  A source that does not compile is CFX_ERR_INVALID_ARGUMENT with the compiler log in
 *      cfx_last_error(); compilation needs no GPU (hipRTC targets gfx950 explicitly), loading the code object does. */
int cfx_integrand_register(const char* name, const char* source, int rank, int* kernel_id);
int cfx_integrand_compile(int kernel_id, int tdim, int ndofs_cell); /* compile a (tdim, dofs per cell) variant now */
/* Vector-valued spaces (bs > 1): CFX_BS = the space's block size and CFX_NDB = CFX_ND * CFX_BS are defined as well;
 * the local tensor is [NDB][NDB] (or [NDB]) with entry (dof i, component a) at i * CFX_BS + a -- the layout of the
 * reference's blocked dofmaps (assemble_matrix_impl.h:137-149).  `w` holds the scalar coefficient's ND dofs for bilinear
 * forms and the ND x CFX_BS dofs of a Function of the form's space for linear forms.
 *
 * Interior-facet integrands -- the reference's kernel call with entity_local_index = {lf0, lf1} and the macro layout
 * [[00, 01], [10, 11]] (assemble_matrix_impl.h:528-542):
 *
 *      __device__ void NAME(double* A,                     macro tensor, zero on entry: [2 NDB][2 NDB] row-major over
 *                                                          [cell 0 dofs, cell 1 dofs]
 *                           const double* w,               packed coefficient of both cells [2][ND], or NULL
 *                           const double* c,               cfx_integral.params[8]
 *                           const double* coordinate_dofs, [2][(TDIM+1)][3]: cell 0, then cell 1
 *                           const int* entity_local_index, {lf0, lf1}
 *                           int nq,
 *                           const double* points0,         [nq][TDIM]: the facet's points in the reference cell of cell 0
 *                           const double* points1,         ... the same physical points in the reference cell of cell 1
 *                           const double* weights);        [nq] physical (facet-measure) weights
 *
 *      over the (c0, lf0, c1, lf1) rows of a CFX_INTERIOR_FACET integral (standard facets; bilinear forms; at most 24
 *      macro dofs: scalar spaces of degree 1 or 2, vector spaces of degree 1).  The points are those of the engine's
 *      reference facet rule of cfx_integral.qdegree, pushed forward from cell 0's facet and pulled back to both cells.
 *      Sparsity, row gather, deactivation work from the entity lists and are unchanged.
 * cfx_integrand_register_variant: registration that validates the source against the named (tdim, dofs per cell, block
 * size) variant instead of (3, 4, 1) -- for sources that only compile for 2-D / degree 2 / vector spaces. */
int cfx_integrand_register_facet(const char* name, const char* source, int* kernel_id);
int cfx_integrand_register_variant(const char* name, const char* source, int rank, int facet, int tdim, int ndofs_cell, int bs,
                                   int* kernel_id);
int cfx_integrand_compile_bs(int kernel_id, int tdim, int ndofs_cell, int bs);

/* ---- deactivation: cpp/cutfemx/fem/deactivate.h:387-418 ------------------- */
/* active_domain(): the two indicators (active cells, active dofs) are the marks of the form's row plan; deactivation
 * works from the marks.  The LISTS (ActiveDomain::active_cells / inactive_dofs) are compacted on the first
 * cfx_active_view call that asks for them; with active_cells = n_active = inactive_dofs = NULL the call returns the
 * number of inactive dofs alone, which needs no list.  The lists are made from the form's plan, whose facet rows alias
 * the form's own interior-facet entity array: ask for them while the form and that array are alive (a released list
 * is detected and refused with CFX_ERR_RUNTIME; the deactivation calls themselves need nothing but the marks). */
int cfx_active_domain(cfx_form_t a, cfx_active_t* out);
int cfx_active_view(cfx_active_t d, const int32_t** active_cells, int64_t* n_active,
                    const int32_t** inactive_dofs, int64_t* n_inactive);
int cfx_deactivate_outside(cfx_active_t d, cfx_pattern_t pattern, double* values,
                           double* b /* or NULL */, double diagonal, double rhs_value);
int cfx_active_destroy(cfx_active_t d);

/* ---- float32 instantiation of the boundary --------------------------------------------------------------
 * python/cutfemx/wrappers/fem.cpp:490-500 declares the runtime assembly for <T, U> in {float, double}^2 and
 * wrappers/cut.cpp:403-407 the cut API for float and double: T = scalar type of MatrixCSR / Vector / Function,
 * U = geometry type.  The *_f32 entry points below are the <float, float> row (a mixed <float, double> or
 * <double, float> caller combines them with the fp64 ones: handles are shared, only the containers differ).
 * f32 is the type of the containers that cross the boundary; every kernel computes in fp64: inputs are widened
 * once on upload (exact), results rounded to nearest once on the way out -- see cutfemx_amd/csrc/cfx_f32.hip for
 * why this engine does not carry a second set of f32-register kernels.  Pointers may be host or HBM as for the
 * fp64 entry points; the accumulate semantics are the same (values += A, b += L, rounded once). */
typedef struct
{
  int32_t tdim, gdim;
  int64_t nq, nr;
  const float* points;       /* [nq*tdim] (HBM), rounded copy owned by the rules handle */
  const float* weights;      /* [nq] (HBM) */
  const int32_t* offsets;    /* [nr+1] (HBM) */
  const int32_t* parent_map; /* [nr] (HBM) */
  int32_t host_width;
  int32_t reserved;
  const int32_t* host_rows;
  const int32_t* host_verts;
} cfx_rules_view_f32;
/* mesh with float32 coordinates (U = float): x [nnodes*3] */
int cfx_mesh_create_f32(int tdim, int gdim, int64_t nnodes, const float* x, int64_t ncells, const int32_t* conn,
                        int cell_stride, cfx_mesh_t* out);
/* cut() / update() with float32 level-set Functions (cut.h:104-181 for T = float).  The widened copies belong to
 * the cut: a caller that changes its f32 values calls cfx_cut_update_f32 with the pointers again (NULL is refused:
 * there is no aliased array to re-read, unlike cfx_cut_update).  Likewise float32 coefficient / point_data arrays
 * are widened once by the binding when a form is created: rebuild the form after changing them in place. */
int cfx_cut_create_f32(cfx_mesh_t mesh, int n_level_sets, const int32_t* ls_dofmap, int ls_ndofs_cell,
                       int64_t ls_ndofs, const float* const* ls_values, const cfx_cut_options* opt, cfx_cut_t* out);
int cfx_cut_update_f32(cfx_cut_t cut, const float* const* ls_values);
/* RuntimeQuadrature<float> fields (runtime_quadrature.h:223-231) */
int cfx_rules_create_f32(cfx_mesh_t mesh, int tdim, int64_t nq, int64_t nr, const float* points, const float* weights,
                         const int32_t* offsets, const int32_t* parent_map, cfx_rules_t* out);
int cfx_rules_view_get_f32(cfx_rules_t r, cfx_rules_view_f32* view);
int cfx_rules_physical_points_f32(cfx_rules_t r, float* out);
int cfx_evaluate_normals_f32(cfx_cut_t cut, int level_set, cfx_rules_t rules, float sign, float* out);
int cfx_evaluate_values_f32(cfx_cut_t cut, int level_set, cfx_rules_t rules, float* out);
/* f32 per-point data / coefficient dof values for a cfx_integral: *out is an fp64 HBM copy to put into
 * cfx_integral.point_data / .coefficient, released with cfx_device_free after the form is destroyed */
int cfx_widen_f32(const float* src, int64_t n, double** out);
/* assemble_matrix / assemble_vector / apply_lifting / set_bc / zero_rows / deactivate_outside for T = float */
int cfx_assemble_matrix_f32(cfx_form_t a, cfx_pattern_t pattern, const int8_t* bc0, const int8_t* bc1, float* values);
int cfx_assemble_matrix_zeroed_f32(cfx_form_t a, cfx_pattern_t pattern, const int8_t* bc0, const int8_t* bc1,
                                   float* values);
int cfx_assemble_vector_f32(cfx_form_t L, float* b);
int cfx_apply_lifting_f32(cfx_form_t a, const int8_t* bc_markers, const float* bc_values, const float* x0, float alpha,
                          float* b);
int cfx_set_bc_f32(int64_t n, const int8_t* bc_markers, const float* bc_values, const float* x0, float alpha, float* b);
int cfx_zero_rows_f32(cfx_pattern_t P, const float* values, float tol, int32_t** rows, int64_t* n_rows);
int cfx_deactivate_outside_f32(cfx_active_t d, cfx_pattern_t pattern, float* values, float* b /* or NULL */,
                               float diagonal, float rhs_value);
/* ---- complex128 scalars: the <std::complex<double>, double> rows of python/cutfemx/wrappers/fem.cpp:490-500;
 *      invariants python/tests/test_complex_assembly.py:24-95.  Values, vectors and Dirichlet data are interleaved
 *      (re, im) doubles; geometry, rules and integrands stay real.  `scales` = the complex constant of every integral
 *      of the form ([2 * n_integrals] re, im; NULL: all 1): kappa in `kappa inner(grad u, grad v) dx`.  A complex
 *      coefficient FUNCTION enters linearly: assemble a form of its real parts with scale s and one of its imaginary
 *      parts with scale i s into the same array (the Python binding does).  Same accumulate semantics as the float64
 *      entry points; zero_first != 0 = MatrixCSR.set_value(0) first. ---------------------------------------------- */
int cfx_assemble_matrix_c128(cfx_form_t a, cfx_pattern_t pattern, const int8_t* bc0, const int8_t* bc1, const double* scales,
                             int zero_first, double* values /* [2 nnz] */);
int cfx_assemble_vector_c128(cfx_form_t L, const double* scales, double* b /* [2 n] */);
int cfx_apply_lifting_c128(cfx_form_t a, const int8_t* bc_markers, const double* bc_values /* [2 n1] */,
                           const double* x0 /* [2 n1] or NULL */, double alpha_re, double alpha_im, const double* scales,
                           double* b /* [2 n] */);
int cfx_set_bc_c128(int64_t n, const int8_t* bc_markers, const double* bc_values, const double* x0, double alpha_re,
                    double alpha_im, double* b);
int cfx_deactivate_outside_c128(cfx_active_t domain, cfx_pattern_t pattern, double* values, double* b, double diag_re,
                                double diag_im, double rhs_re, double rhs_im);
/* ---- complex64: the <std::complex<float>, float> rows of the same table.  Values, vectors and Dirichlet data are
 *      interleaved (re, im) float32 (geometry and level sets then usually come through the *_f32 constructors); the
 *      constants stay double; every entry is formed in fp64 as above and rounded to float32 once (accumulating calls:
 *      widen, add, round). ------------------------------------------------------------------------------------------ */
int cfx_assemble_matrix_c64(cfx_form_t a, cfx_pattern_t pattern, const int8_t* bc0, const int8_t* bc1, const double* scales,
                            int zero_first, float* values /* [2 nnz] */);
int cfx_assemble_vector_c64(cfx_form_t L, const double* scales, float* b /* [2 n] */);
int cfx_apply_lifting_c64(cfx_form_t a, const int8_t* bc_markers, const float* bc_values /* [2 n1] */,
                          const float* x0 /* [2 n1] or NULL */, double alpha_re, double alpha_im, const double* scales,
                          float* b /* [2 n] */);
int cfx_set_bc_c64(int64_t n, const int8_t* bc_markers, const float* bc_values, const float* x0, double alpha_re,
                   double alpha_im, float* b);
int cfx_deactivate_outside_c64(cfx_active_t domain, cfx_pattern_t pattern, float* values, float* b, double diag_re,
                               double diag_im, double rhs_re, double rhs_im);


/* ---- multi-GPU exchange steps (one rank per GPU; RCCL send/recv over xGMI between the ranks that share dofs) ----
 * The reference's collectives on this path (DOLFINx index maps, MPI neighbourhood exchanges):
 *   phi.x.scatter_forward()                      python/demo/demo_poisson.py:157       -> cfx_dist_scatter_forward
 *   A.scatter_reverse(); b.scatter_reverse(add)  python/demo/demo_poisson.py:51-54     -> cfx_dist_scatter_reverse_matrix / _add
 *   indicator.scatter_rev(plus) + scatter_fwd    cpp/cutfemx/fem/deactivate.h:180-181  -> cfx_dist_indicator_or + _forward
 * A communicator is either RCCL (cfx_dist_unique_id on one rank, the 128 bytes handed to all ranks by the launcher's
 * own channel -- MPI_Bcast, torch.distributed, a file -- then cfx_dist_comm_create on every rank after cfx_init on its
 * GPU; librccl.so.1 is loaded on first use) or host-staged (cfx_dist_comm_create_host: the library moves the slices
 * through pinned host memory and the caller's callback -- MPI_Sendrecv, gloo -- moves the bytes) or the caller's own
 * device transport (cfx_dist_comm_create_device: the callback is handed the device segments). */
typedef struct cfx_comm_s* cfx_comm_t;
#define CFX_DIST_ID_BYTES 128 /* sizeof(ncclUniqueId) */
/* n messages out and n in: to / from peers[i], send[i] / recv[i] are host buffers of send_bytes[i] / recv_bytes[i] bytes
 * (either may be 0); returns 0 on success.  All n exchanges may be posted together (they do not depend on each other). */
typedef int (*cfx_host_exchange_fn)(void* user, int n, const int32_t* peers, const void* const* send,
                                    const int64_t* send_bytes, void* const* recv, const int64_t* recv_bytes);
int cfx_dist_unique_id(char id[CFX_DIST_ID_BYTES]);
int cfx_dist_comm_create(int world, int rank, const char id[CFX_DIST_ID_BYTES], cfx_comm_t* out);
int cfx_dist_comm_create_host(int world, int rank, cfx_host_exchange_fn fn, void* user, cfx_comm_t* out);
/* The caller's own GPU-to-GPU transport (a GPU-aware MPI_Isend / MPI_Irecv pair per peer, the launcher's RCCL process
 * group -- torch.distributed on the nccl backend): same contract as cfx_host_exchange_fn with send[i] / recv[i] DEVICE
 * buffers.  The library's stream has been drained when the callback runs; every transfer must have completed (or be
 * ordered before later work on the library's stream) when it returns.  Replaces MPI neighbourhood exchanges of
 * dolfinx::common::Scatterer (python/demo/demo_poisson.py:51-54,157) for callers that already hold a communicator. */
typedef int (*cfx_device_exchange_fn)(void* user, int n, const int32_t* peers, const void* const* send,
                                      const int64_t* send_bytes, void* const* recv, const int64_t* recv_bytes);
int cfx_dist_comm_create_device(int world, int rank, cfx_device_exchange_fn fn, void* user, cfx_comm_t* out);
int cfx_dist_comm_info(cfx_comm_t comm, int* world, int* rank, int* is_rccl);
int cfx_dist_comm_destroy(cfx_comm_t comm);
/* What this rank sends to and receives from one peer in one exchange step, in elements of the array: a contiguous
 * range (offset, count) -- a vertex plane of a slab partition, sent in place -- or, when the index pointer is set, the
 * `count` entries listed by a DEVICE int32 index list (the shared / ghost index lists of an index map; packed and
 * unpacked by a kernel; duplicate-free).  The peer's send count must equal this rank's receive count. */
typedef struct
{
  int32_t peer;
  int32_t reserved;
  int64_t send_offset, send_count;
  const int32_t* send_index; /* or NULL */
  int64_t recv_offset, recv_count;
  const int32_t* recv_index; /* or NULL */
} cfx_dist_exchange;
/* x[recv] = the peer's x[send]: la::Vector::scatter_forward (owners send, ghosts receive) */
int cfx_dist_scatter_forward(cfx_comm_t comm, double* x, int n, const cfx_dist_exchange* ex);
/* x[recv] += the peer's x[send]: la::Vector::scatter_reverse(add) (ghost contributions go to the owners); also serves
 * CSR value arrays when the caller has the value offsets at hand */
int cfx_dist_scatter_reverse_add(cfx_comm_t comm, double* x, int n, const cfx_dist_exchange* ex);
/* la::MatrixCSR::scatter_reverse for row blocks: rows [send_row_lo, send_row_hi) of this rank's pattern are added to
 * rows [recv_row_lo, recv_row_hi) of the peer's -- the same global rows under the two local numberings, built from the
 * same entities on both ranks so that they hold the same columns in the same order (DOLFINx keeps the sparsity of
 * ghost rows on both sides); the slices of the value array are exchanged without packing and their lengths compared. */
typedef struct
{
  int32_t peer;
  int32_t reserved;
  int64_t send_row_lo, send_row_hi;
  int64_t recv_row_lo, recv_row_hi;
} cfx_dist_row_exchange;
int cfx_dist_scatter_reverse_matrix(cfx_comm_t comm, cfx_pattern_t pattern, double* values, int n,
                                    const cfx_dist_row_exchange* ex);
/* 0/1 indicators (int8): indicator[recv] |= the peer's indicator[send], then the owners' values back to the ghosts */
int cfx_dist_indicator_or(cfx_comm_t comm, int8_t* indicator, int n, const cfx_dist_exchange* ex);
int cfx_dist_indicator_forward(cfx_comm_t comm, int8_t* indicator, int n, const cfx_dist_exchange* ex);

#ifdef __cplusplus
}
#endif
#endif /* CUTFEMX_AMD_H */
