"""cutfemx_amd: MI355X-native cut-FEM quadrature-and-assembly engine.

Drop-in for the CutFEMx hot path (classification -> sub-triangulation ->
runtime quadrature -> local tensors + ghost penalty -> CSR) with the reference's
API names (python/cutfemx/__init__.py:19-48).  All compute runs in hand-written
HIP kernels for gfx950 behind the C ABI of include/cutfemx_amd.h; importing the
package needs no GPU, calling into it does.
"""
from . import fem
from . import extensions
from .cut import (CutData, FacetRows, RuntimeQuadratureRules, cut, exterior_facets, full_cell_rules, full_facet_rules,
                  ghost_penalty_facets, interior_facets_for_cells,
                  level_set_value, locate_entities, locate_entities_device, normal, runtime_quadrature,
                  runtime_quadratures, update)
from .mesh import Function, FunctionSpace, Mesh, box_lagrange2_dofmap, box_mesh_arrays, lagrange_dofmap
from .step import forget as forget_step_history, run_step, set_margin as set_step_margin, step

__all__ = [
    "CutData", "FacetRows", "RuntimeQuadratureRules", "cut", "update", "locate_entities",
    "locate_entities_device", "runtime_quadrature", "runtime_quadratures", "full_cell_rules",
    "ghost_penalty_facets", "interior_facets_for_cells", "exterior_facets", "full_facet_rules", "normal", "level_set_value", "Mesh", "FunctionSpace", "Function",
    "box_mesh_arrays", "box_lagrange2_dofmap", "lagrange_dofmap", "fem", "extensions",
    "step", "run_step", "set_step_margin", "forget_step_history",
]
