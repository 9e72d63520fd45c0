"""Cut Poisson problem of python/demo/demo_poisson.py:138-203 expressed with the
engine's integral descriptors: the workload of BASELINE.json configs 1-3.

    a = (grad u, grad v)_Omega
        + (-dn(u) v - dn(v) u + gamma/h u v)_Gamma
        + gamma_g h_avg ([dn u], [dn v])_{F_G}
    L = (f, v)_Omega + (-dn(v) g + gamma/h g v)_Gamma,
    f = gdim pi^2 prod sin(pi x_i),  g = u_exact = prod sin(pi x_i)
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import fem
from .cut import (cut, ghost_penalty_facets, interior_facets_for_cells, locate_entities, locate_entities_device, normal,
                  runtime_quadrature, runtime_quadratures)
from .mesh import FunctionSpace


@dataclass
class PoissonSystem:
    cut_data: object
    inside_cells: tuple
    volume_rules: object
    interface_rules: object
    ghost_facets: object
    normals: object
    a: object
    L: object


def build_forms(V, cut_data, *, order: int = 4, gamma: float = 40.0, gamma_g: float = 0.1,
                ghost_penalty: bool = True, source_degree: int = 4) -> PoissonSystem:
    """locate -> runtime rules -> normals -> forms; everything stays in HBM."""
    inside = locate_entities_device(cut_data, "phi<0")
    rules = runtime_quadratures(cut_data, ["phi<0", "phi=0"], order)     # one pass over the cut cells for both
    volume_rules, interface_rules = rules["phi<0"], rules["phi=0"]
    normals = normal(cut_data, interface_rules, device=True)
    ghost = ghost_penalty_facets(cut_data, "phi<0") if ghost_penalty else None
    P = V.degree
    a_int = [
        fem.Integral(fem.STIFFNESS, cells=inside, rules=volume_rules, qdegree=2 * (P - 1)),
        fem.Integral(fem.NITSCHE, rules=interface_rules, point_data=normals, params=(gamma,)),
    ]
    if ghost is not None and ghost.size > 0:
        a_int.append(fem.Integral(fem.GHOST_GRADJUMP, facets=ghost, params=(gamma_g,), qdegree=2 * (P - 1)))
    L_int = [
        fem.Integral(fem.SOURCE, cells=inside, rules=volume_rules, params=(fem.F_POISSON_RHS, 1.0),
                     qdegree=source_degree),
        fem.Integral(fem.NITSCHE_RHS, rules=interface_rules, point_data=normals,
                     params=(gamma, fem.F_SINPROD, 1.0)),
    ]
    return PoissonSystem(cut_data, inside, volume_rules, interface_rules, ghost, normals,
                         fem.form(a_int, V), fem.form(L_int, V))


@dataclass
class DGPoissonSystem:
    function_space: object
    cell_cut: object
    skeleton_cut: object
    skeleton: object        # interior facets of the active mesh, (c0, lf0, c1, lf1) rows
    omega_facets: object    # ... those inside the domain (standard dS entities)
    facet_rules: object     # runtime rules of the cut skeleton facets (hosted by the facets)
    volume_rules: object
    interface_rules: object
    a: object
    L: object


def build_dg_forms(level_set, degree: int = 1, *, order: int = 4, sigma: float = 10.0, sigma_gamma: float = 20.0,
                   gamma_g: float = 0.1) -> DGPoissonSystem:
    """Cut DG Poisson problem of python/demo/demo_dg_poisson.py:205-277: the DG space of `degree` on the mesh of
    `level_set` (every cell owns its dofs), volume terms on [inside cells, cut-cell rules], the symmetric
    interior penalty on dS(subdomain_data=[skeleton facets inside, rules of the cut skeleton facets]) -- the
    skeleton is cut with the FACETS as hosts -- Nitsche on the interface and the ghost penalty."""
    mesh = level_set.function_space.mesh
    tdim = mesh.tdim
    nd = {1: tdim + 1, 2: (tdim + 1) * (tdim + 2) // 2}[degree]
    ndofs = mesh.num_cells * nd
    V = FunctionSpace(mesh, degree, dofmap=np.arange(ndofs, dtype=np.int32).reshape(mesh.num_cells, nd), ndofs=ndofs)
    cell_cut = cut(level_set)
    inside = locate_entities(cell_cut, "phi<0")
    active = locate_entities(cell_cut, "phi<=0")
    vol = runtime_quadrature(cell_cut, "phi<0", order)
    itf = runtime_quadrature(cell_cut, "phi=0", order)
    nrm = normal(cell_cut, itf)
    skeleton = interior_facets_for_cells(mesh, active)
    skeleton_cut = cut(level_set, skeleton, tdim - 1)
    omega_facets = skeleton.rows[locate_entities(skeleton_cut, "phi<0")]
    facet_rules = runtime_quadrature(skeleton_cut, "phi<0", order)
    ghost = ghost_penalty_facets(cell_cut, "phi<0")
    s2 = degree * degree
    a = [fem.Integral(fem.STIFFNESS, cells=inside, rules=vol, qdegree=2 * (degree - 1)),
         fem.Integral(fem.SIP, facets=omega_facets, rules=facet_rules, params=(sigma * s2,), qdegree=2 * degree),
         fem.Integral(fem.NITSCHE, rules=itf, point_data=nrm, params=(sigma_gamma * s2,))]
    if ghost.size > 0:
        a.append(fem.Integral(fem.GHOST_GRADJUMP, facets=ghost, params=(gamma_g,), qdegree=2 * (degree - 1)))
    L = [fem.Integral(fem.SOURCE, cells=inside, rules=vol, params=(fem.F_POISSON_RHS, 1.0), qdegree=4),
         fem.Integral(fem.NITSCHE_RHS, rules=itf, point_data=nrm, params=(sigma_gamma * s2, fem.F_SINPROD, 1.0))]
    return DGPoissonSystem(V, cell_cut, skeleton_cut, skeleton, omega_facets, facet_rules, vol, itf,
                           fem.form(a, V), fem.form(L, V))
