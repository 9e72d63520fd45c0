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

from . import fem
from .cut import ghost_penalty_facets, locate_entities_device, normal, runtime_quadrature


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
    volume_rules = runtime_quadrature(cut_data, "phi<0", order)
    interface_rules = runtime_quadrature(cut_data, "phi=0", order)
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
