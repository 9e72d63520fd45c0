"""Cell aggregation and the extension-penalty stabilisation: the Python surface of
python/cutfemx/extensions.py on the HIP engine (cpp/cutfemx/extensions/).

    agg = create_cell_aggregation(cut_data, "phi<0", 0.3)
    A   = extension_penalty_matrix(V, cut_data, agg, beta, degree)            # its own matrix
    a   = fem.form([..., extension_penalty_integral(agg, beta, degree)], V)   # or a term of a form

The penalty is an interior-facet-TYPE integral over the (bad, root) pairs (integrand
`EXTENSION_L2`), so sparsity, row-gather assembly, lifting and deactivation treat it like the
ghost-penalty term it replaces.
"""
from __future__ import annotations

import ctypes as C
import numbers
from dataclasses import dataclass
from typing import Any

import numpy as np

from . import _lib, fem
from .cut import CutData, _engine_selector

_POLICIES = {"interior_only": 0, "interior_or_well_cut": 1}


class CellAggregation:
    """cutfemx.extensions.CellAggregation (cell_aggregation.h:24-38); arrays live in HBM and are
    downloaded on first access."""

    def __init__(self, handle, cut_data: CutData):
        self._h, self.cut_data = handle, cut_data
        v = _lib.AggregationView()
        _lib.check(_lib.lib().cfx_cell_aggregation_view_get(handle, C.byref(v)))
        self._view, self._cache = v, {}

    def _get(self, name, ptr, n, dtype):
        if name not in self._cache:
            self._cache[name] = _lib.download(ptr, n, dtype)
        return self._cache[name]

    active_cells = property(lambda s: s._get("active", s._view.active_cells, s._view.n_active, np.int32))
    cut_cells = property(lambda s: s._get("cut", s._view.cut_cells, s._view.n_cut, np.int32))
    interior_cells = property(lambda s: s._get("interior", s._view.interior_cells, s._view.n_interior, np.int32))
    well_posed_cells = property(lambda s: s._get("well", s._view.well_posed_cells, s._view.n_well_posed, np.int32))
    ill_posed_cells = property(lambda s: s._get("ill", s._view.ill_posed_cells, s._view.n_ill_posed, np.int32))
    rootless_cells = property(lambda s: s._get("rootless", s._view.rootless_cells, s._view.n_rootless, np.int32))
    root_cell = property(lambda s: s._get("root", s._view.root_cell, s._view.ncells, np.int32))
    aggregate_id = property(lambda s: s._get("agg", s._view.aggregate_id, s._view.ncells, np.int32))
    propagation_depth = property(lambda s: s._get("depth", s._view.propagation_depth, s._view.ncells, np.int32))
    cut_volume_fraction = property(lambda s: s._get("frac", s._view.cut_volume_fraction, s._view.ncells, np.float64))

    @property
    def num_pairs(self) -> int:
        return int(self._view.n_pairs)

    @property
    def pairs(self) -> np.ndarray:
        """(bad, 0, root, 0) rows of the rooted ill-posed cells (extension_pairs, extension_penalty.cpp:373-392)."""
        return self._get("pairs", self._view.pairs, 4 * self._view.n_pairs, np.int32).reshape(-1, 4)

    def pair_rows(self) -> fem.FacetRows:
        """The pairs as device-resident entity rows of an EXTENSION_L2 integral."""
        return fem.FacetRows(self._view.pairs, self.num_pairs, owner=self)

    def __del__(self):
        try:
            if self._h:
                _lib.load().cfx_cell_aggregation_destroy(self._h)
                self._h = None
        except Exception:
            pass


def create_cell_aggregation(cut_data: CutData, selector: str, volume_fraction_threshold: float, *,
                            root_policy: str = "interior_or_well_cut", max_iterations: int = -1,
                            allow_rootless: bool = False) -> CellAggregation:
    """python/cutfemx/extensions.py:131-165 -> cell_aggregation.cpp:143-270."""
    if not isinstance(cut_data, CutData):
        raise TypeError("create_cell_aggregation expects a cutfemx.CutData object")
    if root_policy not in _POLICIES:
        raise ValueError("Unknown root policy. Expected 'interior_only' or 'interior_or_well_cut'.")
    h = C.c_void_p()
    _lib.check(_lib.lib().cfx_cell_aggregation_create(
        cut_data._h, _engine_selector(cut_data, selector), C.c_double(volume_fraction_threshold),
        _POLICIES[root_policy], int(max_iterations), int(bool(allow_rootless)), C.byref(h)))
    return CellAggregation(h, cut_data)


@dataclass
class ExtensionPenaltyTerm:
    """One extension-penalty contribution (python/cutfemx/extensions.py:100-128)."""
    V: Any
    beta: Any
    quadrature_degree: int
    product: str = "L2"
    cut_data: Any | None = None
    aggregation: Any | None = None

    def __post_init__(self):
        if self.product != "L2":
            raise NotImplementedError("Only L2 extension penalty terms are implemented in v1")

    def with_domain(self, cut_data: CutData, aggregation: CellAggregation) -> "ExtensionPenaltyTerm":
        return ExtensionPenaltyTerm(self.V, self.beta, self.quadrature_degree, self.product, cut_data, aggregation)


def _unpack(V, beta, quadrature_degree, who):
    if isinstance(V, ExtensionPenaltyTerm):
        if beta is not None or quadrature_degree is not None:
            raise ValueError("Do not pass beta/quadrature_degree when using ExtensionPenaltyTerm")
        V, beta, quadrature_degree = V.V, V.beta, V.quadrature_degree
    if beta is None or quadrature_degree is None:
        raise TypeError(f"{who} requires beta and quadrature_degree")
    return V, beta, quadrature_degree


def extension_penalty_integral(aggregation: CellAggregation, beta, quadrature_degree: int) -> fem.Integral:
    """The penalty as one integral of a bilinear form.  `beta`: scalar, or cellwise values (one per
    background cell, evaluated on the bad cell of each pair)."""
    if not isinstance(aggregation, CellAggregation):
        raise TypeError("extension_penalty_integral expects a CellAggregation object")
    if isinstance(beta, numbers.Number):
        return fem.Integral(fem.EXTENSION_L2, facets=aggregation.pair_rows(), params=(float(beta),),
                            qdegree=int(quadrature_degree))
    values = np.ascontiguousarray(beta, dtype=np.float64).ravel()
    if values.size != aggregation._view.ncells:
        raise ValueError("cellwise beta must hold one value per background cell")
    return fem.Integral(fem.EXTENSION_L2, facets=aggregation.pair_rows(), params=(1.0,),
                        qdegree=int(quadrature_degree), point_data=values[aggregation.pairs[:, 0]])


def create_extension_penalty_matrix(V, cut_data: CutData, aggregation: CellAggregation):
    """Zero matrix with the pair sparsity (+ the all-rows diagonal)."""
    if isinstance(V, ExtensionPenaltyTerm):
        V = V.V
    if not isinstance(cut_data, CutData):
        raise TypeError("create_extension_penalty_matrix expects a cutfemx.CutData object")
    if not isinstance(aggregation, CellAggregation):
        raise TypeError("create_extension_penalty_matrix expects a CellAggregation object")
    return fem.create_matrix(fem.form([extension_penalty_integral(aggregation, 1.0, 0)], V))


def assemble_extension_penalty(A, V, cut_data: CutData, aggregation: CellAggregation, beta=None,
                               quadrature_degree=None):
    """Add the penalty into an existing matrix whose pattern contains the pair couplings."""
    if not isinstance(cut_data, CutData):
        raise TypeError("assemble_extension_penalty expects a cutfemx.CutData object")
    if not isinstance(aggregation, CellAggregation):
        raise TypeError("assemble_extension_penalty expects a CellAggregation object")
    V, beta, quadrature_degree = _unpack(V, beta, quadrature_degree, "assemble_extension_penalty")
    fem.assemble_matrix(fem.form([extension_penalty_integral(aggregation, beta, quadrature_degree)], V), A=A)
    return A


def extension_penalty_matrix(V, cut_data: CutData, aggregation: CellAggregation, beta=None, quadrature_degree=None):
    if not isinstance(cut_data, CutData):
        raise TypeError("extension_penalty_matrix expects a cutfemx.CutData object")
    if not isinstance(aggregation, CellAggregation):
        raise TypeError("extension_penalty_matrix expects a CellAggregation object")
    V, beta, quadrature_degree = _unpack(V, beta, quadrature_degree, "extension_penalty_matrix")
    return fem.assemble_matrix(fem.form([extension_penalty_integral(aggregation, beta, quadrature_degree)], V))
