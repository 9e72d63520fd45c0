"""cut / update / locate_entities / runtime_quadrature / ghost_penalty_facets:
the Python surface of python/cutfemx/cut.py backed by the HIP engine.

Same names, argument meaning and error classes as the reference; array results
keep the reference layouts (`RuntimeQuadratureRules`: points[nq,tdim] in parent
reference coordinates, physical weights[nq], int32 offsets[nr+1],
parent_map[nr], kind="per_entity").  Results live in HBM; numpy views are
downloaded on first access.
"""
from __future__ import annotations

import ctypes as C
import re
from collections.abc import Sequence

import numpy as np

from . import _lib
from .mesh import Function, Mesh


class RuntimeQuadratureRules:
    """runintgen.QuadratureRules(kind="per_entity") stand-in backed by HBM arrays
    (python/cutfemx/cut.py:22-57, python/cutfemx/wrappers/cut.cpp:178-240)."""

    kind = "per_entity"

    def __init__(self, handle, mesh: Mesh, dtype=np.float64):
        self._h = handle
        self.mesh = mesh
        self.dtype = np.dtype(dtype)       # RuntimeQuadrature<T>: float32 rules come from a float32 cut / arrays
        v = _lib.RulesView()
        if self.dtype == np.dtype(np.float32):   # same layout, points / weights are the rounded float copies
            _lib.check(_lib.lib().cfx_rules_view_get_f32(handle, C.byref(v)))
        else:
            _lib.check(_lib.lib().cfx_rules_view_get(handle, C.byref(v)))
        self._view = v
        self.tdim, self.gdim = v.tdim, v.gdim
        self.host_width = int(v.host_width)   # 0: hosted by cells; 2 / 4: by exterior / interior facets
        self._cache: dict = {}

    def _counts(self):
        """(total points, rules) as the engine knows them now: capacities while the cutfemx_amd.step that made the
        rules is open (their lengths are still in HBM), exact afterwards."""
        v = _lib.RulesView()
        if self.dtype == np.dtype(np.float32):
            _lib.check(_lib.lib().cfx_rules_view_get_f32(self._h, C.byref(v)))
        else:
            _lib.check(_lib.lib().cfx_rules_view_get(self._h, C.byref(v)))
        return int(v.nq), int(v.nr)

    @property
    def total_points(self) -> int:
        return self._counts()[0]

    @property
    def num_rules(self) -> int:
        return self._counts()[1]

    @classmethod
    def from_arrays(cls, mesh: Mesh, points, weights, offsets, parent_map):
        """Wrap caller-supplied per-entity rules (host or device arrays)."""
        keep: list = []
        nq, nr = len(weights), len(parent_map)
        h = C.c_void_p()
        if _lib.scalar_dtype(points) == np.float32 and _lib.scalar_dtype(weights) == np.float32:
            _lib.check(_lib.lib().cfx_rules_create_f32(
                mesh._h, mesh.tdim, C.c_int64(nq), C.c_int64(nr), _lib.as_ptr(points, np.float32, keep),
                _lib.as_ptr(weights, np.float32, keep), _lib.as_ptr(offsets, np.int32, keep),
                _lib.as_ptr(parent_map, np.int32, keep), C.byref(h)))
            r = cls(h, mesh, np.float32)
            r._keep = [k for k in keep[2:] if _lib.is_device(k)]
            return r
        _lib.check(_lib.lib().cfx_rules_create(
            mesh._h, mesh.tdim, C.c_int64(nq), C.c_int64(nr), _lib.as_ptr(points, np.float64, keep),
            _lib.as_ptr(weights, np.float64, keep), _lib.as_ptr(offsets, np.int32, keep),
            _lib.as_ptr(parent_map, np.int32, keep), C.byref(h)))
        r = cls(h, mesh)
        r._keep = [k for k in keep if _lib.is_device(k)]
        return r

    def _get(self, name, ptr, n, dtype, shape=None):
        if name not in self._cache:
            a = _lib.download(ptr, n, dtype)   # (the properties below resolve pending counts before they size `n`)
            self._cache[name] = a if shape is None else a.reshape(shape)
        return self._cache[name]

    @property
    def points(self):
        _lib.resolve_counts()
        return self._get("points", self._view.points, self.total_points * self.tdim, self.dtype, (-1, self.tdim))

    @property
    def weights(self):
        _lib.resolve_counts()
        return self._get("weights", self._view.weights, self.total_points, self.dtype)

    @property
    def offsets(self):
        _lib.resolve_counts()
        return self._get("offsets", self._view.offsets, self.num_rules + 1, np.int32)

    @property
    def parent_map(self):
        _lib.resolve_counts()
        return self._get("parent_map", self._view.parent_map, self.num_rules, np.int32)

    @property
    def physical_points(self):
        """(gdim, total_nq), as python/cutfemx/cut.py:25-49."""
        if "phys" not in self._cache:
            _lib.resolve_counts()
            out = np.empty((self.total_points, self.gdim), dtype=self.dtype)
            fn = (_lib.lib().cfx_rules_physical_points_f32 if self.dtype == np.dtype(np.float32)
                  else _lib.lib().cfx_rules_physical_points)
            _lib.check(fn(self._h, out.ctypes.data_as(C.c_void_p)))
            self._cache["phys"] = np.ascontiguousarray(out.T)
        return self._cache["phys"]

    def with_physical_points(self):
        _ = self.physical_points
        return self

    @property
    def host_rows(self):
        """Facet-hosted rules: the integration row (cell, local facet[, cell1, local facet1]) of each rule's facet."""
        if not self.host_width:
            raise ValueError("rules hosted by cells have no facet rows")
        return self._get("host_rows", self._view.host_rows, self.num_rules * self.host_width, np.int32,
                         (-1, self.host_width))

    def to_cells(self, side: int = 0) -> "RuntimeQuadratureRules":
        """Facet-hosted rules seen from cell `side` of their facets: cell-hosted rules (points in that cell's
        reference coordinates, parents ascending) for dx-type integrals -- the mapping of
        facet_runtime_quadrature_payload / interior_facet_runtime_quadrature_payload
        (python/cutfemx/_runintgen_adapter.py:605-680)."""
        h = C.c_void_p()
        _lib.check(_lib.lib().cfx_facet_rules_to_cells(self._h, int(side), C.byref(h)))
        return RuntimeQuadratureRules(h, self.mesh, self.dtype)

    def slice_by_parent(self, cell_lo: int, cell_hi: int, device):
        """Zero-copy sub-rule-set of the rules whose parent cell lies in
        [cell_lo, cell_hi) (parents are ascending): `offsets` / `parent_map` are
        aliased with a shift, points / weights / per-point data keep their absolute
        indices.  Used to restrict a form to a rank's owned cells."""
        import torch

        from .dist import as_torch
        parents = as_torch(self._view.parent_map, self.num_rules, "int32", device)
        r0, r1 = (int(v) for v in torch.searchsorted(
            parents, torch.tensor([cell_lo, cell_hi], device=device, dtype=torch.int32)))
        h = C.c_void_p()
        _lib.check(_lib.lib().cfx_rules_create(
            self.mesh._h, self.tdim, C.c_int64(self.total_points), C.c_int64(r1 - r0),
            C.c_void_p(self._view.points), C.c_void_p(self._view.weights),
            C.c_void_p(self._view.offsets + 4 * r0), C.c_void_p(self._view.parent_map + 4 * r0), C.byref(h)))
        sub = RuntimeQuadratureRules(h, self.mesh)
        sub._keep = [self]
        offs = as_torch(self._view.offsets, self.num_rules + 1, "int32", device)
        sub.total_points_owned = int(offs[r1] - offs[r0]) if r1 > r0 else 0
        return sub

    def __del__(self):
        try:
            if self._h:
                _lib.load().cfx_rules_destroy(self._h)
                self._h = None
        except Exception:
            pass


class CutData:
    """Python handle for cut data (python/cutfemx/cut.py:94-146)."""

    def __init__(self, handle, level_sets: Sequence[Function], keep=(), names=None):
        self._h = handle
        self._level_sets = tuple(level_sets)
        self._keep = list(keep)
        self._names = frozen_level_set_names([f.name for f in level_sets]) if names is None else tuple(names)
        self.dtype = np.dtype(np.float64)     # scalar type of the level-set Functions the cut was made from

    def update(self) -> None:
        update(self)

    @property
    def mesh(self) -> Mesh:
        return self._level_sets[0].function_space.mesh

    def _info(self):
        t, g, n, k = C.c_int(), C.c_int(), C.c_int64(), C.c_int()
        _lib.check(_lib.lib().cfx_cut_info(self._h, C.byref(t), C.byref(g), C.byref(n), C.byref(k)))
        return t.value, g.value, n.value, k.value

    @property
    def tdim(self) -> int:
        """Dimension of the hosts (cut.cpp:571): the mesh cells, or tdim - 1 for facet hosts."""
        return self._info()[0]

    @property
    def gdim(self) -> int:
        return self.mesh.gdim

    @property
    def num_local_cells(self) -> int:
        """Number of hosts (cut.cpp:572)."""
        return self._info()[2]

    @property
    def level_set_names(self) -> tuple[str, ...]:
        """Frozen at cut(): renaming a Function afterwards does not change them (test_cut_api.py:750-763)."""
        return self._names

    @property
    def level_sets(self) -> tuple[Function, ...]:
        return self._level_sets

    @property
    def entity_dim(self):
        return getattr(self, "_entity_dim", None)

    @property
    def entities(self):
        return getattr(self, "_entities", None)

    def domain(self, level_set: int = 0) -> np.ndarray:
        """int8 classification per cell: -1 inside, 0 intersected, +1 outside."""
        p = C.c_void_p()
        _lib.check(_lib.lib().cfx_cut_domain(self._h, level_set, C.byref(p)))
        return _lib.download(p.value, self.num_local_cells, np.int8)

    def __del__(self):
        try:
            if self._h:
                _lib.load().cfx_cut_destroy(self._h)
                self._h = None
        except Exception:
            pass


_UNSPECIFIED_NAMES = ("", "u", "f")
_IDENT = re.compile(r"[A-Za-z_][A-Za-z0-9_]*")


def frozen_level_set_names(raw_names: Sequence[str]) -> tuple[str, ...]:
    """Selector names of the level sets, fixed at cut() time (cut.cpp:82-138): a Function's own
    name when it has one (not "", "u", "f"), else "phi", "phi1", ...; real names must be valid
    identifiers and unique; defaults step aside for real names that look like them."""
    real: list[str | None] = []
    seen: set[str] = set()
    for raw in raw_names:
        if raw in _UNSPECIFIED_NAMES:
            real.append(None)
            continue
        if not _IDENT.fullmatch(raw):
            raise ValueError(f"Level-set function name '{raw}' is not a valid selector identifier. "
                             "Expected [A-Za-z_][A-Za-z0-9_]*.")
        if raw in seen:
            raise ValueError(f"Duplicate level-set function name '{raw}'. Level-set selector names must be unique.")
        seen.add(raw)
        real.append(raw)
    used = set(seen)
    names = []
    for i, r in enumerate(real):
        if r is not None:
            names.append(r)
            continue
        candidate = "phi" if i == 0 else f"phi{i}"
        counter = 1 if i == 0 else i + 1
        while candidate in used:
            candidate = f"phi{counter}"
            counter += 1
        used.add(candidate)
        names.append(candidate)
    return tuple(names)


def _engine_selector(cut_data: "CutData", selector: str) -> bytes:
    """Rewrite a selector written with the frozen names into the engine's positional names
    (phi, phi1, ...); keywords pass through, unknown identifiers are user errors."""
    names = cut_data.level_set_names
    table = {n: ("phi" if i == 0 else f"phi{i}") for i, n in enumerate(names)}

    def repl(m):
        word = m.group(0)
        if word in table:
            return table[word]
        if word.lower() in ("and", "or", "not"):
            return word
        raise ValueError(f"invalid selector '{selector}': unknown level set '{word}'")
    return _IDENT.sub(repl, str(selector)).encode()


def _normalise_level_sets(level_set) -> list[Function]:
    # python/cutfemx/cut.py:163-183
    if isinstance(level_set, Function):
        return [level_set]
    if isinstance(level_set, (str, bytes)) or not isinstance(level_set, Sequence):
        raise TypeError("cutfemx.cut expects a Function or a non-empty sequence of Functions")
    level_sets = list(level_set)
    if not level_sets:
        raise ValueError("cutfemx.cut requires at least one level-set function")
    for item in level_sets:
        if not isinstance(item, Function):
            raise TypeError("cutfemx.cut sequence entries must be Function objects")
    return level_sets


def _value_ptrs(level_sets, keep, dtype=np.float64):
    arr = (C.c_void_p * len(level_sets))()
    for i, f in enumerate(level_sets):
        arr[i] = _lib.as_ptr(f.values, dtype, keep)
    return arr


def _level_set_dtype(level_sets):
    """float32 when every level-set Function holds float32 values (declare_cut_api<float>), else float64."""
    return np.float32 if all(_lib.scalar_dtype(f.values) == np.float32 for f in level_sets) else np.float64


def cut(level_set, entities=None, entity_dim=None, *, cut_approximation: str = "auto",
        cut_approximation_order: int = 1, max_refinement_iterations: int = 8,
        edge_max_depth: int = 20, facet_ids=None, entity_geometry=None) -> CutData:
    """Classify all cells -- or the given host entities -- against one or more level sets
    (python/cutfemx/cut.py:186-249).  entity_dim = tdim: `entities` is a cell subset; entity_dim = tdim - 1:
    `entities` are facets as integration rows ((n, 2) exterior / (n, 4) interior array or FacetRows; the engine
    has no global facet numbering), numbered by position or by `facet_ids`; `entity_geometry` (n, tdim)
    optionally fixes the host vertex order (dolfinx entities_to_geometry)."""
    level_sets = _normalise_level_sets(level_set)
    names = frozen_level_set_names([f.name for f in level_sets])
    if entities is None and entity_dim is not None:
        raise ValueError("entity_dim is only valid when entities are supplied")
    if entities is not None and entity_dim is None:
        raise ValueError("entity_dim must be supplied when entities are supplied")
    V = level_sets[0].function_space
    tdim = V.mesh.tdim
    if entities is not None and (entity_dim <= 0 or entity_dim > tdim):
        raise ValueError("cutfemx::cut entity_dim must select positive-dimensional mesh entities")  # cut.cpp:546-550
    if entities is not None and entity_dim < tdim - 1:
        raise NotImplementedError("hosts of codimension > 1 are not implemented")
    for f in level_sets[1:]:
        if f.function_space is not V:
            raise ValueError("all level sets must share one function space")
    if V.bs != 1:
        raise ValueError("level-set function must be scalar")  # cut.cpp:445-460
    opt = _lib.CutOptions(cut_approximation_order, max_refinement_iterations, edge_max_depth, 0)
    keep: list = []
    f32 = _level_set_dtype(level_sets) == np.float32 and not (entities is not None and entity_dim == tdim - 1)
    vals = _value_ptrs(level_sets, keep, np.float32 if f32 else np.float64)
    h = C.c_void_p()
    if entities is not None and entity_dim == tdim - 1:
        # facets as hosts (cut.cpp:540-591, python/tests/test_cut_api.py:171-187, 349-367): the facets are their
        # integration rows; facet id = position in `entities` unless facet_ids says otherwise
        if isinstance(entities, FacetRows):
            rows_ptr, n, width = C.c_void_p(entities.ptr), entities.size, entities.width
            keep.append(entities)
        else:
            rows = np.ascontiguousarray(np.asarray(entities, dtype=np.int32))
            if rows.ndim != 2 or rows.shape[1] not in (2, 4):
                raise ValueError("facet hosts are integration rows: (n, 2) exterior or (n, 4) interior")
            rows_ptr, n, width = rows.ctypes.data_as(C.c_void_p), rows.shape[0], rows.shape[1]
            keep.append(rows)
        ids = None if facet_ids is None else np.ascontiguousarray(np.asarray(facet_ids, dtype=np.int32))
        geom = None if entity_geometry is None else np.ascontiguousarray(np.asarray(entity_geometry, dtype=np.int32))
        if ids is not None and ids.size != n:
            raise ValueError("facet_ids must name every facet")
        if geom is not None and geom.size != n * tdim:
            raise ValueError("entity_geometry must hold tdim vertices per facet")
        _lib.check(_lib.lib().cfx_cut_create_facets(
            V.mesh._h, C.c_int64(n), None if ids is None else ids.ctypes.data_as(C.c_void_p), rows_ptr, int(width),
            None if geom is None else geom.ctypes.data_as(C.c_void_p), len(level_sets), V._dofmap_ptr, V.ndofs_cell,
            C.c_int64(V.ndofs), vals, C.byref(opt), C.byref(h)))
        cd = CutData(h, level_sets, keep=[k for k in keep if _lib.is_device(k)] + [V], names=names)
        cd._entities = entities if ids is None else ids
        cd._entity_dim = int(entity_dim)
        cd._host_rows = entities
        return cd
    if f32:   # the engine widens the values into its own copy: nothing of the caller's is aliased
        _lib.check(_lib.lib().cfx_cut_create_f32(V.mesh._h, len(level_sets), V._dofmap_ptr, V.ndofs_cell,
                                                 C.c_int64(V.ndofs), vals, C.byref(opt), C.byref(h)))
        cd = CutData(h, level_sets, keep=[V], names=names)
        cd.dtype = np.dtype(np.float32)
    else:
        _lib.check(_lib.lib().cfx_cut_create(V.mesh._h, len(level_sets), V._dofmap_ptr, V.ndofs_cell,
                                             C.c_int64(V.ndofs), vals, C.byref(opt), C.byref(h)))
        cd = CutData(h, level_sets, keep=[k for k in keep if _lib.is_device(k)] + [V], names=names)
    if entities is not None:   # cell subset as host (python/tests/test_cut_api.py:160-168)
        cells = np.ascontiguousarray(np.asarray(entities, dtype=np.int32))
        _lib.check(_lib.lib().cfx_cut_restrict(h, cells.ctypes.data_as(C.c_void_p), C.c_int64(cells.size)))
        cd._entities, cd._entity_dim = cells, int(entity_dim)
    return cd


def update(cut_data: CutData) -> None:
    """Re-classify from the current level-set values (python/cutfemx/cut.py:252-254)."""
    keep: list = []
    if cut_data.dtype == np.dtype(np.float32):
        vals = _value_ptrs(cut_data._level_sets, keep, np.float32)
        _lib.check(_lib.lib().cfx_cut_update_f32(cut_data._h, vals))
        return
    vals = _value_ptrs(cut_data._level_sets, keep)
    _lib.check(_lib.lib().cfx_cut_update(cut_data._h, vals))
    cut_data._keep = [k for k in keep if _lib.is_device(k)] + [cut_data._level_sets[0].function_space]


def locate_entities(cut_data: CutData, ls_part: str) -> np.ndarray:
    """Background cells matched by a selector, ascending int32 (cut.cpp:877-924)."""
    p, n = C.c_void_p(), C.c_int64()
    sel = _engine_selector(cut_data, ls_part)
    _lib.check(_lib.lib().cfx_locate_entities(cut_data._h, sel, C.byref(p), C.byref(n)))
    if _lib._step_open:   # the list's length may still be in HBM: fetch it, then ask again
        _lib.resolve_counts()
        _lib.check(_lib.lib().cfx_locate_entities(cut_data._h, sel, C.byref(p), C.byref(n)))
    return _lib.download(p.value, n.value, np.int32)


class DeviceEntities(tuple):
    """(device pointer, count) of a located entity list.  The array belongs to the `CutData` it came from:
    the tuple keeps that object alive (`owner`), and the list is valid until the next `update()` /
    `locate_entities` with the same selector on it -- forms must be rebuilt after an update, as
    python/demo/demo_moving_poisson.py:53-67 does.  Inside a cutfemx_amd.step the count is the list's capacity (its
    length is still in HBM; the engine recognises the list when it comes back in an Integral); `size` asks again."""
    owner = None
    selector = None

    def __new__(cls, ptr, n, owner, selector=None):
        self = super().__new__(cls, (ptr, n))
        self.owner = owner
        self.selector = selector
        return self

    @property
    def size(self) -> int:
        """The list's length as the engine knows it now (exact once the step that made it has ended)."""
        n = C.c_int64()
        _lib.check(_lib.lib().cfx_list_count(C.c_void_p(self[0]), C.c_int64(int(self[1])), C.byref(n)))
        return int(n.value)


def locate_entities_device(cut_data: CutData, ls_part: str):
    """(device pointer, count) of the selector result; owned by `cut_data` (kept alive by the result)."""
    p, n = C.c_void_p(), C.c_int64()
    sel = _engine_selector(cut_data, ls_part)
    _lib.check(_lib.lib().cfx_locate_entities(cut_data._h, sel, C.byref(p), C.byref(n)))
    return DeviceEntities(p.value, n.value, cut_data, sel)


def runtime_quadrature(cut_data: CutData, ls_part: str, order: int, *, backend: str = "straight"):
    """Runtime quadrature on the cut entities of a selector (cut.cpp:1311-1335)."""
    h = C.c_void_p()
    _lib.check(_lib.lib().cfx_runtime_quadrature(cut_data._h, _engine_selector(cut_data, ls_part), int(order), backend.encode(),
                                                 C.byref(h)))
    return RuntimeQuadratureRules(h, cut_data.mesh, cut_data.dtype)


def runtime_quadratures(cut_data: CutData, ls_parts: Sequence[str], order: int, *, backend: str = "straight"):
    """Rules of several selectors of one cut, {selector: rules} (python/cutfemx/cut.py runtime_quadratures,
    cut.h:178-181).  One library call: pairs of plain selectors share the pass over the cut cells."""
    parts = [str(p) for p in ls_parts]
    sels = [_engine_selector(cut_data, p) for p in parts]
    arr = (C.c_char_p * len(sels))(*sels)
    out = (C.c_void_p * len(sels))()
    _lib.check(_lib.lib().cfx_runtime_quadratures(cut_data._h, len(sels), arr, int(order), backend.encode(), out))
    return {p: RuntimeQuadratureRules(C.c_void_p(out[i]), cut_data.mesh, cut_data.dtype) for i, p in enumerate(parts)}


def full_cell_rules(mesh: Mesh, cells, order: int) -> RuntimeQuadratureRules:
    """Whole-cell per-entity rules: reference points, weights*|detJ|
    (python/tests/quadrature_utils.py:12-70)."""
    keep: list = []
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    h = C.c_void_p()
    _lib.check(_lib.lib().cfx_full_cell_rules(mesh._h, _lib.as_ptr(cells, np.int32, keep), C.c_int64(cells.size),
                                              int(order), C.byref(h)))
    return RuntimeQuadratureRules(h, mesh)


class FacetRows:
    """Interior-facet integration rows (cell0, local_facet0, cell1, local_facet1),
    cell0 < cell1 -- what facet_integration_rows produces from raw facet ids
    (python/cutfemx/wrappers/cut.cpp:54-115).  The engine has no global facet
    numbering, so the rows ARE the facet identity."""

    def __init__(self, ptr, n, owner, width: int = 4, requery=None):
        self.ptr, self._n, self._owner, self.width = ptr, int(n), owner, int(width)
        self._requery = requery     # asks the engine for the list's count again (ghost-penalty rows made inside a step)
        self._rows = None

    @property
    def size(self) -> int:
        """Number of rows: the capacity of the list while the cutfemx_amd.step that made it is open, exact afterwards."""
        if self._requery is not None:
            # by the list's identity (cfx_list_count): nothing is recomputed, a list the engine no longer tracks keeps
            # the count it was handed out with
            n = C.c_int64()
            _lib.check(_lib.lib().cfx_list_count(C.c_void_p(self.ptr), C.c_int64(self._n), C.byref(n)))
            self._n = int(n.value)
        return self._n

    @property
    def rows(self) -> np.ndarray:
        if self._rows is None:
            _lib.resolve_counts()
            self._rows = _lib.download(self.ptr, self.width * self.size, np.int32).reshape(-1, self.width)
        return self._rows

    def __len__(self):
        return self.size


class _OwnedRows:
    """Device rows returned by the engine through cfx_device_alloc."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                _lib.load().cfx_device_free(C.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def interior_facets_for_cells(mesh: Mesh, cells, *, include_ghosts: bool = False) -> FacetRows:
    """Interior facets whose two cells both lie in `cells` (python/cutfemx/cut.py:320-337,
    cut.cpp:926-994), as integration rows (c0, lf0, c1, lf1), c0 < c1, ascending."""
    cells = np.ascontiguousarray(np.asarray(cells, dtype=np.int32))
    p, n = C.c_void_p(), C.c_int64()
    _lib.check(_lib.lib().cfx_interior_facets_for_cells(mesh._h, cells.ctypes.data_as(C.c_void_p), C.c_int64(cells.size),
                                                        C.byref(p), C.byref(n)))
    return FacetRows(p.value, n.value, owner=_OwnedRows(p.value))


def exterior_facets(mesh: Mesh) -> FacetRows:
    """Boundary facets as (cell, local facet) integration rows, ascending: dolfinx
    exterior_facet_indices + facet_integration_rows (python/cutfemx/wrappers/cut.cpp:54-115)."""
    p, n = C.c_void_p(), C.c_int64()
    _lib.check(_lib.lib().cfx_exterior_facets(mesh._h, C.byref(p), C.byref(n)))
    return FacetRows(p.value, n.value, owner=_OwnedRows(p.value), width=2)


def full_facet_rules(cut_data: CutData, ls_part: str | None, order: int) -> RuntimeQuadratureRules:
    """Whole-facet rules over the hosts of a facet-hosted cut matching `ls_part` (None: all hosts): the
    standard facets of a mixed [facets, rules] measure (python/tests/test_cut_api.py:527-560)."""
    h = C.c_void_p()
    sel = None if ls_part is None else _engine_selector(cut_data, ls_part)
    _lib.check(_lib.lib().cfx_full_facet_rules(cut_data._h, sel, int(order), C.byref(h)))
    return RuntimeQuadratureRules(h, cut_data.mesh, cut_data.dtype)


def ghost_penalty_facets(cut_data: CutData, selector: str, *, depth: int = 1, include_ghosts: bool = False):
    """Interior facets of the cut-cell stabilisation band (python/cutfemx/cut.py:340-380)."""
    if depth != 1:
        raise NotImplementedError("ghost_penalty_facets currently supports depth=1.")
    p, n = C.c_void_p(), C.c_int64()
    sel = _engine_selector(cut_data, selector)
    _lib.check(_lib.lib().cfx_ghost_penalty_facets(cut_data._h, sel, C.byref(p), C.byref(n)))
    ptr = p.value

    def requery():
        q, m = C.c_void_p(), C.c_int64()
        _lib.check(_lib.lib().cfx_ghost_penalty_facets(cut_data._h, sel, C.byref(q), C.byref(m)))
        return m.value if q.value == ptr else n.value
    # the rows stay in HBM (owned by cut_data until update()/destruction); .rows downloads on demand
    return FacetRows(ptr, n.value, cut_data, requery=requery if _lib._step_open else None)


class QuadratureFunction:
    """Per-point coefficient values aligned with a rule set (python/cutfemx/level_set.py)."""

    def __init__(self, rules: RuntimeQuadratureRules, values: np.ndarray):
        self.rules, self.values = rules, values


def normal(cut_data: CutData, rules: RuntimeQuadratureRules, level_set: int = 0, sign: float = 1.0,
           device: bool = False):
    """Unit normals grad(phi)/|grad(phi)| at the rule points, (nq, gdim)
    (cpp/cutfemx/level_set/normal.h:39-187).  device=True keeps them in HBM."""
    f32 = rules.dtype == np.dtype(np.float32)
    dt = np.float32 if f32 else np.float64
    fn = _lib.lib().cfx_evaluate_normals_f32 if f32 else _lib.lib().cfx_evaluate_normals
    sgn = C.c_float(sign) if f32 else C.c_double(sign)
    if device:
        buf = _lib.DeviceBuffer(rules.total_points * rules.gdim, dt, (rules.total_points, rules.gdim))
        _lib.check(fn(cut_data._h, level_set, rules._h, sgn, C.c_void_p(buf.ptr)))
        return buf
    out = np.empty((rules.total_points, rules.gdim), dtype=dt)
    _lib.check(fn(cut_data._h, level_set, rules._h, sgn, out.ctypes.data_as(C.c_void_p)))
    return out


def level_set_value(cut_data: CutData, rules: RuntimeQuadratureRules, level_set: int = 0) -> np.ndarray:
    """phi at the rule points (cpp/cutfemx/level_set/value.h:34-119)."""
    f32 = rules.dtype == np.dtype(np.float32)
    out = np.empty(rules.total_points, dtype=np.float32 if f32 else np.float64)
    fn = _lib.lib().cfx_evaluate_values_f32 if f32 else _lib.lib().cfx_evaluate_values
    _lib.check(fn(cut_data._h, level_set, rules._h, out.ctypes.data_as(C.c_void_p)))
    return out
