"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (cutfemx_amd/) never imports
this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

INSIDE, INTERSECTED, OUTSIDE = -1, 0, 1
CELL, EXTERIOR_FACET, INTERIOR_FACET = 0, 1, 2
K_MASS, K_STIFFNESS, K_NITSCHE, K_GHOST_GRADJUMP, K_ELASTICITY = 1, 2, 3, 4, 5
K_EXTENSION_L2 = 8
K_JUMP = 9
K_SIP = 10
K_DIV_TEST, K_DIV_TRIAL = 20, 21   # rectangular blocks: scale div(v) p / scale q div(u)
L_SOURCE, L_NITSCHE_RHS = 101, 102
F_ONE, F_SINPROD, F_POISSON_RHS, F_COEFFICIENT = 0, 1, 2, 3


class _Rules(C.Structure):
    _fields_ = [("tdim", C.c_int32), ("nq", C.c_int64), ("nr", C.c_int64),
                ("points", C.c_void_p), ("weights", C.c_void_p),
                ("offsets", C.c_void_p), ("parent_map", C.c_void_p)]


class _Integral(C.Structure):
    _fields_ = [("type", C.c_int32), ("kernel", C.c_int32), ("qdegree", C.c_int32),
                ("point_stride", C.c_int32), ("entities", C.c_void_p),
                ("n_entities", C.c_int64), ("rules", C.c_void_p),
                ("point_data", C.c_void_p), ("params", C.c_double * 8), ("coefficient", C.c_void_p),
                ("n_std", C.c_int64), ("host_verts", C.c_void_p)]


class _Mesh(C.Structure):
    _fields_ = [("tdim", C.c_int32), ("gdim", C.c_int32), ("nnodes", C.c_int64),
                ("ncells", C.c_int64), ("x", C.c_void_p), ("conn", C.c_void_p)]


class _Space(C.Structure):
    _fields_ = [("degree", C.c_int32), ("bs", C.c_int32), ("ndofs_cell", C.c_int32),
                ("ndofs", C.c_int64), ("dofmap", C.c_void_p)]


def build(force: bool = False) -> Path:
    so = _HERE / "liboracle.so"
    srcs = [_HERE / "cfx_oracle.c", _HERE / "cfx_oracle.h", _HERE / "cfx_quadrature_tables.h"]
    if force or not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(_HERE), "-B", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return so


_NATIVE = False


def use_native_build():
    """bench.py's cpu_baseline leg: the same C restatement built `-O3 -march=native` ON THE MACHINE THAT RUNS IT
    (BASELINE.md 3), into oracle/_build/ (never shipped: a -march=native object of another host may not run).
    -ffp-contract=off and -fno-fast-math stay, so results are those of the portable build.  Must be called
    before the first use of the library in the process."""
    global _NATIVE
    if _LIB is not None and not _NATIVE:
        raise RuntimeError("the portable oracle library is already loaded")
    _NATIVE = True


def _build_native() -> Path:
    out = _HERE / "_build"
    out.mkdir(exist_ok=True)
    so = out / "liboracle_native.so"
    src = _HERE / "cfx_oracle.c"
    if so.exists() and so.stat().st_mtime >= src.stat().st_mtime:
        return so                      # (built by the parent process of bench.py's all-cores leg)
    import os
    tmp = out / f"liboracle_native.{os.getpid()}.tmp"
    subprocess.run(["gcc", "-O3", "-march=native", "-std=c11", "-fPIC", "-fno-fast-math", "-ffp-contract=off", "-shared",
                    "-o", str(tmp), str(src), "-lm"], check=True)
    os.replace(tmp, so)                # atomic: concurrent builders never see a partial file
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(str(_build_native() if _NATIVE else build()))
        _LIB.orc_locate_entities.restype = C.c_int64
        _LIB.orc_ghost_penalty_facets.restype = C.c_int64
        _LIB.orc_interior_facets_for_cells.restype = C.c_int64
        _LIB.orc_active_cells.restype = C.c_int64
        _LIB.orc_inactive_dofs.restype = C.c_int64
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class Mesh:
    tdim: int
    x: np.ndarray      # (nnodes, 3) float64
    conn: np.ndarray   # (ncells, tdim+1) int32

    def __post_init__(self):
        self.x = np.ascontiguousarray(self.x, dtype=np.float64)
        self.conn = np.ascontiguousarray(self.conn, dtype=np.int32)
        self.gdim = self.tdim
        self.c = _Mesh(self.tdim, self.gdim, self.x.shape[0], self.conn.shape[0],
                       _p(self.x), _p(self.conn))

    @property
    def ncells(self):
        return self.conn.shape[0]

    @property
    def nnodes(self):
        return self.x.shape[0]


@dataclass
class Space:
    dofmap: np.ndarray
    ndofs: int
    degree: int = 1
    bs: int = 1

    def __post_init__(self):
        self.dofmap = np.ascontiguousarray(self.dofmap, dtype=np.int32)
        self.c = _Space(self.degree, self.bs, self.dofmap.shape[1], self.ndofs, _p(self.dofmap))


@dataclass
class Rules:
    tdim: int
    points: np.ndarray
    weights: np.ndarray
    offsets: np.ndarray
    parent_map: np.ndarray
    kind: str = "per_entity"
    host_rows: np.ndarray | None = None    # facet-hosted rules (8f-4): integration row of each rule's facet
    host_verts: np.ndarray | None = None   # ... and the mesh vertices spanning its reference simplex

    def cstruct(self):
        self.points = np.ascontiguousarray(self.points, dtype=np.float64)
        self.weights = np.ascontiguousarray(self.weights, dtype=np.float64)
        self.offsets = np.ascontiguousarray(self.offsets, dtype=np.int32)
        self.parent_map = np.ascontiguousarray(self.parent_map, dtype=np.int32)
        return _Rules(self.tdim, self.weights.size, self.parent_map.size, _p(self.points),
                      _p(self.weights), _p(self.offsets), _p(self.parent_map))


def _rules_from_c(r: _Rules) -> Rules:
    nq, nr, tdim = r.nq, r.nr, r.tdim

    def arr(ptr, n, dt):
        if n == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))),
                                     shape=(n,)).copy()
    out = Rules(tdim, arr(r.points, nq * tdim, np.float64).reshape(nq, tdim),
                arr(r.weights, nq, np.float64), arr(r.offsets, nr + 1, np.int32),
                arr(r.parent_map, nr, np.int32))
    lib().orc_rules_free(C.byref(r))
    return out


def mesh_box(tdim: int, n: int) -> Mesh:
    nn = (n + 1) ** tdim
    nc = (2 if tdim == 2 else 6) * n ** tdim
    x = np.zeros((nn, 3))
    conn = np.zeros((nc, tdim + 1), dtype=np.int32)
    lib().orc_mesh_box(tdim, n, _p(x), _p(conn))
    return Mesh(tdim, x, conn)


def classify(ls_dofmap, ls_values):
    ls_dofmap = np.ascontiguousarray(ls_dofmap, dtype=np.int32)
    ls_values = np.ascontiguousarray(ls_values, dtype=np.float64)
    dom = np.zeros(ls_dofmap.shape[0], dtype=np.int8)
    lib().orc_classify(C.c_int64(ls_dofmap.shape[0]), ls_dofmap.shape[1], _p(ls_dofmap),
                       _p(ls_values), _p(dom))
    return dom


def locate_entities(domain, selector: str):
    domain = np.ascontiguousarray(domain, dtype=np.int8)
    dom2 = domain.reshape(-1, domain.shape[-1]) if domain.ndim > 1 else domain.reshape(1, -1)
    nls, nc = dom2.shape
    out = np.zeros(nc, dtype=np.int32)
    n = lib().orc_locate_entities(C.c_int64(nc), nls, _p(dom2), selector.encode(), _p(out))
    if n < 0:
        raise ValueError(f"invalid selector {selector!r}")
    return out[:n].copy()


def runtime_quadrature(mesh: Mesh, ls_dofmap, ls_values, domain, selector: str, order: int):
    ls_dofmap = np.ascontiguousarray(ls_dofmap, dtype=np.int32)
    ls_values = np.ascontiguousarray(ls_values, dtype=np.float64)
    domain = np.ascontiguousarray(domain, dtype=np.int8)
    r = _Rules()
    rc = lib().orc_runtime_quadrature(C.byref(mesh.c), _p(ls_dofmap), _p(ls_values), _p(domain),
                                      selector.encode(), order, C.byref(r))
    if rc != 0:
        raise ValueError(f"invalid runtime-quadrature selector {selector!r}")
    return _rules_from_c(r)


def classify_multi(ls_dofmap, ls_values_list):
    """[nls, ncells] classification codes of several level sets on one dofmap."""
    return np.stack([classify(ls_dofmap, v) for v in ls_values_list])


def runtime_quadrature_multi(mesh: Mesh, ls_dofmap, ls_values_list, domains, selector: str, order: int):
    """Rules of one conjunction over several level sets ("phi<0 and phi1>0", "phi=0 and phi1<0")."""
    ls_dofmap = np.ascontiguousarray(ls_dofmap, dtype=np.int32)
    vals = [np.ascontiguousarray(v, dtype=np.float64) for v in ls_values_list]
    ptrs = (C.c_void_p * len(vals))(*[v.ctypes.data for v in vals])
    domains = np.ascontiguousarray(domains, dtype=np.int8)
    r = _Rules()
    rc = lib().orc_runtime_quadrature_multi(C.byref(mesh.c), len(vals), _p(ls_dofmap), ptrs, _p(domains),
                                            selector.encode(), int(order), C.byref(r))
    if rc != 0:
        raise ValueError(f"unsupported selector for runtime quadrature: {selector!r}")
    return _rules_from_c(r)


def full_cell_rules(mesh: Mesh, cells, order: int):
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    r = _Rules()
    lib().orc_full_cell_rules(C.byref(mesh.c), _p(cells), C.c_int64(cells.size), order, C.byref(r))
    return _rules_from_c(r)


def evaluate_normals(mesh, ls_dofmap, ls_values, rules: Rules, sign=1.0):
    ls_dofmap = np.ascontiguousarray(ls_dofmap, dtype=np.int32)
    ls_values = np.ascontiguousarray(ls_values, dtype=np.float64)
    rc = rules.cstruct()
    out = np.zeros((rules.weights.size, mesh.gdim))
    lib().orc_evaluate_normals(C.byref(mesh.c), _p(ls_dofmap), _p(ls_values), C.byref(rc),
                               C.c_double(sign), _p(out))
    return out


def evaluate_values(mesh, ls_dofmap, ls_values, rules: Rules):
    ls_dofmap = np.ascontiguousarray(ls_dofmap, dtype=np.int32)
    ls_values = np.ascontiguousarray(ls_values, dtype=np.float64)
    rc = rules.cstruct()
    out = np.zeros(rules.weights.size)
    lib().orc_evaluate_values(C.byref(mesh.c), _p(ls_dofmap), _p(ls_values), C.byref(rc), _p(out))
    return out


def physical_points(mesh, rules: Rules):
    rc = rules.cstruct()
    out = np.zeros((rules.weights.size, mesh.gdim))
    lib().orc_physical_points(C.byref(mesh.c), C.byref(rc), _p(out))
    return out


def _rows_out(n, ptr):
    if n < 0:
        raise ValueError("invalid selector")
    if n == 0:
        rows = np.zeros((0, 4), dtype=np.int32)
    else:
        rows = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), shape=(n, 4)).copy()
    lib().orc_free(ptr)
    return rows


def ghost_penalty_facets(mesh, domain, selector: str):
    domain = np.ascontiguousarray(domain, dtype=np.int8)
    ptr = C.c_void_p()
    n = lib().orc_ghost_penalty_facets(C.byref(mesh.c), _p(domain), selector.encode(), C.byref(ptr))
    return _rows_out(n, ptr)


def interior_facets_for_cells(mesh, cells):
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    ptr = C.c_void_p()
    n = lib().orc_interior_facets_for_cells(C.byref(mesh.c), _p(cells), C.c_int64(cells.size),
                                            C.byref(ptr))
    return _rows_out(n, ptr)


@dataclass
class Integral:
    type: int
    kernel: int
    entities: np.ndarray | None = None
    rules: Rules | None = None
    point_data: np.ndarray | None = None
    params: tuple = ()
    qdegree: int = 2
    coefficient: np.ndarray | None = None
    _keep: list = field(default_factory=list)

    def cstruct(self):
        ent = np.zeros(0, dtype=np.int32) if self.entities is None else \
            np.ascontiguousarray(self.entities, dtype=np.int32)
        n_ent = ent.size // 4 if self.type == INTERIOR_FACET else ent.size
        n_std, hv = n_ent, None
        if self.type == INTERIOR_FACET and self.rules is not None:
            # facet-hosted rules (8f-4): one entity list [standard rows, the rules' rows]
            ent = np.ascontiguousarray(np.concatenate([ent.reshape(-1, 4), self.rules.host_rows.reshape(-1, 4)]),
                                       dtype=np.int32)
            n_ent = ent.shape[0]
            hvarr = np.ascontiguousarray(self.rules.host_verts, dtype=np.int32)
            hv = _p(hvarr)
        self._keep = [ent] + ([hvarr] if hv is not None else [])
        rptr = None
        if self.rules is not None:
            rc = self.rules.cstruct()
            self._keep.append(rc)
            rptr = C.cast(C.pointer(rc), C.c_void_p)
        pd, stride = None, 0
        if self.point_data is not None:
            pdarr = np.ascontiguousarray(self.point_data, dtype=np.float64)
            self._keep.append(pdarr)
            pd = _p(pdarr)
            stride = 1 if pdarr.ndim == 1 else pdarr.shape[1]
        params = (C.c_double * 8)(*([float(v) for v in self.params] + [0.0] * (8 - len(self.params))))
        co = None
        if self.coefficient is not None:
            carr = np.ascontiguousarray(self.coefficient, dtype=np.float64)
            self._keep.append(carr)
            co = _p(carr)
        return _Integral(self.type, self.kernel, self.qdegree, stride, _p(ent), n_ent, rptr, pd, params, co, n_std, hv)


def _integral_array(integrals):
    arr = (_Integral * len(integrals))(*[i.cstruct() for i in integrals])
    return arr


def create_sparsity(mesh, V: Space, integrals):
    arr = _integral_array(integrals)
    ip, ix = C.c_void_p(), C.c_void_p()
    lib().orc_create_sparsity(C.byref(mesh.c), C.byref(V.c), arr, len(integrals),
                              C.byref(ip), C.byref(ix))
    nrows = V.ndofs * V.bs
    indptr = np.ctypeslib.as_array(C.cast(ip, C.POINTER(C.c_int64)), shape=(nrows + 1,)).copy()
    nnz = int(indptr[-1])
    indices = np.ctypeslib.as_array(C.cast(ix, C.POINTER(C.c_int32)), shape=(nnz,)).copy()
    lib().orc_free(ip)
    lib().orc_free(ix)
    return indptr, indices


def assemble_matrix(mesh, V: Space, integrals, indptr, indices, bc0=None, bc1=None):
    arr = _integral_array(integrals)
    values = np.zeros(indices.size)
    b0 = None if bc0 is None else np.ascontiguousarray(bc0, dtype=np.int8)
    b1 = None if bc1 is None else np.ascontiguousarray(bc1, dtype=np.int8)
    rc = lib().orc_assemble_matrix(C.byref(mesh.c), C.byref(V.c), arr, len(integrals), _p(b0),
                                   _p(b1), _p(indptr), _p(indices), _p(values))
    if rc != 0:
        raise RuntimeError("entry not in sparsity pattern")
    return values


# ---- rectangular forms (test space V0, trial space V1): assemble_matrix_impl.h:68-189, assembler.h:442-560 ----
def create_sparsity2(mesh, V0: Space, V1: Space, integrals):
    arr = _integral_array(integrals)
    ip, ix = C.c_void_p(), C.c_void_p()
    rc = lib().orc_create_sparsity2(C.byref(mesh.c), C.byref(V0.c), C.byref(V1.c), arr, len(integrals), C.byref(ip), C.byref(ix))
    if rc != 0:
        raise ValueError("rectangular forms take cell integrals (and interior-facet integrals between scalar spaces)")
    nrows = V0.ndofs * V0.bs
    indptr = np.ctypeslib.as_array(C.cast(ip, C.POINTER(C.c_int64)), shape=(nrows + 1,)).copy()
    nnz = int(indptr[-1])
    indices = np.ctypeslib.as_array(C.cast(ix, C.POINTER(C.c_int32)), shape=(max(nnz, 1),)).copy()[:nnz]
    lib().orc_free(ip)
    lib().orc_free(ix)
    return indptr, indices


def assemble_matrix2(mesh, V0: Space, V1: Space, integrals, indptr, indices, bc0=None, bc1=None):
    arr = _integral_array(integrals)
    values = np.zeros(indices.size)
    b0 = None if bc0 is None else np.ascontiguousarray(bc0, dtype=np.int8)
    b1 = None if bc1 is None else np.ascontiguousarray(bc1, dtype=np.int8)
    rc = lib().orc_assemble_matrix2(C.byref(mesh.c), C.byref(V0.c), C.byref(V1.c), arr, len(integrals), _p(b0), _p(b1),
                                    _p(indptr), _p(indices), _p(values))
    if rc != 0:
        raise RuntimeError("entry not in sparsity pattern" if rc == -1 else "unsupported integral")
    return values


def tabulate_entity2(mesh, V0: Space, V1: Space, integral: Integral, idx: int, use_rule: bool):
    ic = integral.cstruct()
    m = 2 if integral.type == INTERIOR_FACET else 1     # facets: macro rows / columns of both cells
    Ae = np.zeros((m * V0.dofmap.shape[1] * V0.bs, m * V1.dofmap.shape[1] * V1.bs))
    lib().orc_tabulate_entity2(C.byref(mesh.c), C.byref(V0.c), C.byref(V1.c), C.byref(ic), C.c_int64(idx), int(use_rule), _p(Ae))
    return Ae


def assemble_vector(mesh, V: Space, integrals):
    arr = _integral_array(integrals)
    b = np.zeros(V.ndofs * V.bs)
    lib().orc_assemble_vector(C.byref(mesh.c), C.byref(V.c), arr, len(integrals), _p(b))
    return b


def apply_lifting(mesh, V: Space, integrals, markers, g, b, x0=None, alpha=1.0):
    arr = _integral_array(integrals)
    markers = np.ascontiguousarray(markers, dtype=np.int8)
    g = np.ascontiguousarray(g, dtype=np.float64)
    x0 = None if x0 is None else np.ascontiguousarray(x0, dtype=np.float64)
    lib().orc_apply_lifting(C.byref(mesh.c), C.byref(V.c), arr, len(integrals), _p(markers), _p(g), _p(x0),
                            C.c_double(alpha), _p(b))
    return b


def tabulate_entity(mesh, V: Space, integral: Integral, idx: int, use_rule: bool):
    ic = integral.cstruct()
    nloc = V.dofmap.shape[1] * V.bs * (2 if integral.type == INTERIOR_FACET else 1)
    rank2 = integral.kernel < 100
    Ae = np.zeros((nloc, nloc) if rank2 else (nloc,))
    lib().orc_tabulate_entity(C.byref(mesh.c), C.byref(V.c), C.byref(ic), C.c_int64(idx),
                              int(use_rule), _p(Ae))
    return Ae


def active_cells(integrals, ncells):
    arr = _integral_array(integrals)
    out = np.zeros(ncells, dtype=np.int32)
    n = lib().orc_active_cells(arr, len(integrals), C.c_int64(ncells), _p(out))
    return out[:n].copy()


def inactive_dofs(V: Space, active):
    active = np.ascontiguousarray(active, dtype=np.int32)
    out = np.zeros(V.ndofs * V.bs, dtype=np.int32)
    n = lib().orc_inactive_dofs(C.byref(V.c), _p(active), C.c_int64(active.size), _p(out))
    return out[:n].copy()


def deactivate(inactive, indptr, indices, values, b=None, diagonal=1.0, rhs_value=0.0):
    inactive = np.ascontiguousarray(inactive, dtype=np.int32)
    lib().orc_deactivate(_p(inactive), C.c_int64(inactive.size), 1, _p(indptr), _p(indices),
                         _p(values), _p(b), C.c_double(diagonal), C.c_double(rhs_value))


# ---- 8f-3: cell aggregation and extension penalty (cpp/cutfemx/extensions/) ---------------------
def cell_volumes(mesh: Mesh) -> np.ndarray:
    x, c = mesh.x, mesh.conn
    e = x[c[:, 1:]] - x[c[:, :1]]
    if mesh.tdim == 2:
        return 0.5 * np.abs(e[:, 0, 0] * e[:, 1, 1] - e[:, 0, 1] * e[:, 1, 0])
    return np.abs(np.einsum("ij,ij->i", e[:, 0], np.cross(e[:, 1], e[:, 2]))) / 6.0


def volume_fractions(mesh: Mesh, ls_dofmap, ls_values, domain, selector: str) -> np.ndarray:
    """|selected part of a cut cell| / |cell| (cutcells::output::volume_fractions as used by
    cell_aggregation.cpp:180-186); 0 on cells without a cut part."""
    rules = runtime_quadrature(mesh, ls_dofmap, ls_values, domain, selector, 1)
    part = np.zeros(mesh.ncells)
    np.add.at(part, rules.parent_map, np.add.reduceat(rules.weights, rules.offsets[:-1]) if rules.parent_map.size
              else np.zeros(0))
    return part / cell_volumes(mesh)


def cell_aggregation(mesh: Mesh, ls_dofmap, ls_values, domain, selector: str, threshold: float,
                     root_policy: str = "interior_or_well_cut", max_iterations: int = -1,
                     allow_rootless: bool = False) -> dict:
    text = "".join(selector.split())
    if ("<" in text) == (">" in text) or "=" in text or not text.endswith("0"):
        raise ValueError("CellAggregation v1 expects a strict single level-set selector such as 'phi < 0' or 'phi > 0'.")
    if not 0.0 <= threshold <= 1.0:
        raise ValueError("Volume fraction threshold must be in [0, 1].")
    if root_policy not in ("interior_only", "interior_or_well_cut"):
        raise ValueError("Unknown root policy. Expected 'interior_only' or 'interior_or_well_cut'.")
    rel = -1 if "<" in text else 1
    domain = np.ascontiguousarray(domain, dtype=np.int8)
    frac = volume_fractions(mesh, ls_dofmap, ls_values, domain, text)
    frac = np.where(domain == 0, frac, 0.0)
    root = np.empty(mesh.ncells, dtype=np.int32)
    agg = np.empty(mesh.ncells, dtype=np.int32)
    depth = np.empty(mesh.ncells, dtype=np.int32)
    lib().orc_cell_aggregation.restype = C.c_int64
    nroot = lib().orc_cell_aggregation(C.byref(mesh.c), _p(domain), rel, _p(frac), C.c_double(threshold),
                                       1 if root_policy == "interior_or_well_cut" else 0, int(max_iterations),
                                       _p(root), _p(agg), _p(depth))
    cut = np.flatnonzero(domain == 0).astype(np.int32)
    interior = np.flatnonzero(domain == rel).astype(np.int32)
    well = np.flatnonzero((root == np.arange(mesh.ncells)) & (depth == 0)).astype(np.int32)
    ill = np.setdiff1d(cut, well).astype(np.int32)
    rootless = ill[root[ill] < 0]
    if nroot and not allow_rootless:
        raise RuntimeError("CellAggregation found active ill-posed cells without an admissible root. Adjust the root "
                           "policy or threshold, or explicitly allow rootless aggregation for diagnostics.")
    return dict(active_cells=np.union1d(interior, cut).astype(np.int32), cut_cells=cut, interior_cells=interior,
                well_posed_cells=well, ill_posed_cells=ill, root_cell=root, aggregate_id=agg,
                propagation_depth=depth, rootless_cells=rootless, cut_volume_fraction=frac)


def extension_pairs(agg: dict) -> np.ndarray:
    """(bad, 0, root, 0) rows of the ill-posed cells that found a root (extension_penalty.cpp:373-392)."""
    ill = agg["ill_posed_cells"]
    ill = ill[agg["root_cell"][ill] >= 0]
    rows = np.zeros((ill.size, 4), dtype=np.int32)
    rows[:, 0] = ill
    rows[:, 2] = agg["root_cell"][ill]
    return rows


# ---------------------------------------------------------------------------
# 8f-4 facet hosts: cut(level_set, facets, tdim - 1)  (cpp/cutfemx/cut/cut.cpp:540-591, 788-830, 1022-1063;
# python/tests/test_cut_api.py:171-187, 349-367, 424-496).  Facets are their integration rows.
# ---------------------------------------------------------------------------
def exterior_facets(mesh: Mesh) -> np.ndarray:
    """(cell, local facet) rows of the facets that belong to one cell only, ascending
    (dolfinx exterior_facet_indices + facet_integration_rows, python/cutfemx/wrappers/cut.cpp:54-115)."""
    nv = mesh.tdim + 1
    conn = mesh.conn.reshape(-1, nv)
    keys, owner = [], []
    for lf in range(nv):
        keys.append(np.sort(np.delete(conn, lf, axis=1), axis=1))
        owner.append(np.stack([np.arange(conn.shape[0]), np.full(conn.shape[0], lf)], axis=1))
    keys, owner = np.concatenate(keys), np.concatenate(owner)
    _, inv, counts = np.unique(keys, axis=0, return_inverse=True, return_counts=True)
    rows = owner[counts[inv.ravel()] == 1]
    return rows[np.lexsort((rows[:, 1], rows[:, 0]))].astype(np.int32)


@dataclass
class FacetHosts:
    rows: np.ndarray          # (n, 2) or (n, 4)
    ids: np.ndarray           # (n,) the caller's facet numbers (parent_entities)
    verts: np.ndarray         # (n, tdim) host vertex order
    ls: np.ndarray            # (n, tdim) level-set dofs of the host vertices


def facet_hosts(mesh: Mesh, rows, ls_dofmap, facet_ids=None, entity_geometry=None) -> FacetHosts:
    """build_entity_mesh_view / build_entity_level_sets for facets: host vertex j is the j-th vertex of cell0
    that is not opposite the facet (ascending local index), or the entity_geometry row."""
    tdim, nv = mesh.tdim, mesh.tdim + 1
    rows = np.asarray(rows, dtype=np.int32)
    conn = mesh.conn.reshape(-1, nv)
    lsd = np.asarray(ls_dofmap).reshape(-1, nv)
    n = rows.shape[0]
    verts = np.empty((n, tdim), dtype=np.int32)
    ls = np.empty((n, tdim), dtype=np.int32)
    for i in range(n):
        c, lf = int(rows[i, 0]), int(rows[i, 1])
        if not (0 <= c < conn.shape[0] and 0 <= lf <= tdim):
            raise IndexError("cell index or local facet out of range")
        if entity_geometry is None:
            loc = [k for k in range(nv) if k != lf]
        else:
            loc = []
            for g in np.asarray(entity_geometry).reshape(n, tdim)[i]:
                k = [k for k in range(nv) if k != lf and conn[c, k] == g]
                if not k:
                    raise ValueError("entity_geometry names a vertex that is not on the facet")
                loc.append(k[0])
        verts[i], ls[i] = conn[c, loc], lsd[c, loc]
        if rows.shape[1] == 4:
            c1, lf1 = int(rows[i, 2]), int(rows[i, 3])
            if not (0 <= c1 < conn.shape[0] and 0 <= lf1 <= tdim):
                raise IndexError("cell index or local facet out of range")
            if set(np.delete(conn[c1], lf1)) != set(verts[i]):
                raise ValueError("the two (cell, local facet) pairs of a row are different facets")
    ids = np.arange(n, dtype=np.int32) if facet_ids is None else np.asarray(facet_ids, dtype=np.int32)
    return FacetHosts(rows, ids, verts, ls)


def facet_classify(hosts: FacetHosts, ls_values):
    return classify(hosts.ls, ls_values)


def facet_locate_entities(hosts: FacetHosts, domain, selector: str):
    """locate_entities on facet hosts answers with the parent facet ids (cut.cpp:352-359, 921)."""
    return hosts.ids[locate_entities(domain, selector)]


def facet_runtime_quadrature(mesh: Mesh, hosts: FacetHosts, ls_values, domain, selector, order: int,
                             whole: bool = False) -> Rules:
    out = _Rules()
    rh = C.POINTER(C.c_int32)()
    verts = np.ascontiguousarray(hosts.verts, dtype=np.int32)
    ls = np.ascontiguousarray(hosts.ls, dtype=np.int32)
    ids = np.ascontiguousarray(hosts.ids, dtype=np.int32)
    vals = np.ascontiguousarray(ls_values, dtype=np.float64)
    dom = np.ascontiguousarray(domain, dtype=np.int8)
    rc = lib().orc_facet_runtime_quadrature(C.byref(mesh.c), C.c_int64(verts.shape[0]), _p(verts), _p(ls), _p(ids), _p(vals),
                                            _p(dom), None if selector is None else selector.encode(), int(order),
                                            int(bool(whole)), C.byref(out), C.byref(rh))
    if rc != 0:
        raise ValueError("facet hosts integrate the phi<0 / phi>0 part (single clause)")
    nr = out.nr
    host = np.ctypeslib.as_array(rh, shape=(max(nr, 1),))[:nr].copy()
    lib().orc_free(rh)
    r = _rules_from_c(out)
    r.host_rows = hosts.rows[host]
    r.host_verts = hosts.verts[host]
    return r


def facet_physical_points(mesh: Mesh, rules: Rules) -> np.ndarray:
    """physical_points_for_host_mesh (cut.cpp:1344-1345): (nq, gdim)."""
    hd = rules.tdim
    rule = np.repeat(np.arange(rules.parent_map.size), np.diff(rules.offsets))
    lam = np.concatenate([1.0 - rules.points.sum(axis=1, keepdims=True), rules.points], axis=1)   # (nq, hd+1)
    xv = mesh.x.reshape(-1, 3)[rules.host_verts[rule]]                                          # (nq, hd+1, 3)
    return np.einsum("qj,qjd->qd", lam, xv)[:, :mesh.tdim]


def facet_rules_to_cells(mesh: Mesh, rules: Rules, side: int = 0) -> Rules:
    """The same points in the reference coordinates of cell `side` of each rule's facet, rules stably ordered by
    that cell (facet_runtime_quadrature_payload, python/cutfemx/_runintgen_adapter.py:605-680)."""
    tdim, nv = mesh.tdim, mesh.tdim + 1
    conn = mesh.conn.reshape(-1, nv)
    cells = rules.host_rows[:, 2 * side].astype(np.int64)
    perm = np.argsort(cells, kind="stable")
    counts = np.diff(rules.offsets)
    pts, wts, offs = [], [], [0]
    for r in perm:
        q0, q1 = rules.offsets[r], rules.offsets[r + 1]
        P = rules.points[q0:q1]
        lam = np.concatenate([1.0 - P.sum(axis=1, keepdims=True), P], axis=1)
        X = np.zeros((q1 - q0, tdim))
        for j in range(tdim):
            k = int(np.nonzero(conn[cells[r]] == rules.host_verts[r, j])[0][0])
            if k > 0:
                X[:, k - 1] += lam[:, j]
        pts.append(X); wts.append(rules.weights[q0:q1]); offs.append(offs[-1] + counts[r])
    return Rules(tdim, np.concatenate(pts) if pts else np.zeros((0, tdim)),
                 np.concatenate(wts) if wts else np.zeros(0), np.asarray(offs, dtype=np.int32),
                 cells[perm].astype(np.int32))
