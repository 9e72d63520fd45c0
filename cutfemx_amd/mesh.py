"""Background mesh + P1/P2 Lagrange spaces: the minimal stand-ins for
dolfinx.mesh.Mesh / dolfinx.fem.FunctionSpace / dolfinx.fem.Function that the
CutFEMx hot path reads (flat arrays only: geometry.x stride 3, int32 geometry
dofmap, int32 dofmap; cpp/cutfemx/cut/cut.cpp:500-538,593-636).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

# Kuhn split of a Basix-order hexahedron / quadrilateral
# (cpp/cutfemx/distance/fast_iterative.h:93-94,103-108)
KUHN_TET = np.array([[0, 1, 3, 7], [0, 1, 5, 7], [0, 2, 3, 7], [0, 2, 6, 7], [0, 4, 5, 7], [0, 4, 6, 7]])
KUHN_TRI = np.array([[0, 1, 3], [0, 3, 2]])


def box_mesh_arrays(tdim: int, n: int, lower=None, upper=None):
    """Host arrays (x[nnodes,3], conn[ncells,tdim+1]) of the synthetic box mesh:
    n^tdim cubes on [lower, upper]^tdim (default the unit box), vertex id
    ix+(n+1)(iy+(n+1)iz), each cube Kuhn-split; cell id = 6*cube+k (2*quad+k)."""
    n1 = n + 1
    lo = np.zeros(tdim) if lower is None else np.asarray(lower, dtype=np.float64)
    hi = np.ones(tdim) if upper is None else np.asarray(upper, dtype=np.float64)
    axes = [np.arange(n1, dtype=np.float64) / n * (hi[d] - lo[d]) + lo[d] for d in range(tdim)]
    x = np.zeros((n1 ** tdim, 3))
    if tdim == 2:
        yy, xx = np.meshgrid(axes[1], axes[0], indexing="ij")
        x[:, 0], x[:, 1] = xx.ravel(), yy.ravel()
        iy, ix = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        ix, iy = ix.ravel(), iy.ravel()
        corner = np.stack([(ix + (i & 1)) + n1 * (iy + ((i >> 1) & 1)) for i in range(4)], axis=1)
        conn = corner[:, KUHN_TRI].reshape(-1, 3)
    else:
        zz, yy, xx = np.meshgrid(axes[2], axes[1], axes[0], indexing="ij")
        x[:, 0], x[:, 1], x[:, 2] = xx.ravel(), yy.ravel(), zz.ravel()
        iz, iy, ix = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
        ix, iy, iz = ix.ravel(), iy.ravel(), iz.ravel()
        corner = np.stack([(ix + (i & 1)) + n1 * ((iy + ((i >> 1) & 1)) + n1 * (iz + ((i >> 2) & 1)))
                           for i in range(8)], axis=1)
        conn = corner[:, KUHN_TET].reshape(-1, 4)
    return x, np.ascontiguousarray(conn, dtype=np.int32)


class Mesh:
    """Simplex background mesh resident in HBM."""

    def __init__(self, handle, tdim, gdim, nnodes, ncells, keep=()):
        self._h = handle
        self.tdim, self.gdim = tdim, gdim
        self.num_nodes, self.num_cells = int(nnodes), int(ncells)
        self._keep = list(keep)
        self.dtype = np.dtype(np.float64)      # geometry type of the arrays the mesh was made from

    @classmethod
    def from_arrays(cls, tdim: int, x, conn) -> "Mesh":
        keep: list = []
        nnodes = x.shape[0]
        ncells = conn.shape[0]
        stride = conn.shape[1]
        h = C.c_void_p()
        if _lib.scalar_dtype(x) == np.float32:
            # geometry type U = float (wrappers/cut.cpp:403-407): the engine keeps a widened copy of its own
            _lib.check(_lib.lib().cfx_mesh_create_f32(tdim, tdim, C.c_int64(nnodes), _lib.as_ptr(x, np.float32, keep),
                                                      C.c_int64(ncells), _lib.as_ptr(conn, np.int32, keep), stride,
                                                      C.byref(h)))
            m = cls(h, tdim, tdim, nnodes, ncells, [k for k in keep[1:] if _lib.is_device(k)])
            m.dtype = np.dtype(np.float32)
            return m
        _lib.check(_lib.lib().cfx_mesh_create(tdim, tdim, C.c_int64(nnodes), _lib.as_ptr(x, np.float64, keep),
                                              C.c_int64(ncells), _lib.as_ptr(conn, np.int32, keep), stride,
                                              C.byref(h)))
        # device inputs are aliased by the engine: keep them alive
        return cls(h, tdim, tdim, nnodes, ncells, [k for k in keep if _lib.is_device(k)])

    @classmethod
    def create_box(cls, tdim: int, n: int) -> "Mesh":
        """Synthetic unit-box mesh generated directly in HBM."""
        h = C.c_void_p()
        _lib.check(_lib.lib().cfx_mesh_create_box(tdim, n, C.byref(h)))
        nn, nc = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().cfx_mesh_info(h, None, None, C.byref(nn), C.byref(nc), None, None))
        return cls(h, tdim, tdim, nn.value, nc.value)

    @classmethod
    def create_slab(cls, n: int, z0: int, nz: int) -> "Mesh":
        """Hex layers z0..z0+nz-1 of the n^3 unit-box mesh, generated in HBM."""
        h = C.c_void_p()
        _lib.check(_lib.lib().cfx_mesh_create_slab(n, z0, nz, C.byref(h)))
        nn, nc = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().cfx_mesh_info(h, None, None, C.byref(nn), C.byref(nc), None, None))
        return cls(h, 3, 3, nn.value, nc.value)

    def _info(self):
        x, conn = C.c_void_p(), C.c_void_p()
        _lib.check(_lib.lib().cfx_mesh_info(self._h, None, None, None, None, C.byref(x), C.byref(conn)))
        return x.value, conn.value

    @property
    def x_ptr(self) -> int:
        return self._info()[0]

    @property
    def conn_ptr(self) -> int:
        return self._info()[1]

    @property
    def x(self) -> np.ndarray:
        return _lib.download(self.x_ptr, self.num_nodes * 3, np.float64).reshape(-1, 3)

    @property
    def conn(self) -> np.ndarray:
        return _lib.download(self.conn_ptr, self.num_cells * (self.tdim + 1), np.int32).reshape(-1, self.tdim + 1)

    def __del__(self):
        try:
            if self._h:
                _lib.load().cfx_mesh_destroy(self._h)
                self._h = None
        except Exception:
            pass


# Basix edge numbering of the reference simplices (P2 edge dofs)
_EDGES = {2: [(1, 2), (0, 2), (0, 1)], 3: [(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)]}


def lagrange_dofmap(tdim: int, conn: np.ndarray, num_nodes: int, degree: int):
    """(dofmap, ndofs) of a continuous Lagrange space on a P1 simplex mesh.
    Degree 1: the geometry dofmap.  Degree 2: vertex dofs, then one dof per
    unique edge (numbered by sorted (v0, v1) pair), local order = Basix."""
    conn = np.asarray(conn)
    if degree == 1:
        return np.ascontiguousarray(conn, dtype=np.int32), int(num_nodes)
    edges = np.stack([np.sort(conn[:, list(e)], axis=1) for e in _EDGES[tdim]], axis=1)  # (nc, ne, 2)
    key = edges[..., 0].astype(np.int64) * num_nodes + edges[..., 1]
    uniq, inv = np.unique(key.ravel(), return_inverse=True)
    edge_dofs = inv.reshape(key.shape) + num_nodes
    dofmap = np.concatenate([conn, edge_dofs], axis=1)
    return np.ascontiguousarray(dofmap, dtype=np.int32), int(num_nodes + uniq.size)


def box_lagrange2_dofmap(mesh: "Mesh", n: int, device=None, chunk: int = 1 << 24):
    """P2 dofmap of a Kuhn box / slab mesh built in HBM with torch (no host pass).

    Closed form for the generator's vertex numbering (stride s = n+1): every edge
    joins a vertex `a` to `a + d` with d one of the 7 Kuhn directions
    {x, y, z, xy, xz, yz, xyz}; edge dof = nnodes + 7*a + direction.  Ids of
    directions that leave the box are never referenced by a cell; those rows stay
    inactive (identity after deactivation).  Returns (torch int32 (ncells, 10), ndofs).
    """
    import torch

    from .dist import as_torch
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    s = n + 1
    nn, nc = mesh.num_nodes, mesh.num_cells
    conn = as_torch(mesh.conn_ptr, nc * 4, "int32", device).view(nc, 4)
    table = torch.tensor([1, s, s * s, 1 + s, 1 + s * s, s + s * s, 1 + s + s * s], device=device, dtype=torch.int32)
    out = torch.empty((nc, 10), device=device, dtype=torch.int32)
    out[:, :4] = conn
    for lo in range(0, nc, chunk):
        c = conn[lo:lo + chunk]
        for k, (p, q) in enumerate(_EDGES[3]):
            a = torch.minimum(c[:, p], c[:, q])
            d = (c[:, p] - c[:, q]).abs()
            direction = (d[:, None] == table[None, :]).to(torch.int32).argmax(dim=1).to(torch.int32)
            out[lo:lo + chunk, 4 + k] = nn + 7 * a + direction
    return out, 8 * nn


class FunctionSpace:
    """Continuous Lagrange space (degree 1 or 2, scalar or gdim-vector)."""

    def __init__(self, mesh: Mesh, degree: int = 1, dofmap=None, ndofs: int | None = None, bs: int = 1):
        self.mesh, self.degree, self.bs = mesh, degree, bs
        keep: list = []
        if dofmap is None:
            if degree == 1:
                # P1: the dofmap IS the geometry dofmap already resident in HBM (zero copy)
                self.ndofs, self.ndofs_cell = mesh.num_nodes, mesh.tdim + 1
                ptr = C.c_void_p(mesh.conn_ptr)
            else:
                dm, nd = lagrange_dofmap(mesh.tdim, mesh.conn, mesh.num_nodes, degree)
                self.ndofs, self.ndofs_cell = nd, dm.shape[1]
                ptr = _lib.as_ptr(dm, np.int32, keep)
                self._host_dofmap = dm
        else:
            self.ndofs, self.ndofs_cell = int(ndofs), dofmap.shape[1]
            ptr = _lib.as_ptr(dofmap, np.int32, keep)
        self._keep = [k for k in keep if _lib.is_device(k)] + [mesh]
        self._h = C.c_void_p()
        _lib.check(_lib.lib().cfx_space_create(mesh._h, degree, bs, C.c_int64(self.ndofs), ptr, self.ndofs_cell,
                                               C.byref(self._h)))
        self._dofmap_ptr = ptr

    def static_table_bytes(self) -> dict:
        """HBM held by the mesh-static tables built by the first assembly on this space (cfx_space_static_bytes)."""
        b = (C.c_int64 * 4)()
        _lib.check(_lib.lib().cfx_space_static_bytes(self._h, b))
        # (cell_neighbours: the cell -> cell table + the vertex runs of the culled classification, both the mesh's)
        return dict(dof_cells=int(b[0]), row_stencil=int(b[1]), row_tiles=int(b[2]), cell_neighbours=int(b[3]))

    @property
    def dofmap(self) -> np.ndarray:
        if hasattr(self, "_host_dofmap"):
            return self._host_dofmap
        if self.degree == 1:
            return self.mesh.conn
        raise RuntimeError("dofmap was supplied by the caller")

    def __del__(self):
        try:
            if self._h:
                _lib.load().cfx_space_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Function:
    """Finite element function: a space plus its dof values (numpy or torch)."""

    def __init__(self, V: FunctionSpace, values=None, name: str = "f"):
        self.function_space = V
        self.name = name
        self.values = np.zeros(V.ndofs * V.bs) if values is None else values

    def interpolate(self, f):
        """P1 only: sample f(x) (x of shape (3, n)) at the mesh vertices."""
        if self.function_space.degree != 1:
            raise NotImplementedError("interpolate is provided for P1 spaces")
        x = self.function_space.mesh.x
        self.values = np.ascontiguousarray(f(x.T), dtype=np.float64)
