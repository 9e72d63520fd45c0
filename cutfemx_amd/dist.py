"""Multi-GPU sharding of the hot path: one process per GPU, z-slabs of the
background mesh, RCCL point-to-point over xGMI for the row reduction.

Reference semantics reproduced (python/demo/demo_poisson.py:51-55):

    A.scatter_reverse(); b.scatter_reverse(add); deactivate_outside(A, b, active_domain(a))

DOLFINx gives every rank its owned cells plus facet-sharing ghost cells, lets a
rank assemble its owned cells/facets into owned AND ghost rows, and sends the
ghost-row contributions to the row owners.  Here:

* cells are partitioned by hex layers [z0, z1) (weighted by active cells so the
  sphere does not unbalance the ranks); a rank keeps `halo` extra layers on each
  side (3: the ghost-penalty stencil of a row reaches two cells beyond it);
* a vertex plane shared by two slabs belongs to the lower rank ("lowest rank
  touching the vertex"): rank p owns planes (z0, z1], rank 0 also plane 0;
* a ghost-penalty facet belongs to the rank owning its lower cell
  (`ghost_penalty_facets(include_ghosts=False)`, python/cutfemx/cut.py:367-374);
* sparsity is built from ALL local entities, so the rows a rank sends and the
  rows their owner holds have identical column sets in identical order (local
  ids differ by a constant): a plane of rows is one contiguous slice of the CSR
  value array and the exchange needs no packing.  Per step a rank sends two
  slices (plane z0 down, plane z1+1 up) and receives two.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


def balanced_boundaries(weights, world: int):
    """Split layers 0..n-1 into `world` contiguous chunks of near-equal weight."""
    w = np.asarray(weights, dtype=np.float64)
    n = w.size
    cum = np.concatenate([[0.0], np.cumsum(w)])
    bounds = [0]
    for p in range(1, world):
        target = cum[-1] * p / world
        z = int(np.searchsorted(cum, target))
        z = min(max(z, bounds[-1] + 1), n - (world - p))
        bounds.append(z)
    bounds.append(n)
    return bounds


def sphere_layer_weights(n: int, active_weight: float = 140.0):
    """Cost model per hex layer for the sphere workload: 1 per cell (classification
    stream) + active_weight per cell that is inside or cut (everything else)."""
    c, R = np.array([0.47, 0.43, 0.41]), 0.31
    z = (np.arange(n) + 0.5) / n
    r2 = np.maximum(R * R - (z - c[2]) ** 2, 0.0)           # radius^2 of the sphere's cross-section
    active = np.pi * r2 * n * n * 6.0                        # tets per layer inside the disc
    return 6.0 * n * n + active_weight * active


@dataclass
class SlabPartition:
    n: int
    world: int
    rank: int
    bounds: list
    halo: int = 3

    @classmethod
    def create(cls, n, world, rank, weights=None, halo=3):
        w = sphere_layer_weights(n) if weights is None else weights
        return cls(n, world, rank, balanced_boundaries(w, world), halo)

    # --- hex layers -----------------------------------------------------------
    @property
    def z0(self): return self.bounds[self.rank]
    @property
    def z1(self): return self.bounds[self.rank + 1]
    @property
    def lz0(self): return max(self.z0 - self.halo, 0)
    @property
    def lz1(self): return min(self.z1 + self.halo, self.n)
    @property
    def nz_local(self): return self.lz1 - self.lz0
    # --- numbering ------------------------------------------------------------
    @property
    def plane_size(self): return (self.n + 1) ** 2
    @property
    def cells_per_layer(self): return 6 * self.n * self.n
    @property
    def vertex_offset(self): return self.plane_size * self.lz0          # global = local + offset
    @property
    def cell_offset(self): return self.cells_per_layer * self.lz0
    @property
    def owned_cells(self):
        """[lo, hi) in local cell ids."""
        return (self.cells_per_layer * (self.z0 - self.lz0), self.cells_per_layer * (self.z1 - self.lz0))
    @property
    def owned_rows(self):
        """[lo, hi) in local vertex ids: planes (z0, z1], plus plane 0 on rank 0."""
        first = self.z0 + (0 if self.rank == 0 else 1)
        return (self.plane_size * (first - self.lz0), self.plane_size * (self.z1 + 1 - self.lz0))

    def plane_rows(self, gz):
        """[lo, hi) local vertex ids of global vertex plane gz."""
        return (self.plane_size * (gz - self.lz0), self.plane_size * (gz + 1 - self.lz0))

    def exchanges(self):
        """(peer, send_plane, recv_plane) triples in global plane indices."""
        out = []
        if self.rank > 0:                       # lower neighbour owns plane z0; it sends me its share of plane z0+1
            out.append((self.rank - 1, self.z0, self.z0 + 1))
        if self.rank < self.world - 1:          # I own plane z1; the facets I own reach plane z1+1 of the upper rank
            out.append((self.rank + 1, self.z1 + 1, self.z1))
        return out


def scatter_reverse(values, row_ptr, part: SlabPartition, group=None):
    """Add the ghost-row contributions into their owners (A.scatter_reverse()).

    `values`: torch tensor of CSR values (or a dense vector with row_ptr=None);
    `row_ptr`: callable row -> value index (CSR indptr lookup) or None for vectors.
    Works on any torch.distributed backend (nccl = RCCL on the GPU, gloo in tests).
    """
    import torch
    import torch.distributed as dist
    # gloo (CPU rehearsal of the GPU path) moves device slices through the host
    stage = values.is_cuda and dist.get_backend(group) == "gloo"
    ops, recvs = [], []
    for peer, send_plane, recv_plane in part.exchanges():
        s_lo, s_hi = part.plane_rows(send_plane)
        r_lo, r_hi = part.plane_rows(recv_plane)
        if row_ptr is not None:
            s_lo, s_hi, r_lo, r_hi = row_ptr(s_lo), row_ptr(s_hi), row_ptr(r_lo), row_ptr(r_hi)
        send = values[s_lo:s_hi].cpu() if stage else values[s_lo:s_hi]
        recv = torch.empty(r_hi - r_lo, dtype=values.dtype, device="cpu" if stage else values.device)
        ops.append(dist.P2POp(dist.isend, send, peer, group))
        ops.append(dist.P2POp(dist.irecv, recv, peer, group))
        recvs.append((r_lo, r_hi, recv, s_hi - s_lo, peer))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for r_lo, r_hi, recv, _, _ in recvs:
        values[r_lo:r_hi] += recv.to(values.device)
    return values


class _DevView:
    """Expose a raw HBM pointer to torch without a copy."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


def as_torch(ptr, n, dtype, device):
    import torch
    typestr = {"int32": "<i4", "int64": "<i8", "float64": "<f8", "int8": "|i1"}[dtype]
    if n == 0:
        return torch.empty(0, dtype=getattr(torch, dtype), device=device)
    return torch.as_tensor(_DevView(ptr, n, typestr), device=device)


class DistributedPoisson:
    """The cut Poisson hot path on this rank's slab (GPU)."""

    def __init__(self, part: SlabPartition, device, order=4, gamma=40.0, gamma_g=0.1):
        import torch

        import cutfemx_amd as cfx
        self.torch, self.cfx = torch, cfx
        self.part, self.device, self.order, self.gamma, self.gamma_g = part, device, order, gamma, gamma_g
        self.mesh = cfx.Mesh.create_slab(part.n, part.lz0, part.nz_local)
        self.V = cfx.FunctionSpace(self.mesh, 1)
        n = part.n
        ax = torch.arange(n + 1, device=device, dtype=torch.float64) / n
        az = torch.arange(part.lz0, part.lz1 + 1, device=device, dtype=torch.float64) / n
        cx, cy, cz, R = 0.47, 0.43, 0.41, 0.31
        d2 = (az[:, None, None] - cz) ** 2 + (ax[None, :, None] - cy) ** 2 + (ax[None, None, :] - cx) ** 2
        self.phi = cfx.Function(self.V, (torch.sqrt(d2) - R).reshape(-1).contiguous())
        nn = self.mesh.num_nodes
        self.values = torch.zeros(nn + 40 * int(0.25 * nn + 100000), device=device, dtype=torch.float64)
        self.b = torch.zeros(nn, device=device, dtype=torch.float64)

    def step(self):
        torch, cfx, part = self.torch, self.cfx, self.part
        from . import fem
        from .cut import RuntimeQuadratureRules
        cd = cfx.cut(self.phi)
        inside_ptr, n_inside = cfx.locate_entities_device(cd, "phi<0")
        vol = cfx.runtime_quadrature(cd, "phi<0", self.order)
        itf = cfx.runtime_quadrature(cd, "phi=0", self.order)
        normals = cfx.normal(cd, itf, device=True)
        ghost = cfx.ghost_penalty_facets(cd, "phi<0")
        dev = self.device
        c_lo, c_hi = part.owned_cells
        # --- owned subsets (sorted lists -> slices; facets by their lower cell)
        inside = as_torch(inside_ptr, n_inside, "int32", dev)
        i0, i1 = (int(v) for v in torch.searchsorted(inside, torch.tensor([c_lo, c_hi], device=dev, dtype=torch.int32)))
        vol_o = vol.slice_by_parent(c_lo, c_hi, dev)
        itf_o = itf.slice_by_parent(c_lo, c_hi, dev)
        rows = as_torch(ghost.ptr, 4 * ghost.size, "int32", dev).view(-1, 4)
        ghost_o = rows[(rows[:, 0] >= c_lo) & (rows[:, 0] < c_hi)].contiguous()
        # --- sparsity from ALL local entities, assembly from the owned ones
        a_all = fem.form([fem.Integral(fem.STIFFNESS, cells=(inside_ptr, n_inside), rules=vol, qdegree=0),
                          fem.Integral(fem.NITSCHE, rules=itf, point_data=normals, params=(self.gamma,)),
                          fem.Integral(fem.GHOST_GRADJUMP, facets=ghost, params=(self.gamma_g,), qdegree=0)], self.V)
        a_own = fem.form([fem.Integral(fem.STIFFNESS, cells=inside[i0:i1], rules=vol_o, qdegree=0),
                          fem.Integral(fem.NITSCHE, rules=itf_o, point_data=normals, params=(self.gamma,)),
                          fem.Integral(fem.GHOST_GRADJUMP, facets=ghost_o, params=(self.gamma_g,), qdegree=0)], self.V)
        L_own = fem.form([fem.Integral(fem.SOURCE, cells=inside[i0:i1], rules=vol_o,
                                       params=(fem.F_POISSON_RHS, 1.0), qdegree=4),
                          fem.Integral(fem.NITSCHE_RHS, rules=itf_o, point_data=normals,
                                       params=(self.gamma, fem.F_SINPROD, 1.0))], self.V)
        A = fem.create_matrix(a_all, values=self.values)
        self.values[:A.nnz].zero_()
        self.b.zero_()
        fem.assemble_matrix(a_own, A=A)
        fem.assemble_vector(L_own, self.b)
        # --- A.scatter_reverse(); b.scatter_reverse(add)
        if part.world > 1:
            indptr = as_torch(A._view.indptr, A.nrows + 1, "int64", dev)
            planes = sorted({p for _, s, r in part.exchanges() for p in (s, r)})
            bounds = sorted({b for p in planes for b in part.plane_rows(p)})
            lut = dict(zip(bounds, indptr[torch.tensor(bounds, device=dev)].tolist()))
            scatter_reverse(self.values, lambda row: lut[row], part)
            scatter_reverse(self.b, None, part)
        dom = fem.deactivate_outside(A, self.b, fem.active_domain(a_all))
        r_lo, r_hi = part.owned_rows
        inactive = as_torch(dom._id, dom._ni, "int32", dev)
        n_inactive_owned = int(((inactive >= r_lo) & (inactive < r_hi)).sum())
        return dict(active_dofs_owned=(r_hi - r_lo) - n_inactive_owned, nnz=A.nnz, n_inside=i1 - i0,
                    nq_volume=vol_o.total_points_owned, nq_interface=itf_o.total_points_owned,
                    n_cut=itf_o.num_rules, n_vol_rules=vol_o.num_rules, n_ghost=int(ghost_o.shape[0]),
                    A=A, dom=dom)
