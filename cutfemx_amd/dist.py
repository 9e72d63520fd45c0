"""Multi-GPU sharding of the hot path: one process per GPU, z-slabs of the
background mesh, RCCL point-to-point over xGMI between slab neighbours.

Two modes of `DistributedPoisson`:

* "owner" (default, what bench.py runs): a rank keeps 1 halo layer below and 2
  above its slab, receives the level-set values of the halo vertex planes from
  its neighbours (`phi.x.scatter_forward()`, python/demo/demo_poisson.py:157;
  the only exchange of the step: 2 planes per neighbour) and assembles ALL its
  local entities.  Every cell and ghost-penalty facet that touches an owned row
  is local, so the owned rows are complete without a reverse reduction; halo
  rows are scratch.  One row plan per step, the stencil kernels apply, no
  packing, ~5 % redundant halo work at 8 ranks of the 512^3 sphere.
* "reduce": the reference's flow restated literally (below): owned entities
  only, then `A.scatter_reverse()` / `b.scatter_reverse(add)` of the two
  boundary planes.  Kept as the cross-check of the owner mode.

Reference semantics of the reduce mode (python/demo/demo_poisson.py:51-55):

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


def balanced_boundaries(weights, world: int, halo_lo: int = 0, halo_hi: int = 0):
    """Split layers 0..n-1 into `world` contiguous chunks of near-equal weight.  With halo layers (a rank also works
    on `halo_lo` layers below and `halo_hi` above its chunk) the equal split is refined by moving one boundary one layer
    at a time while that lowers the heaviest rank's weight INCLUDING its halos: the halos of a slab through the middle
    of the sphere cost more than those of a slab near a pole (round 4: ranks 4 and 5 of 8 were the slowest by 3 %)."""
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
    if (halo_lo == 0 and halo_hi == 0) or world == 1:
        return bounds

    # halo-aware: the smallest T for which the layers can be dealt out front to back with every rank's weight --
    # its halo layers included -- at most T (a rank takes as many layers as fit; bisection on T)
    def cost(z0, z1):
        return cum[min(z1 + halo_hi, n)] - cum[max(z0 - halo_lo, 0)]

    def deal(T):
        b, z = [0], 0
        for p in range(world):
            left = world - 1 - p                    # ranks still to come: one layer each at least
            if p == world - 1:
                z1 = n
            else:
                z1 = z + 1
                while z1 + 1 <= n - left and cost(z, z1 + 1) <= T:
                    z1 += 1
            if cost(z, z1) > T:
                return None
            b.append(z1)
            z = z1
        return b
    lo_t, hi_t = 0.0, max(cost(bounds[p], bounds[p + 1]) for p in range(world))
    best = bounds
    for _ in range(60):
        mid = 0.5 * (lo_t + hi_t)
        b = deal(mid)
        if b is None:
            lo_t = mid
        else:
            best, hi_t = b, mid
    return best


def sphere_layer_weights(n: int, active_weight: float = 18.0, cut_weight: float = 1100.0):
    """Cost model per hex layer for the sphere workload, in units of one background cell
    (~3.5 ps on MI355X since the culled classification of round 4 -- selector scans, mark arrays, the
    block-wise sign test): an active (inside) cell costs ~26 of them more for its assembly, a cut cell
    ~1550 for sub-triangulation, runtime quadrature, local tensors and ghost-penalty facets -- picked by
    tools/rank_balance.py on the 8- and 4-rank partitions of the 512^3 case (round 5, after the bulk rows: 18 / 1100,
    slowest rank 2.95 / 5.22 ms against 3.02 / 5.30 with round 4's 26 / 1550; rounds 1-3 used 15 / 400 and 14 / 800
    against a costlier background cell).  The surface of a sphere
    between two parallel planes is 2 pi R dz (Archimedes), so the cut cells are spread evenly
    over the layers that meet the sphere: ~4.7 cut tets per h^2 of surface."""
    import os
    if os.environ.get("CFX_SLAB_WEIGHTS"):       # "active,cut": refitting experiments (tools/rank_balance.py)
        active_weight, cut_weight = (float(v) for v in os.environ["CFX_SLAB_WEIGHTS"].split(","))
    c, R = np.array([0.47, 0.43, 0.41]), 0.31
    z = (np.arange(n) + 0.5) / n
    r2 = np.maximum(R * R - (z - c[2]) ** 2, 0.0)           # radius^2 of the sphere's cross-section
    active = np.pi * r2 * n * n * 6.0                        # tets per layer inside the disc
    cut = np.where(r2 > 0.0, 4.7 * 2.0 * np.pi * R * n, 0.0)
    return 6.0 * n * n + active_weight * active + cut_weight * cut


def _zero(t):
    """t[:] = 0 on the engine's stream with its streaming fill (cfx_device_memset)."""
    import ctypes as C

    from . import _lib
    _lib.check(_lib.lib().cfx_device_memset(C.c_void_p(t.data_ptr()), 0, C.c_size_t(t.element_size() * t.numel())))


class TransportError(RuntimeError):
    """RCCL did not come up on every rank and the caller asked for RCCL or nothing (strict)."""


def decide_transport(backend_is_nccl: bool, rccl_ranks: int, world: int, strict: bool, why: str = "",
                     torch_p2p: bool = True) -> str:
    """What the ranks of a job exchange data through, decided together:
    'rccl'        torch.distributed runs on nccl and the library's own RCCL communicator came up on every rank;
    'rccl-torch'  it did not come up everywhere, but the job's nccl process group is there: the library hands its device
                  segments to a callback that posts them as grouped isend / irecv on that group -- still RCCL over xGMI,
                  driven by torch instead of by the library (`torch_p2p`, on unless CFX_DIST_TORCH_P2P=0);
    'host-staged' the library packs on the GPU, the bytes travel through gloo -- a rehearsal on a gloo job, and never a
                  silent fallback of an nccl job when `strict`: a benchmark that is meant to measure xGMI must not quietly
                  time a host path, so such a job raises TransportError instead."""
    if backend_is_nccl and rccl_ranks == world:
        return "rccl"
    if backend_is_nccl and torch_p2p:
        return "rccl-torch"
    if backend_is_nccl and strict:
        raise TransportError(f"the RCCL communicator came up on {rccl_ranks} of {world} ranks ({why or 'another rank failed'}); "
                             "refusing the host-staged fallback (CFX_DIST_STRICT=1: set CFX_REHEARSE=1 for a one-GPU "
                             "rehearsal over gloo)")
    return "host-staged"


class DistComm:
    """cfx_comm_t of this rank: RCCL over xGMI when torch.distributed runs on the nccl backend (the 128-byte
    ncclUniqueId made by rank 0 travels through torch.distributed, which is only the launcher's side channel
    here), else the library's host-staged mode with a callback that moves the bytes through torch.distributed
    (gloo): what a one-GPU box can run with several ranks.  Data-path exchanges then go through
    cfx_dist_scatter_forward / cfx_dist_scatter_reverse_add / cfx_dist_scatter_reverse_matrix."""

    def __init__(self, group=None, strict: bool | None = None):
        import ctypes as C
        import os
        if strict is None:
            strict = os.environ.get("CFX_DIST_STRICT") == "1"

        import torch
        import torch.distributed as dist

        from . import _lib
        self._lib, self.group = _lib, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self._h = C.c_void_p()
        l = _lib.lib()
        backend_nccl = dist.get_backend(group) == "nccl"
        self.rccl = backend_nccl
        self.rccl_ranks, self.fallback_reason = 0, ""
        self._transport = "host-staged"
        host_group = group
        # CFX_DIST_TRANSPORT: "torch" skips the library's own communicator (exchanges as isend / irecv of device tensors on the
        # job's process group), "device" does the same on any backend (gloo: the callback stages through the host itself -- the
        # one-GPU rehearsal of the device-callback path), "host" forces the host-staged mode
        forced = os.environ.get("CFX_DIST_TRANSPORT", "")
        torch_p2p = os.environ.get("CFX_DIST_TORCH_P2P", "1") != "0"
        if forced == "host":
            self.rccl = False
            if backend_nccl:
                if strict:
                    raise TransportError("CFX_DIST_TRANSPORT=host on an nccl job with CFX_DIST_STRICT=1")
                host_group = dist.new_group(backend="gloo")
        elif forced in ("torch", "device"):
            self.rccl = False
            self._device_callback(group, via_host=not backend_nccl)
            self.fallback_reason = f"CFX_DIST_TRANSPORT={forced}"
        if self.rccl:
            # every rank must end up on the same transport: the ranks agree on whether RCCL came up everywhere, and move
            # together to the next transport otherwise (decide_transport)
            ok, why = 1, ""
            buf = (C.c_char * 128)()
            if self.rank == 0 and l.cfx_dist_unique_id(buf) != 0:
                buf = (C.c_char * 128)()    # all zero: "no id", every rank then skips the RCCL communicator
            obj = [bytes(buf)]
            dist.broadcast_object_list(obj, src=0, group=group)
            try:
                if not any(obj[0]):
                    raise RuntimeError("rank 0 could not make an ncclUniqueId: " + (l.cfx_last_error() or b"").decode())
                uid = (C.c_char * 128).from_buffer_copy(obj[0])
                _lib.check(l.cfx_dist_comm_create(self.world, self.rank, uid, C.byref(self._h)))
            except Exception as e:          # noqa: BLE001 -- reported below, then the collective fallback
                ok, why = 0, f"{type(e).__name__}: {e}"
            flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=group)
            self.rccl_ranks = int(flag.item())
            self.fallback_reason = why
            # (raises TransportError on every rank alike when strict and no RCCL path is left: the job ends instead of
            # timing a host path)
            decided = decide_transport(True, self.rccl_ranks, self.world, strict, why, torch_p2p)
            if decided == "rccl":
                self._transport = "rccl"
            else:
                import sys
                print(f"cutfemx_amd.dist: the library's RCCL communicator is unavailable on some rank ({why or 'another rank'}); "
                      + ("exchanging device buffers through torch.distributed's nccl group instead" if decided == "rccl-torch"
                         else "falling back to host-staged exchanges over gloo"), file=sys.stderr)
                if self._h:
                    l.cfx_dist_comm_destroy(self._h)
                    self._h = C.c_void_p()
                self.rccl = False
                if decided == "rccl-torch":
                    self._device_callback(group, via_host=False)
                else:
                    host_group = dist.new_group(backend="gloo")
        if not self.rccl and not self._h:
            group = host_group
            self.group = group

            def exchange(_user, n, peers, send, send_bytes, recv, recv_bytes):
                try:
                    ops, keep = [], []
                    for i in range(n):
                        if send_bytes[i] > 0:
                            src = (C.c_uint8 * send_bytes[i]).from_address(send[i])
                            t = torch.frombuffer(src, dtype=torch.uint8).clone()
                            keep.append(t)
                            ops.append(dist.P2POp(dist.isend, t, int(peers[i]), group))
                        if recv_bytes[i] > 0:
                            t = torch.empty(recv_bytes[i], dtype=torch.uint8)
                            keep.append((t, recv[i], recv_bytes[i]))
                            ops.append(dist.P2POp(dist.irecv, t, int(peers[i]), group))
                    if ops:
                        for req in dist.batch_isend_irecv(ops):
                            req.wait()
                    for k in keep:
                        if isinstance(k, tuple):
                            C.memmove(k[1], k[0].data_ptr(), k[2])
                    return 0
                except Exception:        # (an exception must not cross the C frame)
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = _lib.HOST_EXCHANGE_FN(exchange)      # kept alive with the communicator
            _lib.check(l.cfx_dist_comm_create_host(self.world, self.rank, self._cb, None, C.byref(self._h)))

    def _device_callback(self, group, via_host: bool):
        """cfx_dist_comm_create_device with a callback that posts the library's device segments as one batch of isend /
        irecv on `group` (nccl: RCCL over xGMI, driven by torch); via_host: the tensors travel as host copies (a gloo
        group -- the one-GPU rehearsal of this path)."""
        import ctypes as C

        import torch
        import torch.distributed as dist
        dev = torch.device("cuda", torch.cuda.current_device())

        def exchange(_user, n, peers, send, send_bytes, recv, recv_bytes):
            try:
                ops, back = [], []
                for i in range(n):
                    if send_bytes[i] > 0:
                        t = as_torch(send[i], send_bytes[i], "uint8", dev)
                        ops.append(dist.P2POp(dist.isend, t.cpu() if via_host else t, int(peers[i]), group))
                    if recv_bytes[i] > 0:
                        t = as_torch(recv[i], recv_bytes[i], "uint8", dev)
                        if via_host:
                            h = torch.empty(recv_bytes[i], dtype=torch.uint8)
                            back.append((t, h))
                            t = h
                        ops.append(dist.P2POp(dist.irecv, t, int(peers[i]), group))
                if ops:
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()
                for t, h in back:
                    t.copy_(h)
                # (the library's stream may not be torch's current one: everything has landed when this returns)
                torch.cuda.current_stream().synchronize()
                return 0
            except Exception:            # (an exception must not cross the C frame)
                import traceback
                traceback.print_exc()
                return 1
        self._cb = self._lib.HOST_EXCHANGE_FN(exchange)     # (same signature; kept alive with the communicator)
        self._lib.check(self._lib.lib().cfx_dist_comm_create_device(self.world, self.rank, self._cb, None, C.byref(self._h)))
        self._transport = "host-callback-device" if via_host else "rccl-torch"

    @property
    def transport(self) -> str:
        """'rccl' (the library's communicator: point-to-point over xGMI), 'rccl-torch' (the same links through
        torch.distributed's nccl group), 'host-staged' (pinned host buffers + gloo)."""
        return self._transport

    def _exchanges(self, triples):
        """[(peer, (send_lo, send_hi), (recv_lo, recv_hi)), ...] -> cfx_dist_exchange array (contiguous ranges)"""
        arr = (self._lib.DistExchange * max(len(triples), 1))()
        self._keep = []
        for i, (peer, snd, rcv) in enumerate(triples):
            arr[i].peer = peer
            for side, part in (("send", snd), ("recv", rcv)):
                if isinstance(part, tuple):                      # contiguous range [lo, hi)
                    setattr(arr[i], side + "_offset", part[0])
                    setattr(arr[i], side + "_count", part[1] - part[0])
                else:                                            # int32 index list in HBM (an index map's shared / ghost list)
                    assert part.is_cuda and part.dtype == self._torch_int32() and part.is_contiguous()
                    self._keep.append(part)
                    setattr(arr[i], side + "_index", part.data_ptr())
                    setattr(arr[i], side + "_count", part.numel())
        return arr

    @staticmethod
    def _torch_int32():
        import torch
        return torch.int32

    def scatter_forward(self, x, triples):
        import ctypes as C
        self._lib.check(self._lib.lib().cfx_dist_scatter_forward(self._h, C.c_void_p(x.data_ptr()), len(triples),
                                                                 self._exchanges(triples)))

    def scatter_reverse_add(self, x, triples):
        import ctypes as C
        self._lib.check(self._lib.lib().cfx_dist_scatter_reverse_add(self._h, C.c_void_p(x.data_ptr()), len(triples),
                                                                     self._exchanges(triples)))

    def scatter_reverse_matrix(self, A, triples):
        """rows (send_lo, send_hi) of this rank's matrix are added to rows (recv_lo, recv_hi) of the peer's"""
        import ctypes as C
        arr = (self._lib.DistRowExchange * max(len(triples), 1))()
        for i, (peer, (s0, s1), (r0, r1)) in enumerate(triples):
            arr[i].peer, arr[i].send_row_lo, arr[i].send_row_hi, arr[i].recv_row_lo, arr[i].recv_row_hi = peer, s0, s1, r0, r1
        self._lib.check(self._lib.lib().cfx_dist_scatter_reverse_matrix(self._h, A._p, C.c_void_p(A.values_ptr),
                                                                        len(triples), arr))

    def indicator_or(self, ind, triples, forward_triples=None):
        import ctypes as C
        l = self._lib.lib()
        self._lib.check(l.cfx_dist_indicator_or(self._h, C.c_void_p(ind.data_ptr()), len(triples), self._exchanges(triples)))
        if forward_triples is not None:
            self._lib.check(l.cfx_dist_indicator_forward(self._h, C.c_void_p(ind.data_ptr()), len(forward_triples),
                                                         self._exchanges(forward_triples)))

    def close(self):
        if self._h:
            self._lib.load().cfx_dist_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class SlabPartition:
    n: int
    world: int
    rank: int
    bounds: list
    halo: int = 3            # layers kept below the slab
    halo_hi: int | None = None  # layers kept above (None: same as below)

    @classmethod
    def create(cls, n, world, rank, weights=None, halo=3, halo_hi=None):
        w = sphere_layer_weights(n) if weights is None else weights
        return cls(n, world, rank, balanced_boundaries(w, world, halo, halo if halo_hi is None else halo_hi), halo, halo_hi)

    @classmethod
    def create_owner(cls, n, world, rank, weights=None):
        """Layout of the owner-computes mode: owned rows are planes (z0, z1]; their cells lie in
        layers z0..z1 and the ghost-penalty facets of those cells reach layers z0-1 and z1+1."""
        return cls.create(n, world, rank, weights, halo=1, halo_hi=2)

    # --- hex layers -----------------------------------------------------------
    @property
    def z0(self): return self.bounds[self.rank]
    @property
    def z1(self): return self.bounds[self.rank + 1]
    @property
    def lz0(self): return max(self.z0 - self.halo, 0)
    @property
    def lz1(self): return min(self.z1 + (self.halo if self.halo_hi is None else self.halo_hi), self.n)
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


def halo_forward(phi, part: SlabPartition, group=None, comm: DistComm | None = None):
    """Fill the halo vertex planes of the level-set vector from their owners
    (`phi.x.scatter_forward()`): planes lz0..z0 come from the rank below, planes
    z1+1..lz1 from the rank above; planes are contiguous slices.
    An array in HBM goes through the C ABI (cfx_dist_scatter_forward on `comm`); a host tensor -- the CPU tests,
    where the oracle stands in for the engine -- through torch.distributed directly."""
    import torch
    import torch.distributed as dist
    if phi.is_cuda:
        if comm is None:
            raise ValueError("halo_forward of a device array needs the rank's DistComm")
        ps, tri = part.plane_size, []
        sl = lambda g0, g1: (ps * (g0 - part.lz0), ps * (g1 + 1 - part.lz0))   # global planes [g0, g1] -> local range
        if part.rank > 0:
            assert part.lz0 > part.bounds[part.rank - 1], "slab thinner than the halo of its neighbour"
            tri.append((part.rank - 1, sl(part.z0 + 1, min(part.z0 + 2, part.n)), sl(part.lz0, part.z0)))
        if part.rank < part.world - 1:
            assert part.z1 - 1 > part.z0 or part.rank == 0, "slab thinner than the halo of its neighbour"
            tri.append((part.rank + 1, sl(part.z1 - 1, part.z1), sl(part.z1 + 1, part.lz1)))
        comm.scatter_forward(phi, tri)
        return phi
    stage = False
    ps = part.plane_size
    ops, recvs = [], []

    def plane_slice(g0, g1):   # global vertex planes [g0, g1] -> local element range
        return ps * (g0 - part.lz0), ps * (g1 + 1 - part.lz0)

    if part.rank > 0:
        below = part.bounds[part.rank - 1]
        assert part.lz0 > below, "slab thinner than the halo of its neighbour"
        r0, r1 = plane_slice(part.lz0, part.z0)                       # owned by rank-1
        up_hi = min(part.z0 + 2, part.n)                              # its halo above: planes z0+1..z0+2
        s0, s1 = plane_slice(part.z0 + 1, up_hi)
        recvs.append((r0, r1, part.rank - 1))
        ops.append((s0, s1, part.rank - 1))
    if part.rank < part.world - 1:
        r0, r1 = plane_slice(part.z1 + 1, part.lz1)                   # owned by rank+1
        s0, s1 = plane_slice(part.z1 - 1, part.z1)                    # its halo below: planes z1-1..z1
        assert part.z1 - 1 > part.z0 or part.rank == 0, "slab thinner than the halo of its neighbour"
        recvs.append((r0, r1, part.rank + 1))
        ops.append((s0, s1, part.rank + 1))
    p2p, bufs = [], []
    for (s0, s1, peer), (r0, r1, _) in zip(ops, recvs):
        send = phi[s0:s1].cpu() if stage else phi[s0:s1]
        recv = torch.empty(r1 - r0, dtype=phi.dtype, device="cpu" if stage else phi.device)
        p2p.append(dist.P2POp(dist.isend, send, peer, group))
        p2p.append(dist.P2POp(dist.irecv, recv, peer, group))
        bufs.append((r0, r1, recv))
    if p2p:
        for req in dist.batch_isend_irecv(p2p):
            req.wait()
    for r0, r1, recv in bufs:
        phi[r0:r1] = recv.to(phi.device)
    return phi


def scatter_reverse(values, row_ptr, part: SlabPartition, group=None):
    """Add the ghost-row contributions into their owners (A.scatter_reverse()).

    `values`: torch tensor of CSR values (or a dense vector with row_ptr=None);
    `row_ptr`: callable row -> value index (CSR indptr lookup) or None for vectors.
    Works on any torch.distributed backend (nccl = RCCL on the GPU, gloo in tests).
    """
    import torch
    import torch.distributed as dist
    if values.is_cuda:
        raise ValueError("device arrays are reduced through DistComm.scatter_reverse_add / scatter_reverse_matrix")
    stage = False
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
    typestr = {"int32": "<i4", "int64": "<i8", "float64": "<f8", "float32": "<f4", "int8": "|i1", "uint8": "|u1"}[dtype]
    if n == 0:
        return torch.empty(0, dtype=getattr(torch, dtype), device=device)
    return torch.as_tensor(_DevView(ptr, n, typestr), device=device)


class DistributedPoisson:
    """The cut Poisson hot path on this rank's slab (GPU)."""

    def __init__(self, part: SlabPartition, device, order=4, gamma=40.0, gamma_g=0.1, mode="owner"):
        import torch

        import cutfemx_amd as cfx
        self.torch, self.cfx = torch, cfx
        self.part, self.device, self.order, self.gamma, self.gamma_g = part, device, order, gamma, gamma_g
        if mode not in ("owner", "reduce"):
            raise ValueError("mode must be 'owner' or 'reduce'")
        self.mode = mode
        import torch.distributed as tdist
        # (no process group: a single process timing one rank's slab, tools/rank_balance.py)
        self.comm = DistComm() if part.world > 1 and tdist.is_available() and tdist.is_initialized() else None
        self.mesh = cfx.Mesh.create_slab(part.n, part.lz0, part.nz_local)
        self.V = cfx.FunctionSpace(self.mesh, 1)
        n = part.n
        ax = torch.arange(n + 1, device=device, dtype=torch.float64) / n
        az = torch.arange(part.lz0, part.lz1 + 1, device=device, dtype=torch.float64) / n
        cx, cy, cz, R = 0.47, 0.43, 0.41, 0.31
        d2 = (az[:, None, None] - cz) ** 2 + (ax[None, :, None] - cy) ** 2 + (ax[None, None, :] - cx) ** 2
        self.phi_values = (torch.sqrt(d2) - R).reshape(-1).contiguous()
        self.phi = cfx.Function(self.V, self.phi_values)
        nn = self.mesh.num_nodes
        # first guess of the CSR value buffer; _matrix() grows it to the real nnz when a step needs more
        self.values = torch.zeros(nn + 40 * int(0.25 * nn + 100000), device=device, dtype=torch.float64)
        self.b = torch.zeros(nn, device=device, dtype=torch.float64)

    def _matrix(self, a):
        """create_matrix(a) over the persistent value buffer, reallocated when the pattern outgrows it."""
        from . import fem
        try:
            return fem.create_matrix(a, values=self.values)
        except ValueError:
            A = fem.create_matrix(a)            # library-owned values: only to learn nnz
            nnz = A.nnz
            del A
            self.values = self.torch.zeros(nnz + nnz // 8, device=self.device, dtype=self.torch.float64)
            return fem.create_matrix(a, values=self.values)

    def step(self):
        return self.step_owner() if self.mode == "owner" else self.step_reduce()

    def exchange_only(self):
        """The level-set halo exchange of a step on its own (what bench.py times as `exchange_ms`)."""
        if self.part.world > 1:
            halo_forward(self.phi_values, self.part, comm=self.comm)

    def step_owner(self):
        """Level-set halo forward, then the serial hot path on the local slab; owned rows are complete."""
        torch, cfx, part, dev = self.torch, self.cfx, self.part, self.device
        from . import fem, poisson
        if part.world > 1:
            # the halo planes hold whatever the neighbours computed: poison them first so that a broken
            # exchange cannot go unnoticed behind the analytic initial values
            ps = part.plane_size
            if part.rank > 0:
                self.phi_values[: ps * (part.z0 + 1 - part.lz0)] = float("nan")
            if part.rank < part.world - 1:
                self.phi_values[ps * (part.z1 + 1 - part.lz0):] = float("nan")
            halo_forward(self.phi_values, part, comm=self.comm)

        def local_path():
            cd = cfx.cut(self.phi)
            system = poisson.build_forms(self.V, cd, order=self.order, gamma=self.gamma, gamma_g=self.gamma_g)
            _zero(self.b)
            import os
            if os.environ.get("CFX_OVERLAP", "0") == "1":
                # (opt-in: the linear form on a second HIP stream beside the pattern + matrix -- measured per rank in
                # bench.py's projection; off by default)
                system.L.prepare()
                with fem.overlap() as lanes:
                    lanes.side(lambda: fem.assemble_vector(system.L, self.b))
                    A = self._matrix(system.a)
                    A.set_value(0.0)
                    fem.assemble_matrix(system.a, A=A)
            else:
                A = self._matrix(system.a)
                A.set_value(0.0)
                fem.assemble_matrix(system.a, A=A)
                fem.assemble_vector(system.L, self.b)
            dom = fem.deactivate_outside(A, self.b, fem.active_domain(system.a))
            return dict(A=A, dom=dom, system=system)
        # the local path is one sync-free step of this rank's loop (cutfemx_amd.step): sizes stay in HBM, one read-back
        # at its end; the exchange above is outside it (CFX_BENCH_STEP=0: every size read back where it is produced)
        import os
        if os.environ.get("CFX_BENCH_STEP", "1") == "0":
            return local_path()
        return cfx.run_step(local_path, key=f"slab-{part.n}-{part.world}-{part.rank}-{self.order}")

    def counters(self, info):
        """Counts of the owned share (active dofs, quadrature points, cut cells ...) of a step's
        result: diagnostics, computed on request so that they stay out of the step itself."""
        if "active_dofs_owned" in info:
            return info
        torch, part, dev = self.torch, self.part, self.device
        system, dom, A = info["system"], info["dom"], info["A"]
        r_lo, r_hi = part.owned_rows
        c_lo, c_hi = part.owned_cells
        inactive = as_torch(dom._view()[2], dom._view()[3], "int32", dev)
        inside = as_torch(system.inside_cells[0], system.inside_cells.size, "int32", dev)
        vr, ir = system.volume_rules, system.interface_rules

        def owned_points(rules):
            parents = as_torch(rules._view.parent_map, rules.num_rules, "int32", dev)
            offs = as_torch(rules._view.offsets, rules.num_rules + 1, "int32", dev)
            k = torch.searchsorted(parents, torch.tensor([c_lo, c_hi], device=dev, dtype=torch.int32))
            return (offs[k[1]] - offs[k[0]]).to(torch.int64), (k[1] - k[0]).to(torch.int64)

        qv, nv = owned_points(vr)
        qi, ni = owned_points(ir)
        ki = torch.searchsorted(inside, torch.tensor([c_lo, c_hi], device=dev, dtype=torch.int32))
        n_in_own = ((inactive >= r_lo) & (inactive < r_hi)).sum()
        counts = torch.stack([n_in_own, (ki[1] - ki[0]).to(torch.int64), qv, qi, ni, nv]).tolist()
        ghost_n = 0 if system.ghost_facets is None else system.ghost_facets.size
        return dict(active_dofs_owned=(r_hi - r_lo) - counts[0], nnz=A.nnz, n_inside=counts[1],
                    nq_volume=counts[2], nq_interface=counts[3], n_cut=counts[4], n_vol_rules=counts[5],
                    n_ghost=ghost_n, A=A, dom=dom)

    def step_reduce(self):
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
        A = self._matrix(a_all)
        A.set_value(0.0)
        _zero(self.b)
        fem.assemble_matrix(a_own, A=A)
        fem.assemble_vector(L_own, self.b)
        # --- A.scatter_reverse(); b.scatter_reverse(add)
        if part.world > 1:
            tri = [(peer, part.plane_rows(sp), part.plane_rows(rp)) for peer, sp, rp in part.exchanges()]
            self.comm.scatter_reverse_matrix(A, tri)      # cfx_dist_scatter_reverse_matrix
            self.comm.scatter_reverse_add(self.b, tri)    # cfx_dist_scatter_reverse_add
        dom = fem.deactivate_outside(A, self.b, fem.active_domain(a_all))
        r_lo, r_hi = part.owned_rows
        inactive = as_torch(dom._view()[2], dom._view()[3], "int32", dev)
        n_inactive_owned = int(((inactive >= r_lo) & (inactive < r_hi)).sum())
        return dict(active_dofs_owned=(r_hi - r_lo) - n_inactive_owned, nnz=A.nnz, n_inside=i1 - i0,
                    nq_volume=vol_o.total_points_owned, nq_interface=itf_o.total_points_owned,
                    n_cut=itf_o.num_rules, n_vol_rules=vol_o.num_rules, n_ghost=int(ghost_o.shape[0]),
                    A=A, dom=dom)
