"""N>1 path on CPU: two gloo ranks, z-slab partition with halo, local assembly
of the owned entities (by the oracle -- the GPU is not available here), then
cutfemx_amd.dist.scatter_reverse.  Every rank's owned rows must equal the rows
of the serial matrix / vector."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, n, q):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist

    from cutfemx_amd.dist import SlabPartition, scatter_reverse
    from helpers import level_set_values, oracle_poisson
    from oracle import pyoracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gm = O.mesh_box(3, n)
        gphi = level_set_values(gm.x, 3)
        gref = oracle_poisson(O, gm, gphi)                       # serial reference
        G = sp.csr_matrix((gref["values"], gref["indices"], gref["indptr"]), shape=(gm.nnodes,) * 2)

        part = SlabPartition.create(n, world, rank, weights=np.ones(n))
        nv, nc = part.plane_size * (part.nz_local + 1), part.cells_per_layer * part.nz_local
        conn = gm.conn[part.cell_offset:part.cell_offset + nc] - part.vertex_offset
        x = gm.x[part.vertex_offset:part.vertex_offset + nv]
        lm = O.Mesh(3, x, conn)
        phi = gphi[part.vertex_offset:part.vertex_offset + nv]
        dom = O.classify(lm.conn, phi)
        assert np.array_equal(dom, gref["domain"][part.cell_offset:part.cell_offset + nc])
        inside = O.locate_entities(dom, "phi<0")
        vol = O.runtime_quadrature(lm, lm.conn, phi, dom, "phi<0", 4)
        itf = O.runtime_quadrature(lm, lm.conn, phi, dom, "phi=0", 4)
        nrm = O.evaluate_normals(lm, lm.conn, phi, itf)
        ghost = O.ghost_penalty_facets(lm, dom, "phi<0")
        V = O.Space(lm.conn, lm.nnodes, 1)

        def forms(ins, v, i, nr, gh):
            a = [O.Integral(O.CELL, O.K_STIFFNESS, entities=ins, rules=v, qdegree=0),
                 O.Integral(O.CELL, O.K_NITSCHE, rules=i, point_data=nr, params=(40.0,)),
                 O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=gh, params=(0.1,), qdegree=0)]
            L = [O.Integral(O.CELL, O.L_SOURCE, entities=ins, rules=v, params=(O.F_POISSON_RHS, 1.0), qdegree=4),
                 O.Integral(O.CELL, O.L_NITSCHE_RHS, rules=i, point_data=nr, params=(40.0, O.F_SINPROD, 1.0))]
            return a, L

        def sub(rules, data, lo, hi):
            keep = (rules.parent_map >= lo) & (rules.parent_map < hi)
            idx = np.flatnonzero(keep)
            counts = np.diff(rules.offsets)[idx]
            pts = np.concatenate([np.arange(rules.offsets[k], rules.offsets[k + 1]) for k in idx]) if idx.size else \
                np.zeros(0, dtype=int)
            r = O.Rules(rules.tdim, rules.points[pts], rules.weights[pts],
                        np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), rules.parent_map[idx])
            return r, (None if data is None else data[pts])

        a_all, _ = forms(inside, vol, itf, nrm, ghost)
        c_lo, c_hi = part.owned_cells
        vol_o, _ = sub(vol, None, c_lo, c_hi)
        itf_o, nrm_o = sub(itf, nrm, c_lo, c_hi)
        a_own, L_own = forms(inside[(inside >= c_lo) & (inside < c_hi)], vol_o, itf_o, nrm_o,
                             ghost[(ghost[:, 0] >= c_lo) & (ghost[:, 0] < c_hi)])
        indptr, indices = O.create_sparsity(lm, V, a_all)
        values = torch.from_numpy(O.assemble_matrix(lm, V, a_own, indptr, indices))
        b = torch.from_numpy(O.assemble_vector(lm, V, L_own))
        scatter_reverse(values, lambda row: int(indptr[row]), part)
        scatter_reverse(b, None, part)

        r_lo, r_hi = part.owned_rows
        A = sp.csr_matrix((values.numpy(), indices, indptr), shape=(lm.nnodes,) * 2)
        rows = np.arange(r_lo, r_hi)
        mine = A[rows].tocoo()
        want = G[rows + part.vertex_offset].tocoo()
        got = sp.csr_matrix((mine.data, (mine.row, mine.col + part.vertex_offset)), shape=(rows.size, gm.nnodes))
        ref = sp.csr_matrix((want.data, (want.row, want.col)), shape=(rows.size, gm.nnodes))
        err = abs(got - ref).max() / abs(ref).max()
        errb = np.abs(b.numpy()[rows] - gref["b"][rows + part.vertex_offset]).max() / np.abs(gref["b"]).max()
        active = O.active_cells(a_all, lm.ncells)
        inactive = O.inactive_dofs(V, active)
        own_inactive = inactive[(inactive >= r_lo) & (inactive < r_hi)] + part.vertex_offset
        gi = gref["inactive"]
        want_inactive = gi[(gi >= r_lo + part.vertex_offset) & (gi < r_hi + part.vertex_offset)]
        q.put((rank, float(err), float(errb), bool(np.array_equal(own_inactive, want_inactive)), rows.size))
    finally:
        dist.destroy_process_group()


def _owner_worker(rank, world, port, n, q):
    """Owner-computes mode: level-set halo forward over gloo, every local entity assembled,
    no reverse reduction; the owned rows must equal the serial ones."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist

    from cutfemx_amd.dist import SlabPartition, halo_forward
    from helpers import level_set_values, oracle_poisson
    from oracle import pyoracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gm = O.mesh_box(3, n)
        gphi = level_set_values(gm.x, 3)
        gref = oracle_poisson(O, gm, gphi)
        G = sp.csr_matrix((gref["values"], gref["indices"], gref["indptr"]), shape=(gm.nnodes,) * 2)
        part = SlabPartition.create_owner(n, world, rank, weights=np.ones(n))
        nv, nc = part.plane_size * (part.nz_local + 1), part.cells_per_layer * part.nz_local
        lm = O.Mesh(3, gm.x[part.vertex_offset:part.vertex_offset + nv],
                    gm.conn[part.cell_offset:part.cell_offset + nc] - part.vertex_offset)
        # level set: owned planes known, halo planes poisoned, then received from the neighbours
        phi = torch.from_numpy(gphi[part.vertex_offset:part.vertex_offset + nv].copy())
        r_lo, r_hi = part.owned_rows
        halo = torch.ones(nv, dtype=torch.bool)
        halo[r_lo:r_hi] = False
        phi[halo] = float("nan")
        halo_forward(phi, part)
        assert np.array_equal(phi.numpy(), gphi[part.vertex_offset:part.vertex_offset + nv])
        ref = oracle_poisson(O, lm, phi.numpy())
        A = sp.csr_matrix((ref["values"], ref["indices"], ref["indptr"]), shape=(lm.nnodes,) * 2)
        rows = np.arange(r_lo, r_hi)
        mine = A[rows].tocoo()
        got = sp.csr_matrix((mine.data, (mine.row, mine.col + part.vertex_offset)), shape=(rows.size, gm.nnodes))
        want = G[rows + part.vertex_offset]
        same_pattern = (got != 0).astype(np.int8).nnz == (want != 0).astype(np.int8).nnz and \
            np.array_equal(np.diff(A.indptr)[rows], np.diff(G.indptr)[rows + part.vertex_offset])
        err = abs(got - want).max() / abs(want).max()
        errb = np.abs(ref["b"][rows] - gref["b"][rows + part.vertex_offset]).max() / np.abs(gref["b"]).max()
        ina = ref["inactive"]
        own_inactive = ina[(ina >= r_lo) & (ina < r_hi)] + part.vertex_offset
        gi = gref["inactive"]
        want_inactive = gi[(gi >= r_lo + part.vertex_offset) & (gi < r_hi + part.vertex_offset)]
        q.put((rank, float(err), float(errb), bool(np.array_equal(own_inactive, want_inactive)) and bool(same_pattern),
               rows.size))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 8), (3, 12)])
def test_owner_mode_matches_serial(oracle, world, n):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_owner_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total_rows = 0
    for rank, err, errb, ok, nrows in res:
        assert err < 1e-12, (rank, err)
        assert errb < 1e-12, (rank, errb)
        assert ok
        total_rows += nrows
    assert total_rows == (n + 1) ** 3


@pytest.mark.parametrize("n", [8])
def test_two_rank_slab_assembly_matches_serial(oracle, n):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total_rows = 0
    for rank, err, errb, inactive_ok, nrows in res:
        assert err < 1e-12, (rank, err)
        assert errb < 1e-12, (rank, errb)
        assert inactive_ok
        total_rows += nrows
    assert total_rows == (n + 1) ** 3  # the owned rows partition the dofs


def test_partition_arithmetic():
    from cutfemx_amd.dist import SlabPartition, balanced_boundaries, sphere_layer_weights
    b = balanced_boundaries(sphere_layer_weights(64), 8)
    assert b[0] == 0 and b[-1] == 64 and all(b[i] < b[i + 1] for i in range(8))
    planes = []
    for r in range(8):
        p = SlabPartition.create(64, 8, r)
        lo, hi = p.owned_rows
        planes += list(range(lo // p.plane_size + p.lz0, hi // p.plane_size + p.lz0))
        assert p.lz0 <= p.z0 < p.z1 <= p.lz1
        for peer, s, rcv in p.exchanges():
            q = SlabPartition.create(64, 8, peer)
            # the peer lists the mirrored exchange: it sends the plane I receive and receives the one I send
            assert (p.rank, rcv, s) in q.exchanges(), (r, peer, s, rcv, q.exchanges())
    assert planes == list(range(65))


def test_slab_boundaries_count_the_halo_layers():
    # the split that counts a rank's halo layers as its work: a partition of the layers, never heavier (halos included)
    # than the equal-weight split, for any world size and layer count -- also when there are barely more layers than ranks
    import numpy as np
    from cutfemx_amd.dist import balanced_boundaries, sphere_layer_weights
    for n in (8, 9, 64, 200):
        w = sphere_layer_weights(n)
        cum = np.concatenate([[0.0], np.cumsum(w)])
        for world in (1, 2, 3, 4, 8):
            for lo, hi in ((1, 2), (3, 3)):
                def heaviest(b):
                    return max(cum[min(b[p + 1] + hi, n)] - cum[max(b[p] - lo, 0)] for p in range(world))
                plain, aware = balanced_boundaries(w, world), balanced_boundaries(w, world, lo, hi)
                assert aware[0] == 0 and aware[-1] == n and len(aware) == world + 1
                assert all(aware[i] < aware[i + 1] for i in range(world)), (n, world, aware)
                assert heaviest(aware) <= heaviest(plain) * (1 + 1e-12), (n, world, plain, aware)
    # uniform weights: nothing to gain, the split stays a partition
    assert balanced_boundaries(np.ones(12), 4, 1, 1)[-1] == 12
