"""GPU rehearsal of the multi-GPU path on one card: two processes, both on
cuda:0, gloo transport (RCCL cannot put two ranks on one device), the real HIP
engine per rank.  Owned rows must equal the serial oracle matrix / vector."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, n, q, mode):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CFX_DEVICE="0")
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist

    from cutfemx_amd.dist import DistributedPoisson, SlabPartition
    from helpers import level_set_values, oracle_poisson
    from oracle import pyoracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gm = O.mesh_box(3, n)
        gref = oracle_poisson(O, gm, level_set_values(gm.x, 3))
        G = sp.csr_matrix((gref["values"], gref["indices"], gref["indptr"]), shape=(gm.nnodes,) * 2)
        part = SlabPartition.create_owner(n, world, rank) if mode == "owner" else SlabPartition.create(n, world, rank)
        dp = DistributedPoisson(part, torch.device("cuda", 0), mode=mode)
        info = dp.step()
        info = dp.counters(dp.step())  # twice: buffers are reused between steps
        A = info["A"]
        M = sp.csr_matrix((A.data, A.indices, A.indptr), shape=(A.nrows, A.nrows))
        r_lo, r_hi = part.owned_rows
        rows = np.arange(r_lo, r_hi)
        inactive = info["dom"].inactive_dofs
        own_inactive = inactive[(inactive >= r_lo) & (inactive < r_hi)]
        # reference with the same deactivation applied
        vals, bref = gref["values"].copy(), gref["b"].copy()
        O.deactivate(gref["inactive"], gref["indptr"], gref["indices"], vals, bref)
        G = sp.csr_matrix((vals, gref["indices"], gref["indptr"]), shape=(gm.nnodes,) * 2)
        mine = M[rows].tocoo()
        got = sp.csr_matrix((mine.data, (mine.row, mine.col + part.vertex_offset)), shape=(rows.size, gm.nnodes))
        ref = G[rows + part.vertex_offset]
        err = abs(got - ref).max() / abs(ref).max()
        b = dp.b.cpu().numpy()
        errb = np.abs(b[rows] - bref[rows + part.vertex_offset]).max() / np.abs(bref).max()
        gi = gref["inactive"]
        want_inactive = gi[(gi >= r_lo + part.vertex_offset) & (gi < r_hi + part.vertex_offset)] - part.vertex_offset
        q.put((rank, float(err), float(errb), bool(np.array_equal(own_inactive, want_inactive)),
               int(info["active_dofs_owned"])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["owner", "reduce"])
def test_two_ranks_on_one_gpu_match_serial(oracle, mode):
    import torch.multiprocessing as mp
    n = 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000) + (7 if mode == "owner" else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from helpers import level_set_values, oracle_poisson
    gm = oracle.mesh_box(3, n)
    gref = oracle_poisson(oracle, gm, level_set_values(gm.x, 3))
    total_active = 0
    for rank, err, errb, inactive_ok, active in res:
        assert err < 1e-12, (rank, err)
        assert errb < 1e-12, (rank, errb)
        assert inactive_ok
        total_active += active
    assert total_active == gm.nnodes - gref["inactive"].size
