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
    if mode.endswith("+device"):
        os.environ["CFX_DIST_TRANSPORT"] = "device"
        mode = mode.split("+")[0]
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


@pytest.mark.parametrize("mode", ["owner", "reduce", "owner+device"])
def test_two_ranks_on_one_gpu_match_serial(oracle, mode):
    import torch.multiprocessing as mp
    n = 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000) + {"owner": 7, "reduce": 0, "owner+device": 13}[mode]
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


def _prim_worker(rank, world, port, q, transport="host"):
    """cfx_dist_* primitives between two ranks on one card (host-staged transport, or the device-callback transport with
    a callback that stages through the host itself): contiguous ranges and index lists, copy / add / or, and the matrix
    row exchange with its size check."""
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CFX_DEVICE="0")
    if transport == "device":
        os.environ["CFX_DIST_TRANSPORT"] = "device"
    import torch
    import torch.distributed as dist

    from cutfemx_amd.dist import DistComm
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        comm = DistComm()
        assert not comm.rccl and comm.world == 2
        assert comm.transport == ("host-callback-device" if transport == "device" else "host-staged")
        peer = 1 - rank
        ok = True
        # scatter_forward, contiguous: my tail [90, 100) <- the peer's head [0, 10)
        x = torch.arange(100, device=dev, dtype=torch.float64) + 1000.0 * rank
        comm.scatter_forward(x, [(peer, (0, 10), (90, 100))])
        ok &= bool(torch.equal(x[90:], torch.arange(10, device=dev, dtype=torch.float64) + 1000.0 * peer))
        ok &= bool(torch.equal(x[:90], torch.arange(90, device=dev, dtype=torch.float64) + 1000.0 * rank))
        # scatter_reverse_add through index lists (an index map's ghost -> owner lists)
        y = torch.ones(50, device=dev, dtype=torch.float64) * (rank + 1)
        send_idx = torch.tensor([49, 3, 17, 20], device=dev, dtype=torch.int32)
        recv_idx = torch.tensor([0, 5, 6, 30], device=dev, dtype=torch.int32)
        comm.scatter_reverse_add(y, [(peer, send_idx, recv_idx)])
        want = torch.ones(50, device=dev, dtype=torch.float64) * (rank + 1)
        want[recv_idx.long()] += float(peer + 1)
        ok &= bool(torch.equal(y, want))
        # indicator: reverse OR into the owner's entries, then forward back to the ghosts
        ind = torch.zeros(20, device=dev, dtype=torch.int8)
        ind[2 + rank] = 1                      # rank 0 marks entry 2, rank 1 marks entry 3
        comm.indicator_or(ind, [(peer, (0, 10), (0, 10))], forward_triples=[(peer, (0, 10), (10, 20))])
        ok &= bool(ind[2] == 1 and ind[3] == 1 and int(ind[:10].sum()) == 2)
        ok &= bool(torch.equal(ind[10:], ind[:10]))    # both ranks hold the same OR-ed values
        # unequal counts are refused by the library (CFX_ERR_INVALID_ARGUMENT for a bad peer)
        try:
            comm.scatter_forward(x, [(rank, (0, 1), (1, 2))])
            ok = False
        except ValueError:
            pass
        comm.close()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["host", "device"])
def test_dist_primitives_two_ranks_host_staged(transport):
    """(device: cfx_dist_comm_create_device -- the library hands DEVICE segments to the caller's transport; on the real
    multi-GPU job that callback posts them on torch.distributed's nccl group, here it copies through the host)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 400) + (500 if transport == "device" else 0)
    procs = [ctx.Process(target=_prim_worker, args=(r, 2, port, q, transport)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_rccl_communicator_single_rank():
    """The RCCL branch as far as one GPU allows: librccl.so.1 resolves at run time, ncclGetUniqueId /
    ncclCommInitRank / ncclCommDestroy work through the C ABI (world size 1: no peer to exchange with)."""
    import ctypes as C

    from cutfemx_amd import _lib
    l = _lib.lib()
    uid = (C.c_char * 128)()
    _lib.check(l.cfx_dist_unique_id(uid))
    assert any(bytes(uid))
    h = C.c_void_p()
    _lib.check(l.cfx_dist_comm_create(1, 0, uid, C.byref(h)))
    w, r, k = C.c_int(), C.c_int(), C.c_int()
    _lib.check(l.cfx_dist_comm_info(h, C.byref(w), C.byref(r), C.byref(k)))
    assert (w.value, r.value, k.value) == (1, 0, 1)
    # an exchange that names this rank itself as the peer is refused
    import torch
    x = torch.zeros(4, device="cuda", dtype=torch.float64)
    ex = (_lib.DistExchange * 1)()
    ex[0].peer, ex[0].send_count, ex[0].recv_count = 0, 1, 1
    assert l.cfx_dist_scatter_forward(h, C.c_void_p(x.data_ptr()), 1, ex) == _lib.ERR_INVALID_ARGUMENT
    _lib.check(l.cfx_dist_scatter_forward(h, C.c_void_p(x.data_ptr()), 0, None))
    _lib.check(l.cfx_dist_comm_destroy(h))
