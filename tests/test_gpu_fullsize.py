"""Parity at BASELINE.json's full sizes (configs 3, 4, 5).

The CPU oracle cannot run 805 M cells, so each full-size result is checked two ways:

* slab parity: the rows / rules / classification that belong to one vertex plane
  of the full problem are compared with the oracle run on a thin slab around that
  plane (same inputs, sliced from the device arrays).  With three layers of cells
  on either side every row of the plane is complete, so the comparison is the
  usual one: classification and CSR columns bit-exact, values to 1e-12.
* size-independent properties of the assembled objects: volume / area against the
  analytic sphere, sum(b) == volume for f = 1 (python/tests/test_cut_api.py:868-869),
  constants in the null space of the stiffness block, rigid translations in the
  null space of the elasticity block, x.(A y) == y.(A x).

Config 5 is an 8-GPU configuration; the single GPU of the test box runs one rank's
share (a 256 x 256 x 32 slab of the 256^3 mesh).
"""
import math

import numpy as np
import pytest

from helpers import oracle_poisson, rel_err

pytestmark = pytest.mark.gpu
RTOL = 1e-12
CENTRE, RADIUS = (0.47, 0.43, 0.41), 0.31
EDGES = [(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)]      # Basix tetrahedron edges


def _torch():
    import torch
    return torch


def _need_hbm(gib):
    import gc
    torch = _torch()
    from cutfemx_amd import _lib
    gc.collect()
    torch.cuda.empty_cache()
    _lib.release_cache()       # blocks cached by earlier tests
    free, _ = torch.cuda.mem_get_info()
    if free < gib * 2 ** 30:
        pytest.skip(f"needs {gib} GiB of free HBM, {free / 2 ** 30:.0f} available")


def level_set(kind, n, z0, nz, device):
    """P1 level-set dof values of the slab z0..z0+nz (vertex id ix + s (iy + s iz))."""
    torch = _torch()
    ax = torch.arange(n + 1, device=device, dtype=torch.float64) / n
    az = torch.arange(z0, z0 + nz + 1, device=device, dtype=torch.float64) / n
    Z, Y, X = az[:, None, None], ax[None, :, None], ax[None, None, :]
    if kind == "sphere":
        d2 = (Z - CENTRE[2]) ** 2 + (Y - CENTRE[1]) ** 2 + (X - CENTRE[0]) ** 2
        return (torch.sqrt(d2) - RADIUS).reshape(-1).contiguous()
    k = 2.0 * math.pi * 4.0
    g = torch.sin(k * X) * torch.cos(k * Y) + torch.sin(k * Y) * torch.cos(k * Z) + torch.sin(k * Z) * torch.cos(k * X)
    return (g + 0.0137).reshape(-1).contiguous()


def p2_dofmap_host(conn, n, nn):
    """numpy twin of cutfemx_amd.box_lagrange2_dofmap for the oracle's slab."""
    s = n + 1
    table = np.array([1, s, s * s, 1 + s, 1 + s * s, s + s * s, 1 + s + s * s], dtype=np.int64)
    out = np.empty((conn.shape[0], 10), dtype=np.int32)
    out[:, :4] = conn
    for k, (p, q) in enumerate(EDGES):
        a = np.minimum(conn[:, p], conn[:, q]).astype(np.int64)
        d = np.abs(conn[:, p].astype(np.int64) - conn[:, q])
        direction = np.argmax(d[:, None] == table[None, :], axis=1)
        out[:, 4 + k] = nn + 7 * a + direction
    return out, 8 * nn


class Numbering:
    """Dof numbering of a slab mesh: vertex dofs [0, nn), P2 edge dofs nn + 7 a + dir."""

    def __init__(self, n, z0, nz, degree):
        self.s2 = (n + 1) ** 2
        self.voff = self.s2 * z0
        self.nn = self.s2 * (nz + 1)
        self.degree = degree

    def to(self, other, ids):
        """ids of this numbering -> ids of `other` (same background mesh)."""
        ids = np.asarray(ids, dtype=np.int64)
        shift = self.voff - other.voff
        if self.degree == 1:
            return ids + shift
        return np.where(ids < self.nn, ids + shift, ids - self.nn + 7 * shift + other.nn)

    def plane_rows(self, k):
        """Scalar dofs attached to vertex plane k (global layer index): the vertices,
        and for P2 the edges whose lower vertex lies in the plane."""
        lo = self.s2 * k - self.voff
        v = np.arange(lo, lo + self.s2, dtype=np.int64)
        if self.degree == 1:
            return v
        return np.concatenate([v, np.arange(self.nn + 7 * lo, self.nn + 7 * (lo + self.s2), dtype=np.int64)])


def rows_of(A, rows):
    """CSR rows `rows` (ascending blocks) of a device matrix, downloading only those."""
    ips, ixs, vas = [], [], []
    breaks = np.flatnonzero(np.diff(rows) != 1) + 1
    for blk in np.split(rows, breaks):
        ip, ix, va = A.row_block(int(blk[0]), int(blk[-1]) + 1)
        ips.append(np.diff(ip)); ixs.append(ix); vas.append(va)
    return np.concatenate(ips), np.concatenate(ixs), np.concatenate(vas)


def compare_plane(A, b, num_gpu, o, num_orc, k0, bs=1):
    """Rows of vertex plane k0: oracle slab (numbering num_orc) vs device (num_gpu)."""
    rows_o = num_orc.plane_rows(k0)
    rows_g = num_orc.to(num_gpu, rows_o)
    if bs > 1:
        rows_o = (rows_o[:, None] * bs + np.arange(bs)).ravel()
        rows_g = (rows_g[:, None] * bs + np.arange(bs)).ravel()
    cnt_g, ix_g, va_g = rows_of(A, rows_g)
    ip, ix, va = o["indptr"], o["indices"], o["values"]
    cnt_o = (ip[rows_o + 1] - ip[rows_o])
    assert np.array_equal(cnt_g, cnt_o), "row lengths differ"
    sel = np.concatenate([np.arange(ip[r], ip[r + 1]) for r in rows_o]) if rows_o.size else np.zeros(0, np.int64)
    cols_o = ix[sel].astype(np.int64)
    cols_as_gpu = num_orc.to(num_gpu, cols_o // bs) * bs + cols_o % bs
    assert np.array_equal(ix_g.astype(np.int64), cols_as_gpu), "CSR columns differ"
    assert rel_err(va_g, va[sel]) < RTOL
    if b is not None:
        assert rel_err(b[rows_g], o["b"][rows_o]) < RTOL
    return int(cnt_g.sum())


def spmv(A, x, absolute=False, chunk=1 << 24):
    """y = A x (|A| x when `absolute`) with torch, one block of rows at a time."""
    torch = _torch()
    ip, ix, va = A.torch_views(x.device)
    y = torch.zeros_like(x)
    for lo in range(0, A.nrows, chunk):
        hi = min(A.nrows, lo + chunk)
        e0, e1 = int(ip[lo]), int(ip[hi])
        rows = torch.repeat_interleave(torch.arange(lo, hi, device=x.device), ip[lo + 1:hi + 1] - ip[lo:hi])
        v = va[e0:e1].abs() if absolute else va[e0:e1]
        y.index_add_(0, rows, v * x[ix[e0:e1].to(torch.int64)])
    return y


def oracle_slab(oracle, cfx, n, z0, nz, kind, device, degree=1):
    slab = cfx.Mesh.create_slab(n, z0, nz)
    om = oracle.Mesh(3, slab.x, slab.conn)
    phi = level_set(kind, n, z0, nz, device).cpu().numpy()
    return om, phi


# --------------------------------------------------------------------------- config 3
@pytest.fixture(scope="class")
def sphere512():
    torch = _torch()
    import cutfemx_amd as cfx
    from cutfemx_amd import poisson
    _need_hbm(110)
    n = 512
    dev = torch.device("cuda", 0)
    mesh = cfx.Mesh.create_box(3, n)
    V = cfx.FunctionSpace(mesh, 1)
    phi = level_set("sphere", n, 0, n, dev)
    cd = cfx.cut(cfx.Function(V, phi))
    system = poisson.build_forms(V, cd, order=4)
    A = cfx.fem.create_matrix(system.a)
    cfx.fem.assemble_matrix(system.a, A=A)
    b = torch.zeros(mesh.num_nodes, device=dev, dtype=torch.float64)
    cfx.fem.assemble_vector(system.L, b)
    yield dict(n=n, dev=dev, mesh=mesh, V=V, phi=phi, cd=cd, system=system, A=A, b=b, cfx=cfx)
    del A, system, cd


class TestConfig3:
    """512^3 P1 sphere system, built once for the three checks below and freed afterwards."""

    def test_cfg3_counts_volume_area(self, sphere512):
        s, torch = sphere512, _torch()
        from cutfemx_amd.dist import as_torch
        n, dev, sysm = s["n"], s["dev"], s["system"]
        dom = s["cd"].domain()
        n_in, n_cut, n_out = int((dom == -1).sum()), int((dom == 0).sum()), int((dom == 1).sum())
        assert n_in + n_cut + n_out == 6 * n ** 3
        vr, ir = sysm.volume_rules, sysm.interface_rules
        assert sysm.inside_cells.size == n_in
        # every cut cell hosts volume rules and interface rules (one per sub-facet), parents ascending
        for rules in (vr, ir):
            parents = rules.parent_map
            assert np.all(np.diff(parents) >= 0) and np.unique(parents).size == n_cut
            assert np.all(dom[parents] == 0)
        wv = as_torch(vr._view.weights, vr.total_points, "float64", dev)
        wi = as_torch(ir._view.weights, ir.total_points, "float64", dev)
        assert float(wv.min()) > 0.0 and float(wi.min()) > 0.0
        volume = float(wv.sum()) + n_in / (6.0 * n ** 3)
        area = float(wi.sum())
        # P1 interpolation of the sphere: O(h^2) geometric error, h = 1/512
        assert abs(volume - 4.0 / 3.0 * math.pi * RADIUS ** 3) < 1e-4 * volume
        assert abs(area - 4.0 * math.pi * RADIUS ** 2) < 1e-4 * area
        s["volume"] = volume


    def test_cfg3_facet_hosts(self, sphere512, oracle):
        """8f-4 at full size: the 3.1 M boundary facets against a plane (closed-form wet area), and the 4.3 M
        ghost-penalty facets as hosts of the sphere (complement property + oracle parity on a slab)."""
        import time
        s, torch = sphere512, _torch()
        cfx, n, dev, mesh, V = s["cfx"], s["n"], s["dev"], s["mesh"], s["V"]
        from cutfemx_amd import _lib
        from cutfemx_amd.dist import as_torch
        t0 = time.perf_counter()
        ext = cfx.exterior_facets(mesh)
        t1 = time.perf_counter()
        assert ext.size == 12 * n * n
        x = torch.arange(n + 1, device=dev, dtype=torch.float64) / n
        plane = (x[None, None, :] - 0.51).expand(n + 1, n + 1, n + 1).reshape(-1).contiguous()
        fcd = cfx.cut(cfx.Function(V, plane), ext, 2)
        run = cfx.runtime_quadrature(fcd, "phi<0", 2)
        std = cfx.full_facet_rules(fcd, "phi<0", 2)
        cells = run.to_cells()
        _lib.check(_lib.lib().cfx_synchronize())
        t2 = time.perf_counter()
        print(f"exterior_facets {1e3 * (t1 - t0):.1f} ms; facet cut + rules + cell view {1e3 * (t2 - t1):.1f} ms")
        assert run.num_rules == 4 * 2 * n                # the 4 faces along x, 2 triangles per boundary square
        wet = float(as_torch(run._view.weights, run.total_points, "float64", dev).sum()) + \
            float(as_torch(std._view.weights, std.total_points, "float64", dev).sum())
        assert abs(wet - (1.0 + 4 * 0.51)) < 1e-11
        assert np.all(np.diff(cells.parent_map) >= 0) and cells.tdim == 3
        # ghost-penalty facets of the sphere as hosts
        ghost = s["system"].ghost_facets
        gcd = cfx.cut(cfx.Function(V, s["phi"]), ghost, 2)
        gin, gout = cfx.runtime_quadrature(gcd, "phi<0", 3), cfx.runtime_quadrature(gcd, "phi>0", 3)
        gall = cfx.full_facet_rules(gcd, "phi=0", 3)
        tot = lambda r: float(as_torch(r._view.weights, r.total_points, "float64", dev).sum())
        assert abs(tot(gin) + tot(gout) - tot(gall)) < 1e-12 * tot(gall)
        ids_cut = cfx.locate_entities(gcd, "phi=0")
        assert set(gin.parent_map.tolist()) <= set(ids_cut.tolist())
        # oracle parity on the facets whose two cells lie in a 6-layer slab through the sphere
        z0, nz = 300, 6
        om, phi = oracle_slab(oracle, cfx, n, z0, nz, "sphere", dev)
        c0, c1 = 6 * n * n * z0, 6 * n * n * (z0 + nz)
        rows = ghost.rows
        inslab = (rows[:, 0] >= c0) & (rows[:, 0] < c1) & (rows[:, 2] >= c0) & (rows[:, 2] < c1)
        assert inslab.sum() > 1000
        local = rows[inslab].copy()
        local[:, 0] -= c0; local[:, 2] -= c0
        ids = np.flatnonzero(inslab).astype(np.int32)
        H = oracle.facet_hosts(om, local, om.conn, facet_ids=ids)
        odom = oracle.facet_classify(H, phi)
        assert np.array_equal(gcd.domain()[inslab], odom)
        oR = oracle.facet_runtime_quadrature(om, H, phi, odom, "phi<0", 3)
        sel = np.isin(gin.parent_map, ids)
        assert np.array_equal(gin.parent_map[sel], oR.parent_map)
        offs = gin.offsets
        w, pts = gin.weights, gin.points
        take = np.concatenate([np.arange(offs[r], offs[r + 1]) for r in np.flatnonzero(sel)])
        assert rel_err(w[take], oR.weights) < RTOL and np.abs(pts[take] - oR.points).max() < 1e-13

    @pytest.mark.parametrize("k0", [210, 366])
    def test_cfg3_plane_rows_match_oracle_slab(self, sphere512, oracle, k0):
        s = sphere512
        cfx, n, dev = s["cfx"], s["n"], s["dev"]
        z0, nz = k0 - 3, 6
        om, phi = oracle_slab(oracle, cfx, n, z0, nz, "sphere", dev)
        voff = (n + 1) ** 2 * z0
        assert np.array_equal(phi, s["phi"][voff:voff + phi.size].cpu().numpy())   # identical inputs
        o = oracle_poisson(oracle, om, phi)
        # classification of the slab's cells, bit-exact
        c0, c1 = 6 * n * n * z0, 6 * n * n * (z0 + nz)
        dom = s["cd"].domain()[c0:c1]
        assert np.array_equal(dom, o["domain"])
        # runtime rules of the slab's cut cells
        from cutfemx_amd import _lib
        for rules, want in ((s["system"].volume_rules, o["vol"]), (s["system"].interface_rules, o["itf"])):
            parents = rules.parent_map
            r0, r1 = np.searchsorted(parents, [c0, c1])
            assert np.array_equal(parents[r0:r1] - c0, want.parent_map)
            offs = rules.offsets
            q0, q1 = int(offs[r0]), int(offs[r1])
            assert np.array_equal(offs[r0:r1 + 1] - q0, want.offsets)
            w = _lib.download(rules._view.weights + 8 * q0, q1 - q0, np.float64)
            p = _lib.download(rules._view.points + 24 * q0, 3 * (q1 - q0), np.float64).reshape(-1, 3)
            assert rel_err(w, want.weights) < RTOL
            assert np.abs(p - want.points).max() < 1e-13
        full, slab = Numbering(n, 0, n, 1), Numbering(n, z0, nz, 1)
        nnz = compare_plane(s["A"], s["b"].cpu().numpy(), full, o, slab, k0)
        assert nnz > (n + 1) ** 2     # the plane crosses the sphere: more than the diagonal
        # inactive dofs of the plane
        act = cfx.fem.active_domain(s["system"].a)
        ina = act.inactive_dofs
        lo = (n + 1) ** 2 * k0
        mine = ina[(ina >= lo) & (ina < lo + (n + 1) ** 2)]
        want = o["inactive"]
        want = want[(want >= lo - voff) & (want < lo - voff + (n + 1) ** 2)] + voff
        assert np.array_equal(mine, want)


    def test_cfg3_sync_free_steps_of_the_bench_match_the_plain_sequence_and_the_oracle_slab(self, sphere512, oracle):
        """The 512^3 number of bench.py is a sync-free step (`cutfemx_amd.run_step(hot_path_step)`: grids sized by the
        previous step's counts, lengths read from HBM, the three-launch count / scan / write chains of meshes with more
        than 512 tiles per site, values stored into the caller's buffer).  Two such steps; the SECOND one -- the
        speculative kind the bench times -- must give the counts and the `indptr` of the plain sequence exactly, and
        the rows of two vertex planes must match the oracle slab as `test_cfg3_plane_rows_match_oracle_slab` asks."""
        import sys
        from pathlib import Path
        sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
        import bench
        from cutfemx_amd import poisson
        s, torch = sphere512, _torch()
        cfx, n, dev, mesh, V = s["cfx"], s["n"], s["dev"], s["mesh"], s["V"]
        values = torch.full((int(mesh.num_nodes) + 40 * int(0.2 * mesh.num_nodes + 100000),), 3.0e33, device=dev,
                            dtype=torch.float64)
        b = torch.zeros(mesh.num_nodes, device=dev, dtype=torch.float64)
        f = cfx.Function(V, s["phi"])
        key = "test-cfg3-bench-step"
        cfx.forget_step_history(key)
        infos = []
        for k in range(2):
            info = {}
            res = cfx.run_step(lambda: bench.hot_path_step(cfx, poisson, V, f, values, b, 4, None, False), key=key, info=info)
            infos.append(info)
            if k == 0:
                del res
        import os
        if os.environ.get("CFX_STEP_SPECULATE") != "0":
            assert infos[1]["published"] > 5 and infos[1]["passes"] == 1, infos
        c, sysm = res.counts(), s["system"]
        assert c["nnz"] == s["A"].nnz and c["n_inside"] == sysm.inside_cells.size
        assert c["n_cut"] == sysm.interface_rules.num_rules and c["n_vol_rules"] == sysm.volume_rules.num_rules
        assert c["nq_volume"] == sysm.volume_rules.total_points and c["nq_interface"] == sysm.interface_rules.total_points
        assert c["n_ghost"] == sysm.ghost_facets.size
        ip_a, _, _ = res.A.torch_views(dev)
        ip_b, _, _ = s["A"].torch_views(dev)
        assert bool(torch.equal(ip_a, ip_b))
        assert float(values[:c["nnz"]].abs().max()) < 1e30          # nothing stale behind the stored rows
        full = Numbering(n, 0, n, 1)
        bh = b.cpu().numpy()
        plain_dom = cfx.fem.active_domain(sysm.a)
        for k0 in (210, 366):
            z0, nz = k0 - 3, 6
            om, phi = oracle_slab(oracle, cfx, n, z0, nz, "sphere", dev)
            o = oracle_poisson(oracle, om, phi)
            oracle.deactivate(o["inactive"], o["indptr"], o["indices"], o["values"], o["b"])   # the step deactivates
            # (rows of the plane are complete in the slab; their inactive set is the slab's restricted to the plane)
            nnz = compare_plane(res.A, bh, full, o, Numbering(n, z0, nz, 1), k0)
            assert nnz > (n + 1) ** 2
            c0, c1 = 6 * n * n * z0, 6 * n * n * (z0 + nz)
            assert np.array_equal(res.system.cut_data.domain()[c0:c1], o["domain"])
        assert res.dom.num_active_dofs == plain_dom.num_active_dofs
        del res

    def test_cfg3_size_independent_properties(self, sphere512):
        s, torch = sphere512, _torch()
        cfx, dev, sysm, V = s["cfx"], s["dev"], s["system"], s["V"]
        fem = cfx.fem
        # (1) sum(b) == |Omega_h| for f = 1
        L1 = fem.form([fem.Integral(fem.SOURCE, cells=sysm.inside_cells, rules=sysm.volume_rules,
                                    params=(fem.F_ONE, 1.0), qdegree=1)], V)
        b1 = torch.zeros(V.ndofs, device=dev, dtype=torch.float64)
        fem.assemble_vector(L1, b1)
        vr = sysm.volume_rules
        from cutfemx_amd.dist import as_torch
        volume = float(as_torch(vr._view.weights, vr.total_points, "float64", dev).sum()) \
            + sysm.inside_cells.size / (6.0 * s["n"] ** 3)
        assert abs(float(b1.sum()) - volume) < 1e-12 * volume
        # (2) constants lie in the null space of the stiffness block (cut cells included)
        aK = fem.form([fem.Integral(fem.STIFFNESS, cells=sysm.inside_cells, rules=sysm.volume_rules, qdegree=0)], V)
        K = fem.assemble_matrix(aK)
        one = torch.ones(V.ndofs, device=dev, dtype=torch.float64)
        _, _, kv = K.torch_views(dev)
        assert float(spmv(K, one).abs().max()) < 1e-11 * float(kv.abs().max())
        del K, aK
        # (3) symmetry of the full operator through two matrix-vector products
        idx = torch.arange(V.ndofs, device=dev, dtype=torch.float64)
        x, y = torch.sin(0.37 * idx), torch.cos(0.11 * idx + 0.3)
        xAy, yAx = float(torch.dot(x, spmv(s["A"], y))), float(torch.dot(y, spmv(s["A"], x)))
        scale = float(torch.dot(x.abs(), spmv(s["A"], y.abs(), absolute=True)))
        assert abs(xAy - yAx) < 1e-12 * scale



# --------------------------------------------------------------------------- config 4
def test_cfg4_p2_gyroid_256(oracle):
    torch = _torch()
    import cutfemx_amd as cfx
    from cutfemx_amd import poisson
    from cutfemx_amd.dist import as_torch
    _need_hbm(120)
    n, dev, fem = 256, torch.device("cuda", 0), cfx.fem
    mesh = cfx.Mesh.create_box(3, n)
    Vphi = cfx.FunctionSpace(mesh, 1)
    phi = level_set("gyroid", n, 0, n, dev)
    cd = cfx.cut(cfx.Function(Vphi, phi))
    dm, ndofs = cfx.box_lagrange2_dofmap(mesh, n, dev)
    V = cfx.FunctionSpace(mesh, 2, dofmap=dm, ndofs=ndofs)
    sysm = poisson.build_forms(V, cd, order=4)
    A = fem.assemble_matrix(sysm.a)
    b = torch.zeros(ndofs, device=dev, dtype=torch.float64)
    fem.assemble_vector(sysm.L, b)
    # slab parity around a mid plane
    k0 = 117
    z0, nz = k0 - 3, 6
    om, phis = oracle_slab(oracle, cfx, n, z0, nz, "gyroid", dev)
    odm, ond = p2_dofmap_host(om.conn, n, om.nnodes)
    o = oracle_poisson(oracle, om, phis, degree=2, dofmap=odm, ndofs=ond)
    c0, c1 = 6 * n * n * z0, 6 * n * n * (z0 + nz)
    assert np.array_equal(cd.domain()[c0:c1], o["domain"])
    nnz = compare_plane(A, b.cpu().numpy(), Numbering(n, 0, n, 2), o, Numbering(n, z0, nz, 2), k0)
    assert nnz > 8 * (n + 1) ** 2
    # properties: mass matrix sums to the volume, sum(b_1) too, stiffness annihilates constants
    vr = sysm.volume_rules
    volume = float(as_torch(vr._view.weights, vr.total_points, "float64", dev).sum()) \
        + sysm.inside_cells.size / (6.0 * n ** 3)
    del A, sysm.a
    M = fem.assemble_matrix(fem.form([fem.Integral(fem.MASS, cells=sysm.inside_cells, rules=vr, qdegree=4)], V))
    assert abs(float(M.torch_views(dev)[2].sum()) - volume) < 1e-12 * volume
    del M
    L1 = fem.form([fem.Integral(fem.SOURCE, cells=sysm.inside_cells, rules=vr, params=(fem.F_ONE, 1.0), qdegree=2)], V)
    b1 = torch.zeros(ndofs, device=dev, dtype=torch.float64)
    fem.assemble_vector(L1, b1)
    assert abs(float(b1.sum()) - volume) < 1e-12 * volume
    K = fem.assemble_matrix(fem.form([fem.Integral(fem.STIFFNESS, cells=sysm.inside_cells, rules=vr, qdegree=2)], V))
    one = torch.ones(ndofs, device=dev, dtype=torch.float64)
    assert float(spmv(K, one).abs().max()) < 1e-11 * float(K.torch_views(dev)[2].abs().max())


# --------------------------------------------------------------------------- config 5
def test_cfg5_p2_vector_elasticity_rank_share(oracle):
    """configs[4]: a = sigma(u):eps(v) dx_solid + gamma (2 mu + lambda) h_avg [grad u.n].[grad v.n] dS_ghost on the
    P2 vector space (python/demo/demo_elasticity.py:214-235), strong Dirichlet data lifted through it (:67-93)."""
    torch = _torch()
    import cutfemx_amd as cfx
    from cutfemx_amd.dist import as_torch
    _need_hbm(60)
    n, dev, fem = 256, torch.device("cuda", 0), cfx.fem
    gz0, gnz = 89, 32                      # one of eight ranks' share of the 256^3 mesh
    E, nu = 1.0e3, 0.3
    mu, lmbda = E / (2.0 * (1.0 + nu)), E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu))
    gpar = 0.05 * (2.0 * mu + lmbda)
    mesh = cfx.Mesh.create_slab(n, gz0, gnz)
    Vphi = cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(Vphi, level_set("sphere", n, gz0, gnz, dev)))
    dm, nd = cfx.box_lagrange2_dofmap(mesh, n, dev)
    V = cfx.FunctionSpace(mesh, 2, dofmap=dm, ndofs=nd, bs=3)
    inside = cfx.locate_entities_device(cd, "phi<0")
    vol = cfx.runtime_quadrature(cd, "phi<0", 2)
    ghost = cfx.ghost_penalty_facets(cd, "phi<0")
    assert ghost.size > 10000
    ga = [fem.Integral(fem.ELASTICITY, cells=inside, rules=vol, params=(E, nu), qdegree=2),
          fem.Integral(fem.GHOST_GRADJUMP, facets=ghost, params=(gpar,), qdegree=2),
          fem.Integral(fem.MASS, cells=inside, rules=vol, qdegree=4)]
    a = fem.form(ga, V)
    A = fem.assemble_matrix(a)
    # strong Dirichlet data on a scattered set of dofs (all three components), lifted through a
    num_g = Numbering(n, gz0, gnz, 2)
    ids = torch.arange(nd, device=dev, dtype=torch.int64)
    marked = ((ids * 2654435761) % 11 == 0)
    markers = marked.to(torch.int8).repeat_interleave(3).contiguous()
    gval = torch.sin(0.013 * torch.arange(3 * nd, device=dev, dtype=torch.float64)) + 0.25
    b = torch.zeros(3 * nd, device=dev, dtype=torch.float64)
    fem.apply_lifting(b, a, markers, gval, alpha=0.8)
    # slab parity: the rows of plane k0 need the cells around it and the partners of their ghost facets
    k0 = 105
    z0, nz = k0 - 3, 6
    om, phis = oracle_slab(oracle, cfx, n, z0, nz, "sphere", dev)
    O = oracle
    dom = O.classify(om.conn, phis)
    c0, c1 = 6 * n * n * (z0 - gz0), 6 * n * n * (z0 + nz - gz0)
    assert np.array_equal(cd.domain()[c0:c1], dom)
    odm, ond = p2_dofmap_host(om.conn, n, om.nnodes)
    oV = O.Space(odm, ond, 2, 3)
    oin = O.locate_entities(dom, "phi<0")
    ovol = O.runtime_quadrature(om, om.conn, phis, dom, "phi<0", 2)
    oghost = O.ghost_penalty_facets(om, dom, "phi<0")
    rows = ghost.rows
    inslab = (rows[:, 0] >= c0) & (rows[:, 0] < c1) & (rows[:, 2] >= c0) & (rows[:, 2] < c1)
    # a facet is in the band by the classification of its two cells alone: the slab's facets are the share's
    got_rows = rows[inslab].copy()
    got_rows[:, 0] -= c0
    got_rows[:, 2] -= c0
    assert np.array_equal(got_rows, oghost)
    oa = [O.Integral(O.CELL, O.K_ELASTICITY, entities=oin, rules=ovol, params=(E, nu), qdegree=2),
          O.Integral(O.INTERIOR_FACET, O.K_GHOST_GRADJUMP, entities=oghost, params=(gpar,), qdegree=2),
          O.Integral(O.CELL, O.K_MASS, entities=oin, rules=ovol, qdegree=4)]
    ip, ix = O.create_sparsity(om, oV, oa)
    num_o = Numbering(n, z0, nz, 2)
    ids_o = np.arange(ond, dtype=np.int64)
    ids_g = num_o.to(num_g, ids_o)
    m_o = np.repeat(((ids_g * 2654435761) % 11 == 0).astype(np.int8), 3)
    g_o = np.sin(0.013 * (ids_g[:, None] * 3 + np.arange(3)).ravel().astype(np.float64)) + 0.25
    ob = O.apply_lifting(om, oV, oa, m_o, g_o, np.zeros(3 * ond), alpha=0.8)
    o = dict(indptr=ip, indices=ix, values=O.assemble_matrix(om, oV, oa, ip, ix), b=ob)
    nnz = compare_plane(A, b.cpu().numpy(), num_g, o, num_o, k0, bs=3)
    assert nnz > 3 * 8 * (n + 1) ** 2
    # rigid translations in the null space of the elasticity block and of the ghost-penalty block;
    # mass sums to 3 |Omega_h|
    del A, a
    K = fem.assemble_matrix(fem.form(ga[:2], V))
    kmax = float(K.torch_views(dev)[2].abs().max())
    for comp in range(3):
        t = torch.zeros(3 * nd, device=dev, dtype=torch.float64)
        t[comp::3] = 1.0
        assert float(spmv(K, t).abs().max()) < 1e-10 * kmax
    del K
    M = fem.assemble_matrix(fem.form(ga[2:], V))
    volume = float(as_torch(vol._view.weights, vol.total_points, "float64", dev).sum()) \
        + inside.size / (6.0 * n ** 3)
    assert abs(float(M.torch_views(dev)[2].sum()) - 3.0 * volume) < 1e-12 * volume
