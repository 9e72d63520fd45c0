"""GPU parity: every stage of the hot path, HIP engine (through the C ABI) vs the
CPU oracle on the same inputs.  Integer/index results bit-exact; floating point
within 1e-12 relative (north_star tolerance)."""
import numpy as np
import pytest

from helpers import level_set_values, oracle_poisson, rel_err, scrambled_mesh

pytestmark = pytest.mark.gpu

RTOL = 1e-12

CASES = [(2, 16, "sphere"), (2, 64, "sphere"), (3, 8, "sphere"), (3, 20, "sphere"), (3, 12, "gyroid"),
         (2, 24, "sphere-scrambled"), (3, 10, "sphere-scrambled"), (3, 10, "gyroid-scrambled")]


@pytest.fixture(scope="module", params=CASES, ids=lambda c: f"{c[0]}d-n{c[1]}-{c[2]}")
def case(request, oracle):
    import cutfemx_amd as cfx
    tdim, n, kind = request.param
    O = oracle
    scrambled = kind.endswith("-scrambled")
    kind = kind.replace("-scrambled", "")
    om = scrambled_mesh(O, tdim, n) if scrambled else O.mesh_box(tdim, n)   # unstructured numbering + geometry
    phi = level_set_values(om.x, tdim, kind)
    ref = oracle_poisson(O, om, phi)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    f = cfx.Function(V, phi)
    cd = cfx.cut(f)
    from cutfemx_amd import poisson
    sysm = poisson.build_forms(V, cd)
    return dict(O=O, om=om, phi=phi, ref=ref, mesh=mesh, V=V, cd=cd, sys=sysm, tdim=tdim, scrambled=scrambled)


def test_box_mesh_generator(case):
    import cutfemx_amd as cfx
    if case["scrambled"]:
        pytest.skip("not the generator's mesh")
    tdim = case["tdim"]
    n = round((case["om"].nnodes) ** (1.0 / tdim)) - 1
    m = cfx.Mesh.create_box(tdim, n)
    assert np.array_equal(m.conn, case["om"].conn)
    assert np.array_equal(m.x, case["om"].x)
    x, conn = cfx.box_mesh_arrays(tdim, n)
    assert np.array_equal(conn, case["om"].conn) and np.array_equal(x, case["om"].x)


def test_classification_bit_exact(case):
    assert np.array_equal(case["cd"].domain(), case["ref"]["domain"])


@pytest.mark.parametrize("sel", ["phi<0", "phi>0", "phi=0", "phi<=0", "phi>=0", "phi<0 or phi=0"])
def test_locate_entities(case, sel):
    import cutfemx_amd as cfx
    got = cfx.locate_entities(case["cd"], sel)
    want = case["O"].locate_entities(case["ref"]["domain"], sel)
    assert got.dtype == np.int32 and np.array_equal(got, want)


@pytest.mark.parametrize("sel", ["phi<0", "phi>0", "phi=0"])
@pytest.mark.parametrize("order", [1, 2, 4])
def test_runtime_quadrature(case, sel, order):
    import cutfemx_amd as cfx
    O, om = case["O"], case["om"]
    want = O.runtime_quadrature(om, om.conn, case["phi"], case["ref"]["domain"], sel, order)
    got = cfx.runtime_quadrature(case["cd"], sel, order)
    assert got.kind == "per_entity" and got.tdim == case["tdim"]
    assert np.array_equal(got.offsets, want.offsets) and got.offsets.dtype == np.int32
    assert np.array_equal(got.parent_map, want.parent_map) and got.parent_map.dtype == np.int32
    assert got.offsets[0] == 0 and got.offsets[-1] == got.weights.size
    assert np.allclose(got.points, want.points, rtol=0, atol=1e-14)
    assert rel_err(got.weights, want.weights) < RTOL
    assert np.all(got.weights >= 0)


def test_normals_and_values(case):
    import cutfemx_amd as cfx
    O, om = case["O"], case["om"]
    rules = case["sys"].interface_rules
    n = cfx.normal(case["cd"], rules)
    assert rel_err(n, case["ref"]["normals"]) < RTOL
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-13)
    v = cfx.level_set_value(case["cd"], rules)
    assert np.max(np.abs(v)) < 1e-13  # interface points sit on phi_h = 0
    want = O.evaluate_values(om, om.conn, case["phi"], case["ref"]["itf"])
    assert np.allclose(v, want, atol=1e-14)
    pp = rules.physical_points
    assert np.allclose(pp.T, O.physical_points(om, case["ref"]["itf"]), atol=1e-14)


def test_ghost_penalty_facets(case):
    got = case["sys"].ghost_facets.rows
    want = case["ref"]["ghost"]
    assert np.array_equal(got, want)
    assert np.all(got[:, 0] < got[:, 2])


def test_sparsity_bit_exact(case):
    import cutfemx_amd as cfx
    A = cfx.fem.create_matrix(case["sys"].a)
    assert np.array_equal(A.indptr, case["ref"]["indptr"]) and A.indptr.dtype == np.int64
    assert np.array_equal(A.indices, case["ref"]["indices"]) and A.indices.dtype == np.int32


def test_assemble_matrix(case):
    import cutfemx_amd as cfx
    A = cfx.fem.assemble_matrix(case["sys"].a)
    want = case["ref"]["values"]
    assert rel_err(A.data, want) < RTOL
    M = A.to_scipy()
    assert abs(M - M.T).max() < 1e-11 * abs(M).max()


def test_assemble_vector(case):
    import cutfemx_amd as cfx
    b = cfx.fem.assemble_vector(case["sys"].L)
    assert rel_err(b, case["ref"]["b"]) < RTOL


def test_local_tensors(case):
    import cutfemx_amd as cfx
    O, om, ref = case["O"], case["om"], case["ref"]
    a = case["sys"].a
    for integral, (oi, use_rule, count) in enumerate([(0, False, len(ref["inside"])), (1, True, ref["itf"].parent_map.size)]):
        for idx in np.unique(np.linspace(0, max(count - 1, 0), 5).astype(int)) if count else []:
            got = cfx.fem.tabulate_entity(a, oi, int(idx), use_rule)
            want = O.tabulate_entity(om, ref["V"], ref["a"][oi], int(idx), use_rule)
            assert rel_err(got, want) < RTOL
    nvol = ref["vol"].parent_map.size
    for idx in np.unique(np.linspace(0, max(nvol - 1, 0), 5).astype(int)) if nvol else []:
        got = cfx.fem.tabulate_entity(a, 0, int(idx), True)
        want = O.tabulate_entity(om, ref["V"], ref["a"][0], int(idx), True)
        assert rel_err(got, want) < RTOL
    if len(ref["ghost"]):
        for idx in np.unique(np.linspace(0, len(ref["ghost"]) - 1, 5).astype(int)):
            got = cfx.fem.tabulate_entity(a, 2, int(idx), False)
            want = O.tabulate_entity(om, ref["V"], ref["a"][2], int(idx), False)
            assert rel_err(got, want) < RTOL


def test_active_domain_and_deactivation(case):
    import cutfemx_amd as cfx
    ref = case["ref"]
    A = cfx.fem.assemble_matrix(case["sys"].a)
    b = cfx.fem.assemble_vector(case["sys"].L)
    dom = cfx.fem.active_domain(case["sys"].a)
    assert np.array_equal(dom.active_cells, ref["active"])
    assert np.array_equal(dom.inactive_dofs, ref["inactive"])
    cfx.fem.deactivate_outside(A, b, dom)
    vals, bb = ref["values"].copy(), ref["b"].copy()
    case["O"].deactivate(ref["inactive"], ref["indptr"], ref["indices"], vals, bb)
    assert rel_err(A.data, vals) < RTOL and rel_err(b, bb) < RTOL
    M = A.to_scipy()
    inact = ref["inactive"]
    assert np.allclose(M.diagonal()[inact], 1.0) and np.all(b[inact] == 0.0)


@pytest.mark.parametrize("tdim,n", [(2, 37), (3, 13)])
def test_implicit_structured_classification(oracle, tdim, n, monkeypatch):
    """CFX_IMPLICIT_BOX=1: on a generated box mesh the classification computes the Kuhn connectivity from the
    cube index instead of streaming it ("implicit-structured"): same domain array, bit for bit."""
    import cutfemx_amd as cfx
    O = oracle
    om = O.mesh_box(tdim, n)
    mesh = cfx.Mesh.create_box(tdim, n)
    assert np.array_equal(mesh.conn, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    for kind in ("sphere", "gyroid"):
        phi = level_set_values(om.x, tdim, kind)
        phi[::7] = 0.0                                   # exact zeros: those cells are intersected
        want = O.classify(om.conn, phi)
        monkeypatch.setenv("CFX_IMPLICIT_BOX", "1")
        cd = cfx.cut(cfx.Function(V, phi))
        assert np.array_equal(cd.domain(), want)
        for sel in ("phi<0", "phi=0", "phi>0"):
            assert np.array_equal(cfx.locate_entities(cd, sel), O.locate_entities(want, sel))
        monkeypatch.delenv("CFX_IMPLICIT_BOX")
        assert np.array_equal(cfx.cut(cfx.Function(V, phi)).domain(), want)


def test_device_memset_any_size_and_alignment():
    """cfx_device_memset is the library's own fill kernel (16 B stores + a bytewise tail; unaligned pointers go
    to hipMemsetAsync): every size / offset combination must touch exactly its range."""
    import ctypes as C

    import cutfemx_amd as cfx  # noqa: F401
    from cutfemx_amd import _lib
    n = 1 << 16
    buf = _lib.DeviceBuffer(n, np.uint8)
    host = np.full(n, 0xAB, dtype=np.uint8)
    l = _lib.lib()
    rng = np.random.default_rng(5)
    cases = [(0, 0), (0, 1), (0, 15), (0, 16), (0, 17), (16, 4095), (32, 4096), (1, 100), (7, 33), (48, n - 48)]
    cases += [(int(rng.integers(0, 1000)), int(rng.integers(0, 30000))) for _ in range(20)]
    for off, size in cases:
        _lib.check(l.cfx_copy(C.c_void_p(buf.ptr), host.ctypes.data_as(C.c_void_p), C.c_size_t(n)))
        byte = int(rng.integers(0, 256))
        _lib.check(l.cfx_device_memset(C.c_void_p(buf.ptr + off), byte, C.c_size_t(size)))
        got = _lib.download(buf.ptr, n, np.uint8)
        want = host.copy()
        want[off:off + size] = byte
        assert np.array_equal(got, want), (off, size)


def test_overlapped_assembly_matches(case):
    """L next to a on two HIP streams (cfx_overlap_*): same matrix, same vector."""
    import torch

    import cutfemx_amd as cfx
    fem, sysm, ref = cfx.fem, case["sys"], case["ref"]
    b = torch.zeros(case["V"].ndofs, device="cuda", dtype=torch.float64)
    for _ in range(2):          # twice: the second pass runs with every table cached
        b.zero_()
        sysm.L.prepare()
        with fem.overlap() as lanes:
            lanes.side(lambda: fem.assemble_vector(sysm.L, b))
            A = fem.create_matrix(sysm.a)
            fem.assemble_matrix(sysm.a, A=A)
        assert np.array_equal(A.indptr, ref["indptr"]) and np.array_equal(A.indices, ref["indices"])
        assert rel_err(A.data, ref["values"]) < RTOL
        assert rel_err(b.cpu().numpy(), ref["b"]) < RTOL
    with pytest.raises(RuntimeError):
        with fem.overlap():
            with fem.overlap():
                pass
