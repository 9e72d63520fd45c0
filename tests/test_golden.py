"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py
from the oracle): the oracle must reproduce them bit for bit on CPU; the HIP
engine must match them on the GPU."""
from pathlib import Path

import numpy as np
import pytest

from helpers import oracle_dg_poisson, oracle_poisson, rel_err

ALL = sorted((Path(__file__).parent / "golden").glob("*.npz"))
GOLD = [p for p in ALL if not p.stem.startswith("dg_")]
GOLD_DG = [p for p in ALL if p.stem.startswith("dg_")]


@pytest.mark.parametrize("path", GOLD, ids=lambda p: p.stem)
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    O = oracle
    m = O.Mesh(int(g["tdim"]), g["x"], g["conn"])
    ref = oracle_poisson(O, m, g["phi"], order=4)
    for k in ("domain", "inside", "ghost", "indptr", "indices", "active", "inactive"):
        assert np.array_equal(ref[k], g[k]), k
    for rk, gk in (("vol", "vol"), ("itf", "itf")):
        assert np.array_equal(ref[rk].offsets, g[gk + "_offsets"])
        assert np.array_equal(ref[rk].parent_map, g[gk + "_parent"])
        assert np.array_equal(ref[rk].points, g[gk + "_points"])
        assert np.array_equal(ref[rk].weights, g[gk + "_weights"])
    assert np.array_equal(ref["values"], g["values"]) and np.array_equal(ref["b"], g["b"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=lambda p: p.stem)
def test_engine_matches_golden(path):
    import cutfemx_amd as cfx
    from cutfemx_amd import poisson
    g = np.load(path)
    tdim = int(g["tdim"])
    mesh = cfx.Mesh.from_arrays(tdim, g["x"], g["conn"])
    V = cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(V, g["phi"]))
    assert np.array_equal(cd.domain(), g["domain"])
    assert np.array_equal(cfx.locate_entities(cd, "phi<0"), g["inside"])
    s = poisson.build_forms(V, cd, order=4)
    for rules, k in ((s.volume_rules, "vol"), (s.interface_rules, "itf")):
        assert np.array_equal(rules.offsets, g[k + "_offsets"])
        assert np.array_equal(rules.parent_map, g[k + "_parent"])
        assert np.allclose(rules.points, g[k + "_points"], rtol=0, atol=1e-14)
        assert rel_err(rules.weights, g[k + "_weights"]) < 1e-12
    assert np.array_equal(s.ghost_facets.rows, g["ghost"])
    A = cfx.fem.assemble_matrix(s.a)
    b = cfx.fem.assemble_vector(s.L)
    assert np.array_equal(A.indptr, g["indptr"]) and np.array_equal(A.indices, g["indices"])
    assert rel_err(A.data, g["values"]) < 1e-12 and rel_err(b, g["b"]) < 1e-12
    dom = cfx.fem.active_domain(s.a)
    assert np.array_equal(dom.active_cells, g["active"]) and np.array_equal(dom.inactive_dofs, g["inactive"])


@pytest.mark.parametrize("path", GOLD_DG, ids=lambda p: p.stem)
def test_oracle_reproduces_dg_golden(oracle, path):
    g = np.load(path)
    O = oracle
    m = O.Mesh(int(g["tdim"]), g["x"], g["conn"])
    s = oracle_dg_poisson(O, m, g["phi"], degree=int(g["degree"]))
    fr = s["facet_rules"]
    for got, k in ((s["skeleton"], "skeleton"), (s["fdom"], "facet_domain"), (s["omega_facets"], "omega_facets"),
                   (fr.points, "fr_points"), (fr.weights, "fr_weights"), (fr.offsets, "fr_offsets"),
                   (fr.parent_map, "fr_parent"), (fr.host_rows, "fr_rows")):
        assert np.array_equal(got, g[k]), k
    ip, ix = O.create_sparsity(m, s["V"], s["a"])
    assert np.array_equal(ip, g["indptr"]) and np.array_equal(ix, g["indices"])
    assert np.array_equal(O.assemble_matrix(m, s["V"], s["a"], ip, ix), g["values"])
    assert np.array_equal(O.assemble_vector(m, s["V"], s["L"]), g["b"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD_DG, ids=lambda p: p.stem)
def test_engine_matches_dg_golden(path):
    import cutfemx_amd as cfx
    from cutfemx_amd import poisson
    g = np.load(path)
    tdim = int(g["tdim"])
    mesh = cfx.Mesh.from_arrays(tdim, g["x"], g["conn"])
    f = cfx.Function(cfx.FunctionSpace(mesh, 1), g["phi"])
    s = poisson.build_dg_forms(f, int(g["degree"]))
    assert np.array_equal(s.skeleton.rows, g["skeleton"]) and np.array_equal(s.skeleton_cut.domain(), g["facet_domain"])
    assert np.array_equal(s.omega_facets, g["omega_facets"])
    fr = s.facet_rules
    assert np.array_equal(fr.offsets, g["fr_offsets"]) and np.array_equal(fr.parent_map, g["fr_parent"])
    assert np.array_equal(fr.host_rows, g["fr_rows"])
    assert np.allclose(fr.points, g["fr_points"], rtol=0, atol=1e-14) and rel_err(fr.weights, g["fr_weights"]) < 1e-12
    A = cfx.fem.assemble_matrix(s.a)
    b = cfx.fem.assemble_vector(s.L)
    assert np.array_equal(A.indptr, g["indptr"]) and np.array_equal(A.indices, g["indices"])
    assert rel_err(A.data, g["values"]) < 1e-12 and rel_err(b, g["b"]) < 1e-12
