"""Several level sets on the GPU: runtime_quadrature(cut([phi, phi1]), "phi<0 and phi1>0", k)
(cpp/cutfemx/cut/cut.h:122-181, docs/user-guide/element-classification.md:145-160) against the oracle, and the
rules used in a form (two-material mass / stiffness split by a second level set)."""
import numpy as np
import pytest

from helpers import level_set_values, rel_err, scrambled_mesh

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def second_level_set(x, tdim):
    # an oblique plane through the sphere / circle of helpers.level_set_values
    return 0.9 * (x[:, 0] - 0.52) + 0.4 * (x[:, 1] - 0.41) + (0.3 * (x[:, 2] - 0.38) if tdim == 3 else 0.0)


@pytest.fixture(scope="module", params=[(2, 24, False), (3, 10, False), (2, 16, True), (3, 8, True)],
                ids=lambda c: f"{c[0]}d-n{c[1]}{'-scrambled' if c[2] else ''}")
def case(request, oracle):
    import cutfemx_amd as cfx
    tdim, n, scr = request.param
    O = oracle
    om = scrambled_mesh(O, tdim, n) if scr else O.mesh_box(tdim, n)
    phis = [level_set_values(om.x, tdim), second_level_set(om.x, tdim)]
    dom = O.classify_multi(om.conn, phis)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut([cfx.Function(V, p) for p in phis])
    return dict(O=O, om=om, phis=phis, dom=dom, mesh=mesh, V=V, cd=cd, tdim=tdim)


SELECTORS = ["phi<0 and phi1<0", "phi<0 and phi1>0", "phi>0 and phi1<0", "phi1>0 and phi<0", "phi<=0 and phi1>=0",
             "phi=0 and phi1<0", "phi=0 and phi1>0", "phi1=0 and phi<0", "phi1<0", "phi1=0"]


@pytest.mark.parametrize("sel", SELECTORS)
@pytest.mark.parametrize("order", [1, 3])
def test_multi_level_set_rules_match_the_oracle(case, sel, order):
    import cutfemx_amd as cfx
    O, om = case["O"], case["om"]
    assert np.array_equal(case["cd"].domain(0), case["dom"][0]) and np.array_equal(case["cd"].domain(1), case["dom"][1])
    want = O.runtime_quadrature_multi(om, om.conn, case["phis"], case["dom"], sel, order)
    got = cfx.runtime_quadrature(case["cd"], sel, order)
    assert got.tdim == case["tdim"]
    assert np.array_equal(got.offsets, want.offsets) and got.offsets.dtype == np.int32
    assert np.array_equal(got.parent_map, want.parent_map) and got.parent_map.dtype == np.int32
    assert want.parent_map.size > 0
    assert np.abs(got.points - want.points).max() <= 2e-14
    assert rel_err(got.weights, want.weights) < RTOL
    assert np.all(got.weights >= 0)


def test_two_material_forms_with_multi_level_set_rules(case):
    """Mass + stiffness over {phi<0} split by the second level set into two materials: standard cells from
    locate_entities with the same selector, cut cells through the multi-level-set rules; the two material
    matrices add up to the one-level-set matrix, and each matches the oracle."""
    import cutfemx_amd as cfx
    O, om, cd, V = case["O"], case["om"], case["cd"], case["V"]
    oV = O.Space(om.conn, om.nnodes, 1)
    mats = []
    for sel, rho in [("phi<0 and phi1<0", 1.0), ("phi<0 and phi1>0", 1.0)]:
        cells = cfx.locate_entities(cd, sel)
        rules = cfx.runtime_quadrature(cd, sel, 2)
        ocells = O.locate_entities(case["dom"], sel)
        orules = O.runtime_quadrature_multi(om, om.conn, case["phis"], case["dom"], sel, 2)
        assert np.array_equal(cells, ocells)
        ga = [cfx.fem.Integral(cfx.fem.MASS, cells=cells, rules=rules, qdegree=2),
              cfx.fem.Integral(cfx.fem.STIFFNESS, cells=cells, rules=rules, qdegree=0)]
        oa = [O.Integral(O.CELL, O.K_MASS, entities=ocells, rules=orules, qdegree=2),
              O.Integral(O.CELL, O.K_STIFFNESS, entities=ocells, rules=orules, qdegree=0)]
        ip, ix = O.create_sparsity(om, oV, oa)
        want = O.assemble_matrix(om, oV, oa, ip, ix)
        A = cfx.fem.assemble_matrix(cfx.fem.form(ga, V))
        assert np.array_equal(A.indptr, ip) and np.array_equal(A.indices, ix)
        assert rel_err(A.data, want) < RTOL
        mats.append(A.to_scipy())
    # phi<0 alone, cut by the first level set only
    cells = cfx.locate_entities(cd, "phi<0")
    rules = cfx.runtime_quadrature(cd, "phi<0", 2)
    whole = cfx.fem.assemble_matrix(cfx.fem.form(
        [cfx.fem.Integral(cfx.fem.MASS, cells=cells, rules=rules, qdegree=2),
         cfx.fem.Integral(cfx.fem.STIFFNESS, cells=cells, rules=rules, qdegree=0)], V)).to_scipy()
    diff = abs(mats[0] + mats[1] - whole).max()
    assert diff < 1e-11 * abs(whole).max()


def test_multi_level_set_selector_errors(case):
    import cutfemx_amd as cfx
    cd = case["cd"]
    with pytest.raises(ValueError):
        cfx.runtime_quadrature(cd, "phi<0 or phi1<0", 2)       # not one conjunction
    with pytest.raises(ValueError):
        cfx.runtime_quadrature(cd, "phi=0 and phi1=0", 2)      # codimension 2
    with pytest.raises(ValueError):
        cfx.runtime_quadrature(cd, "phi<0 and phi2<0", 2)      # unknown level set


@pytest.mark.parametrize("tdim,n", [(2, 16), (3, 8)])
def test_runtime_quadratures_plural_equals_the_single_calls(oracle, tdim, n):
    """runtime_quadratures(cut_data, parts, order) (python/cutfemx/cut.py, cut.h:178-181): pairs of plain selectors
    share one pass over the cut cells; the rule sets are the ones the single calls return, bit for bit, whatever the
    mix of selectors."""
    import cutfemx_amd as cfx
    from helpers import level_set_values
    om = oracle.mesh_box(tdim, n)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(V, level_set_values(om.x, tdim)))
    parts = ["phi<0", "phi=0", "phi>0", "phi<=0", "phi=0"]
    got = cfx.runtime_quadratures(cd, parts[:4], 3)
    assert list(got) == parts[:4]
    for p in parts[:4]:
        one = cfx.runtime_quadrature(cd, p, 3)
        r = got[p]
        assert np.array_equal(r.offsets, one.offsets) and np.array_equal(r.parent_map, one.parent_map)
        assert np.array_equal(r.points, one.points) and np.array_equal(r.weights, one.weights)
    odd = cfx.runtime_quadratures(cd, ["phi>0"], 2)                     # a single selector: the single path
    assert np.array_equal(odd["phi>0"].weights, cfx.runtime_quadrature(cd, "phi>0", 2).weights)
    with pytest.raises(ValueError):
        cfx.runtime_quadratures(cd, ["phi<0", "psi<0"], 2)              # unknown level set in the second selector
    assert cfx.runtime_quadratures(cd, [], 2) == {}
