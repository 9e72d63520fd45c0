"""Linear forms by cell block (cfx::VecBlocks) where the box meshes of the other tests do not reach:
cells in random order (a block's union of dofs is then far longer than the block: every segment-sum round and the long
dof -> partials lists run), forms of rule integrals alone, bitwise reproducibility, and the block path against the
per-cell-record path on the same form (reference loop: /root/reference/cpp/dolfinx_custom_data/fem/assemble_vector_impl.h,
cited in oracle/cfx_oracle.c `orc_assemble_vector`)."""
import os

import numpy as np
import pytest

from helpers import level_set_values, rel_err

pytestmark = pytest.mark.gpu
RTOL = 1e-12


class env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def shuffled_case(oracle, tdim, n, degree, seed=3):
    import cutfemx_amd as cfx
    O = oracle
    box = O.mesh_box(tdim, n)
    perm = np.random.default_rng(seed).permutation(box.conn.shape[0])
    om = O.Mesh(tdim, box.x, box.conn[perm])
    phi = level_set_values(om.x, tdim, "sphere")
    dofmap, ndofs = cfx.lagrange_dofmap(tdim, om.conn, om.nnodes, degree)
    oV = O.Space(dofmap, ndofs, degree, 1)
    mesh = cfx.Mesh.from_arrays(tdim, om.x, om.conn)
    V = cfx.FunctionSpace(mesh, degree, dofmap=None if degree == 1 else dofmap, ndofs=ndofs)
    Vphi = V if degree == 1 else cfx.FunctionSpace(mesh, 1)
    cd = cfx.cut(cfx.Function(Vphi, phi))
    dom = O.classify(om.conn, phi)
    inside = O.locate_entities(dom, "phi<0")
    ovol = O.runtime_quadrature(om, om.conn, phi, dom, "phi<0", 4)
    oitf = O.runtime_quadrature(om, om.conn, phi, dom, "phi=0", 4)
    onrm = O.evaluate_normals(om, om.conn, phi, oitf)
    vol = cfx.runtime_quadrature(cd, "phi<0", 4)
    itf = cfx.runtime_quadrature(cd, "phi=0", 4)
    nrm = cfx.normal(cd, itf)
    return dict(O=O, om=om, oV=oV, V=V, inside=inside, ovol=ovol, oitf=oitf, onrm=onrm, vol=vol, itf=itf, nrm=nrm)


def forms(c, field, with_source=True, with_nitsche=True):
    import cutfemx_amd as cfx
    O = c["O"]
    oL, gL = [], []
    if with_source:
        oL.append(O.Integral(O.CELL, O.L_SOURCE, entities=c["inside"], rules=c["ovol"], params=(field[0], 1.25), qdegree=4))
        gL.append(cfx.fem.Integral(cfx.fem.SOURCE, cells=c["inside"], rules=c["vol"], params=(field[1], 1.25), qdegree=4))
    if with_nitsche:
        oL.append(O.Integral(O.CELL, O.L_NITSCHE_RHS, rules=c["oitf"], point_data=c["onrm"], params=(10.0, O.F_SINPROD, 0.5)))
        gL.append(cfx.fem.Integral(cfx.fem.NITSCHE_RHS, rules=c["itf"], point_data=c["nrm"], params=(10.0, cfx.fem.F_SINPROD, 0.5)))
    return oL, gL


@pytest.mark.parametrize("tdim,n,degree", [(2, 14, 1), (3, 6, 1), (2, 10, 2), (3, 5, 2)])
def test_shuffled_cells_by_block_and_by_record(oracle, tdim, n, degree):
    """Random cell order: the block path (forced: CFX_VEC_BLOCKS=2), the per-cell-record path (=0) and the oracle agree;
    two block runs are bitwise equal."""
    import cutfemx_amd as cfx
    c = shuffled_case(oracle, tdim, n, degree)
    O = c["O"]
    for field in [(O.F_SINPROD, cfx.fem.F_SINPROD), (O.F_ONE, cfx.fem.F_ONE)]:   # series source term / generic integrand
        oL, gL = forms(c, field)
        want = O.assemble_vector(c["om"], c["oV"], oL)
        with env(CFX_VEC_BLOCKS="2"):
            b1 = cfx.fem.assemble_vector(cfx.fem.form(gL, c["V"], rank=1))
            b2 = cfx.fem.assemble_vector(cfx.fem.form(gL, c["V"], rank=1))
        with env(CFX_VEC_BLOCKS="0"):
            b0 = cfx.fem.assemble_vector(cfx.fem.form(gL, c["V"], rank=1))
        assert rel_err(b1, want) < RTOL and rel_err(b0, want) < RTOL
        if os.environ.get("CFX_ASSEMBLY") != "atomic":   # (the entity-parallel FP64-atomic kernels add in arrival order)
            assert np.array_equal(b1, b2)


@pytest.mark.parametrize("tdim,n,degree", [(3, 6, 1), (3, 5, 2)])
def test_rule_integrals_alone_and_uncut_alone(oracle, tdim, n, degree):
    """A form of runtime-rule integrals only (no uncut entity anywhere: the complex split of a Nitsche datum is one) and a
    form of uncut entities only take the block path too."""
    import cutfemx_amd as cfx
    c = shuffled_case(oracle, tdim, n, degree, seed=5)
    O = c["O"]
    field = (O.F_SINPROD, cfx.fem.F_SINPROD)
    with env(CFX_VEC_BLOCKS="2"):
        for kw in (dict(with_source=False), dict(with_nitsche=False)):
            oL, gL = forms(c, field, **kw)
            want = O.assemble_vector(c["om"], c["oV"], oL)
            b = cfx.fem.assemble_vector(cfx.fem.form(gL, c["V"], rank=1))
            assert rel_err(b, want) < RTOL
        # uncut entities without rules at all
        oL = [O.Integral(O.CELL, O.L_SOURCE, entities=c["inside"], params=(O.F_SINPROD, 2.0), qdegree=4)]
        gL = [cfx.fem.Integral(cfx.fem.SOURCE, cells=c["inside"], params=(cfx.fem.F_SINPROD, 2.0), qdegree=4)]
        want = O.assemble_vector(c["om"], c["oV"], oL)
        assert rel_err(cfx.fem.assemble_vector(cfx.fem.form(gL, c["V"], rank=1)), want) < RTOL


def test_accumulates_into_b(oracle):
    """assemble_vector adds to the caller's b (assemble_vector_impl.h: b is not zeroed)."""
    import cutfemx_amd as cfx
    c = shuffled_case(oracle, 3, 5, 2, seed=7)
    O = c["O"]
    oL, gL = forms(c, (O.F_SINPROD, cfx.fem.F_SINPROD))
    want = O.assemble_vector(c["om"], c["oV"], oL)
    L = cfx.fem.form(gL, c["V"], rank=1)
    b = np.full(want.shape, 0.5)
    got = cfx.fem.assemble_vector(L, b)
    assert rel_err(np.asarray(got), want + 0.5) < RTOL
