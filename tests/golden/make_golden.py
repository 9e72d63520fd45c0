#!/usr/bin/env python3
"""Generate the committed golden vectors from the CPU oracle.

The reference cannot be built or imported in this image (its dependencies are
absent) and holds no per-point fixtures of its own, so these
vectors come from the oracle restatement, which tests/test_oracle_kat.py pins
against the reference's integral-level invariants.  Run from the repo root:

    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from helpers import level_set_values, oracle_dg_poisson, oracle_poisson  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

CASES = {"circle_2d_n8": (2, 8, "sphere"), "sphere_3d_n4": (3, 4, "sphere"), "gyroid_3d_n6": (3, 6, "gyroid")}

for name, (tdim, n, kind) in CASES.items():
    m = O.mesh_box(tdim, n)
    phi = level_set_values(m.x, tdim, kind)
    ref = oracle_poisson(O, m, phi, order=4)
    np.savez_compressed(
        Path(__file__).parent / f"{name}.npz",
        tdim=tdim, n=n, x=m.x, conn=m.conn, phi=phi, domain=ref["domain"], inside=ref["inside"],
        vol_points=ref["vol"].points, vol_weights=ref["vol"].weights, vol_offsets=ref["vol"].offsets,
        vol_parent=ref["vol"].parent_map, itf_points=ref["itf"].points, itf_weights=ref["itf"].weights,
        itf_offsets=ref["itf"].offsets, itf_parent=ref["itf"].parent_map, normals=ref["normals"],
        ghost=ref["ghost"], indptr=ref["indptr"], indices=ref["indices"], values=ref["values"], b=ref["b"],
        active=ref["active"], inactive=ref["inactive"])
    print(name, m.ncells, "cells", int((ref["domain"] == 0).sum()), "cut", ref["indices"].size, "nnz")

# facet hosts + DG skeleton terms: python/demo/demo_dg_poisson.py on small meshes
DG_CASES = {"dg_circle_2d_n8": (2, 8, "sphere", 1), "dg_sphere_3d_n4": (3, 4, "sphere", 1)}
for name, (tdim, n, kind, degree) in DG_CASES.items():
    m = O.mesh_box(tdim, n)
    phi = level_set_values(m.x, tdim, kind)
    s = oracle_dg_poisson(O, m, phi, degree=degree)
    ip, ix = O.create_sparsity(m, s["V"], s["a"])
    values = O.assemble_matrix(m, s["V"], s["a"], ip, ix)
    b = O.assemble_vector(m, s["V"], s["L"])
    fr = s["facet_rules"]
    np.savez_compressed(
        Path(__file__).parent / f"{name}.npz",
        tdim=tdim, n=n, degree=degree, x=m.x, conn=m.conn, phi=phi, skeleton=s["skeleton"], facet_domain=s["fdom"],
        omega_facets=s["omega_facets"], fr_points=fr.points, fr_weights=fr.weights, fr_offsets=fr.offsets,
        fr_parent=fr.parent_map, fr_rows=fr.host_rows, indptr=ip, indices=ix, values=values, b=b)
    print(name, s["skeleton"].shape[0], "skeleton facets", fr.parent_map.size, "cut", ix.size, "nnz")
