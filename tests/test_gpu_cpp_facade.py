"""The C++ facade (include/cutfemx_amd.hpp) compiled with the host compiler
against libcutfemx_amd.so, run on the GPU and compared with the oracle."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from helpers import level_set_values, oracle_poisson, rel_err

ROOT = Path(__file__).resolve().parent.parent


def _build(tmp_path):
    exe = tmp_path / "poisson_facade"
    cmd = ["g++", "-std=c++20", "-O1", "-I", str(ROOT / "include"), str(ROOT / "tests/cpp/poisson_facade.cpp"),
           "-o", str(exe), "-L", str(ROOT / "cutfemx_amd"), "-lcutfemx_amd",
           f"-Wl,-rpath,{ROOT / 'cutfemx_amd'}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return exe


def _read(path):
    out, buf, o = [], path.read_bytes(), 0
    dtypes = [np.int8, np.int32, np.int32, np.int32, np.float64, np.float64, np.int32, np.int64, np.int32,
              np.float64, np.float64, np.int32,
              np.int32, np.int32, np.int32, np.float64, np.float64, np.int32, np.float64, np.float64]
    for dt in dtypes:
        n = int(np.frombuffer(buf, dtype=np.int64, count=1, offset=o)[0]); o += 8
        out.append(np.frombuffer(buf, dtype=dt, count=n, offset=o).copy()); o += n * np.dtype(dt).itemsize
    return out


def test_facade_compiles_without_gpu(tmp_path):
    """Header + library link with plain g++ (no HIP headers needed on the caller's side)."""
    assert _build(tmp_path).exists()


@pytest.mark.gpu
@pytest.mark.parametrize("tdim,n", [(2, 16), (3, 8)])
def test_facade_matches_oracle(oracle, tmp_path, tdim, n):
    exe = _build(tmp_path)
    out = tmp_path / "facade.bin"
    subprocess.run([str(exe), str(tdim), str(n), str(out)], check=True)
    (dom, inside, voff, vpar, vw, iw, ghost, ip, ix, A, b, inactive,
     ext, cut_facets, fpar, fw, fphys, cpar, cpts, sw) = _read(out)
    O = oracle
    om = O.mesh_box(tdim, n)
    ref = oracle_poisson(O, om, level_set_values(om.x, tdim))
    vals, bb = ref["values"].copy(), ref["b"].copy()
    O.deactivate(ref["inactive"], ref["indptr"], ref["indices"], vals, bb)
    assert np.array_equal(dom, ref["domain"]) and np.array_equal(inside, ref["inside"])
    assert np.array_equal(voff, ref["vol"].offsets) and np.array_equal(vpar, ref["vol"].parent_map)
    assert rel_err(vw, ref["vol"].weights) < 1e-12 and rel_err(iw, ref["itf"].weights) < 1e-12
    assert np.array_equal(ghost.reshape(-1, 4), ref["ghost"])
    assert np.array_equal(ip, ref["indptr"]) and np.array_equal(ix, ref["indices"])
    assert rel_err(A, vals) < 1e-12 and rel_err(b, bb) < 1e-12
    assert np.array_equal(inactive, ref["inactive"])
    # facets as hosts: boundary facets against the plane x = 0.51
    plane = om.x[:, 0] - 0.51
    orows = O.exterior_facets(om)
    assert np.array_equal(ext.reshape(-1, 2), orows)
    H = O.facet_hosts(om, orows, om.conn)
    fdom = O.facet_classify(H, plane)
    assert np.array_equal(cut_facets, O.facet_locate_entities(H, fdom, "phi=0"))
    oR = O.facet_runtime_quadrature(om, H, plane, fdom, "phi<0", 2)
    assert np.array_equal(fpar, oR.parent_map) and rel_err(fw, oR.weights) < 1e-12
    assert np.max(np.abs(fphys.reshape(-1, tdim) - O.facet_physical_points(om, oR))) < 1e-13
    oC = O.facet_rules_to_cells(om, oR, 0)
    assert np.array_equal(cpar, oC.parent_map) and np.max(np.abs(cpts.reshape(-1, tdim) - oC.points)) < 1e-13
    oS = O.facet_runtime_quadrature(om, H, plane, fdom, "phi<0", 2, whole=True)
    assert rel_err(sw, oS.weights) < 1e-12
    # wet part of the boundary of the unit box: the face x = 0 plus 0.51 of the 2 (tdim - 1) faces along x
    assert abs(fw.sum() + sw.sum() - (1 + 2 * (tdim - 1) * 0.51)) < 1e-12
