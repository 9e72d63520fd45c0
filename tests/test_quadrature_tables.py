"""The generated reference-simplex rules are exact to their nominal degree
(independent check with exact rational moments), positive and interior."""
import math
import re
from fractions import Fraction
from itertools import product
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def load_tables(path):
    text = path.read_text()
    out = {}
    for tdim in (1, 2, 3):
        offs = [int(v) for v in re.search(rf"cfx_quad_offset_{tdim}d\[\d+\] = \{{([^}}]*)\}}", text).group(1).split(",")]
        pts = [float.fromhex(v.strip()) for v in
               re.search(rf"cfx_quad_points_{tdim}d\[\d+\] = \{{([^}}]*)\}}", text).group(1).split(",") if v.strip()]
        wts = [float.fromhex(v.strip()) for v in
               re.search(rf"cfx_quad_weights_{tdim}d\[\d+\] = \{{([^}}]*)\}}", text).group(1).split(",") if v.strip()]
        out[tdim] = (offs, np.array(pts).reshape(-1, tdim), np.array(wts))
    return out


@pytest.mark.parametrize("which", ["cutfemx_amd/csrc/cfx_quadrature_tables.h", "oracle/cfx_quadrature_tables.h"])
def test_rules_exact_positive_interior(which):
    tables = load_tables(ROOT / which)
    for tdim, (offs, pts, wts) in tables.items():
        for degree in range(len(offs) - 1):
            p, w = pts[offs[degree]:offs[degree + 1]], wts[offs[degree]:offs[degree + 1]]
            assert np.all(w > 0) and np.all(p > 0) and np.all(p.sum(axis=1) < 1)
            for e in product(range(degree + 1), repeat=tdim):
                if sum(e) > degree:
                    continue
                exact = Fraction(math.prod(math.factorial(k) for k in e), math.factorial(sum(e) + tdim))
                got = float(np.sum(w * np.prod(p ** np.array(e), axis=1)))
                assert abs(got - float(exact)) < 4e-16, (tdim, degree, e)


def test_both_copies_identical():
    a = (ROOT / "cutfemx_amd/csrc/cfx_quadrature_tables.h").read_text()
    b = (ROOT / "oracle/cfx_quadrature_tables.h").read_text()
    assert a == b
