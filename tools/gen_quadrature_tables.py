#!/usr/bin/env python3
"""Generate the reference-simplex quadrature tables used by the engine.

The reference obtains its sub-simplex rules from CutCells / Basix (third party,
absent here).  Any rule that is exact to the requested
degree yields the same integrals up to round-off, so this script *derives*
fully symmetric, positive-weight, interior rules from their moment equations
instead of copying a table:

  1. pick an orbit structure (centroid / S21 / S111 for triangles,
     S4 / S31 / S22 / S211 for tetrahedra; degree 4 on tetrahedra: no positive rule with 11 points -- the size of the
     reference's Basix / Xiao-Gimbutas rule -- has the full symmetry (the only fully symmetric structure with 11
     points, S4 + S31 + S22, has a negative weight: Keast's rule), so that rule keeps the symmetry about ONE vertex
     only: orbits A1 / A3 of the permutations of barycentric coordinates 1..3, a two-parameter family of positive
     interior 11-point rules, pinned by fixing the two axis points),
  2. solve "all monomials of degree <= d are integrated exactly" with
     scipy.least_squares from many random starts (double precision),
  3. polish the solution with mpmath Newton at 40 digits,
  4. verify exactness on every monomial at 40 digits, positivity and
     interiority, then round to double.

Degrees without a compact symmetric solution fall back to the Stroud conical
product of Gauss-Jacobi rules (exact by construction, positive, interior).

Output: a C header with one flat table per (tdim, degree) used verbatim by the
HIP kernels (cutfemx_amd/csrc/cfx_quadrature_tables.h) and by the CPU oracle
(oracle/cfx_quadrature_tables.h).  Points are reference coordinates
(x[,y[,z]]), weights sum to the reference measure (1, 1/2, 1/6).

Run:  python tools/gen_quadrature_tables.py
"""
from __future__ import annotations

import itertools
import math
import sys
from pathlib import Path

import mpmath as mp
import numpy as np
from scipy.optimize import least_squares

mp.mp.dps = 40
MAX_DEGREE = 8  # table covers degree 0..MAX_DEGREE for every tdim
ROOT = Path(__file__).resolve().parent.parent


# ---------------------------------------------------------------------------
# orbits (barycentric), as functions of their free parameters
# ---------------------------------------------------------------------------
def orbit_points(tdim: int, kind: str, params):
    """Return list of barycentric tuples for one orbit."""
    if tdim == 2:
        if kind == "S3":
            t = mp.mpf(1) / 3 if _USE_MP else 1.0 / 3.0
            return [(t, t, t)]
        if kind == "S21":
            a = params[0]
            base = (a, a, 1 - 2 * a)
            return sorted(set(itertools.permutations(base)), key=_key)
        if kind == "S111":
            a, b = params
            base = (a, b, 1 - a - b)
            return sorted(set(itertools.permutations(base)), key=_key)
    if tdim == 3:
        if kind == "S4":
            t = 0.25
            return [(t, t, t, t)]
        if kind == "S31":
            a = params[0]
            base = (a, a, a, 1 - 3 * a)
            return sorted(set(itertools.permutations(base)), key=_key)
        if kind == "S22":
            a = params[0]
            base = (a, a, 0.5 - a, 0.5 - a)
            return sorted(set(itertools.permutations(base)), key=_key)
        if kind == "S211":
            a, b = params
            base = (a, a, b, 1 - 2 * a - b)
            return sorted(set(itertools.permutations(base)), key=_key)
        # symmetry about vertex 0 only (permutations of barycentric coordinates 1, 2, 3)
        if kind == "A1":      # on the axis through vertex 0 and the centroid of the opposite facet
            a = params[0]
            r = (1 - a) / 3
            return [(a, r, r, r)]
        if kind == "A3":      # (a; b, b, c): three points
            a, b = params
            c = 1 - a - 2 * b
            return [(a, b, b, c), (a, b, c, b), (a, c, b, b)]
    raise ValueError(kind)


_USE_MP = False


def _key(t):
    return tuple(float(v) for v in t)


NPARAM = {"S3": 0, "S21": 1, "S111": 2, "S4": 0, "S31": 1, "S22": 1, "S211": 2, "A1": 1, "A3": 2}
NPTS = {"S3": 1, "S21": 3, "S111": 6, "S4": 1, "S31": 4, "S22": 6, "S211": 12, "A1": 1, "A3": 3}


def unpack(tdim, orbits, z):
    """z = [w_0, params_0..., w_1, params_1 ...] -> (bary points, weights)."""
    pts, wts = [], []
    k = 0
    for kind in orbits:
        w = z[k]
        k += 1
        p = [z[k + i] for i in range(NPARAM[kind])]
        k += NPARAM[kind]
        for b in orbit_points(tdim, kind, p):
            pts.append(b)
            wts.append(w)
    return pts, wts


def monomials(tdim, degree):
    return [e for e in itertools.product(range(degree + 1), repeat=tdim)
            if sum(e) <= degree]


def exact_moment(e):
    """Integral of x^e over the reference simplex (vertices 0, e_i)."""
    num = 1
    for k in e:
        num *= math.factorial(k)
    return mp.mpf(num) / math.factorial(sum(e) + len(e))


def residual(tdim, orbits, degree, z, use_mp=False):
    global _USE_MP
    _USE_MP = use_mp
    pts, wts = unpack(tdim, orbits, z)
    out = []
    for e in monomials(tdim, degree):
        s = 0
        for b, w in zip(pts, wts):
            term = w
            # reference coords x_i = barycentric b[i+1]
            for i, k in enumerate(e):
                term = term * b[i + 1] ** k
            s = s + term
        ex = exact_moment(e)
        out.append(s - (ex if use_mp else float(ex)))
    return out


_BASE_LABELS = {"S3": "aaa", "S21": "aac", "S111": "abc",
                "S4": "aaaa", "S31": "aaac", "S22": "aacc", "S211": "aabc"}


def _label_perms(kind):
    return sorted(set(itertools.permutations(_BASE_LABELS[kind])))


def _base_values(kind, p):
    if kind == "S3":
        return {"a": 1.0 / 3.0}
    if kind == "S21":
        return {"a": p[0], "c": 1 - 2 * p[0]}
    if kind == "S111":
        return {"a": p[0], "b": p[1], "c": 1 - p[0] - p[1]}
    if kind == "S4":
        return {"a": 0.25}
    if kind == "S31":
        return {"a": p[0], "c": 1 - 3 * p[0]}
    if kind == "S22":
        return {"a": p[0], "c": 0.5 - p[0]}
    if kind == "S211":
        return {"a": p[0], "b": p[1], "c": 1 - 2 * p[0] - p[1]}
    raise ValueError(kind)


def fast_residual(tdim, orbits, degree, z, mons, exact):
    rows, wts = [], []
    k = 0
    for kind in orbits:
        w = z[k]
        k += 1
        p = z[k:k + NPARAM[kind]]
        k += NPARAM[kind]
        vals = _base_values(kind, p)
        for perm in _label_perms(kind):
            rows.append([vals[c] for c in perm])
            wts.append(w)
    B = np.array(rows)[:, 1:]
    W = np.array(wts)
    out = np.empty(len(mons))
    for m, e in enumerate(mons):
        t = W.copy()
        for i, kk in enumerate(e):
            if kk:
                t = t * B[:, i] ** kk
        out[m] = t.sum() - exact[m]
    return out


def solve_symmetric(tdim, orbits, degree, seed=0, tries=150):
    mons = monomials(tdim, degree)
    exact = np.array([float(exact_moment(e)) for e in mons])
    found = 0
    rng = np.random.default_rng(seed)
    nz = sum(1 + NPARAM[k] for k in orbits)
    vol = 1.0 / math.factorial(tdim)
    best = None
    for _ in range(tries):
        z0 = []
        for kind in orbits:
            z0.append(vol / sum(NPTS[k] for k in orbits) * rng.uniform(0.3, 2.0))
            if kind in ("S21",):
                z0.append(rng.uniform(0.02, 0.49))
            elif kind == "S111":
                a = rng.uniform(0.02, 0.6)
                z0 += [a, rng.uniform(0.02, (1 - a) * 0.9)]
            elif kind == "S31":
                z0.append(rng.uniform(0.02, 0.32))
            elif kind == "S22":
                z0.append(rng.uniform(0.02, 0.48))
            elif kind == "S211":
                a = rng.uniform(0.02, 0.45)
                z0 += [a, rng.uniform(0.02, (1 - 2 * a) * 0.9)]
        z0 = np.array(z0)
        try:
            sol = least_squares(
                lambda z: fast_residual(tdim, orbits, degree, z, mons, exact),
                z0, xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=200)
        except Exception:
            continue
        if np.max(np.abs(sol.fun)) > 1e-13:
            continue
        pts, wts = unpack(tdim, orbits, sol.x)
        if min(wts) <= 1e-6 * vol:
            continue
        if min(min(b) for b in pts) <= 1e-4:
            continue
        # prefer the most "balanced" rule (largest minimum weight)
        score = min(wts)
        if best is None or score > best[0]:
            best = (score, sol.x.copy())
        found += 1
        if found >= 4:
            break
    if best is None:
        return None
    return polish(tdim, orbits, degree, best[1])


def polish(tdim, orbits, degree, z):
    """Newton / Gauss-Newton in 40-digit arithmetic on the moment equations."""
    z = mp.matrix([mp.mpf(float(v)) for v in z])
    n = len(z)
    for _ in range(30):
        r = mp.matrix(residual(tdim, orbits, degree, list(z), use_mp=True))
        if max(abs(v) for v in r) < mp.mpf(10) ** (-36):
            break
        m = len(r)
        J = mp.zeros(m, n)
        h = mp.mpf(10) ** (-20)
        for j in range(n):
            zp = z.copy()
            zp[j] += h
            zm = z.copy()
            zm[j] -= h
            rp = mp.matrix(residual(tdim, orbits, degree, list(zp), use_mp=True))
            rm = mp.matrix(residual(tdim, orbits, degree, list(zm), use_mp=True))
            for i in range(m):
                J[i, j] = (rp[i] - rm[i]) / (2 * h)
        # least-squares step (moment equations are redundant by symmetry)
        dz = mp.lu_solve(J.T * J, -(J.T * r))
        z = z + dz
    r = residual(tdim, orbits, degree, list(z), use_mp=True)
    err = max(abs(v) for v in r)
    if err > mp.mpf(10) ** (-30):
        return None
    pts, wts = unpack(tdim, orbits, list(z))
    return ([[float(v) for v in b[1:]] for b in pts], [float(w) for w in wts])


# ---------------------------------------------------------------------------
# rules symmetric about one vertex (orbits A1 / A3): the axis parameters are fixed, the rest is solved
# ---------------------------------------------------------------------------
AXIS_RULES = {
    # (tdim, degree): (orbits, {index of z: fixed value}) -- z = [w, params...] per orbit as in unpack()
    (3, 4): (["A1", "A1", "A3", "A3", "A3"], {1: 0.06, 3: 0.75}),
}


def solve_axis(tdim, degree, seed=0, tries=400):
    orbits, fixed = AXIS_RULES[(tdim, degree)]
    nz = sum(1 + NPARAM[k] for k in orbits)
    free = [i for i in range(nz) if i not in fixed]

    def full(y, cast=float):
        z = [None] * nz
        for i, v in fixed.items():
            z[i] = cast(v)
        for i, v in zip(free, y):
            z[i] = v
        return z
    mons = monomials(tdim, degree)
    exact = np.array([float(exact_moment(e)) for e in mons])
    E = np.array(mons)

    def fres(y):
        pts, wts = unpack(tdim, orbits, full(list(y)))
        P = np.array([[float(v) for v in b[1:]] for b in pts])
        W = np.array([float(w) for w in wts])
        M = np.prod(P[:, None, :] ** E[None, :, :], axis=2)
        return (W @ M - exact) / np.maximum(exact, 1e-3)
    rng = np.random.default_rng(seed)
    vol = 1.0 / math.factorial(tdim)
    npts = sum(NPTS[k] for k in orbits)
    best = None
    found = 0
    for _ in range(tries):
        z0 = []
        for kind in orbits:
            z0.append(vol / npts * rng.uniform(0.3, 2.0))
            if kind == "A1":
                z0.append(rng.uniform(0.02, 0.9))
            else:
                a = rng.uniform(0.02, 0.8)
                z0 += [a, rng.uniform(0.02, (1 - a) / 2 * 0.95)]
        y0 = np.array([z0[i] for i in free])
        try:
            sol = least_squares(fres, y0, xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=300)
        except Exception:
            continue
        if np.max(np.abs(sol.fun)) > 1e-12:
            continue
        pts, wts = unpack(tdim, orbits, full(list(sol.x)))
        if min(wts) <= 1e-3 * vol / npts or min(min(b) for b in pts) <= 0.02:
            continue
        score = min(min(wts) * npts / vol, 10 * min(min(b) for b in pts))
        found += 1
        if best is None or score > best[0]:
            best = (score, sol.x.copy())
        if found >= 6:
            break
    if best is None:
        return None
    # Newton on the free unknowns at 40 digits (the moment equations are redundant: normal equations)
    y = mp.matrix([mp.mpf(float(v)) for v in best[1]])
    n = len(y)
    for _ in range(30):
        r = mp.matrix(residual(tdim, orbits, degree, full(list(y), mp.mpf), use_mp=True))
        if max(abs(v) for v in r) < mp.mpf(10) ** (-36):
            break
        m = len(r)
        J = mp.zeros(m, n)
        hstep = mp.mpf(10) ** (-20)
        for j in range(n):
            yp, ym = y.copy(), y.copy()
            yp[j] += hstep
            ym[j] -= hstep
            rp = mp.matrix(residual(tdim, orbits, degree, full(list(yp), mp.mpf), use_mp=True))
            rm = mp.matrix(residual(tdim, orbits, degree, full(list(ym), mp.mpf), use_mp=True))
            for i in range(m):
                J[i, j] = (rp[i] - rm[i]) / (2 * hstep)
        y = y + mp.lu_solve(J.T * J, -(J.T * r))
    r = residual(tdim, orbits, degree, full(list(y), mp.mpf), use_mp=True)
    if max(abs(v) for v in r) > mp.mpf(10) ** (-30):
        return None
    pts, wts = unpack(tdim, orbits, full(list(y), mp.mpf))
    return ([[float(v) for v in b[1:]] for b in pts], [float(w) for w in wts])


# ---------------------------------------------------------------------------
# Gauss-Jacobi conical product (fallback, any degree)
# ---------------------------------------------------------------------------
def gauss_jacobi_01(n, alpha):
    """n-point Gauss rule on [0,1] for weight (1-x)^alpha, 40-digit."""
    # Golub-Welsch on [-1,1] with weight (1-t)^alpha (beta = 0), mapped to [0,1]
    a, b = mp.mpf(alpha), mp.mpf(0)
    Jm = mp.zeros(n, n)
    for k in range(n):
        if k == 0:
            ak = (b - a) / (a + b + 2)
        else:
            ak = (b * b - a * a) / ((2 * k + a + b) * (2 * k + a + b + 2))
        Jm[k, k] = ak
        if k + 1 < n:
            kk = k + 1
            if kk == 1:
                bk = 4 * (1 + a) * (1 + b) / ((2 + a + b) ** 2 * (3 + a + b))
            else:
                bk = (4 * kk * (kk + a) * (kk + b) * (kk + a + b)
                      / ((2 * kk + a + b) ** 2 * (2 * kk + a + b + 1) * (2 * kk + a + b - 1)))
            Jm[k, k + 1] = Jm[k + 1, k] = mp.sqrt(bk)
    ev, evec = mp.eigsy(Jm)
    mu0 = mp.mpf(2) ** (a + b + 1) * mp.gamma(a + 1) * mp.gamma(b + 1) / mp.gamma(a + b + 2)
    xs, ws = [], []
    for i in range(n):
        xs.append((ev[i] + 1) / 2)
        ws.append(mu0 * evec[0, i] ** 2 / mp.mpf(2) ** (a + 1))
    order = sorted(range(n), key=lambda i: xs[i])
    return [xs[i] for i in order], [ws[i] for i in order]


def conical(tdim, degree):
    n = degree // 2 + 1
    if tdim == 1:
        x, w = gauss_jacobi_01(n, 0)
        return [[float(v)] for v in x], [float(v) for v in w]
    if tdim == 2:
        x0, w0 = gauss_jacobi_01(n, 1)
        x1, w1 = gauss_jacobi_01(n, 0)
        pts, wts = [], []
        for i in range(n):
            for j in range(n):
                pts.append([float(x0[i]), float(x1[j] * (1 - x0[i]))])
                wts.append(float(w0[i] * w1[j]))
        return pts, wts
    x0, w0 = gauss_jacobi_01(n, 2)
    x1, w1 = gauss_jacobi_01(n, 1)
    x2, w2 = gauss_jacobi_01(n, 0)
    pts, wts = [], []
    for i in range(n):
        for j in range(n):
            for k in range(n):
                pts.append([float(x0[i]),
                            float(x1[j] * (1 - x0[i])),
                            float(x2[k] * (1 - x0[i]) * (1 - x1[j]))])
                wts.append(float(w0[i] * w1[j] * w2[k]))
    return pts, wts


# ---------------------------------------------------------------------------
# rule selection
# ---------------------------------------------------------------------------
SYMMETRIC = {
    # (tdim, degree): orbit structure to try
    (2, 1): ["S3"],
    (2, 2): ["S21"],
    (2, 3): ["S21", "S21"],
    (2, 4): ["S21", "S21"],
    (2, 5): ["S3", "S21", "S21"],
    (2, 6): ["S21", "S21", "S111"],
    (2, 7): ["S21", "S21", "S21", "S111"],
    (2, 8): ["S3", "S21", "S21", "S21", "S111"],
    (3, 1): ["S4"],
    (3, 2): ["S31"],
    (3, 3): ["S31", "S31"],
    (3, 4): ["S4", "S31", "S22"],
    (3, 5): ["S31", "S31", "S22"],
    (3, 6): ["S31", "S31", "S31", "S211"],
}


def check_rule(tdim, degree, pts, wts):
    worst = mp.mpf(0)
    for e in monomials(tdim, degree):
        s = mp.mpf(0)
        for p, w in zip(pts, wts):
            t = mp.mpf(w)
            for i, k in enumerate(e):
                t *= mp.mpf(p[i]) ** k
            s += t
        worst = max(worst, abs(s - exact_moment(e)))
    return float(worst)


CACHE = Path(__file__).resolve().parent / "quadrature_cache.json"


def build():
    import json
    cache = json.loads(CACHE.read_text()) if CACHE.exists() else {}
    raw = {}
    for tdim in (1, 2, 3):
        for d in range(1, MAX_DEGREE + 1):
            key = f"{tdim},{d}"
            if key in cache:
                raw[(tdim, d)] = (cache[key]["points"], cache[key]["weights"])
                continue
            rule = None
            if (tdim, d) in AXIS_RULES:
                rule = solve_axis(tdim, d, seed=100 * tdim + d)
                if rule is None:
                    print(f"  tdim={tdim} degree={d}: axis-symmetric solve failed", file=sys.stderr)
            if rule is None and tdim > 1 and (tdim, d) in SYMMETRIC:
                rule = solve_symmetric(tdim, SYMMETRIC[(tdim, d)], d, seed=100 * tdim + d)
                if rule is None:
                    print(f"  tdim={tdim} degree={d}: symmetric solve failed -> conical",
                          file=sys.stderr)
            if rule is None:
                rule = conical(tdim, d)
            raw[(tdim, d)] = rule
            cache[key] = {"points": rule[0], "weights": rule[1]}
            CACHE.write_text(json.dumps(cache, indent=0))
            print(f"solved tdim={tdim} degree={d}: {len(rule[1])} points", file=sys.stderr,
                  flush=True)
    # degree d -> the rule with the fewest points among those exact to >= d
    rules = {}
    for tdim in (1, 2, 3):
        for degree in range(0, MAX_DEGREE + 1):
            d = max(degree, 1)
            best = min(range(d, MAX_DEGREE + 1), key=lambda k: (len(raw[(tdim, k)][1]), k))
            pts, wts = raw[(tdim, best)]
            err = check_rule(tdim, d, pts, wts)
            assert err < 2e-16, (tdim, d, err)
            assert min(wts) > 0
            assert all(min(p) > 0 and sum(p) < 1 for p in pts)
            rules[(tdim, degree)] = (pts, wts)
            print(f"tdim={tdim} degree={degree}: {len(wts)} points (rule of degree {best}), "
                  f"moment error {err:.2e}", file=sys.stderr, flush=True)
    return rules


def emit(rules) -> str:
    out = []
    out.append("// GENERATED by tools/gen_quadrature_tables.py -- do not edit.")
    out.append("// Reference-simplex quadrature rules: symmetric positive interior rules")
    out.append("// derived from their moment equations (40-digit Newton), Gauss-Jacobi")
    out.append("// conical products where no compact symmetric rule was found.")
    out.append("// Layout: points[npts*tdim] (reference coords), weights[npts] summing to")
    out.append("// the reference measure (1, 1/2, 1/6).  Degree d -> rule exact to >= d.")
    out.append("#ifndef CFX_QUADRATURE_TABLES_H")
    out.append("#define CFX_QUADRATURE_TABLES_H")
    out.append(f"#define CFX_QUAD_MAX_DEGREE {MAX_DEGREE}")
    maxn = {t: max(len(rules[(t, d)][1]) for d in range(MAX_DEGREE + 1)) for t in (1, 2, 3)}
    for t in (1, 2, 3):
        out.append(f"#define CFX_QUAD_MAX_POINTS_{t}D {maxn[t]}")
    out.append("#ifndef CFX_QUAD_TABLE_QUALIFIER")
    out.append("#define CFX_QUAD_TABLE_QUALIFIER static const")
    out.append("#endif")
    for tdim in (1, 2, 3):
        offs = [0]
        allp, allw = [], []
        for d in range(MAX_DEGREE + 1):
            pts, wts = rules[(tdim, d)]
            offs.append(offs[-1] + len(wts))
            allp += [v for p in pts for v in p]
            allw += wts
        out.append(f"CFX_QUAD_TABLE_QUALIFIER int cfx_quad_offset_{tdim}d[{MAX_DEGREE + 2}] = {{"
                   + ", ".join(str(o) for o in offs) + "};")
        out.append(f"CFX_QUAD_TABLE_QUALIFIER double cfx_quad_points_{tdim}d[{len(allp)}] = {{")
        for i in range(0, len(allp), tdim):
            out.append("  " + ", ".join(float(v).hex() for v in allp[i:i + tdim]) + ",")
        out.append("};")
        out.append(f"CFX_QUAD_TABLE_QUALIFIER double cfx_quad_weights_{tdim}d[{len(allw)}] = {{")
        for w in allw:
            out.append(f"  {float(w).hex()},")
        out.append("};")
    out.append("#endif")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    text = emit(build())
    for rel in ("cutfemx_amd/csrc/cfx_quadrature_tables.h", "oracle/cfx_quadrature_tables.h"):
        p = ROOT / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
        print("wrote", p, file=sys.stderr)
