#!/usr/bin/env python3
"""Differential soak: the same moving-domain step run (a) as a sync-free step (cutfemx_amd.run_step) and (b) as the plain
sequence with every size read back, on small meshes with a level set that wanders, breathes, leaves the mesh (no domain
at all) and swallows it (no cut cell) -- spaces P1, P2 and P1-vector, 2-D and 3-D.  Every step compares nnz, indptr,
indices bit for bit and values / b to 1e-12.  usage: python tools/soak_fuzz.py [steps] [seed] [big]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cutfemx_amd as cfx
from cutfemx_amd import _lib, poisson
fem = cfx.fem
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)


RTC = {}
MLS = {}


class SkipStep(Exception):
    pass


vary = {"ghost": True, "order": 4, "peek": False}   # CFX_FUZZ_VARY=1: the body of the step changes from step to step


def build(V, cd, kind):
    if kind == "poisson":
        s = poisson.build_forms(V, cd, order=vary["order"], ghost_penalty=vary["ghost"])
        if vary["peek"]:
            _ = s.inside_cells.size + s.volume_rules.num_rules     # sizes asked for in mid-step: resolved on demand
        return s.a, s.L
    inside = cfx.locate_entities_device(cd, "phi<0")
    vol = cfx.runtime_quadrature(cd, "phi<0", 2)
    ghost = cfx.ghost_penalty_facets(cd, "phi<0")
    ints = [fem.Integral(fem.ELASTICITY, cells=inside, rules=vol, params=(1.0e3, 0.3), qdegree=0)]
    if ghost.size > 0:
        ints.append(fem.Integral(fem.GHOST_GRADJUMP, facets=ghost, params=(5.0,), qdegree=0))
    return fem.form(ints, V), None


def one(V, f, kind, state):
    if kind == "dg":
        # the cut DG system on the facet hosts of the skeleton (python/demo/demo_dg_poisson.py): its own cuts every step
        g = poisson.build_dg_forms(f, 1)
        A = fem.assemble_matrix(g.a)
        b = fem.assemble_vector(g.L)
        return A, b, fem.deactivate_outside(A, b, fem.active_domain(g.a))
    if kind == "stokes":
        # cut Stokes blocks on (P2 vector, P1) with both ghost penalties, deactivated and merged into one matrix
        # (python/tests/test_assembly_stokes.py:98-142): rectangular forms, block helpers
        VU, VP = V
        if state.get("cd") is None:
            state["cd"] = cfx.cut(f)
        else:
            cfx.update(state["cd"])
        cd = state["cd"]
        inside = cfx.locate_entities_device(cd, "phi<0")
        vol = cfx.runtime_quadrature(cd, "phi<0", 4)
        ghost = cfx.ghost_penalty_facets(cd, "phi<0")
        i00 = [fem.Integral(fem.STIFFNESS, cells=inside, rules=vol, qdegree=2)]
        # (no branch on ghost.size that changes the FORM: inside a step a size is a capacity, and an integral over a list
        # that turns out empty adds nothing -- but replacing it by another integral would)
        i11 = [fem.Integral(fem.MASS, cells=inside, rules=vol, params=(0.0,), qdegree=2)]
        if ghost.size > 0:
            i00.append(fem.Integral(fem.GHOST_GRADJUMP, facets=ghost, params=(0.1,), qdegree=2))
            i11.append(fem.Integral(fem.GHOST_GRADJUMP, facets=ghost, params=(-0.05, 3.0), qdegree=2))
        a00 = fem.form(i00, VU)
        a01 = fem.form([fem.Integral(fem.DIV_TEST, cells=inside, rules=vol, params=(-1.0,), qdegree=3)], VU, trial_space=VP)
        a10 = fem.form([fem.Integral(fem.DIV_TRIAL, cells=inside, rules=vol, params=(-1.0,), qdegree=3)], VP, trial_space=VU)
        a11 = fem.form(i11, VP)
        blocks = [[fem.assemble_matrix(a00), fem.assemble_matrix(a01)], [fem.assemble_matrix(a10), fem.assemble_matrix(a11)]]
        dom_p = fem.active_domain(fem.form([fem.Integral(fem.MASS, cells=inside, rules=vol, qdegree=2)], VP))
        doms = [fem.active_domain(a00), dom_p]
        fem.deactivate_outside_blocks(blocks, doms)
        state["dbg"] = dict(block_nnz=[b.nnz for row in blocks for b in row], n_ghost=ghost.size, n_inside=inside.size,
                            n_rules=vol.num_rules)
        return fem.merge_blocks(blocks), None, doms[0]
    if kind == "rtc":
        # the Poisson system with its stiffness and Nitsche integrands REGISTERED from source (hipRTC): the generated-kernel
        # path of the reference (Form.h:59-75) inside steps
        if "ids" not in RTC:
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
            from test_user_integrands import NITSCHE_SRC, STIFFNESS_SRC
            RTC["ids"] = (fem.register_integrand("user_stiffness", STIFFNESS_SRC), fem.register_integrand("user_nitsche", NITSCHE_SRC))
        ks, kn = RTC["ids"]
        if state.get("cd") is None:
            state["cd"] = cfx.cut(f)
        else:
            cfx.update(state["cd"])
        s0 = poisson.build_forms(V, state["cd"], order=3)
        ints = [fem.Integral(ks, cells=s0.inside_cells, rules=s0.volume_rules, qdegree=0),
                fem.Integral(kn, rules=s0.interface_rules, point_data=s0.normals, params=(40.0,))]
        if s0.ghost_facets is not None and s0.ghost_facets.size > 0:
            ints.append(fem.Integral(fem.GHOST_GRADJUMP, facets=s0.ghost_facets, params=(0.1,), qdegree=0))
        a = fem.form(ints, V)
        A = fem.create_matrix(a)
        fem.assemble_matrix(a, A=A)
        b = fem.assemble_vector(s0.L)
        return A, b, fem.deactivate_outside(A, b, fem.active_domain(a))
    if kind == "mls":
        # two level sets: the material "phi<0 and phi1>0" (a sphere with a half space cut off), stiffness + mass over
        # [located cells, multi-level-set rules] (python/tests/test_multi_level_set_quadrature.py)
        f1 = state.setdefault("f1", cfx.Function(f.function_space if hasattr(f, "function_space") else V, (MLS["x0"] - 0.55)))
        cd = cfx.cut([f, f1])
        sel = "phi<0 and phi1>0"
        cells = cfx.locate_entities_device(cd, sel)
        rules = cfx.runtime_quadrature(cd, sel, 3)
        a = fem.form([fem.Integral(fem.STIFFNESS, cells=cells, rules=rules, qdegree=0),
                      fem.Integral(fem.MASS, cells=cells, rules=rules, qdegree=2)], V)
        A = fem.create_matrix(a)
        fem.assemble_matrix(a, A=A)
        return A, None, fem.deactivate_outside(A, None, fem.active_domain(a))
    if kind == "extension":
        cd = cfx.cut(f)
        agg = cfx.extensions.create_cell_aggregation(cd, "phi<0", 0.6, allow_rootless=True)
        if agg.num_pairs == 0:
            raise SkipStep()      # (nothing to penalise)
        A = cfx.extensions.extension_penalty_matrix(V, cd, agg, 2.5, 2)
        s = poisson.build_forms(V, cd, order=3)
        return A, None, fem.active_domain(s.a)
    if state.get("cd") is None:
        state["cd"] = cfx.cut(f)
    else:
        cfx.update(state["cd"])
    a, L = build(V, state["cd"], kind)
    A = fem.create_matrix(a)
    fem.assemble_matrix(a, A=A)
    b = fem.assemble_vector(L) if L is not None else None
    dom = fem.deactivate_outside(A, b, fem.active_domain(a))
    return A, b, dom


bad = 0
if os.environ.get("CFX_FUZZ_MARGIN"):      # e.g. 0.97: capacities BELOW the previous counts -- nearly every speculative pass is void
    cfx.set_step_margin(float(os.environ["CFX_FUZZ_MARGIN"]), 0)
big = len(sys.argv) > 3 and sys.argv[3] == "big"    # larger meshes: the row tiles, the culled classification and the bulk rows engage
huge = len(sys.argv) > 3 and sys.argv[3] == "huge"  # ... and the unfused count / scan / write triples (> 512 tiles per site)
for tdim, n, degree, bs, kind in [(3, 136, 1, 1, "poisson"), (3, 56, 2, 1, "poisson")] if huge else [(3, 48, 1, 1, "poisson"), (3, 40, 1, 1, "poisson+rough"), (3, 24, 2, 1, "poisson"), (3, 28, 1, 3, "elasticity"), (2, 200, 1, 1, "poisson"),
                                  (2, 96, 1, 1, "dg"), (3, 12, 1, 1, "dg"), (3, 10, 2, 3, "stokes"), (2, 60, 2, 2, "stokes"), (3, 24, 1, 1, "rtc"),
                                  (3, 24, 1, 1, "mls"), (2, 120, 1, 1, "extension")] if big else [(3, 10, 1, 1, "poisson"), (2, 20, 1, 1, "poisson"), (3, 6, 2, 1, "poisson"), (3, 8, 1, 3, "elasticity"),
                                  (2, 14, 2, 1, "poisson"), (3, 8, 1, 1, "poisson+scrambled"), (2, 12, 2, 1, "poisson+scrambled"),
                                  (3, 12, 1, 1, "poisson+rough"), (2, 30, 1, 1, "poisson+rough"), (3, 7, 2, 1, "poisson+rough"),
                                  (3, 9, 1, 3, "elasticity+rough"), (2, 18, 1, 1, "dg"), (3, 5, 1, 1, "dg"), (2, 22, 1, 1, "extension"),
                                  (2, 12, 2, 2, "stokes"), (3, 5, 2, 3, "stokes"), (3, 8, 1, 1, "rtc"), (2, 16, 1, 1, "rtc"), (3, 8, 1, 1, "mls"), (2, 20, 1, 1, "mls")]:
    if os.environ.get("CFX_FUZZ_ONLY") and os.environ["CFX_FUZZ_ONLY"] not in kind:
        continue
    rng = np.random.default_rng([seed, tdim, n, degree, bs, len(kind)])     # (every configuration its own stream: CFX_FUZZ_ONLY replays it)
    x, conn = cfx.box_mesh_arrays(tdim, n)
    rough = kind.endswith("+rough")          # a level set with islands, holes and necks whose phases wander (no sphere)
    if rough:
        kind = kind.split("+")[0]
    if kind.endswith("+scrambled"):
        # no locality left: vertices and cells renumbered at random, local vertex order of every cell permuted
        kind = kind.split("+")[0]
        pv = rng.permutation(x.shape[0])
        inv = np.empty_like(pv); inv[pv] = np.arange(pv.size)
        x = x[pv]
        conn = inv[conn][rng.permutation(conn.shape[0])]
        conn = np.ascontiguousarray(np.take_along_axis(conn, rng.permuted(np.tile(np.arange(tdim + 1), (conn.shape[0], 1)), axis=1), axis=1)).astype(np.int32)
    mesh = cfx.Mesh.from_arrays(tdim, x, conn)
    Vphi = cfx.FunctionSpace(mesh, 1)
    V = Vphi if (degree == 1 and bs == 1) else cfx.FunctionSpace(mesh, degree, bs=bs)
    if kind == "stokes":
        V = (cfx.FunctionSpace(mesh, 2, bs=tdim), Vphi)
    xt = torch.tensor(x[:, :tdim].copy(), device=dev)
    MLS["x0"] = x[:, 0].copy()
    phi = torch.empty(x.shape[0], device=dev, dtype=torch.float64)
    f = cfx.Function(Vphi, phi)
    key = f"fuzz-{tdim}-{n}-{degree}-{bs}-{kind}{'-rough' if rough else ''}"
    cfx.forget_step_history(key)
    sa, sb = {"cd": None}, {"cd": None}
    c, R, redo, empty, t0 = np.full(tdim, 0.5), 0.3, 0, 0, time.perf_counter()
    for k in range(steps):
        c = np.clip(c + rng.normal(0.0, 0.03, tdim) + 0.05 * (0.5 - c), 0.0, 1.0)
        R = float(np.clip(R + rng.normal(0.0, 0.03) + 0.05 * (0.3 - R), 0.05, 0.9))
        if k % 37 == 36: R = -0.05          # no domain at all
        if k % 53 == 52: R = 2.0            # the whole mesh inside: no cut cell
        if rough:
            if k == 0:
                kv = [torch.tensor(rng.uniform(2.0, 9.0, size=tdim), device=dev) for _ in range(3)]
                ph = [rng.uniform(0.0, 6.28, size=tdim) for _ in range(3)]
            val = torch.full((x.shape[0],), 0.3 - R, device=dev, dtype=torch.float64)
            for q in range(3):
                ph[q] = ph[q] + rng.normal(0.0, 0.08, tdim)
                val += 0.35 * torch.prod(torch.sin(kv[q] * xt + torch.tensor(ph[q], device=dev)), dim=1)
            if R in (-0.05, 2.0):
                val = val * 0.0 + (1.0 if R < 0 else -1.0)      # no domain / the whole mesh
            phi.copy_(val)
        else:
            phi.copy_(torch.linalg.norm(xt - torch.tensor(c, device=dev), dim=1) - R)
        if os.environ.get("CFX_FUZZ_VARY") == "1":
            vary.update(ghost=bool(rng.integers(0, 4) > 0), order=int(rng.choice([2, 3, 4])), peek=bool(rng.integers(0, 5) == 0))
        info = {}
        try:
            A1, b1, d1 = cfx.run_step(lambda: one(V, f, kind, sa), key=key, info=info)
            A2, b2, d2 = one(V, f, kind, sb)
        except SkipStep:
            continue
        except Exception as e:              # noqa: BLE001 -- reported, the run goes on with fresh handles
            sa, sb = {"cd": None}, {"cd": None}
            cfx.forget_step_history(key)
            if (float(phi.min()) > 0.0 or kind == "mls") and "no active background cells" in str(e):   # (mls: the conjunction may be empty)
                empty += 1                   # (the reference's own error for a domain without cells, deactivate.h:150-155)
                continue
            bad += 1
            print(f"{key} step {k} R={R:.3f}: {type(e).__name__}: {str(e)[:200]}", flush=True)
            sa, sb = {"cd": None}, {"cd": None}
            cfx.forget_step_history(key)
            continue
        redo += info.get("passes", 1) - 1
        ok = A1.nnz == A2.nnz and np.array_equal(A1.indptr, A2.indptr) and np.array_equal(A1.indices, A2.indices)
        if ok and A1.nnz > 0:
            sc = max(float(np.abs(A2.data).max()), 1e-300)
            ok = float(np.abs(A1.data - A2.data).max()) <= 1e-12 * sc
        if ok and b1 is not None:
            bb1, bb2 = np.asarray(b1.cpu() if hasattr(b1, "cpu") else b1), np.asarray(b2.cpu() if hasattr(b2, "cpu") else b2)
            ok = float(np.abs(bb1 - bb2).max()) <= 1e-12 * max(float(np.abs(bb2).max()), 1e-300)
        ok = ok and np.array_equal(d1.inactive_dofs, d2.inactive_dofs)
        if not ok:
            bad += 1
            print(f"{key} step {k} R={R:.3f}: in-step result differs from the plain sequence (nnz {A1.nnz} / {A2.nnz}) "
                  f"{sa.get('dbg')} / {sb.get('dbg')} passes {info.get('passes')}", flush=True)
        del A1, b1, d1, A2, b2, d2
    m = _lib.memory_stats()
    print(f"{key}: {steps} steps, {redo} repeated, {empty} with no domain (refused as the reference does), {bad} bad so far, {1e3 * (time.perf_counter() - t0) / steps:.2f} ms per double step, "
          f"engine in use {m['in_use'] / 2**20:.1f} MiB", flush=True)
print("bad:", bad)
sys.exit(1 if bad else 0)
