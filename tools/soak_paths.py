#!/usr/bin/env python3
"""Soak run over the paths the bench does not touch, with a moving level set: DG Poisson on facet hosts (2-D), extension
penalty + cell aggregation, a vector-valued degree-1 elasticity system with lifting, a degree-2 scalar system -- every
handle created and dropped every step.  Prints the engine's HBM in use / cached / peak and the host RSS: flat = no leak.
usage: python tools/soak_paths.py [steps] [every]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cutfemx_amd as cfx
from cutfemx_amd import _lib, poisson
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
every = int(sys.argv[2]) if len(sys.argv) > 2 else 50
fem, ext = cfx.fem, cfx.extensions
dev = torch.device("cuda:0")
x2, c2 = cfx.box_mesh_arrays(2, 48)
m2 = cfx.Mesh.from_arrays(2, x2, c2)
x3, c3 = cfx.box_mesh_arrays(3, 14)
m3 = cfx.Mesh.from_arrays(3, x3, c3)
V2 = cfx.FunctionSpace(m2, 1)
V3 = cfx.FunctionSpace(m3, 1)
V3v = cfx.FunctionSpace(m3, 1, bs=3)
V3q = cfx.FunctionSpace(m3, 2)
xt2, xt3 = torch.tensor(x2[:, :2].copy(), device=dev), torch.tensor(x3[:, :3].copy(), device=dev)
t0 = time.perf_counter()
for k in range(steps):
    cx, R = 0.48 + 0.07 * math.sin(0.31 * k), 0.29 + 0.03 * math.sin(0.13 * k)
    phi2 = (torch.linalg.norm(xt2 - torch.tensor([cx, 0.47], device=dev), dim=1) - R).cpu().numpy()
    phi3 = (torch.linalg.norm(xt3 - torch.tensor([cx, 0.47, 0.52], device=dev), dim=1) - R).cpu().numpy()
    # DG Poisson on the cut skeleton (facet-hosted rules, dS terms)
    g = poisson.build_dg_forms(cfx.Function(V2, phi2), 1)
    A = fem.assemble_matrix(g.a); b = fem.assemble_vector(g.L)
    fem.deactivate_outside(A, b, fem.active_domain(g.a))
    # extension penalty on aggregated cells
    cd2 = cfx.cut(cfx.Function(V2, phi2))
    agg = ext.create_cell_aggregation(cd2, "phi<0", 0.6, allow_rootless=True)
    if agg.num_pairs > 0:
        E = ext.extension_penalty_matrix(V2, cd2, agg, 2.5, 2)
    # vector elasticity with ghost penalty and lifting, 3-D
    cd3 = cfx.cut(cfx.Function(V3, phi3))
    inside = cfx.locate_entities_device(cd3, "phi<0")
    vol = cfx.runtime_quadrature(cd3, "phi<0", 2)
    ghost = cfx.ghost_penalty_facets(cd3, "phi<0")
    ints = [fem.Integral(fem.ELASTICITY, cells=inside, rules=vol, params=(1.0e3, 0.3), qdegree=0)]
    if ghost.size > 0:
        ints.append(fem.Integral(fem.GHOST_GRADJUMP, facets=ghost, params=(5.0,), qdegree=0))
    a = fem.form(ints, V3v)
    Av = fem.assemble_matrix(a)
    nd = m3.num_nodes
    markers = (np.arange(3 * nd) % 7 == 0).astype(np.int8)
    bv = np.zeros(3 * nd)
    fem.apply_lifting(bv, a, markers, np.full(3 * nd, 0.5), alpha=1.0)
    fem.deactivate_outside(Av, None, fem.active_domain(a))
    # degree 2, scalar (hashed pattern rows, reuse cache)
    sysq = poisson.build_forms(V3q, cd3, order=4)
    Aq = fem.assemble_matrix(sysq.a); bq = fem.assemble_vector(sysq.L)
    nnz = (A.nnz, Av.nnz, Aq.nnz)
    del g, A, b, cd2, agg, cd3, inside, vol, ghost, a, Av, sysq, Aq, bq
    if k % every == every - 1:
        torch.cuda.synchronize()
        m = _lib.memory_stats()
        try:
            import psutil
            rss = psutil.Process().memory_info().rss / 2**20
        except ImportError:
            rss = float("nan")
        print(f"step {k + 1:5d}: host RSS {rss:8.1f} MiB in_use {m['in_use'] / 2**20:8.1f} MiB cached {m['cached'] / 2**20:8.1f} MiB "
              f"peak {m['peak'] / 2**20:8.1f} MiB nnz {nnz} {1e3 * (time.perf_counter() - t0) / (k + 1):.2f} ms/step", flush=True)
