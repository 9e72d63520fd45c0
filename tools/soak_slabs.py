#!/usr/bin/env python3
"""Differential soak of the multi-GPU slab path with a MOVING interface: the N ranks of the z-slab partition are stepped
one after the other on one GPU (sync-free steps, halo values written analytically instead of exchanged) while a sphere
travels through the slabs, enters and leaves them; every step the rows a rank OWNS are compared with the same rows of
the whole-mesh system (plain sequence): pattern bit for bit, values / right-hand side to 1e-12.
usage: python tools/soak_slabs.py [n] [world] [steps] [seed]   (CFX_FUZZ_MARGIN=0.98: forced overflow)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch
import cutfemx_amd as cfx
from cutfemx_amd import poisson
from cutfemx_amd.dist import DistributedPoisson, SlabPartition
fem = cfx.fem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
world = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
if os.environ.get("CFX_FUZZ_MARGIN"):
    cfx.set_step_margin(float(os.environ["CFX_FUZZ_MARGIN"]), 0)
ax = torch.arange(n + 1, device=dev, dtype=torch.float64) / n


def sphere(z_lo, z_hi, c, R):
    az = torch.arange(z_lo, z_hi + 1, device=dev, dtype=torch.float64) / n
    d2 = (az[:, None, None] - c[2]) ** 2 + (ax[None, :, None] - c[1]) ** 2 + (ax[None, None, :] - c[0]) ** 2
    return (torch.sqrt(d2) - R).reshape(-1)


mesh = cfx.Mesh.create_box(3, n)
V = cfx.FunctionSpace(mesh, 1)
phi = torch.empty((n + 1) ** 3, device=dev, dtype=torch.float64)
f = cfx.Function(V, phi)
ranks = []
for r in range(world):
    part = SlabPartition.create_owner(n, world, r, weights=np.ones(n))
    dp = DistributedPoisson(part, dev, order=3, mode="owner")
    dp.part.world = 1            # no exchange: the halo planes are written below
    ranks.append(dp)
bad, c, R = 0, np.array([0.5, 0.5, 0.5]), 0.25
for k in range(steps):
    c = np.clip(c + rng.normal(0.0, 0.03, 3) + np.array([0.0, 0.0, 0.05 * math.sin(0.23 * k)]), 0.05, 0.95)
    R = float(np.clip(R + rng.normal(0.0, 0.02) + 0.05 * (0.22 - R), 0.08, 0.4))
    phi.copy_(sphere(0, n, c, R))
    s = poisson.build_forms(V, cfx.cut(f), order=3)
    A = fem.assemble_matrix(s.a); b = fem.assemble_vector(s.L)
    fem.deactivate_outside(A, b, fem.active_domain(s.a))
    G = sp.csr_matrix((A.data, A.indices, A.indptr), shape=(A.nrows, A.nrows))
    bg = np.asarray(b)
    for r, dp in enumerate(ranks):
        p = dp.part
        dp.phi_values.copy_(sphere(p.lz0, p.lz1, c, R))
        try:
            info = dp.step()
        except ValueError as e:
            if "no active background cells" in str(e):
                continue           # (the interface is not in this slab, nor any inside cell: refused as the reference does)
            raise
        Al = info["A"]
        M = sp.csr_matrix((Al.data, Al.indices, Al.indptr), shape=(Al.nrows, Al.nrows))
        r_lo, r_hi = p.owned_rows
        rows = np.arange(r_lo, r_hi)
        mine = M[rows].tocoo()
        got = sp.csr_matrix((mine.data, (mine.row, mine.col + p.vertex_offset)), shape=(rows.size, G.shape[1]))
        ref = G[rows + p.vertex_offset]
        got.sort_indices(); ref.sort_indices()
        ok = np.array_equal(got.indptr, ref.indptr) and np.array_equal(got.indices, ref.indices)
        if ok and ref.nnz:
            ok = float(np.abs(got.data - ref.data).max()) <= 1e-12 * float(np.abs(ref.data).max())
        bl = dp.b.cpu().numpy()
        if ok:
            ok = float(np.abs(bl[rows] - bg[rows + p.vertex_offset]).max()) <= 1e-12 * max(float(np.abs(bg).max()), 1e-300)
        if not ok:
            bad += 1
            print(f"step {k} rank {r}: owned rows differ from the whole-mesh system (R={R:.3f}, c={c})", flush=True)
        del info, Al
print(f"n={n} world={world}: {steps} steps, bad {bad}")
sys.exit(1 if bad else 0)
