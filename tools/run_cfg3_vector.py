"""configs[3] (256^3 gyroid, P2): forms once, assemble_vector three times (the script of PMC passes over the vector kernels)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import cutfemx_amd as cfx
from cutfemx_amd import poisson, fem
from test_gpu_fullsize import level_set
dev = torch.device('cuda', 0)
n = 256
mesh = cfx.Mesh.create_box(3, n)
Vphi = cfx.FunctionSpace(mesh, 1)
cd = cfx.cut(cfx.Function(Vphi, level_set('gyroid', n, 0, n, dev)))
dm, nd = cfx.box_lagrange2_dofmap(mesh, n, dev)
V = cfx.FunctionSpace(mesh, 2, dofmap=dm, ndofs=nd)
s = poisson.build_forms(V, cd, order=4)
b = torch.zeros(nd, device=dev, dtype=torch.float64)
for _ in range(3):
    fem.assemble_vector(s.L, b)
torch.cuda.synchronize()
