import sys, cProfile, pstats, io, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.chdir('/root/repo')
import torch, bench
import cutfemx_amd as cfx
from cutfemx_amd import poisson
dev = torch.device('cuda', 0)
n = 64
mesh = cfx.Mesh.create_box(3, n); V = cfx.FunctionSpace(mesh, 1)
phi = cfx.Function(V, bench.sphere_level_set(torch, n, dev))
vals = torch.zeros(int(mesh.num_nodes) + 40 * int(0.2 * mesh.num_nodes + 100000), device=dev, dtype=torch.float64)
b = torch.zeros(mesh.num_nodes, device=dev, dtype=torch.float64)
step = lambda: bench.hot_path_step(cfx, poisson, V, phi, vals, b, 4)
for _ in range(5): step()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18); print(s.getvalue()[:3500])
