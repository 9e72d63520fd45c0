"""usage: python tools/host_time.py [n]: host-side enqueue time of one sync-free step against its wall time (is the
step bound by the host -- Python, ctypes, hipLaunchKernel -- or by the GPU?)"""
import sys, time
sys.path.insert(0, '.')
import torch
import cutfemx_amd as cfx
from cutfemx_amd import poisson, _lib
from bench import sphere_level_set, hot_path_step
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda', 0)
mesh = cfx.Mesh.create_box(3, n); V = cfx.FunctionSpace(mesh, 1)
phi = cfx.Function(V, sphere_level_set(torch, n, dev))
vals = torch.zeros(int(mesh.num_nodes) + 40 * int(0.2 * mesh.num_nodes + 100000), device=dev, dtype=torch.float64)
b = torch.zeros(mesh.num_nodes, device=dev, dtype=torch.float64)
body = lambda: hot_path_step(cfx, poisson, V, phi, vals, b, 4, None, False)
for _ in range(5):
    out = cfx.run_step(body, key='host-time')
enq, tot = [], []
for _ in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with cfx.step('host-time') as s:
        out = body()
        t1 = time.perf_counter()       # everything enqueued, nothing waited for
    t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
enq.sort(); tot.sort()
print(f"n={n}: host enqueue {1e3*enq[len(enq)//2]:.3f} ms, step (to the end of cfx_step_end) {1e3*tot[len(tot)//2]:.3f} ms")
