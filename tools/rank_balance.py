"""Time every rank's slab of the N-rank partition one after the other on one GPU (no exchange):
checks the cost model behind SlabPartition (the slowest rank sets the multi-GPU step)."""
import sys, time
sys.path.insert(0, '.')
import torch
from cutfemx_amd.dist import SlabPartition, DistributedPoisson
n, world = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device('cuda', 0)
out = []
for r in range(world):
    part = SlabPartition.create_owner(n, world, r)
    dp = DistributedPoisson(part, dev, mode="owner")
    dp.part.world = 1            # skip the exchange: the analytic halo values are already in place
    for _ in range(2):
        info = dp.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        info = dp.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    info = dp.counters(info)
    out.append((r, part.z0, part.z1, round(1e3 * dt, 3), info['active_dofs_owned'], info['n_cut']))
    del dp, info
    from cutfemx_amd import _lib
    _lib.release_cache(); torch.cuda.empty_cache()
for o in out: print(o)
print('max ms', max(o[3] for o in out), 'sum ms', sum(o[3] for o in out))
