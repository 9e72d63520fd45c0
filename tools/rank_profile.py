"""Kernel-time sum vs wall time of one rank's slab step (fixed overheads of the multi-GPU step)."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import torch
from cutfemx_amd.dist import SlabPartition, DistributedPoisson
from cutfemx_amd import _lib
n, world, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device('cuda', 0)
part = SlabPartition.create_owner(n, world, r)
dp = DistributedPoisson(part, dev, mode="owner")
dp.part.world = 1
for _ in range(3): dp.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): dp.step()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10
l = _lib.lib(); _lib.check(l.cfx_profile_enable(1)); _lib.check(l.cfx_profile_reset())
for _ in range(5): dp.step()
tot, nl, rows = 0.0, 0, []
for i in range(l.cfx_profile_count()):
    name, ms, cnt = C.c_char_p(), C.c_double(), C.c_int64()
    _lib.check(l.cfx_profile_get(i, C.byref(name), C.byref(ms), C.byref(cnt)))
    if cnt.value: tot += ms.value / 5; nl += cnt.value / 5; rows.append((ms.value / 5, name.value.decode(), cnt.value / 5))
print('rank', r, 'wall ms', round(1e3 * wall, 3), 'kernel sum ms', round(tot, 3), 'launches', nl)
for ms, name, c in sorted(rows, reverse=True)[:14]: print('   %-24s %7.3f ms x%g' % (name, ms, c))
