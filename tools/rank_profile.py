"""Per-kernel HIP-event profile of ONE rank's slab of the N-rank partition (no exchange): python tools/rank_profile.py n world rank"""
import ctypes as C
import sys

sys.path.insert(0, '.')
import torch

from cutfemx_amd import _lib
from cutfemx_amd.dist import DistributedPoisson, SlabPartition

n, world, rank = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device('cuda', 0)
part = SlabPartition.create_owner(n, world, rank)
dp = DistributedPoisson(part, dev, mode="owner")
dp.part.world = 1
for _ in range(3):
    dp.step()
torch.cuda.synchronize()
_lib.check(_lib.lib().cfx_profile_enable(1))
_lib.check(_lib.lib().cfx_profile_reset())
K = 5
for _ in range(K):
    dp.step()
torch.cuda.synchronize()
rows = []
for i in range(_lib.lib().cfx_profile_count()):
    nm, ms, cnt = C.c_char_p(), C.c_double(), C.c_int64()
    _lib.check(_lib.lib().cfx_profile_get(i, C.byref(nm), C.byref(ms), C.byref(cnt)))
    if cnt.value:
        rows.append((ms.value / K, nm.value.decode(), cnt.value / K))
print(f"rank {rank} of {world} at {n}^3: kernel sum {sum(r[0] for r in rows):.3f} ms in {sum(r[2] for r in rows):.0f} launches")
for ms, name, cnt in sorted(rows, reverse=True)[:24]:
    print(f"  {name:26s} {ms:7.3f} ms x{cnt:.0f}")
