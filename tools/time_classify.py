"""Classification alone at n^3 for three level sets: all negative (every block uniform), the bench's sphere, a plane."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
import cutfemx_amd as cfx
from cutfemx_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
mesh = cfx.Mesh.create_box(3, n)
V = cfx.FunctionSpace(mesh, 1)
sphere = bench.sphere_level_set(torch, n, dev)
cases = {"all negative": -torch.ones_like(sphere), "sphere": sphere, "shifted sphere": sphere + 0.11}
for name, phi in cases.items():
    f = cfx.Function(V, phi)
    cd = cfx.cut(f)              # (the first one builds the summary)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().cfx_profile_enable(1))
    _lib.check(_lib.lib().cfx_profile_reset())
    for _ in range(5):
        cfx.update(cd)
    torch.cuda.synchronize()
    out = {}
    for i in range(_lib.lib().cfx_profile_count()):
        nm, ms, cnt = C.c_char_p(), C.c_double(), C.c_int64()
        _lib.check(_lib.lib().cfx_profile_get(i, C.byref(nm), C.byref(ms), C.byref(cnt)))
        if cnt.value:
            out[nm.value.decode()] = round(ms.value / cnt.value, 4)
    _lib.check(_lib.lib().cfx_profile_enable(0))
    print(f"{name:16s}", {k: v for k, v in out.items() if k in ("classify", "sign_codes", "locate_entities")})
