import sys,time,os
sys.path.insert(0,".")
import torch
from cutfemx_amd.dist import DistributedPoisson, SlabPartition
from cutfemx_amd import _lib
dev=torch.device("cuda",0)
for r in (3,7):
  for ov in ("0","1"):
    os.environ["CFX_OVERLAP"]=ov
    part=SlabPartition.create_owner(512,8,r); dp=DistributedPoisson(part,dev,mode="owner"); dp.part.world=1
    for _ in range(3): dp.step()
    torch.cuda.synchronize(); s0=_lib.sync_count(); t0=time.perf_counter()
    for _ in range(6): dp.step()
    torch.cuda.synchronize(); print("rank",r,"overlap",ov, round(1e3*(time.perf_counter()-t0)/6,3),"ms  syncs/step",(_lib.sync_count()-s0)/6, flush=True)
    del dp
