import os, sys
sys.path.insert(0, "/root/repo")
with_torch = len(sys.argv) > 1
if with_torch:
    import torch
    torch.cuda.init()
import dashing_amd
c = dashing_amd.Context(0)
uid = dashing_amd.comm_unique_id()
c.comm_init(uid, 0, 1)
print("comm ok", c.comm_rank(), flush=True)
for l in open("/proc/self/maps"):
    if ("rccl" in l or "amdhip" in l) and "r-xp" in l:
        print(l.split()[-1])
c.comm_destroy(); c.close()
