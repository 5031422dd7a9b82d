import sys, time, torch
sys.path.insert(0,'/root/repo')
from dance_amd.graph import morton_order
g=torch.Generator(device='cuda').manual_seed(0)
x=torch.randn(1_000_000,50,device='cuda',generator=g)
for i in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter(); p=morton_order(x); torch.cuda.synchronize(); print(round((time.perf_counter()-t0)*1e3,2),'ms')
print(p[:5].tolist())
