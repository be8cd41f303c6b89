import torch, time, numpy as np
n=2<<30
a=np.ones(n//4,dtype=np.int32)
d=torch.empty(n//4,dtype=torch.int32,device='cuda')
t=torch.from_numpy(a)
for r in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter(); d.copy_(t); torch.cuda.synchronize(); dt=time.perf_counter()-t0
print("pageable H2D GB/s", n/dt/1e9)
p=torch.empty(n//4,dtype=torch.int32).pin_memory()
for r in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter(); d.copy_(p,non_blocking=True); torch.cuda.synchronize(); dt=time.perf_counter()-t0
print("pinned H2D GB/s", n/dt/1e9)
for r in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter(); p.copy_(d,non_blocking=True); torch.cuda.synchronize(); dt=time.perf_counter()-t0
print("pinned D2H GB/s", n/dt/1e9)
b=np.empty(n//4,dtype=np.int32); tb=torch.from_numpy(b)
for r in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter(); tb.copy_(d); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print("pageable D2H GB/s", n/dt/1e9)
t0=time.perf_counter(); c=np.empty(n//4,dtype=np.int32); c[:]=1; print("first touch GB/s", n/(time.perf_counter()-t0)/1e9)
t0=time.perf_counter(); c[:]=a; print("memcpy GB/s", n/(time.perf_counter()-t0)/1e9)
