import torch, time
x = torch.empty(9_800_000_000, dtype=torch.uint8, device="cuda")
y = torch.empty_like(x)
x.fill_(3)
for n in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    y.copy_(x); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"torch copy 9.8 GB: {dt*1e3:.3f} ms  {2*x.numel()/dt/1e12:.2f} TB/s read+write")
x32 = x.view(torch.int32)
for n in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = x32.sum(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"torch sum (read only) 9.8 GB: {dt*1e3:.3f} ms  {x.numel()/dt/1e12:.2f} TB/s")
for n in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    y.fill_(1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"torch fill (write only) 9.8 GB: {dt*1e3:.3f} ms  {x.numel()/dt/1e12:.2f} TB/s")
