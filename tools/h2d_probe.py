import torch, time
h = torch.empty((64,50,2054)).pin_memory(); d = torch.empty((64,50,2054), device="cuda:0")
for _ in range(3): d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
print("H2D pinned 26.3 MB: %.2f ms = %.1f GB/s" % (dt*1e3, 26.3e-3/dt))
h2 = torch.empty((64,50,2054))
t0=time.perf_counter()
for _ in range(5): d.copy_(h2)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
print("H2D pageable: %.2f ms = %.1f GB/s" % (dt*1e3, 26.3e-3/dt))
