import time, torch
torch.cuda.init()
x = torch.zeros(1, device="cuda:0")
keep = []
for i in range(12):
    t0 = time.perf_counter(); h = torch.empty(8 + i, dtype=torch.int32, pin_memory=True); dt = time.perf_counter() - t0
    keep.append(h)
    print(f"pinned alloc {i}: {1e3*dt:.2f} ms")
