import time, torch
for mb in (16, 64, 256):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"H2D pinned {mb} MiB: {n / dt / 1e9:.1f} GB/s")
    for _ in range(3): h.copy_(d, non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): h.copy_(d, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"D2H pinned {mb} MiB: {n / dt / 1e9:.1f} GB/s")
