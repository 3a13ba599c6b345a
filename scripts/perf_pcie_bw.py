import torch, time
for mb in (1, 4, 16, 64):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    e = time.perf_counter() - t
    t = time.perf_counter()
    for _ in range(20): h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    e2 = time.perf_counter() - t
    print("%d MB: H2D %.1f GB/s, D2H %.1f GB/s" % (mb, 20 * n / e / 1e9, 20 * n / e2 / 1e9))
