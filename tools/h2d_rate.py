import torch, time
for mb in (2, 8, 32, 128):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    reps = max(4, 2048 // mb)
    t0 = time.perf_counter()
    for _ in range(reps): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mb:4d} MiB pinned H2D: {reps * n / dt / 1e9:.1f} GB/s")
