"""raw host-to-device copy rate of the box (pinned and pageable), for the PCIe-inclusive note in DESIGN.md"""
import time, torch
n = 2388787200  # one 64-GOP 1080p batch
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, h in (("pinned", torch.empty(n, dtype=torch.uint8).pin_memory()), ("pageable", torch.empty(n, dtype=torch.uint8))):
    h.fill_(7)
    for _ in range(2):
        d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("%s H2D %.1f ms  %.1f GB/s" % (name, dt * 1e3, n / dt / 1e9))
