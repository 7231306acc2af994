#!/usr/bin/env python3
"""Host<->device copy rates from page-locked memory on this box (what bounds the host-inclusive rate)."""
import time
import torch
dev = torch.device("cuda", 0)
for mb in (16, 128, 512):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
    d2 = torch.empty(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for name, fn in (("h2d", lambda: d.copy_(h, non_blocking=True)), ("d2h", lambda: h.copy_(d, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        print(f"{name} {mb} MB: {5 * n / (time.perf_counter() - t) / 1e9:.1f} GB/s")
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        with torch.cuda.stream(s1):
            d.copy_(h, non_blocking=True)
        with torch.cuda.stream(s2):
            h2.copy_(d2, non_blocking=True)
    torch.cuda.synchronize()
    print(f"both directions at once {mb} MB each: {5 * n / (time.perf_counter() - t) / 1e9:.1f} GB/s per direction")
