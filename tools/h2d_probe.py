#!/usr/bin/env python3
"""tools/h2d_probe.py: what the host -> HBM link of this box gives, to put the PCIe-inclusive figures against:
pageable vs pinned source, one big copy vs many 1..16 MB pieces, and the CPU side (threads copying into pinned memory)."""
import threading
import time

import numpy as np
import torch

N = 403 * 1024 * 1024            # the bytes of 131 072 CIFAR-10 records
dev = torch.device("cuda")
dst = torch.empty(N, dtype=torch.uint8, device=dev)
pageable = torch.from_numpy(np.random.default_rng(0).integers(0, 256, N, dtype=np.uint8))
pinned = torch.empty(N, dtype=torch.uint8).pin_memory()
pinned.copy_(pageable)


def best(f, reps=5):
    t = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    return min(t)


for name, src in (("pageable", pageable), ("pinned", pinned)):
    dt = best(lambda: dst.copy_(src, non_blocking=True))
    print("%-9s one copy of %d MB: %.2f ms = %.1f GB/s" % (name, N >> 20, dt * 1e3, N / dt / 1e9), flush=True)
for mb in (1, 2, 4, 8, 16, 64):
    p = mb << 20

    def pieces():
        for o in range(0, N, p):
            dst[o:o + p].copy_(pinned[o:o + p], non_blocking=True)
    dt = best(pieces)
    print("pinned    %3d MB pieces (%4d copies): %.2f ms = %.1f GB/s, %.1f us per call" % (mb, (N + p - 1) // p, dt * 1e3, N / dt / 1e9, dt / ((N + p - 1) // p) * 1e6), flush=True)
# CPU side: T threads memcpy pageable -> pinned (numpy releases the GIL in copyto)
a, b = pageable.numpy(), pinned.numpy()
for T in (1, 2, 4, 8, 12, 16):
    def run():
        part = (N + T - 1) // T
        th = [threading.Thread(target=lambda lo=lo: np.copyto(b[lo:lo + part], a[lo:lo + part])) for lo in range(0, N, part)]
        for x in th:
            x.start()
        for x in th:
            x.join()
    t = []
    for _ in range(4):
        t0 = time.perf_counter()
        run()
        t.append(time.perf_counter() - t0)
    print("memcpy pageable -> pinned, %2d threads: %.2f ms = %.1f GB/s" % (T, min(t) * 1e3, N / min(t) / 1e9), flush=True)
