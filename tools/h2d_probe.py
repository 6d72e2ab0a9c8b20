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
# a file in the page cache, mapped: does the runtime's pageable path take it at link speed?
import mmap, os, tempfile
with tempfile.NamedTemporaryFile(dir="/tmp", suffix=".bin") as f:
    f.write(a.tobytes())
    f.flush()
    for populate in (False, True):
        t = []
        for _ in range(4):
            t0 = time.perf_counter()
            fd = os.open(f.name, os.O_RDONLY)
            mm = mmap.mmap(fd, N, flags=mmap.MAP_SHARED | (mmap.MAP_POPULATE if populate else 0), prot=mmap.PROT_READ)
            src = torch.frombuffer(mm, dtype=torch.uint8)
            t1 = time.perf_counter()
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            t.append((time.perf_counter() - t0, time.perf_counter() - t1))
            del src
            mm.close()
            os.close(fd)
        b = min(t)
        print("mmap'ed file (populate=%s) -> HBM: %.2f ms incl. mmap (copy alone %.2f ms) = %.1f GB/s" % (populate, b[0] * 1e3, b[1] * 1e3, N / b[0] / 1e9), flush=True)
    # pread into pinned memory, T threads
    fd = os.open(f.name, os.O_RDONLY)
    pv = memoryview(b if False else pinned.numpy())
    for T in (1, 2, 4, 8, 12, 16):
        part = (N + T - 1) // T

        def rd(lo):
            hi = min(lo + part, N)
            o = lo
            while o < hi:
                o += os.preadv(fd, [pv[o:hi]], o)
        tt = []
        for _ in range(4):
            th = [threading.Thread(target=rd, args=(lo,)) for lo in range(0, N, part)]
            t0 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            tt.append(time.perf_counter() - t0)
        print("pread page cache -> pinned, %2d threads: %.2f ms = %.1f GB/s" % (T, min(tt) * 1e3, N / min(tt) / 1e9), flush=True)
    os.close(fd)
