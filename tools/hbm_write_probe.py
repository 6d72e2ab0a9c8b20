#!/usr/bin/env python3
"""tools/hbm_write_probe.py: what a plain streaming store / copy reaches on this GPU (torch fill_ and copy_ on buffers the
size of layer 0's output, 900 x 8 B x 131 072 images = 944 MB) -- the yardstick for k_conv0_mfma, whose traffic is
dominated by its 7 200 B per image of stores."""
import torch

n = 131072 * 900 * 8
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
src = torch.empty(131072 * 3072, dtype=torch.uint8, device="cuda")


def timed(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


t = timed(lambda: a.fill_(7))
print("fill   %4d MB: %7.1f us  %.2f TB/s written" % (n >> 20, t, n / t / 1e6))
t = timed(lambda: b.copy_(a))
print("copy   %4d MB: %7.1f us  %.2f TB/s read + %.2f TB/s written" % (n >> 20, t, n / t / 1e6, n / t / 1e6))
av = a.view(torch.int64)
t = timed(lambda: torch.sum(av))
print("read   %4d MB: %7.1f us  %.2f TB/s read" % (n >> 20, t, n / t / 1e6))
m = src.numel()
t = timed(lambda: (src.fill_(1), a.fill_(2)))
print("fill %d MB + fill %d MB back to back: %7.1f us" % (m >> 20, n >> 20, t))
