#!/usr/bin/env python3
"""tools/host_path_rate.py: end-to-end rate of the HOST-buffer entry point (PCIe-inclusive):
numpy array in host memory -> bnn_mi355x_inference_buffer -> int32 classes in host memory."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_lib as gl  # noqa: E402

net = gl.Net(sys.argv[1] if len(sys.argv) > 1 else "cnvW1A1", "cifar10" if len(sys.argv) < 3 else sys.argv[2])
for n in (10000, 131072, 524288):
    imgs = np.random.default_rng(0).integers(0, 256, (n, net.isz), dtype=np.uint8)
    net.classify(imgs[:1000], 10)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        net.classify(imgs, 10)
        best = min(best, time.perf_counter() - t)
    print("n=%7d  end-to-end %.2f ms  %.2f M img/s  (%.1f GB/s of input)   device-only %.3f us/img" % (
        n, best * 1e3, n / best / 1e6, n * net.isz / best / 1e9, net.usec))
