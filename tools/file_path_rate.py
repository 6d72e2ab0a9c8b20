#!/usr/bin/env python3
"""End-to-end rate of the reference's file ABI: inference_multiple(path) on a CIFAR-10 / MNIST file in the page cache.
usage: file_path_rate.py [network [n_images]]"""
import ctypes as C, os, sys, tempfile, time
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_lib as gl
net = sys.argv[1] if len(sys.argv) > 1 else "cnvW1A1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
cnv = net.startswith("cnv")
L = gl.load(net)
L.load_parameters(gl.param_dir("cifar10" if cnv else "mnist", net).encode())
rng = np.random.default_rng(0)
with tempfile.NamedTemporaryFile(dir="/tmp", suffix=".bin") as f:
    if cnv:
        rec = rng.integers(0, 256, (n, 3073), dtype=np.uint8)
        f.write(rec.tobytes())
    else:
        f.write((0x803).to_bytes(4, "big") + n.to_bytes(4, "big") + (28).to_bytes(4, "big") + (28).to_bytes(4, "big"))
        f.write(rng.integers(0, 256, (n, 784), dtype=np.uint8).tobytes())
    f.flush()
    cnt, usec = C.c_int(0), C.c_float(0)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        p = L.inference_multiple(f.name.encode(), 10, C.byref(cnt), C.byref(usec), 0)
        dt = time.perf_counter() - t0
        assert p and cnt.value == n
        L.free_results(p)
        best = min(best, dt)
print("%s: inference_multiple on a %d-image file: %.1f ms end to end = %.2f M img/s (device stages %.1f ms)"
      % (net, n, best * 1e3, n / best / 1e6, usec.value * n / 1e3))
