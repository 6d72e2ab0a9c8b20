#!/usr/bin/env python3
"""Wall time of one fault-injection campaign through the reference ABI (inference_multiple_with_faults).
usage: fault_campaign_rate.py [network [n_images [flips]]]"""
import ctypes as C, os, sys, tempfile, time
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_lib as gl
net = sys.argv[1] if len(sys.argv) > 1 else "cnvW1A1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
flips = int(sys.argv[3]) if len(sys.argv) > 3 else 100
cnv = net.startswith("cnv")
L = gl.load(net)
pdir = gl.param_dir("cifar10" if cnv else "mnist", net).encode()
rng = np.random.default_rng(0)
with tempfile.NamedTemporaryFile(dir="/tmp", suffix=".bin") as f:
    if cnv:
        f.write(rng.integers(0, 256, (n, 3073), dtype=np.uint8).tobytes())
    else:
        f.write((0x803).to_bytes(4, "big") + n.to_bytes(4, "big") + (28).to_bytes(4, "big") * 2)
        f.write(rng.integers(0, 256, (n, 784), dtype=np.uint8).tobytes())
    f.flush()
    cnt, usec = C.c_int(0), C.c_float(0)
    best = 1e9
    for rep in range(3):
        L.load_parameters(pdir)  # a campaign starts from clean parameters
        L.bnn_mi355x_set_fault_seed(1234 + rep)
        t0 = time.perf_counter()
        p = L.inference_multiple_with_faults(f.name.encode(), 10, C.byref(cnt), C.byref(usec), flips, 1, -1, None, 0)
        dt = time.perf_counter() - t0
        assert p and cnt.value == n
        L.free_results(p)
        best = min(best, dt)
print("%s: %d images, %d bit flips: %.1f ms per campaign (%.2f M img/s), device stages %.1f ms"
      % (net, n, flips, best * 1e3, n / best / 1e6, usec.value * n / 1e3))
