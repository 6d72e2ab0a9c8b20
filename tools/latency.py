#!/usr/bin/env python3
"""tools/latency.py: single-image latency through the reference ABI (inference(path, ...)):
device time reported in usecPerImage and wall time of the whole call (file read + H2D + stages + D2H)."""
import ctypes as C
import os
import sys
import time

import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_lib as gl  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
only = sys.argv[1:]   # optional: the networks to time
for network, dataset, f in (("cnvW1A1", "cifar10", "deer.cifar"), ("cnvW2A2", "cifar10", "deer.cifar"), ("lfcW1A1", "mnist", "3.image-idx3-ubyte"),
                            ("lfcW1A2", "mnist", "3.image-idx3-ubyte")):
    if only and network not in only:
        continue
    net = gl.Net(network, dataset)
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    os.dup2(devnull, 1)          # the ABI prints like the reference does
    usec = C.c_float(0)
    path, uref = os.path.join(G, f).encode(), C.byref(usec)   # (built once: the call is what is timed, not Python's string handling)
    dev, wall = [], []
    for _ in range(30):
        t = time.perf_counter()
        net.L.inference(path, None, 10, uref)
        wall.append((time.perf_counter() - t) * 1e6)
        dev.append(usec.value)
    os.dup2(saved, 1)
    print("%s single image: device %.1f us (min of 30), whole call %.1f us" % (network, min(dev), min(wall)))
