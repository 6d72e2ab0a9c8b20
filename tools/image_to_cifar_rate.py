#!/usr/bin/env python3
"""Pictures -> CIFAR-10 records: PIL on the host (the reference's image_to_cifar) vs the device path
(bnn_mi355x_images_to_cifar), decoded pictures already in host memory.  usage: image_to_cifar_rate.py [W H [N]]"""
import io, os, sys, time
import numpy as np
import torch  # noqa: F401  (HIP runtime order)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bnn-pynq_amd"))
from PIL import Image
import bnn

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
clf = bnn.CnvClassifier(bnn.NETWORK_CNVW1A1, "cifar10", bnn.RUNTIME_SW)
rng = np.random.default_rng(0)
imgs = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB") for _ in range(n)]
clf.images_to_cifar(imgs[:2])
t0 = time.perf_counter(); dev = clf.images_to_cifar(imgs); t_dev = time.perf_counter() - t0
# the C ABI call alone (pictures already numpy arrays)
import ctypes as C
arrs = [np.ascontiguousarray(np.asarray(im)) for im in imgs]
ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
ws, hs, bs = (C.c_int * n)(*[w] * n), (C.c_int * n)(*[h] * n), (C.c_int * n)(*[3] * n)
out = np.empty((n, 3073), np.uint8)
t0 = time.perf_counter()
assert clf.bnn.interface.bnn_mi355x_images_to_cifar(ptrs, ws, hs, bs, None, n, out.ctypes.data) == 0
t_abi = time.perf_counter() - t0
assert (out == dev).all()
t0 = time.perf_counter()
host = []
for im in imgs:
    buf = io.BytesIO(); clf.image_to_cifar(im.copy(), buf); host.append(np.frombuffer(buf.getvalue(), np.uint8))
t_host = time.perf_counter() - t0
assert (np.stack(host) == dev).all()
print("%d pictures %dx%d RGB: PIL %.2f ms/picture; device path %.2f ms/picture from PIL images (of which PIL->numpy %.2f), "
      "%.2f ms/picture for the C call alone (upload from pageable memory included); records identical"
      % (n, w, h, 1e3 * t_host / n, 1e3 * t_dev / n, 1e3 * (t_dev - t_abi) / n, 1e3 * t_abi / n))
