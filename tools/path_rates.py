#!/usr/bin/env python3
"""tools/path_rates.py NETWORK N [REPS]: end-to-end rates of the two entry points that take HOST data, same process,
same images: bnn_mi355x_inference_buffer (pageable host array -> classes) and inference_multiple(path) (file in the
page cache -> classes), next to the device-resident rate.  The chunk plan is the library's (BNN_MI355X_CHUNKS=
head:tail:max overrides it for A/B runs)."""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_lib as gl  # noqa: E402

net = sys.argv[1] if len(sys.argv) > 1 else "cnvW1A1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cnv = net.startswith("cnv")
N = gl.Net(net, "cifar10" if cnv else "mnist")
L = N.L
devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)
imgs = np.random.default_rng(0).integers(0, 256, (n, N.isz), dtype=np.uint8)
d = torch.from_numpy(imgs).cuda()
cls = torch.zeros(n, dtype=torch.int32, device="cuda")
L.bnn_mi355x_reserve(n)
for _ in range(3):
    L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, cls.data_ptr(), None, None, None)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, cls.data_ptr(), None, None, None)
torch.cuda.synchronize()
resident = (time.perf_counter() - t) / 5
want = cls.cpu().numpy()
N.classify(imgs, 10)
host = []
for _ in range(reps):
    t = time.perf_counter()
    got = N.classify(imgs, 10)
    host.append(time.perf_counter() - t)
assert (got == want).all()
with tempfile.NamedTemporaryFile(dir="/tmp", suffix=".bin") as f:
    if cnv:
        rec = np.empty((n, 3073), np.uint8)
        rec[:, 0] = 1
        rec[:, 1:] = imgs
        f.write(rec.tobytes())
        del rec
    else:
        f.write((0x803).to_bytes(4, "big") + n.to_bytes(4, "big") + (28).to_bytes(4, "big") * 2 + imgs.tobytes())
    f.flush()
    cnt, usec = C.c_int(0), C.c_float(0)
    filet = []
    os.dup2(devnull, 1)
    for _ in range(reps + 1):
        t = time.perf_counter()
        p = L.inference_multiple(f.name.encode(), 10, C.byref(cnt), C.byref(usec), 0)
        filet.append(time.perf_counter() - t)
        assert p and cnt.value == n
        res = np.ctypeslib.as_array(p, (n,)).copy()
        L.free_results(p)
    os.dup2(saved, 1)
    assert (res == want).all()
bases = (C.c_int * 128)()
k = L.bnn_mi355x_chunk_plan(n, 1, bases, 128)
print("%s n=%d plan=%s chunks=%d | resident %.2f ms (%.2f M/s) | host buffer best %.2f ms (%.2f M/s) median %.2f | file best %.2f ms (%.2f M/s) median %.2f"
      % (net, n, os.environ.get("BNN_MI355X_CHUNKS", "default"), k - 1, resident * 1e3, n / resident / 1e6, min(host) * 1e3, n / min(host) / 1e6,
         sorted(host)[len(host) // 2] * 1e3, min(filet[1:]) * 1e3, n / min(filet[1:]) / 1e6, sorted(filet[1:])[len(filet[1:]) // 2] * 1e3), flush=True)
