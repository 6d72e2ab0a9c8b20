#!/usr/bin/env python3
"""tools/small_call_sweep.py NETWORK N [PLAN ...]: the two host-data entry points at the reference's own call size
(10 000 records: classify_cifars / classify_mnists on a test-set file), one line per chunk plan
(BNN_MI355X_CHUNKS=head:tail:max:growth, read by the library at every call; "default" = the shipped plan): best and
median wall time of REPS calls of bnn_mi355x_inference_buffer and of inference_multiple(path), next to the resident
rate.  Process-wide switches (BNN_MI355X_FEEDER_PIECE_MB, BNN_MI355X_FEED_HOST, BNN_MI355X_LANES) are taken from the
environment of the run and printed."""
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
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
plans = sys.argv[3:] or ["default"]
reps = int(os.environ.get("REPS", "9"))
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
res = []
for _ in range(5):
    t = time.perf_counter()
    L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, cls.data_ptr(), None, None, None)
    torch.cuda.synchronize()
    res.append(time.perf_counter() - t)
want = cls.cpu().numpy()
f = tempfile.NamedTemporaryFile(dir="/tmp", suffix=".bin")
if cnv:
    rec = np.empty((n, 3073), np.uint8)
    rec[:, 0] = 1
    rec[:, 1:] = imgs
    f.write(rec.tobytes())
    del rec
else:
    f.write((0x803).to_bytes(4, "big") + n.to_bytes(4, "big") + (28).to_bytes(4, "big") * 2 + imgs.tobytes())
f.flush()
sw = {k: os.environ[k] for k in ("BNN_MI355X_FEEDER_PIECE_MB", "BNN_MI355X_FEED_HOST", "BNN_MI355X_LANES", "BNN_MI355X_FEEDER_THREADS") if k in os.environ}
print("%s n=%d resident best %.3f ms (%.2f M/s)  switches %s" % (net, n, min(res) * 1e3, n / min(res) / 1e6, sw or "none"), flush=True)
for plan in plans:
    if plan == "default":
        os.environ.pop("BNN_MI355X_CHUNKS", None)
    else:
        os.environ["BNN_MI355X_CHUNKS"] = plan
    bases = (C.c_int * 256)()
    kb = L.bnn_mi355x_chunk_plan(n, 0, bases, 256)
    pb = [bases[i + 1] - bases[i] for i in range(kb - 1)]
    kf = L.bnn_mi355x_chunk_plan(n, 1, bases, 256)
    pf = [bases[i + 1] - bases[i] for i in range(kf - 1)]
    host, filet = [], []
    for _ in range(reps + 1):
        t = time.perf_counter()
        got = N.classify(imgs, 10)
        host.append(time.perf_counter() - t)
    assert (got == want).all()
    cnt, usec = C.c_int(0), C.c_float(0)
    os.dup2(devnull, 1)
    for _ in range(reps + 1):
        t = time.perf_counter()
        p = L.inference_multiple(f.name.encode(), 10, C.byref(cnt), C.byref(usec), 0)
        filet.append(time.perf_counter() - t)
        assert p and cnt.value == n
        r = np.ctypeslib.as_array(p, (n,)).copy()
        L.free_results(p)
    os.dup2(saved, 1)
    assert (r == want).all()
    host, filet = sorted(host[1:]), sorted(filet[1:])
    print("  plan %-22s buffer %s: best %.3f ms (%.2f M/s) median %.3f | file %s: best %.3f ms (%.2f M/s) median %.3f  usec/img %.4f"
          % (plan, pb, host[0] * 1e3, n / host[0] / 1e6, host[len(host) // 2] * 1e3, pf, filet[0] * 1e3, n / filet[0] / 1e6,
             filet[len(filet) // 2] * 1e3, usec.value), flush=True)
