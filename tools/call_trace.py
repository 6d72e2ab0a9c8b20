#!/usr/bin/env python3
"""tools/call_trace.py NETWORK N: host-side time stamps (BNN_MI355X_TRACE=1, microseconds since the start of the call, on
stderr) of the last of 4 calls each of bnn_mi355x_inference_buffer, inference_multiple(path) and inference(path):
where the wall time of a host-data call goes that no device timeline shows -- file open, worker wake-up, first piece,
every chunk's arrival and launch, the final wait."""
import ctypes as C
import os
import sys
import tempfile
import time

os.environ["BNN_MI355X_TRACE"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_lib as gl  # noqa: E402

net = sys.argv[1] if len(sys.argv) > 1 else "cnvW1A1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
cnv = net.startswith("cnv")
N = gl.Net(net, "cifar10" if cnv else "mnist")
L = N.L
imgs = np.random.default_rng(0).integers(0, 256, (n, N.isz), dtype=np.uint8)
f = tempfile.NamedTemporaryFile(dir="/tmp", suffix=".bin")
if cnv:
    rec = np.empty((n, 3073), np.uint8)
    rec[:, 0] = 1
    rec[:, 1:] = imgs
    f.write(rec.tobytes())
else:
    f.write((0x803).to_bytes(4, "big") + n.to_bytes(4, "big") + (28).to_bytes(4, "big") * 2 + imgs.tobytes())
f.flush()
devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)
for kind in ("buffer", "file", "single"):
    for rep in range(4):
        time.sleep(0.02)
        sys.stderr.write("---- %s %s n=%d call %d\n" % (net, kind, n if kind != "single" else 1, rep))
        sys.stderr.flush()
        t = time.perf_counter()
        if kind == "buffer":
            N.classify(imgs, 10)
        else:
            os.dup2(devnull, 1)
            if kind == "file":
                cnt = C.c_int(0)
                p = L.inference_multiple(f.name.encode(), 10, C.byref(cnt), None, 0)
                L.free_results(p)
            else:
                L.inference(f.name.encode(), None, 10, None)
            os.dup2(saved, 1)
        sys.stderr.write("     wall %.1f us\n" % ((time.perf_counter() - t) * 1e6))
