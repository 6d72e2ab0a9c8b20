#!/usr/bin/env python3
"""tools/host_timeline.py NETWORK N [file]: a few calls of the host-buffer (or, with `file`, the file) entry point, to be run
under `rocprofv3 --kernel-trace --memory-copy-trace`; tools/timeline_summary.py then shows where the last call's time
went on the device (copies, stages, gaps)."""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_lib as gl  # noqa: E402

net = sys.argv[1] if len(sys.argv) > 1 else "cnvW1A1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
use_file = len(sys.argv) > 3 and sys.argv[3] == "file"
cnv = net.startswith("cnv")
N = gl.Net(net, "cifar10" if cnv else "mnist")
imgs = np.random.default_rng(0).integers(0, 256, (n, N.isz), dtype=np.uint8)
if use_file:
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
for rep in range(4):
    time.sleep(0.05)          # a visible gap between the calls in the trace
    t = time.perf_counter()
    if use_file:
        cnt = C.c_int(0)
        os.dup2(devnull, 1)
        p = N.L.inference_multiple(f.name.encode(), 10, C.byref(cnt), None, 0)
        os.dup2(saved, 1)
        N.L.free_results(p)
    else:
        N.classify(imgs, 10)
    print("call %d: %.2f ms" % (rep, (time.perf_counter() - t) * 1e3), flush=True)
