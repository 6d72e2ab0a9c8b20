#!/usr/bin/env python3
"""tools/plan_ab.py NETWORK N ROUNDS PLAN [PLAN ...]: chunk plans (BNN_MI355X_CHUNKS, read by the library at every call;
"default" = the shipped plan) compared INTERLEAVED -- every round runs each plan once, buffer call then file call -- so that
clock and box drift hit all plans alike; median, best and the 25-75 % range per plan over the rounds.  (Sequential blocks of
calls per plan, as tools/small_call_sweep.py runs them, differ by 3-5 % between identical plans on these boxes.)"""
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

net, n, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
plans = sys.argv[4:] or ["default"]
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
    del rec
else:
    f.write((0x803).to_bytes(4, "big") + n.to_bytes(4, "big") + (28).to_bytes(4, "big") * 2 + imgs.tobytes())
f.flush()
devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)
T = {p: {"buffer": [], "file": []} for p in plans}


PER_CALL = ("BNN_MI355X_NO_CHUNK_TIMING",)   # switches the library reads at every call: "env:NAME=VALUE" as a plan name


def setplan(p):
    os.environ.pop("BNN_MI355X_CHUNKS", None)
    for k in PER_CALL:
        os.environ.pop(k, None)
    if p.startswith("env:"):
        k, v = p[4:].split("=")
        os.environ[k] = v
    elif p != "default":
        os.environ["BNN_MI355X_CHUNKS"] = p


usec, cnt = C.c_float(0), C.c_int(0)
want = None
os.dup2(devnull, 1)
for r in range(rounds + 2):
    for p in (plans if r % 2 == 0 else plans[::-1]):
        setplan(p)
        t = time.perf_counter()
        q = L.bnn_mi355x_inference_buffer(imgs.ctypes.data, n, 10, C.byref(usec), 0)
        tb = time.perf_counter() - t
        got = np.ctypeslib.as_array(q, (n,)).copy()
        L.free_results(q)
        t = time.perf_counter()
        q = L.inference_multiple(f.name.encode(), 10, C.byref(cnt), C.byref(usec), 0)
        tf = time.perf_counter() - t
        got2 = np.ctypeslib.as_array(q, (n,)).copy()
        L.free_results(q)
        if want is None:
            want = got
        assert (got == want).all() and (got2 == want).all(), "classes differ under plan " + p
        if r >= 2:
            T[p]["buffer"].append(tb)
            T[p]["file"].append(tf)
os.dup2(saved, 1)
print("%s n=%d, %d interleaved rounds" % (net, n, rounds))
for p in plans:
    setplan(p)
    b = (C.c_int * 256)()
    k = L.bnn_mi355x_chunk_plan(n, 0, b, 256)
    sizes = [b[i + 1] - b[i] for i in range(k - 1)]
    out = "  %-24s %s" % (p, sizes if len(sizes) < 12 else "%d chunks" % len(sizes))
    for kind in ("buffer", "file"):
        v = np.sort(np.array(T[p][kind])) * 1e3
        out += " | %s median %.3f ms (%.2f M/s) best %.3f  q25-q75 %.3f-%.3f" % (kind, np.median(v), n / np.median(v) / 1e3, v[0], v[len(v) // 4], v[3 * len(v) // 4])
    print(out)
