#!/usr/bin/env python3
"""tools/lfc_stamps.py [N]: where one k_lfc_block_s launch spends its time (diagnostic build: tools/build_variant.sh
stamps "-DBNN_LFC_STAMPS" lfcW1A1; run with BNN_MI355X_LIBDIR=bnn-pynq_amd/build/variants/stamps).  Wave 0 of every
block writes the 100 MHz wall clock at entry, behind the binarise hand-off, behind layer 0, behind each further
hand-off and at exit; this prints, over the blocks of the LAST of a series of launches, the spread of the entry
times, and min / median / max of every phase."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gpu_lib as gl  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = gl.load("lfcW1A1")
L.load_parameters(gl.param_dir("mnist", "lfcW1A1").encode())
imgs = torch.randint(0, 256, (n, 784), dtype=torch.uint8, device="cuda")
cls = torch.zeros(n, dtype=torch.int32, device="cuda")
L.bnn_mi355x_reserve(n)
for _ in range(20):
    L.bnn_mi355x_inference_device(imgs.data_ptr(), n, 10, cls.data_ptr(), None, None, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    L.bnn_mi355x_inference_device(imgs.data_ptr(), n, 10, cls.data_ptr(), None, None, None)
torch.cuda.synchronize()
print("n=%d: %.1f us per call back to back" % (n, (time.perf_counter() - t0) / 50 * 1e6))
L.bnn_mi355x_debug_lfc_stamps.argtypes = [C.c_void_p]
st = np.zeros((1024, 8), np.uint64)
assert L.bnn_mi355x_debug_lfc_stamps(st.ctypes.data) == 0
ipb = (n + 511) // 512
blocks = (n + ipb - 1) // ipb
s = st[:blocks, :7].astype(np.int64)
t = (s - s[:, 0].min()) / 100.0          # us since the first block's entry
names = ["entry", "binarise+handoff", "layer 0", "handoff", "layer 1+handoff", "layer 2+handoff", "layer 3+exit"]
print("%d blocks of %d images; entry times: min 0  median %.2f  p90 %.2f  max %.2f us; exits: min %.2f median %.2f max %.2f us"
      % (blocks, ipb, np.median(t[:, 0]), np.percentile(t[:, 0], 90), t[:, 0].max(), t[:, 6].min(), np.median(t[:, 6]), t[:, 6].max()))
for i in range(1, 7):
    d = t[:, i] - t[:, i - 1]
    print("  %-18s min %6.2f  median %6.2f  max %6.2f us" % (names[i], d.min(), np.median(d), d.max()))
life = t[:, 6] - t[:, 0]
print("  block lifetime     min %6.2f  median %6.2f  max %6.2f us;  first entry -> last exit %.2f us" % (life.min(), np.median(life), life.max(), t[:, 6].max()))
order = np.argsort(t[:, 0])
print("  entry time by block id (every 50th):", " ".join("%d:%.1f" % (b, t[b, 0]) for b in range(0, blocks, 50)))

# ---- per-wave stamps: [block][wave][16]
L.bnn_mi355x_debug_lfc_wstamps.argtypes = [C.c_void_p]
ws = np.zeros((1024, 16, 16), np.uint64)
assert L.bnn_mi355x_debug_lfc_wstamps(ws.ctypes.data) == 0
w = (ws[:blocks].astype(np.int64) - s[:, 0].min()) / 100.0
wn = ["entry", "prologue done", "handoff0 done", "L0 rows in", "L0 image 0 done", "L0 loop done", "L0 handoff done", "L1 rows in", "L1 image 0 done",
      "L1 loop done", "L1 handoff done", "L2 rows in", "L2 image 0 done", "L2 loop done", "L2 handoff done", "exit"]
print("per-wave phase lengths over all %d waves (us): min / p50 / p90 / p99 / max" % (blocks * 16))
for i in range(1, 16):
    d = (w[:, :, i] - w[:, :, i - 1]).reshape(-1)
    print("  %-18s %6.2f %6.2f %6.2f %6.2f %6.2f" % (wn[i], d.min(), np.percentile(d, 50), np.percentile(d, 90), np.percentile(d, 99), d.max()))
ex = t[:, 6]
print("exit time percentiles over blocks: p50 %.1f p75 %.1f p90 %.1f p95 %.1f p99 %.1f max %.1f" % tuple(np.percentile(ex, q) for q in (50, 75, 90, 95, 99, 100)))
late = np.argsort(-ex)[:8]
print("eight latest blocks (id, id mod 8, exit):", " ".join("%d/%d/%.1f" % (b, b % 8, ex[b]) for b in late))
print("blocks later than median + 5 us: %d; by id mod 8: %s" % ((ex > np.median(ex) + 5).sum(), np.bincount(np.nonzero(ex > np.median(ex) + 5)[0] % 8, minlength=8).tolist()))
b = int(late[0])
print("latest block %d, per wave (rows: waves 0..15; columns: the 16 stamps, us since first entry):" % b)
for wv in range(16):
    print("   w%-2d " % wv + " ".join("%6.1f" % x for x in w[b, wv]))
b = int(np.argsort(ex)[len(ex) // 2])
print("a median block %d:" % b)
for wv in range(16):
    print("   w%-2d " % wv + " ".join("%6.1f" % x for x in w[b, wv]))
