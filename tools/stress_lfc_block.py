#!/usr/bin/env python3
"""Stress of the in-kernel hand-offs of k_lfc_block_s (vector stores -> L2 -> scalar cache, once per layer and
block): many batches of random sizes inside its policy range (and beyond it with BNN_MI355X_LFC_BLOCK_MAX),
back to back on two streams that interleave a second, CNV, library's work (uneven load on the CUs), every
output word compared with the CPU restatement.  usage: stress_lfc_block.py [rounds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_lib as gl  # noqa: E402
import oracle_lib as ol  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
L = gl.load("lfcW1A1")
L.load_parameters(gl.param_dir("mnist", "lfcW1A1").encode())
o = ol.Oracle("lfcW1A1", ol.param_dir("mnist", "lfcW1A1"))
C = gl.load("cnvW1A1")
C.load_parameters(gl.param_dir("cifar10", "cnvW1A1").encode())
hi = int(os.environ.get("BNN_MI355X_LFC_BLOCK_MAX", "131072"))
cimgs = torch.randint(0, 256, (3000, 3072), dtype=torch.uint8, device="cuda")
ccls = torch.zeros(3000, dtype=torch.int32, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
L.bnn_mi355x_reserve(min(hi, 131072))
C.bnn_mi355x_reserve(3000)
t0 = time.time()
total = 0
for r in range(rounds):
    n = int(rng.integers(4097, min(hi, 131072) + 1))
    imgs = rng.integers(0, 256, (n, 784), dtype=np.uint8)
    if r % 3 == 0:
        imgs = np.where(rng.random(imgs.shape) < 0.12, 255, 0).astype(np.uint8)
    d = torch.from_numpy(imgs).cuda()
    w = torch.zeros(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    reps = int(rng.integers(1, 4))
    for k in range(reps):   # the same batch several times back to back (stale scalar-cache lines of the previous pass), CNV work beside it
        if r % 2:
            assert C.bnn_mi355x_inference_device(cimgs.data_ptr(), int(rng.integers(1, 3000)), 10, ccls.data_ptr(), None, None, s2.cuda_stream) == 0
        assert L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, None, None, w.data_ptr(), s1.cuda_stream) == 0
    torch.cuda.synchronize()
    want = o.words_fast(imgs)
    got = w.cpu().numpy().view(np.uint64)
    bad = np.flatnonzero(got != want)
    if bad.size:
        print("MISMATCH round %d n=%d: %d words differ, first at image %d: got %x want %x" % (r, n, bad.size, bad[0], got[bad[0]], want[bad[0]]))
        sys.exit(1)
    total += n * reps
    if r % 25 == 24:
        print("round %d ok, %d images so far, %.0f s" % (r + 1, total, time.time() - t0), flush=True)
print("stress ok: %d rounds, %d images" % (rounds, total))
