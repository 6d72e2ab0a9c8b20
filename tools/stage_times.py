#!/usr/bin/env python3
"""tools/stage_times.py NETWORK BATCH [BATCH ...]: per-stage HIP-event device times (us) of one batch through
bnn_mi355x_inference_device (profiling on: the staged form at every size), and the wall time per call with
profiling off (the shipped dispatch policy, launch gaps included)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gpu_lib as gl  # noqa: E402

net = sys.argv[1]
is_cnv = net.startswith("cnv")
L = gl.load(net)
L.load_parameters(gl.param_dir("cifar10" if is_cnv else "mnist", net).encode())
isz = L.bnn_mi355x_image_bytes()
for n in [int(x) for x in sys.argv[2:]]:
    imgs = torch.randint(0, 256, (n, isz), dtype=torch.uint8, device="cuda")
    cls = torch.zeros(n, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream

    def call():
        assert L.bnn_mi355x_inference_device(imgs.data_ptr(), n, 10, cls.data_ptr(), None, None, s) == 0

    R = 50 if n <= 20000 else 10
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        call()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / R * 1e6
    L.bnn_mi355x_profile(1)
    for _ in range(R):
        call()
    torch.cuda.synchronize()
    ms = (C.c_float * 16)()
    nc = C.c_int(0)
    k = L.bnn_mi355x_profile_read(ms, 16, C.byref(nc))
    L.bnn_mi355x_profile(0)
    print("%s n=%d wall=%.1f us/call (%.2f M img/s) | staged: %s sum=%.1f us" % (
        net, n, wall, n / wall, " ".join("%s=%.1f" % (L.bnn_mi355x_stage_name(i).decode().replace(" ", "_"), ms[i] / R * 1e3)
                                         for i in range(k)), sum(ms[:k]) / R * 1e3), flush=True)
