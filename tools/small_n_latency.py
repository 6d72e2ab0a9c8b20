import sys, os, time, ctypes as C, numpy as np
import torch
sys.path.insert(0, "tests")
import gpu_lib as gl, oracle_lib as ol
net = gl.Net("cnvW1A1", "cifar10"); o = ol.Oracle("cnvW1A1", ol.param_dir("cifar10", "cnvW1A1"))
for n in (1, 8, 16, 24, 32, 48, 64, 65, 128):
    imgs = np.random.default_rng(n).integers(0, 256, (n, 3072), dtype=np.uint8)
    assert (net.raw(imgs) == o.scores_fast(imgs)).all()
    ts = []
    usec = C.c_float(0)
    for _ in range(40):
        t = time.perf_counter()
        p = net.L.bnn_mi355x_inference_buffer(imgs.ctypes.data, n, 10, C.byref(usec), 0)
        ts.append((time.perf_counter() - t) * 1e6)
        net.L.free_results(p)
    ts.sort()
    print("n=%d whole call median %.1f us best %.1f, usecPerImage*n %.1f" % (n, ts[len(ts)//2], ts[0], usec.value * n))
