#!/usr/bin/env python3
"""1 048 576 CIFAR-10-shaped images on one GPU through the device-pointer and the host-buffer entry points."""
import sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, "tests")
import gpu_lib as gl
L = gl.load("cnvW1A1"); L.load_parameters(gl.param_dir("cifar10", "cnvW1A1").encode())
n = 1 << 20
g = torch.Generator(device="cuda"); g.manual_seed(5)
d = torch.randint(0, 256, (n, 3072), dtype=torch.uint8, device="cuda", generator=g)
cls = torch.zeros(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize(); t = time.perf_counter()
assert L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, cls.data_ptr(), None, None, None) == 0
torch.cuda.synchronize(); dt = time.perf_counter() - t
print("device API: 1,048,576 images in %.1f ms = %.2f M img/s" % (dt * 1e3, n / dt / 1e6), flush=True)
h = d.cpu().numpy()
usec = C.c_float(0); t = time.perf_counter()
p = L.bnn_mi355x_inference_buffer(h.ctypes.data, n, 10, C.byref(usec), 0)
dt = time.perf_counter() - t
got = np.ctypeslib.as_array(p, (n,)).copy(); L.free_results(p)
print("host buffer: %.1f ms = %.2f M img/s; classes equal: %s; histogram %s" % (dt * 1e3, n / dt / 1e6, bool((got == cls.cpu().numpy()).all()), np.bincount(got, minlength=10).tolist()))
