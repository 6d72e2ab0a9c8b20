#!/usr/bin/env python3
"""tools/two_lane_probe.py [NETWORK] [N]: what running the chunks of a host-path call on TWO compute lanes (two streams, two
workspaces) would buy, before building it: the chunk plan of N images walked (a) on one stream, (b) alternating over two
streams -- the second lane is a second copy of the library (its own workspace), images resident in HBM, no copies."""
import ctypes as C
import os
import shutil
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gpu_lib as gl  # noqa: E402
from bnn import abi  # noqa: E402

net = sys.argv[1] if len(sys.argv) > 1 else "cnvW1A1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
is_cnv = net.startswith("cnv")
pdir = gl.param_dir("cifar10" if is_cnv else "mnist", net).encode()
A = gl.load(net)
tmp = tempfile.mkdtemp()
libs = [A]
for i in range(max(int(os.environ.get("LANES", "2")), 2) - 1):   # LANES=3: a third copy of the library, three lanes
    shutil.copy(gl.lib_path(net), os.path.join(tmp, "lane%d.so" % (i + 2)))
    B = C.CDLL(os.path.join(tmp, "lane%d.so" % (i + 2)))
    abi.declare_legacy(B)
    abi.declare_extensions(B)
    libs.append(B)
for L in libs:
    L.load_parameters(pdir)
isz = A.bnn_mi355x_image_bytes()
bases = (C.c_int * 256)()
halves = os.environ.get("HALVES")  # HALVES=1: instead of the chunk plans, the batch cut into two equal halves (what a fork-join inside
                                   # bnn_mi355x_inference_device would run), against the same batch as one call
for from_file in ((0,) if halves else (0, 1)):
    k = A.bnn_mi355x_chunk_plan(n, from_file, bases, 256)
    plan = [bases[i] for i in range(k)]
    if halves:  # HALVES=k: k equal pieces (multiples of 256 images)
        k = max(int(halves), 2)
        plan = [min(((n * i // k) + 255) & ~255, n) for i in range(k)] + [n]
        if os.environ.get("SPLIT"):  # SPLIT=0.4: two pieces, the first 40 % of the batch
            plan = [0, (int(n * float(os.environ["SPLIT"])) + 255) & ~255, n]
    imgs = torch.randint(0, 256, (n, isz), dtype=torch.uint8, device="cuda")
    cls = torch.zeros(n, dtype=torch.int32, device="cuda")
    s = [torch.cuda.Stream() for _ in libs]

    def walk(lanes):
        for c in range(len(plan) - 1):
            lane = c % lanes
            L = libs[lane]
            m = plan[c + 1] - plan[c]
            rc = L.bnn_mi355x_inference_device(C.c_void_p(imgs.data_ptr() + plan[c] * isz), m, 10, C.c_void_p(cls.data_ptr() + 4 * plan[c]), None, None,
                                               C.c_void_p(s[lane].cuda_stream))
            assert rc == 0
        torch.cuda.synchronize()

    for lanes in (1, len(libs), 1, len(libs)):
        walk(lanes); walk(lanes)
        ref = cls.clone() if lanes == 1 else ref
        t = []
        for _ in range(7):
            t0 = time.perf_counter(); walk(lanes); t.append(time.perf_counter() - t0)
        assert (cls == ref).all()
        print("%s n=%d plan(from_file=%d) %d chunks, %d lane(s): best %.3f ms median %.3f ms" % (net, n, from_file, len(plan) - 1, lanes, min(t) * 1e3, sorted(t)[3] * 1e3), flush=True)
# one launch of everything, for reference
t = []
for _ in range(7):
    t0 = time.perf_counter()
    assert A.bnn_mi355x_inference_device(C.c_void_p(imgs.data_ptr()), n, 10, C.c_void_p(cls.data_ptr()), None, None, C.c_void_p(s[0].cuda_stream)) == 0
    torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
print("one call of %d images: best %.3f ms" % (n, min(t) * 1e3))
shutil.rmtree(tmp)
