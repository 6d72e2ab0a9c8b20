#!/usr/bin/env python3
"""Device time of one batch through bnn_mi355x_inference_device over a range of batch sizes, no stage events.
usage: BATCHES=1,256,... batch_sweep.py [network]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_lib as gl
net = sys.argv[1] if len(sys.argv) > 1 else "cnvW1A1"
cnv = net.startswith("cnv")
L = gl.load(net)
L.load_parameters(gl.param_dir("cifar10" if cnv else "mnist", net).encode())
dev = torch.device("cuda", 0)
default = "1,16,64,256,1024,4096,10000,32768,131072"
for batch in [int(x) for x in os.environ.get("BATCHES", default).split(",")]:
    imgs = torch.randint(0, 256, (batch, 3072 if cnv else 784), dtype=torch.uint8, device=dev)
    cls = torch.zeros(batch, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    L.bnn_mi355x_reserve(batch)
    reps = 50 if batch <= 4096 else 10
    for _ in range(3):
        L.bnn_mi355x_inference_device(imgs.data_ptr(), batch, 10, cls.data_ptr(), None, None, s)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        for _ in range(reps):
            L.bnn_mi355x_inference_device(imgs.data_ptr(), batch, 10, cls.data_ptr(), None, None, s)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    print("%s batch %6d: %9.1f us  %8.3f Mimg/s" % (net, batch, best * 1e6, batch / best / 1e6), flush=True)
