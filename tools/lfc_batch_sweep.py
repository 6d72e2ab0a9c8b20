#!/usr/bin/env python3
"""Device time of one LFC batch through bnn_mi355x_inference_device at a range of batch sizes, without stage
events (so the runtime is free to pick the one-launch form).  usage: BATCHES=129,1024,... lfc_ab.py [network]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_lib as gl
net = sys.argv[1] if len(sys.argv) > 1 else "lfcW1A1"
L = gl.load(net)
L.load_parameters(gl.param_dir("mnist", net).encode())
dev = torch.device("cuda", 0)
for batch in [int(x) for x in os.environ.get("BATCHES", "1000,10000,32768,131072").split(",")]:
    imgs = torch.randint(0, 256, (batch, 784), dtype=torch.uint8, device=dev)
    cls = torch.zeros(batch, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    L.bnn_mi355x_reserve(batch)
    for _ in range(5):
        L.bnn_mi355x_inference_device(imgs.data_ptr(), batch, 10, cls.data_ptr(), None, None, s)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(50):
            L.bnn_mi355x_inference_device(imgs.data_ptr(), batch, 10, cls.data_ptr(), None, None, s)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 50)
    print("%s batch %6d: %8.1f us  %7.1f Mimg/s  checksum %d" % (net, batch, best * 1e6, batch / best / 1e6, int(cls.sum().item())), flush=True)
