#!/usr/bin/env python3
"""Random batch sizes around every dispatch limit of kernels.hip, all five networks, raw outputs and classes
against the CPU restatement (test infrastructure; not part of the pytest suite because of its run time)."""
import os, sys
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_lib as gl
import oracle_lib as ol

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
edges = {"cnv": [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 1500, 2049],
         "lfc": [1, 2, 63, 65, 255, 257, 511, 513, 1023, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 6000, 8191, 12289, 20001,
                 32767, 32768, 32769, 33000]}
for net, ds in (("cnvW1A1", "cifar10"), ("cnvW1A2", "cifar10"), ("cnvW2A2", "cifar10"), ("lfcW1A1", "mnist"), ("lfcW1A2", "mnist")):
    N = gl.Net(net, ds)
    o = ol.Oracle(net, ol.param_dir(ds, net))
    kind = net[:3]
    sizes = edges[kind] + [int(x) for x in rng.integers(1, 1600 if kind == "cnv" else 5000, 12)]
    if kind == "lfc":  # the one-launch block kernel's range (lfcW1A1: 4 097 .. 32 768 images)
        sizes += [int(x) for x in rng.integers(4097, 34000, 10)]
    for n in sizes:
        imgs = rng.integers(0, 256, (n, N.isz), dtype=np.uint8)
        if rng.random() < 0.3:
            imgs = np.where(rng.random(imgs.shape) < 0.1, 255, 0).astype(np.uint8)
        raw = N.raw(imgs)
        ref = o.scores_fast(imgs) if o.is_cnv else o.words_fast(imgs)
        assert (raw == ref).all(), (net, n)
        assert (N.classify(imgs, 10) == o.classes_batched(imgs, 10)).all(), (net, n, "classes")
        d = torch.from_numpy(imgs).cuda(); cls = torch.zeros(n, dtype=torch.int32, device="cuda")
        assert N.L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, cls.data_ptr(), None, None, None) == 0
        torch.cuda.synchronize()
        assert (cls.cpu().numpy() == o.classes_batched(imgs, 10)).all(), (net, n, "device classes")
    print(net, "ok:", len(sizes), "sizes", flush=True)
print("fuzz ok")
