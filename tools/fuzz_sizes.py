#!/usr/bin/env python3
"""tools/fuzz_sizes.py [SEED] [sizes|files|lanes]: random batch sizes around every dispatch limit of kernels.hip, all five
networks, raw outputs and classes against the CPU restatement; files either side of the pinned ring's threshold; host
buffers and files on two compute lanes (test infrastructure; not part of the pytest suite because of its run time)."""
import os, sys
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_lib as gl
import oracle_lib as ol

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
only = sys.argv[2] if len(sys.argv) > 2 else "all"
edges = {"cnv": [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 1500, 2049,
                 8185, 8191, 8192, 8193, 8199, 9001,       # layer 0: lane per pixel below 8 192 images, blocks of 8 from there
                 16383, 16384, 16385, 16641],             # the device call forks over two lanes from 16 384 images
         "lfc": [1, 2, 63, 65, 255, 257, 511, 513, 1023, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 6000, 8191, 12289, 20001,
                 32767, 32768, 32769, 33000]}
nets_filter = os.environ.get("NETS")  # e.g. NETS=lfcW1A1,lfcW1A2: the "sizes" section for those networks only
for net, ds in (("cnvW1A1", "cifar10"), ("cnvW1A2", "cifar10"), ("cnvW2A2", "cifar10"), ("lfcW1A1", "mnist"), ("lfcW1A2", "mnist")) if only in ("all", "sizes") else ():
    if nets_filter and net not in nets_filter.split(","):
        continue
    N = gl.Net(net, ds)
    o = ol.Oracle(net, ol.param_dir(ds, net))
    kind = net[:3]
    sizes = edges[kind] + [int(x) for x in rng.integers(1, 1600 if kind == "cnv" else 5000, 12)]
    if kind == "lfc":  # the one-launch block kernel's range (lfcW1A1: 4 097 .. 131 072 images)
        sizes += [int(x) for x in rng.integers(4097, 34000, 10)] + [int(x) for x in rng.integers(34000, 131073, 4)] + [131071, 131072, 131073]
    else:              # the tile form of layer 0 with ragged last blocks
        sizes += [int(x) for x in rng.integers(8192, 20000, 4)]
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
# the file entry point either side of the size from which worker threads feed a ring of pinned pieces (24 MB), odd counts
import ctypes as C, tempfile
# (round 4: the ring from 2 MB = 682 records on, pieces of 85 / 170 records, readers' preadv batches of 512 records, a two-sided
# chunk plan below 32 768 images; the LFC nets direct up to 1 024 images, binarised by worker threads beyond, chunks of 8 192 ...)
extra_c = [1, 2, 85, 86, 170, 171, 511, 512, 513, 681, 682, 683, 1023, 1024, 1025, 1537, 2047, 4097] + [int(x) for x in rng.integers(600, 40000, 10)]
extra_l = [1, 2, 255, 256, 257, 1023, 1024, 1025, 2049, 8191, 8192, 8193, 16383, 16384, 16385] + [int(x) for x in rng.integers(900, 140000, 8)]
for net, ds, rec, nn in (("cnvW1A1", "cifar10", 3073, [7809, 7811, 8200, 9999, 12345, 33001, 70003] + extra_c), ("cnvW2A2", "cifar10", 3073, extra_c[-6:]),
                         ("lfcW1A1", "mnist", 784, [30000, 32101, 40001, 100003] + extra_l), ("lfcW1A2", "mnist", 784, extra_l[-10:])) if only in ("all", "files") else ():
    N = gl.Net(net, ds)
    o = ol.Oracle(net, ol.param_dir(ds, net))
    for n in nn:
        imgs = rng.integers(0, 256, (n, N.isz), dtype=np.uint8)
        with tempfile.NamedTemporaryFile(dir="/tmp", suffix=".bin") as f:
            if rec == 3073:
                r = np.empty((n, 3073), np.uint8); r[:, 0] = rng.integers(0, 256, n); r[:, 1:] = imgs; f.write(r.tobytes())
            else:
                f.write((0x803).to_bytes(4, "big") + n.to_bytes(4, "big") + (28).to_bytes(4, "big") * 2 + imgs.tobytes())
            f.flush()
            cnt = C.c_int(0)
            p = N.L.inference_multiple(f.name.encode(), 10, C.byref(cnt), None, 0)
            assert p and cnt.value == n, (net, n)
            got = np.ctypeslib.as_array(p, (n,)).copy()
            N.L.free_results(p)
        assert (got == o.classes_batched(imgs, 10)).all(), (net, n, "file classes")
    print(net, "files ok:", nn, flush=True)
# host buffers and files of random sizes on two compute lanes (three or more chunks), classes against the device entry point
# (one pass on the caller's stream), which the sections above have compared with the restatement
for net, ds, rec, lo, hi in (("cnvW1A1", "cifar10", 3073, 5000, 150000), ("cnvW1A2", "cifar10", 3073, 5000, 60000), ("lfcW1A1", "mnist", 784, 20000, 400000)) if only in ("all", "lanes") else ():
    N = gl.Net(net, ds)
    for n in [int(x) for x in rng.integers(lo, hi, 8)]:
        imgs = rng.integers(0, 256, (n, N.isz), dtype=np.uint8)
        d = torch.from_numpy(imgs).cuda(); cls = torch.zeros(n, dtype=torch.int32, device="cuda")
        assert N.L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, cls.data_ptr(), None, None, None) == 0
        torch.cuda.synchronize()
        want = cls.cpu().numpy()
        del d
        assert (N.classify(imgs, 10) == want).all(), (net, n, "buffer, two lanes")
        with tempfile.NamedTemporaryFile(dir="/tmp", suffix=".bin") as f:
            if rec == 3073:
                r = np.empty((n, 3073), np.uint8); r[:, 0] = 7; r[:, 1:] = imgs; f.write(r.tobytes()); del r
            else:
                f.write((0x803).to_bytes(4, "big") + n.to_bytes(4, "big") + (28).to_bytes(4, "big") * 2 + imgs.tobytes())
            f.flush()
            cnt = C.c_int(0)
            p = N.L.inference_multiple(f.name.encode(), 10, C.byref(cnt), None, 0)
            assert p and cnt.value == n, (net, n)
            got = np.ctypeslib.as_array(p, (n,)).copy()
            N.L.free_results(p)
        assert (got == want).all(), (net, n, "file, two lanes")
        print(net, n, "two lanes ok", flush=True)
print("fuzz ok")
