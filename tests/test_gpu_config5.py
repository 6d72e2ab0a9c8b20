"""BASELINE config 5's workload on the one GPU a test box has: CNV-W1A1 on 1 048 576 synthetic 32x32x3 images,
generated on the device in shards of 131 072 (seeded per shard: SURVEY 8(d) C5), cut into the 8 contiguous rank
shards of bnn.multigpu.shard_bounds(1048576, 8) and walked through the device entry point one rank shard at a
time -- what rank g of the 8-GPU job runs on its own GPU.  (The 8-GPU split itself needs 8 GPUs: the driver's
SCALE run; bench.py's N-rank line validates itself.)"""
import ctypes as C

import numpy as np
import pytest

import gpu_lib as gl
import oracle_lib as ol
from bnn import multigpu as mg

pytestmark = pytest.mark.gpu

N, GEN, RANKS = 1048576, 131072, 8


def test_million_images_in_eight_rank_shards():
    import torch
    net = gl.Net("cnvW1A1", "cifar10")
    L = net.L
    o = ol.Oracle("cnvW1A1", ol.param_dir("cifar10", "cnvW1A1"))
    imgs = torch.empty((N, 3072), dtype=torch.uint8, device="cuda")           # 3.2 GB of the 288
    for s in range(N // GEN):
        g = torch.Generator(device="cuda")
        g.manual_seed(2 * 1000 + s)                                            # C5: seed 2, one stream per generation shard
        imgs[s * GEN:(s + 1) * GEN] = torch.randint(0, 256, (GEN, 3072), dtype=torch.uint8, device="cuda", generator=g)
    bounds = mg.shard_bounds(N, RANKS)
    assert bounds[0] == (0, 131072) and bounds[-1] == (N - 131072, N)
    assert L.bnn_mi355x_reserve(131072) == 0
    cls = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    for lo, hi in bounds:                                                      # rank g's call, with rank g's pointers
        assert L.bnn_mi355x_inference_device(imgs[lo:].data_ptr(), hi - lo, 10, cls[lo:].data_ptr(), None, None, st) == 0, \
            L.bnn_mi355x_last_error()
    torch.cuda.synchronize()
    got = cls.cpu().numpy()
    assert got.min() >= 0 and got.max() <= 9
    # a seeded sample of every rank shard + the images either side of every shard boundary, against the oracle
    rng = np.random.default_rng(5)
    pick = []
    for lo, hi in bounds:
        pick += list(lo + rng.choice(hi - lo, 96, replace=False)) + [lo, lo + 1, lo + 2, hi - 3, hi - 2, hi - 1]
    pick = np.array(sorted(set(pick)))
    host = imgs[torch.from_numpy(pick).cuda()].cpu().numpy()
    assert (got[pick] == o.classes_batched(host, 10)).all()
    # size-independent property: the whole million in ONE call (the library walks it in passes of 131 072, not at the
    # rank boundaries' offsets in general) gives the same 1 048 576 classes as the eight rank calls
    whole = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    assert L.bnn_mi355x_inference_device(imgs.data_ptr(), N, 10, whole.data_ptr(), None, None, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(whole, cls)
    # ... and a split that is NOT aligned with anything (7 ranks: shards of 149 796 / 149 797 images)
    odd = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    for lo, hi in mg.shard_bounds(N, 7):
        assert L.bnn_mi355x_inference_device(imgs[lo:].data_ptr(), hi - lo, 10, odd[lo:].data_ptr(), None, None, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(odd, cls)
    # the same images from HOST memory through inference_buffer (the PCIe-inclusive entry point), two rank shards
    for lo, hi in (bounds[0], bounds[5]):
        h = imgs[lo:hi].cpu().numpy()
        p = L.bnn_mi355x_inference_buffer(h.ctypes.data, hi - lo, 10, None, 0)
        assert p, L.bnn_mi355x_last_error()
        res = np.ctypeslib.as_array(p, (hi - lo,)).copy()
        L.free_results(p)
        assert (res == got[lo:hi]).all()
    # checksum of checksums: the class histogram of the million equals the sum of the shards' histograms
    hist = np.bincount(got, minlength=10)
    assert hist.sum() == N and (sum(np.bincount(got[lo:hi], minlength=10) for lo, hi in bounds) == hist).all()
