"""CPU, world_size 2, gloo: the N>1 path of bnn/multigpu.py -- the packed-parameter broadcast
(rank 0 packs with the product's host-only packer, rank 1 receives byte-identical data), contiguous
sharding, and result gathering.  The per-shard compute is stood in for by the oracle here (no GPU
in this container); on the GPU box the same helpers drive the HIP library (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_images, ret):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "bnn-pynq_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gpu_lib as gl
        import oracle_lib as ol
        from bnn import multigpu as mg
        lib = gl.load("lfcW1A1")
        pdir = gl.param_dir("mnist", "lfcW1A1")
        calls = []
        real_broadcast = dist.broadcast
        dist.broadcast = lambda *a, **k: (calls.append(a[0].numel()), real_broadcast(*a, **k))[1]
        blob = mg.distribute_params(lib, pdir, upload=False)          # the one collective of the job
        dist.broadcast = real_broadcast
        assert calls == [lib.bnn_mi355x_params_bytes()]               # ONE broadcast, of exactly the blob (no size exchange)
        assert (blob == gl.pack_params("lfcW1A1", pdir)).all()        # every rank holds rank 0's bytes
        imgs = np.random.default_rng(0).integers(0, 256, (n_images, 784), dtype=np.uint8)
        lo, hi = mg.shard_bounds(n_images, world)[rank]
        o = ol.Oracle("lfcW1A1", ol.param_dir("mnist", "lfcW1A1"))
        local = torch.from_numpy(o.classes_batched(imgs[lo:hi], 10))
        full = mg.gather_classes(local, n_images)
        want = o.classes_batched(imgs, 10)
        ret[rank] = bool((full.numpy() == want).all()) and (hi - lo) > 0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [64, 101])
def test_two_rank_broadcast_shard_gather(n_images):
    world, port = 2, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, port, n_images, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_shard_bounds_partition():
    sys.path.insert(0, os.path.join(ROOT, "bnn-pynq_amd"))
    from bnn import multigpu as mg
    for n in (0, 1, 7, 8, 1000, 1048576):
        for w in (1, 2, 4, 8):
            b = mg.shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
