#!/usr/bin/env python3
"""Classify one CIFAR-10-format file on all GPUs of a node: contiguous shards, one RCCL broadcast of the
packed parameters, no other communication until the class indices are gathered.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        examples/multigpu_classify.py test_batch.bin
    (one-GPU rehearsal: add --gloo --nproc-per-node 2; every rank then computes on cuda:0)
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bnn-pynq_amd"))
from bnn import abi, multigpu  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--network", default="cnvW1A1")
    ap.add_argument("--params", default="cifar10")
    ap.add_argument("--gloo", action="store_true")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = 0 if a.gloo else int(os.environ.get("LOCAL_RANK", 0)) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("gloo" if a.gloo else "nccl", rank=rank, world_size=world)
    lib = abi.load(a.network)
    lib.bnn_mi355x_set_device(local)
    pdir = os.path.join(abi.PARAM_ROOT, a.params, a.network)
    multigpu.distribute_params(lib, pdir, device=None if a.gloo else dev)
    ncls = len(open(os.path.join(pdir, "classes.txt")).read().split("\n"))
    rec = lib.bnn_mi355x_image_bytes() + 1                       # CIFAR-10 record: label byte + image
    n = os.path.getsize(a.path) // rec
    lo, hi = multigpu.shard_bounds(n, world)[rank]
    shard = np.fromfile(a.path, np.uint8, count=(hi - lo) * rec, offset=lo * rec).reshape(hi - lo, rec)[:, 1:]
    shard = np.ascontiguousarray(shard)                          # keep it referenced while the library reads it
    usec = C.c_float(0)
    p = lib.bnn_mi355x_inference_buffer(shard.ctypes.data, hi - lo, ncls, C.byref(usec), 0)
    if not p:
        sys.exit(lib.bnn_mi355x_last_error().decode())
    mine = torch.from_numpy(np.ctypeslib.as_array(p, shape=(max(hi - lo, 1),))[: hi - lo].astype(np.int32))
    lib.free_results(p)
    full = multigpu.gather_classes(mine if a.gloo else mine.to(dev), n)
    if rank == 0:
        print("%d images on %d ranks, %.3f us/image on rank 0; class histogram %s" % (
            n, world, usec.value, np.bincount(full.cpu().numpy(), minlength=ncls).tolist()))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
