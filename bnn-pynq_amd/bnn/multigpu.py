"""Batch data-parallelism over the GPUs of one node (not in the reference: a PYNQ board has one
accelerator).  One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo"
on CPU for tests).  Images are independent and the weights read-only, so the ONLY collective is
the one-time broadcast of the packed parameter blob; shards are contiguous ranges of the batch and
no per-layer communication exists.

    dist.init_process_group("nccl"); torch.cuda.set_device(local_rank)
    lib = bnn.abi.load("cnvW1A1")          # ctypes handle with the ABI declared
    lib.bnn_mi355x_set_device(local_rank)
    distribute_params(lib, "/path/to/params/cifar10/cnvW1A1")   # rank 0 reads the files
    lo, hi = shard_bounds(n_images, world)[rank]
"""
import numpy as np


def shard_bounds(n_items, world_size):
    """contiguous, near-equal shards [lo, hi) of a batch; the first n % world get one more"""
    base, extra = divmod(int(n_items), int(world_size))
    out, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def pack_params(lib, param_dir):
    """param directory -> packed blob (numpy uint8).  Host only: touches no GPU."""
    n = lib.bnn_mi355x_pack_params(param_dir.encode(), None, 0)
    if n == 0:
        raise RuntimeError(lib.bnn_mi355x_last_error().decode())
    blob = np.zeros(n, np.uint8)
    if lib.bnn_mi355x_pack_params(param_dir.encode(), blob.ctypes.data, n) != n:
        raise RuntimeError(lib.bnn_mi355x_last_error().decode())
    return blob


def broadcast_blob(blob, src=0, device=None, group=None):
    """rank `src` passes the blob, the others None; everyone returns the same numpy uint8 array.
    Two broadcasts (size, bytes) over whatever backend the group uses."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    size = torch.tensor([blob.size if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(size, src, group=group)
    if rank == src:
        buf = torch.from_numpy(np.ascontiguousarray(blob)).to(dev)
    else:
        buf = torch.empty(int(size.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src, group=group)
    return buf.cpu().numpy()


def distribute_params(lib, param_dir, device=None, group=None, upload=True):
    """load_parameters() for a multi-GPU job: rank 0 reads and repacks the reference's param files,
    every rank receives the blob over the process group and (upload=True) hands it to its own
    library instance / GPU.  Returns the blob."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    blob = pack_params(lib, param_dir) if rank == 0 else None
    blob = broadcast_blob(blob, 0, device, group)
    if upload:
        if lib.bnn_mi355x_import_params(blob.ctypes.data, blob.size) != 0:
            raise RuntimeError(lib.bnn_mi355x_last_error().decode())
    return blob


def gather_classes(local_classes, n_total, group=None):
    """per-rank int32 class vectors (shards in rank order) -> the full vector on every rank"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    bounds = shard_bounds(n_total, world)
    longest = max(hi - lo for lo, hi in bounds)
    pad = torch.zeros(longest, dtype=torch.int32, device=local_classes.device)
    pad[: local_classes.numel()] = local_classes.to(torch.int32)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][: hi - lo] for r, (lo, hi) in enumerate(bounds)])
