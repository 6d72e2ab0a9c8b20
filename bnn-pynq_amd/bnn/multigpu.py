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


def broadcast_blob(lib, blob, src=0, device=None, group=None):
    """ONE collective: rank `src` passes the packed blob (numpy uint8), the others None; every rank returns
    the received bytes as a uint8 torch tensor on `device` (CPU when None).  The size needs no exchange: it
    is a function of the network alone (``bnn_mi355x_params_bytes``)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    size = int(lib.bnn_mi355x_params_bytes())
    if rank == src:
        if blob.size != size:
            raise RuntimeError("packed blob has %d bytes, this network's layout has %d" % (blob.size, size))
        buf = torch.from_numpy(np.ascontiguousarray(blob)).to(dev)
    else:
        buf = torch.empty(size, dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src, group=group)
    return buf


def distribute_params(lib, param_dir, device=None, group=None, upload=True):
    """load_parameters() for a multi-GPU job: rank 0 reads and repacks the reference's param files, every
    rank receives the blob in one broadcast over the process group and (upload=True) hands it to its own
    library instance -- straight from HBM when the group's tensors live there (RCCL), from host memory
    otherwise (gloo).  Returns the blob as a numpy array (host copy; the GPU path makes it on demand)."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    blob = pack_params(lib, param_dir) if rank == 0 else None
    buf = broadcast_blob(lib, blob, 0, device, group)
    if upload:
        import_blob(lib, buf)
    return buf.cpu().numpy()


def import_blob(lib, buf):
    """hand a received blob (uint8 torch tensor, on the GPU or on the host) to this rank's library instance"""
    import torch
    if buf.is_cuda:
        stream = torch.cuda.current_stream(buf.device).cuda_stream
        rc = lib.bnn_mi355x_import_params_device(buf.data_ptr(), buf.numel(), stream)
    else:
        host = buf.numpy()
        rc = lib.bnn_mi355x_import_params(host.ctypes.data, host.size)
    if rc != 0:
        raise RuntimeError(lib.bnn_mi355x_last_error().decode())


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
