#!/usr/bin/env python3
"""The multi-GPU helpers over the real RCCL backend with a world of one rank (run from the repository root)."""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, "tests"); sys.path.insert(0, "bnn-pynq_amd")
import gpu_lib as gl
from bnn import multigpu as mg
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
L = gl.load("cnvW1A1"); L.bnn_mi355x_set_device(0)
blob = mg.distribute_params(L, gl.param_dir("cifar10", "cnvW1A1"), device=dev)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
imgs = torch.randint(0, 256, (1000, 3072), dtype=torch.uint8, device=dev); cls = torch.zeros(1000, dtype=torch.int32, device=dev)
assert L.bnn_mi355x_inference_device(imgs.data_ptr(), 1000, 10, cls.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream) == 0
full = mg.gather_classes(cls, 1000)
torch.cuda.synchronize()
print("nccl world=1 ok: blob", blob.size, "bytes, max", float(t.item()), "classes", int(full.sum().item()))
dist.destroy_process_group()
