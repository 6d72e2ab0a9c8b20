import sys, ctypes as C
import torch
sys.path.insert(0, "tests")
import gpu_lib as gl
L = gl.load("cnvW1A1"); L.load_parameters(gl.param_dir("cifar10", "cnvW1A1").encode())
for n in (1, 10, 64):
    imgs = torch.randint(0, 256, (n, 3072), dtype=torch.uint8, device="cuda")
    cls = torch.zeros(n, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(5): L.bnn_mi355x_inference_device(imgs.data_ptr(), n, 10, cls.data_ptr(), None, None, s)
    torch.cuda.synchronize()
    L.bnn_mi355x_profile(1)
    R = 50
    for _ in range(R): L.bnn_mi355x_inference_device(imgs.data_ptr(), n, 10, cls.data_ptr(), None, None, s)
    torch.cuda.synchronize()
    ms = (C.c_float * 16)(); nc = C.c_int(0)
    k = L.bnn_mi355x_profile_read(ms, 16, C.byref(nc)); L.bnn_mi355x_profile(0)
    print(n, " ".join("%s=%.1f" % (L.bnn_mi355x_stage_name(i).decode().split()[-1], ms[i] / R * 1e3) for i in range(k)), "sum=%.1f us" % (sum(ms[:k]) / R * 1e3))
