"""ctypes binding of the PRODUCT library (include/bnn_mi355x.h) for the tests
and bench.py.  Everything here goes through the C ABI; nothing touches oracle/."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# BNN_MI355X_LIBDIR: kernel-tuning experiments point this at an alternative build of the same ABI
LIB_DIR = os.environ.get("BNN_MI355X_LIBDIR") or os.path.join(ROOT, "bnn-pynq_amd", "bnn", "libraries", "mi355x")
PARAM_ROOT = os.path.join(ROOT, "bnn-pynq_amd", "bnn", "params")

LEGACY = ["load_parameters", "inference", "inference_multiple", "inference_multiple_with_faults",
          "free_results", "deinit"]
EXT = ["bnn_mi355x_network", "bnn_mi355x_image_bytes", "bnn_mi355x_last_error", "bnn_mi355x_set_device",
       "bnn_mi355x_pack_params", "bnn_mi355x_export_params", "bnn_mi355x_import_params",
       "bnn_mi355x_inference_buffer", "bnn_mi355x_inference_raw", "bnn_mi355x_inference_device",
       "bnn_mi355x_reserve", "bnn_mi355x_set_fault_seed", "bnn_mi355x_last_faults", "bnn_mi355x_plan_faults",
       "bnn_mi355x_pack_params_faulty", "bnn_mi355x_debug_stage_output", "bnn_mi355x_profile", "bnn_mi355x_profile_read", "bnn_mi355x_stage_name"]


def lib_path(network, runtime="python_sw"):
    return os.path.join(LIB_DIR, "%s-%s-mi355x.so" % (runtime, network))


_cache = {}


def load(network, runtime="python_sw"):
    key = (network, runtime)
    if key in _cache:
        return _cache[key]
    path = lib_path(network, runtime)
    if not os.path.exists(path):
        raise RuntimeError("product library missing: %s (run `make -C bnn-pynq_amd`)" % path)
    L = C.CDLL(path)
    ip, fp = C.POINTER(C.c_int), C.POINTER(C.c_float)
    L.load_parameters.argtypes = [C.c_char_p]
    L.load_parameters.restype = None
    L.inference.argtypes = [C.c_char_p, ip, C.c_int, fp]
    L.inference_multiple.argtypes = [C.c_char_p, C.c_int, ip, fp, C.c_int]
    L.inference_multiple.restype = ip
    L.inference_multiple_with_faults.argtypes = [C.c_char_p, C.c_int, ip, fp, C.c_uint, C.c_int, C.c_int, ip, C.c_uint]
    L.inference_multiple_with_faults.restype = ip
    L.free_results.argtypes = [ip]
    L.free_results.restype = None
    L.deinit.restype = None
    L.bnn_mi355x_network.restype = C.c_char_p
    L.bnn_mi355x_last_error.restype = C.c_char_p
    L.bnn_mi355x_set_device.argtypes = [C.c_int]
    L.bnn_mi355x_pack_params.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
    L.bnn_mi355x_pack_params.restype = C.c_size_t
    L.bnn_mi355x_export_params.argtypes = [C.c_void_p, C.c_size_t]
    L.bnn_mi355x_export_params.restype = C.c_size_t
    L.bnn_mi355x_import_params.argtypes = [C.c_void_p, C.c_size_t]
    L.bnn_mi355x_inference_buffer.argtypes = [C.c_void_p, C.c_int, C.c_int, fp, C.c_int]
    L.bnn_mi355x_inference_buffer.restype = ip
    L.bnn_mi355x_inference_raw.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, fp]
    L.bnn_mi355x_inference_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p]
    L.bnn_mi355x_reserve.argtypes = [C.c_int]
    L.bnn_mi355x_set_fault_seed.argtypes = [C.c_ulonglong]
    L.bnn_mi355x_last_faults.argtypes = [ip, C.c_int]
    L.bnn_mi355x_plan_faults.argtypes = [C.c_ulonglong, C.c_int, C.c_uint, C.c_int, C.c_int, ip, C.c_uint, ip, C.c_int]
    L.bnn_mi355x_pack_params_faulty.argtypes = [C.c_char_p, ip, C.c_int, C.c_void_p, C.c_size_t]
    L.bnn_mi355x_pack_params_faulty.restype = C.c_size_t
    L.bnn_mi355x_debug_stage_output.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.bnn_mi355x_debug_stage_output.restype = C.c_long
    L.bnn_mi355x_profile.argtypes = [C.c_int]
    L.bnn_mi355x_profile_read.argtypes = [fp, C.c_int, ip]
    L.bnn_mi355x_stage_name.argtypes = [C.c_int]
    L.bnn_mi355x_stage_name.restype = C.c_char_p
    _cache[key] = L
    return L


def param_dir(dataset, network):
    return os.path.join(PARAM_ROOT, dataset, network)


def pack_params(network, pdir):
    """host-only: param directory -> blob bytes (no GPU touched)"""
    L = load(network)
    n = L.bnn_mi355x_pack_params(pdir.encode(), None, 0)
    if n == 0:
        raise RuntimeError(L.bnn_mi355x_last_error().decode())
    buf = np.zeros(n, np.uint8)
    assert L.bnn_mi355x_pack_params(pdir.encode(), buf.ctypes.data, n) == n
    return buf


class Net:
    """a loaded network of the product library (GPU)"""

    def __init__(self, network, dataset):
        self.L = load(network)
        self.network = network
        self.is_cnv = network.startswith("cnv")
        self.isz = self.L.bnn_mi355x_image_bytes()
        self.L.load_parameters(param_dir(dataset, network).encode())
        err = self.L.bnn_mi355x_last_error().decode()
        if err:
            raise RuntimeError(err)

    def raw(self, imgs):
        """CNV: int16 scores [n,64]; LFC: uint64 words [n]"""
        imgs = np.ascontiguousarray(imgs, np.uint8).reshape(-1, self.isz)
        n = imgs.shape[0]
        usec = C.c_float(0)
        if self.is_cnv:
            out = np.zeros((n, 64), np.int16)
            rc = self.L.bnn_mi355x_inference_raw(imgs.ctypes.data, n, out.ctypes.data, None, C.byref(usec))
        else:
            out = np.zeros(n, np.uint64)
            rc = self.L.bnn_mi355x_inference_raw(imgs.ctypes.data, n, None, out.ctypes.data, C.byref(usec))
        if rc != 0:
            raise RuntimeError(self.L.bnn_mi355x_last_error().decode())
        self.usec = usec.value
        return out

    def classify(self, imgs, ncls, detail=False):
        imgs = np.ascontiguousarray(imgs, np.uint8).reshape(-1, self.isz)
        n = imgs.shape[0]
        usec = C.c_float(0)
        p = self.L.bnn_mi355x_inference_buffer(imgs.ctypes.data, n, ncls, C.byref(usec), 1 if detail else 0)
        if not p:
            raise RuntimeError(self.L.bnn_mi355x_last_error().decode())
        cnt = n * (ncls if (detail and self.is_cnv) else 1)
        out = np.ctypeslib.as_array(p, shape=(max(cnt, 1),))[:cnt].astype(np.int32, copy=True)
        self.L.free_results(p)
        self.usec = usec.value
        return out
