"""Test / benchmark helpers on top of the product's own ctypes binding (bnn/abi.py).
Everything here goes through the C ABI; nothing touches oracle/."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bnn-pynq_amd"))
from bnn import abi  # noqa: E402

LEGACY, EXT = abi.LEGACY, abi.EXT
LIB_DIR, PARAM_ROOT = abi.LIB_DIR, abi.PARAM_ROOT
lib_path, load = abi.lib_path, abi.load


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
