"""ctypes binding of oracle/_build/libbnn_oracle.so (the CPU restatement).

Test infrastructure: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only.  The product package never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
BUILD_DIR = os.path.join(ORACLE_DIR, "_build")
PARAM_ROOT = os.path.join(ROOT, "bnn-pynq_amd", "bnn", "params")
GOLDEN = os.path.join(ROOT, "tests", "golden")

_lib = None


def build():
    """(re)build the oracle with make; cheap when up to date."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(BUILD_DIR, "libbnn_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp, u8p, i8p, i16p, u64p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_int8), \
        C.POINTER(C.c_int16), C.POINTER(C.c_uint64)
    L.bnn_oracle_create.restype = vp
    L.bnn_oracle_create.argtypes = [C.c_char_p, C.c_char_p]
    L.bnn_oracle_destroy.argtypes = [vp]
    L.bnn_oracle_is_cnv.argtypes = [vp]
    L.bnn_oracle_num_layers.argtypes = [vp]
    L.bnn_oracle_cnv_scores_ref.argtypes = [vp, u8p, i16p]
    L.bnn_oracle_lfc_word_ref.restype = C.c_uint64
    L.bnn_oracle_lfc_word_ref.argtypes = [vp, u8p]
    L.bnn_oracle_layer_ref.argtypes = [vp, u8p, C.c_int, i8p, C.c_int]
    L.bnn_oracle_cnv_scores_fast.argtypes = [vp, u8p, C.c_int, i16p, C.c_int]
    L.bnn_oracle_lfc_words_fast.argtypes = [vp, u8p, C.c_int, u64p, C.c_int]
    L.bnn_oracle_decode_cnv_batched.argtypes = [i16p, C.c_int]
    L.bnn_oracle_decode_cnv_single.argtypes = [i16p, C.c_int]
    L.bnn_oracle_decode_lfc_batched.argtypes = [C.c_uint64, C.c_int]
    L.bnn_oracle_decode_lfc_single.argtypes = [C.c_uint64, C.c_int]
    L.bnn_oracle_lfc_single_hot.argtypes = [C.c_uint64, C.c_int]
    L.bnn_oracle_quantise_u8.argtypes = [C.c_int]
    L.bnn_oracle_parse_cifar10.argtypes = [C.c_char_p, C.POINTER(u8p)]
    L.bnn_oracle_parse_mnist.argtypes = [C.c_char_p, C.POINTER(u8p)]
    L.bnn_oracle_free.argtypes = [vp]
    for f in ("weight", "threshold"):
        getattr(L, "bnn_oracle_" + f).argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.bnn_oracle_layer_mw.argtypes = [vp, C.c_int]
    L.bnn_oracle_apply_fault.argtypes = [vp] + [C.c_int] * 7
    L.bnn_oracle_layer_mh.argtypes = [vp, C.c_int]
    L.bnn_oracle_lfc_binarize.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)]
    L.bnn_oracle_lfc_binarize.restype = None
    _lib = L
    return L


def param_dir(dataset, network):
    return os.path.join(PARAM_ROOT, dataset, network)


def num_classes(dataset, network):
    with open(os.path.join(param_dir(dataset, network), "classes.txt")) as f:
        return len([c.strip() for c in f.readlines()])


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class Oracle:
    """One loaded network of the CPU restatement."""

    def __init__(self, network, pdir):
        self.L = lib()
        self.network = network
        self.h = self.L.bnn_oracle_create(network.encode(), pdir.encode())
        if not self.h:
            raise RuntimeError("oracle: cannot load %s from %s" % (network, pdir))
        self.is_cnv = bool(self.L.bnn_oracle_is_cnv(self.h))
        self.nl = self.L.bnn_oracle_num_layers(self.h)
        self.isz = 3072 if self.is_cnv else 784

    def close(self):
        if self.h:
            self.L.bnn_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- faithful -----------------------------------------------------------
    def scores_ref(self, img):
        img = np.ascontiguousarray(img, np.uint8).reshape(-1)
        assert img.size == 3072
        s = np.zeros(64, np.int16)
        self.L.bnn_oracle_cnv_scores_ref(self.h, _u8(img), s.ctypes.data_as(C.POINTER(C.c_int16)))
        return s

    def word_ref(self, px):
        px = np.ascontiguousarray(px, np.uint8).reshape(-1)
        assert px.size == 784
        return int(self.L.bnn_oracle_lfc_word_ref(self.h, _u8(px)))

    def layer_ref(self, img, layer):
        img = np.ascontiguousarray(img, np.uint8).reshape(-1)
        assert img.size == self.isz
        out = np.zeros(30 * 30 * 64, np.int8)
        n = self.L.bnn_oracle_layer_ref(self.h, _u8(img), layer,
                                        out.ctypes.data_as(C.POINTER(C.c_int8)), out.size)
        assert n >= 0
        return out[:n].copy()

    def binarize(self, imgs):
        """binarizeAndPack of every image: uint64 [n, 13]"""
        imgs = np.ascontiguousarray(imgs, np.uint8).reshape(-1, 784)
        out = np.zeros((imgs.shape[0], 13), np.uint64)
        for i in range(imgs.shape[0]):
            self.L.bnn_oracle_lfc_binarize(_u8(imgs[i]), out[i].ctypes.data_as(C.POINTER(C.c_uint64)))
        return out

    # -- fast ---------------------------------------------------------------
    def scores_fast(self, imgs, nthreads=0):
        imgs = np.ascontiguousarray(imgs, np.uint8).reshape(-1, 3072)
        s = np.zeros((imgs.shape[0], 64), np.int16)
        if imgs.shape[0]:
            self.L.bnn_oracle_cnv_scores_fast(self.h, _u8(imgs), imgs.shape[0],
                                              s.ctypes.data_as(C.POINTER(C.c_int16)), nthreads)
        return s

    def words_fast(self, imgs, nthreads=0):
        imgs = np.ascontiguousarray(imgs, np.uint8).reshape(-1, 784)
        w = np.zeros(imgs.shape[0], np.uint64)
        if imgs.shape[0]:
            self.L.bnn_oracle_lfc_words_fast(self.h, _u8(imgs), imgs.shape[0],
                                             w.ctypes.data_as(C.POINTER(C.c_uint64)), nthreads)
        return w

    # -- batched classes the way the ABI returns them ------------------------
    def classes_batched(self, imgs, ncls, nthreads=0):
        L = self.L
        if self.is_cnv:
            s = self.scores_fast(imgs, nthreads)
            return np.array([L.bnn_oracle_decode_cnv_batched(
                s[i].ctypes.data_as(C.POINTER(C.c_int16)), ncls) for i in range(len(s))], np.int32)
        w = self.words_fast(imgs, nthreads)
        return np.array([L.bnn_oracle_decode_lfc_batched(int(x), ncls) for x in w], np.int32)

    def apply_fault(self, rec):
        """rec = (image, target, layer, mem, ind, thresh, bit, word_size); returns the row changed"""
        return self.L.bnn_oracle_apply_fault(self.h, *[int(x) for x in rec[1:8]])

    def weights(self, layer):
        mh, mw = self.L.bnn_oracle_layer_mh(self.h, layer), self.L.bnn_oracle_layer_mw(self.h, layer)
        W = np.zeros((mh, mw), np.int8)
        for n in range(mh):
            for j in range(mw):
                W[n, j] = self.L.bnn_oracle_weight(self.h, layer, n, j)
        return W


def decode_cnv_batched(scores, ncls):
    s = np.ascontiguousarray(scores, np.int16)
    return lib().bnn_oracle_decode_cnv_batched(s.ctypes.data_as(C.POINTER(C.c_int16)), ncls)


def decode_cnv_single(scores, ncls):
    s = np.ascontiguousarray(scores, np.int16)
    return lib().bnn_oracle_decode_cnv_single(s.ctypes.data_as(C.POINTER(C.c_int16)), ncls)


def read_cifar(path):
    d = np.fromfile(path, np.uint8)
    n = d.size // 3073
    return d[: n * 3073].reshape(n, 3073)[:, 1:].copy()


def read_mnist(path):
    d = np.fromfile(path, np.uint8)
    n = int.from_bytes(d[4:8].tobytes(), "big")
    return d[16:16 + n * 784].reshape(n, 784).copy()
