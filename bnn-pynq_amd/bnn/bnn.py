"""Python classifier API of the MI355X-native BNN-PYNQ runtime.

Mirrors the public surface of the reference's ``bnn/bnn.py`` (names, argument
meaning, printed messages, return types) so that code and notebooks written
against ``bnn.LfcClassifier`` / ``bnn.CnvClassifier`` run unchanged:

  reference                          here
  ---------------------------------  -------------------------------------------
  bnn.py:37-53   RUNTIME_*/NETWORK_*  same constants
  bnn.py:55-65   PLATFORM / *_DIR     PLATFORM = "mi355x" (no $BOARD needed)
  bnn.py:67-79   cffi cdef + dlopen   ctypes binding of the same six symbols (_CDEF)
  bnn.py:82-91   available_params     same
  bnn.py:94-207  PynqBNN              same methods; no bitstream download
  bnn.py:210-345 CnvClassifier        same 13 classify_* methods + image_to_cifar
  bnn.py:348-388 LfcClassifier        same

The shared object it loads, ``libraries/<PLATFORM>/<runtime>-<network>-
<PLATFORM>.so``, is the HIP runtime built from ``../csrc`` (C ABI:
``include/bnn_mi355x.h``).  There is no CPU implementation behind this module:
if the library is missing the constructor raises.

Additions (not in the reference): ``PynqBNN.inference_array`` /
``classify_array`` for images already in a numpy array, and ``bnn.multigpu``.
"""
import ctypes
import os
import tempfile

import numpy as np

from . import abi

try:  # the reference imports PIL unconditionally; keep the module importable without it
    from PIL import Image
except ImportError:  # pragma: no cover
    Image = None

RUNTIME_HW = "python_hw"
RUNTIME_SW = "python_sw"

NETWORK_CNVW1A1 = "cnvW1A1"
NETWORK_CNVW1A1_INTERLEAVED = "cnvW1A1-interleaved"
NETWORK_CNVW1A1_RESILIENT_INTERLEAVED = "cnvW1A1-resilient-interleaved"
NETWORK_CNVW1A1_TMR = "cnvW1A1-TMR"
NETWORK_CNVW1A2 = "cnvW1A2"
NETWORK_CNVW1A2_INTERLEAVED = "cnvW1A2-interleaved"
NETWORK_CNVW1A2_RESILIENT_INTERLEAVED = "cnvW1A2-resilient-interleaved"
NETWORK_CNVW2A2 = "cnvW2A2"
NETWORK_CNVW2A2_INTERLEAVED = "cnvW2A2-interleaved"
NETWORK_CNVW2A2_RESILIENT_INTERLEAVED = "cnvW2A2-resilient-interleaved"
NETWORK_CNVW2A2_TMR = "cnvW2A2-TMR"
NETWORK_LFCW1A1 = "lfcW1A1"
NETWORK_LFCW1A2 = "lfcW1A2"
NETWORK_LFCW1A2_INTERLEAVED = "lfcW1A2-interleaved"

# The reference derives PLATFORM from $BOARD (Ultra96 / Pynq-Z1 / Pynq-Z2); a GPU
# host has no such variable.  BNN_PLATFORM overrides the directory name.
PLATFORM = abi.PLATFORM

BNN_ROOT_DIR = os.path.dirname(os.path.realpath(__file__))
BNN_LIB_DIR = abi.LIB_DIR
BNN_BIT_DIR = os.path.join(BNN_ROOT_DIR, "bitstreams", PLATFORM)  # kept for name compatibility; unused
BNN_PARAM_DIR = os.path.join(BNN_ROOT_DIR, "params")

# the reference's cdef, verbatim signatures: the contract both sides bind to
_CDEF = """
void load_parameters(const char* path);
int inference(const char* path, int results[64], int number_class, float *usecPerImage);
int* inference_multiple(const char* path, int number_class, int *image_number, float *usecPerImage, int enable_detail);
int* inference_multiple_with_faults(const char* path, int number_class, int *image_number, float *usecPerImage, unsigned int flip_count, int word_size, int target, int* target_layers, unsigned int num_targets);
void free_results(int * result);
void deinit();
"""

_libraries = {}


def _open_library(dllname):
    """dlopen once per process and name, like the reference's _libraries cache"""
    if dllname not in _libraries:
        _libraries[dllname] = abi.load_path(os.path.join(BNN_LIB_DIR, dllname))
    return _libraries[dllname]


def _param_name(network, dataset_dir):
    """directory name holding `network`'s parameters: the hardened variants (-TMR, -interleaved,
    -resilient-interleaved) ship byte-identical copies of their base network's files in the
    reference; this package keeps one copy and resolves the variant to its base when absent"""
    names = os.listdir(dataset_dir)
    if network in names:
        return network
    base = network.split("-")[0]
    return base if base != network and base in names else None


def available_params(network):
    """datasets for which a parameter set of `network` is installed"""
    found = []
    for dataset in os.listdir(BNN_PARAM_DIR):
        dpath = os.path.join(BNN_PARAM_DIR, dataset)
        if os.path.isdir(dpath) and _param_name(network, dpath):
            found.append(dataset)
    return found


class PynqBNN:
    """Interface object onto the per-network shared library."""

    def __init__(self, runtime, network, load_overlay=True):
        # RUNTIME_HW meant "download the FPGA bitstream" in the reference; on
        # MI355X both runtime names resolve to the same HIP library.
        self.bitstream_name = None
        dllname = "{0}-{1}-{2}.so".format(runtime, network, PLATFORM)
        self.interface = _open_library(dllname)
        self.num_classes = 0
        self.classes = []
        self.usecPerImage = 0.0

    def __del__(self):
        try:
            self.interface.deinit()
        except Exception:
            pass

    def load_parameters(self, params):
        if not os.path.isabs(params):
            params = os.path.join(BNN_PARAM_DIR, params)
            if not os.path.isdir(params) and os.path.isdir(os.path.dirname(params)):
                name = _param_name(os.path.basename(params), os.path.dirname(params))
                if name:
                    params = os.path.join(os.path.dirname(params), name)
        if os.path.isdir(params):
            self.interface.load_parameters(params.encode())
            with open(os.path.join(params, "classes.txt")) as f:
                self.classes = [c.strip() for c in f.readlines()]
        else:
            print("\nERROR: No such parameter directory \"" + params + "\"")

    def _report_single(self, usec):
        print("Inference took %.2f microseconds" % usec)
        print("Classification rate: %.2f images per second" % (1000000.0 / usec))
        self.usecPerImage = usec

    def _report_multi(self, usec, count):
        print("Inference took %.2f microseconds, %.2f usec per image" % (usec * count, usec))
        print("Classification rate: %.2f images per second" % (1000000.0 / usec))
        self.usecPerImage = usec

    def inference(self, path):
        usec = ctypes.c_float(0)
        cls = self.interface.inference(path.encode(), None, len(self.classes), ctypes.byref(usec))
        if cls < 0:
            raise RuntimeError("inference failed: see stderr")
        self._report_single(usec.value)
        return cls

    def detailed_inference(self, path):
        n = len(self.classes)
        details = (ctypes.c_int * max(n, 64))()
        usec = ctypes.c_float(0)
        if self.interface.inference(path.encode(), details, n, ctypes.byref(usec)) < 0:
            raise RuntimeError("inference failed: see stderr")
        self._report_single(usec.value)
        return np.array(details[:n], dtype=np.int32)

    def _collect(self, ptr, count):
        if not ptr:
            raise RuntimeError("inference failed: see stderr")
        arr = np.ctypeslib.as_array(ptr, shape=(count,)).astype(np.int32, copy=True) if count else np.zeros(0, np.int32)
        self.interface.free_results(ptr)
        return arr

    def inference_multiple(self, path):
        size = ctypes.c_int(0)
        usec = ctypes.c_float(0)
        ptr = self.interface.inference_multiple(path.encode(), len(self.classes), ctypes.byref(size),
                                                ctypes.byref(usec), 0)
        result = self._collect(ptr, size.value)
        self._report_multi(usec.value, size.value)
        return result

    def inference_multiple_with_faults(self, path, num_faults, word_size, target_type, target_layers=[]):
        if len(target_layers) == 0:
            targets = None
        else:
            tl = np.array(target_layers, dtype=np.int32)
            targets = tl.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
        size = ctypes.c_int(0)
        usec = ctypes.c_float(0)
        ptr = self.interface.inference_multiple_with_faults(
            path.encode(), len(self.classes), ctypes.byref(size), ctypes.byref(usec), num_faults, word_size,
            target_type, targets, len(target_layers))
        result = self._collect(ptr, size.value)
        self._report_multi(usec.value, size.value)
        return result

    def inference_multiple_detail(self, path):
        size = ctypes.c_int(0)
        usec = ctypes.c_float(0)
        ptr = self.interface.inference_multiple(path.encode(), len(self.classes), ctypes.byref(size),
                                                ctypes.byref(usec), 1)
        result = self._collect(ptr, size.value * len(self.classes))
        self._report_multi(usec.value, size.value)
        return result

    # -- extension: images already in memory ----------------------------------
    def inference_array(self, images, detail=False):
        """images: uint8 array, n x 3072 (planar CHW) for cnv*, n x 784 for lfc*"""
        if not self.interface.has_extensions:
            raise RuntimeError("this runtime library has no in-memory entry point")
        isz = self.interface.bnn_mi355x_image_bytes()
        a = np.ascontiguousarray(images, dtype=np.uint8).reshape(-1, isz)
        usec = ctypes.c_float(0)
        ptr = self.interface.bnn_mi355x_inference_buffer(a.ctypes.data, a.shape[0], len(self.classes),
                                                         ctypes.byref(usec), 1 if detail else 0)
        count = a.shape[0] * (len(self.classes) if detail else 1)
        result = self._collect(ptr, count)
        self.usecPerImage = usec.value
        return result

    def class_name(self, index):
        return self.classes[index]


class CnvClassifier:
    """CNV networks on CIFAR-10 formatted (32x32x3) images."""

    def __init__(self, network, params, runtime=RUNTIME_HW):
        if params in available_params(network):
            self.net = network
            self.params = params
            self.runtime = runtime
            self.usecPerImage = 0.0
            self.bnn = PynqBNN(runtime, network)
            self.bnn.load_parameters(os.path.join(params, network))
            self.classes = self.bnn.classes
        else:
            print("ERROR: parameters are not availlable for {0}".format(network))

    def image_to_cifar(self, img, fp):
        """append one CIFAR-10 record (label byte + R, G, B planes) for `img`.

        Same procedure as the reference (thumbnail to 32x32 with the ANTIALIAS
        = LANCZOS filter from the full-resolution image, centred on a
        transparent white canvas).  `reducing_gap=None` keeps modern Pillow
        from pre-shrinking JPEGs, which is what reproduces
        tests/Test_image/deer.bin."""
        img.thumbnail((32, 32), Image.LANCZOS, reducing_gap=None)
        canvas = Image.new("RGBA", (32, 32), (255, 255, 255, 0))
        canvas.paste(img, (int((32 - img.size[0]) / 2), int((32 - img.size[1]) / 2)))
        px = np.array(canvas)
        fp.write(np.identity(1, dtype=np.uint8).tobytes())
        for ch in range(3):
            fp.write(px[:, :, ch].flatten().tobytes())

    def images_to_cifar(self, imgs):
        """CIFAR-10 records (uint8 array [n, 3073]) of a list of PIL images (or decoded pictures as
        uint8 arrays [H, W, 3] / [H, W, 4] / [H, W], which skips the PIL -> numpy copy), the resampling done on
        the GPU (``bnn_mi355x_images_to_cifar``): same bytes as :meth:`image_to_cifar` writes, but
        the caller's images are left as they are (the reference's ``thumbnail`` shrinks them in
        place).  Modes other than RGB, RGBA and L (palette, LA, ...) take the PIL route on the host."""
        import io
        from concurrent.futures import ThreadPoolExecutor
        recs = np.empty((len(imgs), 3073), dtype=np.uint8)
        # a library with only the reference's six symbols (the reference's own .so) has no such entry point
        on_device = hasattr(self.bnn.interface, "bnn_mi355x_images_to_cifar")

        def decoded(img):
            """uint8 array for the device route, None for the PIL route"""
            if isinstance(img, np.ndarray):  # a decoded picture
                if img.dtype != np.uint8 or not (img.ndim == 2 or (img.ndim == 3 and img.shape[2] in (3, 4))):
                    raise ValueError("pictures given as arrays must be uint8 [H, W, 3] (RGB), [H, W, 4] (RGBA) or [H, W] (L)")
                return np.ascontiguousarray(img) if on_device else None
            if on_device and img.mode in ("RGB", "RGBA", "L"):
                return np.ascontiguousarray(np.asarray(img))  # (decodes lazily opened files: PIL releases the GIL there)
            return None

        # a bounded number of decoded pictures in memory at a time; decoding on a few threads
        CHUNK = 32
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
            for base in range(0, len(imgs), CHUNK):
                part = imgs[base:base + CHUNK]
                arrays = list(pool.map(decoded, part)) if len(part) > 1 else [decoded(part[0])]
                dev = [(base + i, a) for i, a in enumerate(arrays) if a is not None]
                for i, a in enumerate(arrays):
                    if a is None:  # the reference's own procedure, on the host
                        img = part[i]
                        buf = io.BytesIO()
                        self.image_to_cifar(Image.fromarray(img) if isinstance(img, np.ndarray) else img.copy(), buf)
                        recs[base + i] = np.frombuffer(buf.getvalue(), dtype=np.uint8)
                n = len(dev)
                if n:
                    arrs = [a for _, a in dev]
                    ptrs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrs])
                    ws = (ctypes.c_int * n)(*[a.shape[1] for a in arrs])
                    hs = (ctypes.c_int * n)(*[a.shape[0] for a in arrs])
                    bs = (ctypes.c_int * n)(*[1 if a.ndim == 2 else a.shape[2] for a in arrs])
                    out = np.empty((n, 3073), dtype=np.uint8)
                    lib = self.bnn.interface
                    if lib.bnn_mi355x_images_to_cifar(ptrs, ws, hs, bs, None, n, out.ctypes.data) != 0:
                        raise RuntimeError(lib.bnn_mi355x_last_error().decode())
                    recs[[i for i, _ in dev]] = out
        return recs

    def _with_tmp(self, imgs, fn):
        with tempfile.NamedTemporaryFile() as tmp:
            tmp.write(self.images_to_cifar(list(imgs)).tobytes())
            tmp.flush()
            result = fn(tmp.name)
        self.usecPerImage = self.bnn.usecPerImage
        return result

    def classify_image(self, img):
        return self._with_tmp([img], self.bnn.inference)

    def classify_cifar(self, path):
        result = self.bnn.inference(path)
        self.usecPerImage = self.bnn.usecPerImage
        return result

    def classify_image_details(self, img):
        return self._with_tmp([img], self.bnn.detailed_inference)

    def classify_cifar_details(self, path):
        result = self.bnn.detailed_inference(path)
        self.usecPerImage = self.bnn.usecPerImage
        return result

    def classify_path(self, path):
        return self.classify_image(Image.open(path))

    def classify_images(self, imgs):
        return self._with_tmp(imgs, self.bnn.inference_multiple)

    def classify_images_with_faults(self, imgs, num_faults, word_size, target_type, target_layers=[]):
        return self._with_tmp(imgs, lambda p: self.bnn.inference_multiple_with_faults(
            p, num_faults, word_size, target_type, target_layers))

    def classify_cifars(self, path):
        result = self.bnn.inference_multiple(path)
        self.usecPerImage = self.bnn.usecPerImage
        return result

    def classify_cifars_with_faults(self, path, num_faults, word_size, target_type, target_layers=[]):
        result = self.bnn.inference_multiple_with_faults(path, num_faults, word_size, target_type, target_layers)
        self.usecPerImage = self.bnn.usecPerImage
        return result

    def classify_images_details(self, imgs):
        return self._with_tmp(imgs, self.bnn.inference_multiple_detail)

    def classify_cifars_details(self, path):
        result = self.bnn.inference_multiple_detail(path)
        self.usecPerImage = self.bnn.usecPerImage
        return result

    def classify_paths(self, paths):
        return self.classify_images([Image.open(p) for p in paths])

    # extension
    def classify_array(self, images, detail=False):
        result = self.bnn.inference_array(images, detail)
        self.usecPerImage = self.bnn.usecPerImage
        return result

    def class_name(self, index):
        return self.bnn.classes[index]


class LfcClassifier:
    """LFC networks on MNIST formatted (28x28) images."""

    def __init__(self, network, params, runtime=RUNTIME_HW):
        if params in available_params(network):
            self.net = network
            self.params = params
            self.runtime = runtime
            self.usecPerImage = 0.0
            self.bnn = PynqBNN(runtime, network)
            self.bnn.load_parameters(os.path.join(params, network))
            self.classes = self.bnn.classes
        else:
            print("ERROR: parameters are not availlable for {0}".format(network))

    def classify_mnist(self, mnist_format_file):
        result = self.bnn.inference(mnist_format_file)
        self.usecPerImage = self.bnn.usecPerImage
        return result

    def classify_mnists(self, mnist_format_file):
        result = self.bnn.inference_multiple(mnist_format_file)
        self.usecPerImage = self.bnn.usecPerImage
        return result

    def classify_mnists_with_faults(self, mnist_format_file, num_faults, flip_word, target_type, target_layers=[]):
        result = self.bnn.inference_multiple_with_faults(mnist_format_file, num_faults, flip_word, target_type,
                                                         target_layers)
        self.usecPerImage = self.bnn.usecPerImage
        return result

    # extension
    def classify_array(self, images):
        result = self.bnn.inference_array(images)
        self.usecPerImage = self.bnn.usecPerImage
        return result

    def class_name(self, index):
        return self.bnn.classes[index]
