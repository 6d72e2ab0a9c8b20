"""ctypes binding of the runtime's C ABI (``include/bnn_mi355x.h``): the reference's six symbols
(``bnn/bnn.py:69-77`` binds the same ones through cffi) plus the ``bnn_mi355x_*`` extensions.
Used by ``bnn.py``, ``multigpu.py``, ``bench.py`` and the tests."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.realpath(__file__))
PLATFORM = os.environ.get("BNN_PLATFORM", "mi355x")
# BNN_MI355X_LIBDIR: kernel-tuning experiments point this at an alternative build of the same ABI
LIB_DIR = os.environ.get("BNN_MI355X_LIBDIR") or os.path.join(ROOT, "libraries", PLATFORM)
PARAM_ROOT = os.path.join(ROOT, "params")

LEGACY = ["load_parameters", "inference", "inference_multiple", "inference_multiple_with_faults",
          "free_results", "deinit"]
EXT = ["bnn_mi355x_network", "bnn_mi355x_image_bytes", "bnn_mi355x_last_error", "bnn_mi355x_set_device",
       "bnn_mi355x_pack_params", "bnn_mi355x_export_params", "bnn_mi355x_import_params",
       "bnn_mi355x_inference_buffer", "bnn_mi355x_inference_raw", "bnn_mi355x_inference_device",
       "bnn_mi355x_reserve", "bnn_mi355x_set_fault_seed", "bnn_mi355x_last_faults", "bnn_mi355x_plan_faults",
       "bnn_mi355x_pack_params_faulty", "bnn_mi355x_debug_stage_output", "bnn_mi355x_profile",
       "bnn_mi355x_profile_read", "bnn_mi355x_stage_name", "bnn_mi355x_thumbnail_size", "bnn_mi355x_images_to_cifar",
       "bnn_mi355x_params_bytes", "bnn_mi355x_import_params_device", "bnn_mi355x_params_crc", "bnn_mi355x_chunk_plan",
       "bnn_mi355x_binarize_pack"]


def lib_path(network, runtime="python_sw", lib_dir=None):
    return os.path.join(lib_dir or LIB_DIR, "%s-%s-%s.so" % (runtime, network, PLATFORM))


def declare_legacy(L):
    """argument / return types of the reference's cdef"""
    ip, fp = C.POINTER(C.c_int), C.POINTER(C.c_float)
    L.load_parameters.argtypes = [C.c_char_p]
    L.load_parameters.restype = None
    L.inference.argtypes = [C.c_char_p, ip, C.c_int, fp]
    L.inference.restype = C.c_int
    L.inference_multiple.argtypes = [C.c_char_p, C.c_int, ip, fp, C.c_int]
    L.inference_multiple.restype = ip
    L.inference_multiple_with_faults.argtypes = [C.c_char_p, C.c_int, ip, fp, C.c_uint, C.c_int, C.c_int, ip, C.c_uint]
    L.inference_multiple_with_faults.restype = ip
    L.free_results.argtypes = [ip]
    L.free_results.restype = None
    L.deinit.argtypes = []
    L.deinit.restype = None


def declare_extensions(L):
    ip, fp = C.POINTER(C.c_int), C.POINTER(C.c_float)
    L.bnn_mi355x_network.restype = C.c_char_p
    L.bnn_mi355x_image_bytes.restype = C.c_int
    L.bnn_mi355x_last_error.restype = C.c_char_p
    L.bnn_mi355x_set_device.argtypes = [C.c_int]
    L.bnn_mi355x_pack_params.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
    L.bnn_mi355x_pack_params.restype = C.c_size_t
    L.bnn_mi355x_export_params.argtypes = [C.c_void_p, C.c_size_t]
    L.bnn_mi355x_export_params.restype = C.c_size_t
    L.bnn_mi355x_import_params.argtypes = [C.c_void_p, C.c_size_t]
    L.bnn_mi355x_params_bytes.argtypes = []
    L.bnn_mi355x_params_bytes.restype = C.c_size_t
    L.bnn_mi355x_import_params_device.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.bnn_mi355x_params_crc.argtypes = []
    L.bnn_mi355x_params_crc.restype = C.c_uint
    L.bnn_mi355x_inference_buffer.argtypes = [C.c_void_p, C.c_int, C.c_int, fp, C.c_int]
    L.bnn_mi355x_inference_buffer.restype = ip
    L.bnn_mi355x_inference_raw.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, fp]
    L.bnn_mi355x_inference_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p]
    L.bnn_mi355x_reserve.argtypes = [C.c_int]
    L.bnn_mi355x_chunk_plan.argtypes = [C.c_int, C.c_int, ip, C.c_int]
    L.bnn_mi355x_binarize_pack.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
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
    L.bnn_mi355x_thumbnail_size.argtypes = [C.c_int, C.c_int, ip, ip]
    L.bnn_mi355x_images_to_cifar.argtypes = [C.POINTER(C.c_void_p), ip, ip, ip, C.POINTER(C.c_long), C.c_int, C.c_void_p]


_cache = {}


def load(network, runtime="python_sw", lib_dir=None):
    return load_path(lib_path(network, runtime, lib_dir))


def load_path(path):
    """dlopen once per process and path (like the reference's _libraries cache) and declare the ABI"""
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise RuntimeError("runtime library %s not found: build it with `make -C bnn-pynq_amd` "
                           "(the MI355X runtime has no CPU fallback)" % path)
    L = C.CDLL(path)
    declare_legacy(L)
    L.has_extensions = hasattr(L, "bnn_mi355x_inference_buffer")
    if L.has_extensions:
        declare_extensions(L)
    _cache[path] = L
    return L
