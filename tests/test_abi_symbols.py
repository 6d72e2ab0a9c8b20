"""CPU: the product C-ABI library loads (no GPU needed for dlopen) and exports
every symbol include/bnn_mi355x.h declares -- and nothing else."""
import ctypes
import os
import re
import subprocess

import pytest

import gpu_lib as gl

HEADER = os.path.join(gl.ROOT, "include", "bnn_mi355x.h")
NETWORKS = ["cnvW1A1", "cnvW1A2", "cnvW2A2", "lfcW1A1", "lfcW1A2"]
# the fork's hardened overlays (bnn.py:41-53): same compute, built under their own names
VARIANTS = ["cnvW1A1-interleaved", "cnvW1A1-resilient-interleaved", "cnvW1A1-TMR", "cnvW1A2-interleaved",
            "cnvW1A2-resilient-interleaved", "cnvW2A2-interleaved", "cnvW2A2-resilient-interleaved", "cnvW2A2-TMR",
            "lfcW1A2-interleaved"]


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", src)))


def test_header_declares_the_reference_cdef():
    """the six symbols of bnn/bnn.py:69-77 are all declared"""
    d = declared_symbols()
    for s in gl.LEGACY:
        assert s in d
    assert sorted(gl.LEGACY + gl.EXT) == d


@pytest.mark.parametrize("network", NETWORKS)
@pytest.mark.parametrize("runtime", ["python_sw", "python_hw"])
def test_library_exports(network, runtime):
    check_exports(network, runtime)


@pytest.mark.parametrize("network", ["cnvW1A1-TMR", "lfcW1A2-interleaved"])
def test_variant_library_exports(network, variant_libs):
    """`make variants` (built on demand): the same ABI under the hardened overlays' names"""
    check_exports(network, "python_sw")
    for v in VARIANTS:
        for rt in ("python_sw", "python_hw"):
            assert os.path.exists(gl.lib_path(v, rt)), v


def check_exports(network, runtime):
    path = gl.lib_path(network, runtime)
    assert os.path.exists(path), "build with `make -C bnn-pynq_amd`"
    lib = ctypes.CDLL(path)
    for s in declared_symbols():
        assert hasattr(lib, s), s
    lib.bnn_mi355x_network.restype = ctypes.c_char_p
    assert lib.bnn_mi355x_network().decode() == network
    lib.bnn_mi355x_image_bytes.restype = ctypes.c_int
    assert lib.bnn_mi355x_image_bytes() == (3072 if network.startswith("cnv") else 784)
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert exported == declared_symbols()


def test_no_gpu_fails_loudly(capfd):
    """without a HIP device the product must refuse to compute (no CPU fallback)"""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    lib = gl.load("lfcW1A1")
    lib.load_parameters(gl.param_dir("mnist", "lfcW1A1").encode())
    assert lib.bnn_mi355x_last_error() != b""
    n = ctypes.c_int(0)
    p = lib.inference_multiple(os.path.join(gl.ROOT, "tests", "golden", "3.image-idx3-ubyte").encode(), 10,
                               ctypes.byref(n), None, 0)
    assert not p
    assert lib.inference(b"/nonexistent", None, 10, None) == -1
