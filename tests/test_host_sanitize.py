"""CPU: the product's host code (parameter packing, fault application, blob validation, resampling tables, the LFC
input binariser)
under AddressSanitizer + UBSan.  Device sanitizers are not available on the GPU pool, so this is where
memory errors in the host half would show."""
import os
import shutil
import subprocess

import pytest

import gpu_lib as gl

CSRC = os.path.join(gl.ROOT, "bnn-pynq_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    srcs = [os.path.join(gl.ROOT, "tests", "host_sanitize", "main.cpp")] + \
           [os.path.join(CSRC, f) for f in ("topology.cpp", "packed_params.cpp", "pack_inputs.cpp", "faults.cpp", "resample.cpp")]
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-ffp-contract=off", "-I", CSRC, "-o", exe] + srcs, check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe, os.path.join(gl.ROOT, "bnn-pynq_amd", "bnn", "params")], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host sanitize run ok" in out.stdout
