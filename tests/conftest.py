import os
import sys

import pytest

# torch (device memory / torch.distributed plumbing) bundles its own HIP runtime with the same
# SONAME as /opt/rocm's: import it BEFORE the product library is dlopen'ed so that the process
# holds exactly one HIP runtime ("One HIP runtime per process").
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "bnn-pynq_amd"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_product():
    """the runtime libraries normally travel with the tree (built by __graft_entry__.build()); if a
    checkout arrives without them and hipcc is here, build them rather than fail every test"""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "bnn-pynq_amd", "bnn", "libraries", "mi355x", "python_sw-cnvW1A1-mi355x.so")
    if not os.path.exists(lib) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "bnn-pynq_amd")], check=True)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """tests are the only place (besides smoke()/bench cpu_baseline) that touch oracle/"""
    import oracle_lib
    if not os.path.exists(os.path.join(oracle_lib.BUILD_DIR, "libbnn_oracle.so")):
        oracle_lib.build()


@pytest.fixture(scope="session")
def variant_libs():
    """the hardened overlays' libraries (cnvW1A1-TMR, ...) are not part of the default build: `make variants`.
    Always run it -- make is incremental -- so that a variant library built before the last change of the runtime
    sources is never what the tests load (round 2 ran stale objects that way)."""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "bnn-pynq_amd", "bnn", "libraries", "mi355x", "python_hw-lfcW1A2-interleaved-mi355x.so")
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "bnn-pynq_amd"), "variants"], check=True)
    assert os.path.exists(lib), "variant libraries missing and no hipcc to build them"
