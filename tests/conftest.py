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
def _build_oracle():
    """tests are the only place (besides smoke()/bench cpu_baseline) that touch oracle/"""
    import oracle_lib
    if not os.path.exists(os.path.join(oracle_lib.BUILD_DIR, "libbnn_oracle.so")):
        oracle_lib.build()
