"""CPU: host logic of the Python classifier shim (bnn-pynq_amd/bnn/bnn.py), exercised through the
SAME six-symbol C ABI served by the oracle's per-network libraries (oracle/_build/
python_sw-<net>-oracle.so) -- BASELINE.json configs[0]: "LFC-W1A1 MNIST single-image classify via
the SW-runtime .so on CPU (plumbing, no GPU)".  The product never does this: the test points the
shim's BNN_LIB_DIR / PLATFORM at the oracle build by monkeypatching, to check ownership
(free_results), NULL results, enable_detail, image_to_cifar and the mirrored API surface."""
import os

import numpy as np
import pytest
from PIL import Image

import oracle_lib as ol


@pytest.fixture()
def bnn_on_oracle(monkeypatch):
    import bnn
    from bnn import bnn as mod
    monkeypatch.setattr(mod, "BNN_LIB_DIR", ol.BUILD_DIR)
    monkeypatch.setattr(mod, "PLATFORM", "oracle")
    monkeypatch.setattr(mod, "_libraries", {})
    return bnn


def test_api_surface_matches_reference_names():
    import bnn
    for name in ["RUNTIME_HW", "RUNTIME_SW", "NETWORK_CNVW1A1", "NETWORK_CNVW1A2", "NETWORK_CNVW2A2", "NETWORK_LFCW1A1",
                 "NETWORK_LFCW1A2", "NETWORK_CNVW1A1_TMR", "NETWORK_LFCW1A2_INTERLEAVED", "available_params",
                 "PynqBNN", "CnvClassifier", "LfcClassifier", "BNN_ROOT_DIR", "BNN_LIB_DIR", "BNN_BIT_DIR", "BNN_PARAM_DIR"]:
        assert hasattr(bnn, name), name
    cnv = ["classify_image", "classify_cifar", "classify_image_details", "classify_cifar_details", "classify_path",
           "classify_images", "classify_images_with_faults", "classify_cifars", "classify_cifars_with_faults",
           "classify_images_details", "classify_cifars_details", "classify_paths", "class_name", "image_to_cifar"]
    for m in cnv:
        assert callable(getattr(bnn.CnvClassifier, m)), m
    for m in ["classify_mnist", "classify_mnists", "classify_mnists_with_faults", "class_name"]:
        assert callable(getattr(bnn.LfcClassifier, m)), m
    for m in ["load_parameters", "inference", "detailed_inference", "inference_multiple",
              "inference_multiple_with_faults", "inference_multiple_detail", "class_name"]:
        assert callable(getattr(bnn.PynqBNN, m)), m
    assert (bnn.RUNTIME_HW, bnn.RUNTIME_SW) == ("python_hw", "python_sw")
    assert sorted(bnn.available_params(bnn.NETWORK_CNVW1A1)) == ["cifar10", "road-signs", "streetview"]
    assert sorted(bnn.available_params(bnn.NETWORK_LFCW1A1)) == ["chars_merged", "mnist"]


def test_lfc_single_image_plumbing(bnn_on_oracle):
    """configs[0]: tests/Test_image/3.image-idx3-ubyte -> 3 (tests/test_qnn.py:44-73)"""
    bnn = bnn_on_oracle
    clf = bnn.LfcClassifier(bnn.NETWORK_LFCW1A1, "mnist", bnn.RUNTIME_SW)
    assert len(clf.classes) == 10
    path = os.path.join(ol.GOLDEN, "3.image-idx3-ubyte")
    assert clf.classify_mnist(path) == 3
    out = clf.classify_mnists(path)
    assert out.dtype == np.int32 and out.tolist() == [3]
    assert clf.usecPerImage > 0 and clf.class_name(3) == "3"
    assert clf.classify_mnists_with_faults(path, 0, 1, -1).tolist() == [3]


def test_cnv_details_and_batches(bnn_on_oracle, tmp_path):
    bnn = bnn_on_oracle
    clf = bnn.CnvClassifier(bnn.NETWORK_CNVW1A1, "cifar10", bnn.RUNTIME_SW)
    deer = os.path.join(ol.GOLDEN, "deer.cifar")
    assert clf.classify_cifar(deer) == 4
    d = clf.classify_cifar_details(deer)
    assert d.tolist() == [234, 231, 265, 248, 410, 257, 224, 262, 226, 233]   # CNV-QNN_Cifar10.ipynb:164-173
    # a multi-record file: details are flattened image by image
    multi = tmp_path / "two.bin"
    multi.write_bytes(open(deer, "rb").read() + open(os.path.join(ol.GOLDEN, "car.cifar"), "rb").read())
    assert clf.classify_cifars(str(multi)).tolist() == [4, 1]
    dd = clf.classify_cifars_details(str(multi)).reshape(2, 10)
    assert dd[0].tolist() == d.tolist() and dd[1].tolist() == [258, 417, 233, 206, 238, 215, 222, 238, 236, 249]
    assert clf.class_name(4) == "Deer"


def test_image_to_cifar_reproduces_reference_fixture(bnn_on_oracle):
    """image_to_cifar(deer.jpg) interior == tests/Test_image/deer.bin; classify_image -> recorded scores"""
    bnn = bnn_on_oracle
    clf = bnn.CnvClassifier(bnn.NETWORK_CNVW1A1, "cifar10", bnn.RUNTIME_SW)
    # rebuild the picture from the committed record: a 32x32 RGB image passes through image_to_cifar unchanged
    rec = np.frombuffer(open(os.path.join(ol.GOLDEN, "deer.cifar"), "rb").read(), np.uint8)
    assert rec[0] == 1 and rec.size == 3073
    rgb = rec[1:].reshape(3, 32, 32).transpose(1, 2, 0)
    img = Image.fromarray(rgb, "RGB")
    assert clf.classify_image(img) == 4
    assert clf.classify_image_details(Image.fromarray(rgb, "RGB")).tolist() == \
        [234, 231, 265, 248, 410, 257, 224, 262, 226, 233]
    assert clf.classify_images([Image.fromarray(rgb, "RGB")] * 3).tolist() == [4, 4, 4]


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import bnn
    from bnn import bnn as mod
    monkeypatch.setattr(mod, "BNN_LIB_DIR", str(tmp_path))
    monkeypatch.setattr(mod, "_libraries", {})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        bnn.PynqBNN(bnn.RUNTIME_SW, bnn.NETWORK_LFCW1A1)


def test_variant_constants_resolve_to_base_parameters():
    """the hardened overlays' names (bnn.py:41-53) find the base network's (byte-identical) parameter files"""
    import bnn
    assert bnn.NETWORK_CNVW1A1_TMR == "cnvW1A1-TMR" and bnn.NETWORK_LFCW1A2_INTERLEAVED == "lfcW1A2-interleaved"
    assert set(bnn.available_params(bnn.NETWORK_CNVW1A1_TMR)) == set(bnn.available_params(bnn.NETWORK_CNVW1A1))
    assert "mnist" in bnn.available_params(bnn.NETWORK_LFCW1A2_INTERLEAVED)
    assert bnn.available_params("cnvW9A9-TMR") == []
