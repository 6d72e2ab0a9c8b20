"""Pin the CPU restatement (oracle/) against the outputs RECORDED in the
reference's notebooks and tests (tests/golden/expected.json, produced by
tests/golden/make_fixtures.py).  This is what makes parity 'pinned': the
reference SW runtime itself cannot be built or run here."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

EXP = json.load(open(os.path.join(ol.GOLDEN, "expected.json")))
_cache = {}


def oracle(network, dataset):
    k = (network, dataset)
    if k not in _cache:
        _cache[k] = ol.Oracle(network, ol.param_dir(dataset, network))
    return _cache[k]


def load_input(name):
    p = os.path.join(ol.GOLDEN, name)
    return ol.read_mnist(p) if name.endswith("ubyte") else ol.read_cifar(p)


@pytest.mark.parametrize("e", EXP["scores"], ids=lambda e: e["input"] + "-" + e["network"])
def test_recorded_scores(e):
    """40 recorded 16-bit class scores (4 vectors x 10 classes), exact."""
    o = oracle(e["network"], e["params"])
    img = load_input(e["input"])[0]
    ref = o.scores_ref(img)[:10]
    fast = o.scores_fast(img[None])[0, :10]
    assert ref.tolist() == e["scores"], e["source"]
    assert fast.tolist() == e["scores"], e["source"]


@pytest.mark.parametrize("e", EXP["classes"], ids=lambda e: e["input"] + "-" + e["network"] + "-" + e["params"])
def test_recorded_classes(e):
    o = oracle(e["network"], e["params"])
    ncls = ol.num_classes(e["params"], e["network"])
    img = load_input(e["input"])[0]
    if o.is_cnv:
        s = o.scores_ref(img)
        assert ol.decode_cnv_single(s, ncls) == e["class"], e["source"]
        assert ol.decode_cnv_batched(s, ncls) == e["class"], e["source"]
    else:
        w = o.word_ref(img)
        L = ol.lib()
        assert L.bnn_oracle_decode_lfc_single(w, ncls) == e["class"], e["source"]
        assert L.bnn_oracle_decode_lfc_batched(w, ncls) == e["class"], e["source"]
    assert o.classes_batched(img[None], ncls)[0] == e["class"]


def test_lfc_raw_word_is_one_hot_3():
    """raw output word of 3.image-idx3-ubyte on lfcW1A1 is 0b1000"""
    o = oracle("lfcW1A1", "mnist")
    w = o.word_ref(load_input("3.image-idx3-ubyte")[0])
    assert w & 0x3FF == 0b1000


def test_derived_deer_bin_scores():
    """restatement outputs on the reference's 0-padded deer.bin (not recorded by
    the reference; listed in SURVEY 8(c) as derived fixtures) -- regression guard"""
    exp = {"cnvW1A1": [228, 229, 251, 250, 412, 253, 230, 262, 222, 239],
           "cnvW1A2": [-26, -44, -32, -8, 264, 2, -12, -32, -46, -36],
           "cnvW2A2": [-22, -27, -16, -10, 239, 1, -9, -17, -15, -27]}
    img = load_input("deer.bin")[0]
    for net, s in exp.items():
        assert oracle(net, "cifar10").scores_ref(img)[:10].tolist() == s
