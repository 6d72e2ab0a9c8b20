"""CPU: bnn/params_io.py (Python-3 restatement of finnthesizer's packing half) against the shipped
parameter files (byte equality after a read -> write round trip) and against the oracle's loader;
oracle fast == faithful on random parameter sets."""
import filecmp
import os

import numpy as np
import pytest

import oracle_lib as ol
import random_params
from bnn import params_io

SETS = [("cnvW1A1", "cifar10"), ("cnvW1A2", "cifar10"), ("cnvW2A2", "cifar10"), ("lfcW1A1", "mnist"),
        ("lfcW1A2", "mnist"), ("cnvW1A1", "road-signs")]


@pytest.mark.parametrize("network,dataset", SETS, ids=lambda x: x)
def test_round_trip_is_byte_identical(network, dataset, tmp_path):
    src = ol.param_dir(dataset, network)
    W, T = params_io.read_params(src, network)
    params_io.write_params(str(tmp_path), network, W, T)
    names = [f for f in os.listdir(src) if f.endswith(".bin")]
    assert len(names) > 50
    match, mismatch, errors = filecmp.cmpfiles(src, str(tmp_path), names, shallow=False)
    assert not mismatch and not errors and len(match) == len(names)


def test_reader_matches_oracle_loader():
    W, T = params_io.read_params(ol.param_dir("cifar10", "cnvW2A2"), "cnvW2A2")
    o = ol.Oracle("cnvW2A2", ol.param_dir("cifar10", "cnvW2A2"))
    for l in (0, 3, 8):
        assert (o.weights(l) == W[l]).all()
    assert all(o.L.bnn_oracle_threshold(o.h, 1, n, i) == np.int16(T[1][n, i]) for n in range(64) for i in range(2))


@pytest.mark.parametrize("network", ["cnvW1A1", "cnvW1A2", "cnvW2A2", "lfcW1A1", "lfcW1A2"])
def test_oracle_fast_equals_faithful_on_random_params(network, tmp_path):
    random_params.make(str(tmp_path), network, seed=3)
    o = ol.Oracle(network, str(tmp_path))
    rng = np.random.default_rng(5)
    if o.is_cnv:
        imgs = rng.integers(0, 256, (2, 3072), dtype=np.uint8)
        fast = o.scores_fast(imgs)
        for i in range(2):
            assert fast[i].tolist() == o.scores_ref(imgs[i]).tolist()
        assert np.unique(fast).size > 8           # the random thresholds do not saturate the net
    else:
        imgs = rng.integers(0, 256, (64, 784), dtype=np.uint8)
        fast = o.words_fast(imgs)
        assert [int(x) for x in fast] == [o.word_ref(imgs[i]) for i in range(64)]
        assert np.unique(fast).size > 8
