"""CPU: binarizeAndPack on the host (csrc/pack_inputs.cpp, exported as bnn_mi355x_binarize_pack) against the oracle's
restatement of the reference's (bnn/src/library/host/foldedmv-offload.cpp:82-98).  This is what the host-data entry points of
the LFC libraries run on their worker threads; the GPU tests then compare whole classifications."""
import numpy as np
import pytest

import gpu_lib as gl
import oracle_lib as ol


def pack(L, imgs):
    imgs = np.ascontiguousarray(imgs, np.uint8).reshape(-1, 784)
    out = np.full((imgs.shape[0], 13), 0xA5A5A5A5A5A5A5A5, np.uint64)   # every word must be written, padding bits included
    assert L.bnn_mi355x_binarize_pack(imgs.ctypes.data, imgs.shape[0], out.ctypes.data) == 0
    return out


def numpy_pack(imgs):
    """the definition, spelled out: bit i of the image = (pixel i >= 128), 13 little-endian words, bits 784..831 zero"""
    imgs = np.asarray(imgs, np.uint8).reshape(-1, 784)
    bits = np.zeros((imgs.shape[0], 832), np.uint8)
    bits[:, :784] = imgs >= 128
    return np.packbits(bits, axis=1, bitorder="little").view("<u8")


@pytest.mark.parametrize("network", ["lfcW1A1", "lfcW1A2"])
def test_binarize_pack_matches_the_definition_and_the_oracle(network):
    L = gl.load(network)
    rng = np.random.default_rng(5)
    edges = np.array([0, 1, 126, 127, 128, 129, 254, 255], np.uint8)
    batches = [rng.integers(0, 256, (257, 784), dtype=np.uint8),
               edges[rng.integers(0, 8, (64, 784))],                      # only values next to the decision level
               np.zeros((3, 784), np.uint8), np.full((3, 784), 255, np.uint8), np.full((2, 784), 127, np.uint8),
               np.full((2, 784), 128, np.uint8),
               (np.arange(784 * 5) % 256).astype(np.uint8).reshape(5, 784)]
    one_hot = np.zeros((784, 784), np.uint8)                              # a single bright pixel at every position
    one_hot[np.arange(784), np.arange(784)] = 200
    batches.append(one_hot)
    o = ol.Oracle(network, ol.param_dir("mnist", network))
    for imgs in batches:
        got = pack(L, imgs)
        assert (got == numpy_pack(imgs)).all()
        assert (got == o.binarize(imgs)).all()
    assert pack(L, np.zeros((0, 784), np.uint8)).shape == (0, 13)


def test_binarize_pack_unaligned_source_and_destination():
    """the feeder binarises from wherever the caller's buffer or the file offset lands: any byte alignment"""
    L = gl.load("lfcW1A1")
    rng = np.random.default_rng(6)
    raw = rng.integers(0, 256, 784 * 9 + 64, dtype=np.uint8)
    for off in (0, 1, 3, 7, 16, 31, 33):
        imgs = raw[off:off + 784 * 9]
        dst = np.zeros(13 * 9 * 8 + 16, np.uint8)
        for doff in (0, 8, 4, 1):
            view = dst[doff:doff + 13 * 9 * 8]
            assert L.bnn_mi355x_binarize_pack(imgs.ctypes.data, 9, view.ctypes.data) == 0
            assert (view.view("<u8").reshape(9, 13) == numpy_pack(imgs)).all() if doff % 8 == 0 else \
                (np.frombuffer(view.tobytes(), "<u8").reshape(9, 13) == numpy_pack(imgs)).all()


def test_binarize_pack_is_refused_by_the_cnv_libraries():
    L = gl.load("cnvW1A1")
    assert L.bnn_mi355x_binarize_pack(None, 0, None) == -1
    assert b"CNV" in L.bnn_mi355x_last_error()
