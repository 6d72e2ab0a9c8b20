"""Fault injection: the planner's address arithmetic and the read-modify-write on the
memories against the oracle's restatement of inject_fault_impl (CPU), and whole campaigns on the
GPU replayed fault by fault in the oracle.  The reference seeds from std::random_device, so WHICH
faults are drawn cannot be compared with it ("parity unpinned" for the random stream); what a given
fault does to the memories and to every later classification is checked bit for bit."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import gpu_lib as gl
import oracle_lib as ol
from bnn import params_io

NETS = [("cnvW1A1", "cifar10"), ("cnvW1A2", "cifar10"), ("cnvW2A2", "cifar10"), ("lfcW1A1", "mnist"), ("lfcW1A2", "mnist")]


def plan(network, seed, n_images, flips, word_size, target, layers=()):
    L = gl.load(network)
    rec = (C.c_int * (8 * flips))()
    tl = (C.c_int * max(len(layers), 1))(*layers)
    n = L.bnn_mi355x_plan_faults(seed, n_images, flips, word_size, target, tl, len(layers), rec, flips)
    return np.array(rec[: 8 * n], np.int32).reshape(n, 8)


@pytest.mark.parametrize("network,dataset", NETS, ids=lambda x: x)
def test_plan_is_deterministic_sorted_and_in_range(network, dataset):
    a = plan(network, 7, 1000, 200, 1, -1)
    assert (a == plan(network, 7, 1000, 200, 1, -1)).all() and not (a == plan(network, 8, 1000, 200, 1, -1)).all()
    assert a.shape == (200, 8) and (np.diff(a[:, 0]) >= 0).all() and a[:, 0].min() >= 0 and a[:, 0].max() < 1000
    lay = params_io.layout(network)
    for img, target, l, mem, ind, thresh, bit, ws in a:
        L = lay[l]
        assert 0 <= mem < L["pe"]
        if target == 0:
            assert 0 <= ind < L["wmem"] and 0 <= bit < L["simd"] * L["wbits"] and thresh == 0
        else:
            assert L["nthr"] > 0 and 0 <= ind < L["tmem"] and 0 <= thresh < L["nthr"] and 0 <= bit < 24
    # targets honour the request; layer choice is weighted by memory size (largest layer is hit most)
    assert (plan(network, 3, 50, 300, 1, 0)[:, 1] == 0).all() and (plan(network, 3, 50, 300, 1, 1)[:, 1] == 1).all()
    w = plan(network, 5, 50, 4000, 1, 0)
    sizes = [L["wbits"] * L["simd"] * L["pe"] * L["wmem"] for L in lay]
    counts = np.bincount(w[:, 2], minlength=len(lay))
    assert abs(counts[int(np.argmax(sizes))] / 4000.0 - max(sizes) / float(sum(sizes))) < 0.05
    only = plan(network, 5, 50, 100, 4, -1, layers=(1, 2))
    assert set(only[:, 2]) <= {1, 2} and (only[:, 7] == 4).all()


@pytest.mark.parametrize("network,dataset", NETS, ids=lambda x: x)
def test_faulty_blob_matches_oracle_memories(network, dataset):
    """product: read files -> apply faults -> pack blob.  oracle: load -> apply the same faults.
    Every row the faults touched must decode to the oracle's faulty weights / thresholds."""
    pdir = gl.param_dir(dataset, network)
    recs = np.concatenate([plan(network, 11, 10, 150, 1, -1), plan(network, 12, 10, 60, 3, 1), plan(network, 13, 10, 60, 8, 0)])
    L = gl.load(network)
    flat = np.ascontiguousarray(recs.reshape(-1), np.int32)
    fp = flat.ctypes.data_as(C.POINTER(C.c_int))
    size = L.bnn_mi355x_pack_params_faulty(pdir.encode(), fp, len(recs), None, 0)
    blob = np.zeros(size, np.uint8)
    assert L.bnn_mi355x_pack_params_faulty(pdir.encode(), fp, len(recs), blob.ctypes.data, size) == size
    clean = gl.pack_params(network, pdir)
    assert (blob != clean).any()
    o = ol.Oracle(network, pdir)
    touched = set()
    for r in recs:
        row = o.apply_fault(r)
        assert row >= 0
        touched.add((int(r[2]), row))
    # decode the touched rows of the blob and compare with the oracle's (now faulty) matrices
    for l, n in sorted(touched):
        off, rd, rows, kw = struct.unpack_from("<4I", blob, 32 + 16 * l)
        R = blob[off + n * rd * 4: off + (n + 1) * rd * 4].view(np.uint32)
        W = np.array([o.L.bnn_oracle_weight(o.h, l, n, j) for j in range(o.L.bnn_oracle_layer_mw(o.h, l))], np.int8)
        if kw == 0:
            taps = R[2:9].copy().view(np.int8)[:27]
            assert (taps == W.reshape(3, 3, 3).transpose(2, 0, 1).reshape(27)).all()
        else:
            wq = R[2:].copy().view(np.uint64)
            bits = lambda x: np.unpackbits(np.ascontiguousarray(x).view(np.uint8), bitorder="little").astype(np.int8)
            if network.endswith("A1") or (network == "lfcW1A2" and l == 0):
                assert (bits(wq) == (W > 0)).all()
            elif "W2" in network:   # {sign, non-zero} plane pairs, then the "weight is -2" plane, flag, pad
                pairs, two, flag = wq[: 2 * kw], wq[2 * kw: 3 * kw], R[2 + 6 * kw]
                assert (bits(pairs[0::2]) == (W < 0)).all() and (bits(pairs[1::2]) == (W != 0)).all()
                assert (bits(two) == (W == -2)).all() and flag == int((W == -2).any())
            else:
                assert (bits(wq) == (W < 0)).all()
    # rows no fault touched are byte-identical to the clean blob (the matrix-pipe copy of layer 0
    # is rebuilt whenever a layer-0 row changes: checked in test_pack_params.py, excluded here)
    same = np.ones(size, bool)
    l0m = struct.unpack_from("<I", blob, 24)[0]
    if l0m:
        same[l0m: l0m + 2 * 64 * 32 + 4096] = False      # pixel-form and tile-form operands
    for l, n in touched:
        off, rd, rows, kw = struct.unpack_from("<4I", blob, 32 + 16 * l)
        same[off + n * rd * 4: off + (n + 1) * rd * 4] = False
    assert (blob[same] == clean[same]).all()


def test_layer0_threshold_fault_quirk():
    """reading a CNV layer-0 threshold returns its INTEGER part and writing reinterprets the word with
    8 fraction bits (top.cpp:84,143): a fault there rescales the threshold by 2^-8 -- mirrored, not 'fixed'"""
    o = ol.Oracle("cnvW1A1", ol.param_dir("cifar10", "cnvW1A1"))
    before = o.L.bnn_oracle_threshold(o.h, 0, 5, 0)
    o.apply_fault((0, 1, 0, 5 % 16, 5 // 16, 0, 0, 1))
    after = o.L.bnn_oracle_threshold(o.h, 0, 5, 0)
    want = ((before >> 8) ^ 1) & 0xFFFFFF
    want = want - (1 << 24) if want & 0x800000 else want
    assert after == want


@pytest.mark.gpu
@pytest.mark.parametrize("network,dataset", NETS, ids=lambda x: x)
@pytest.mark.parametrize("target,word_size", [(-1, 1), (0, 4), (1, 1)])
def test_campaign_on_gpu_replayed_in_oracle(network, dataset, target, word_size, tmp_path):
    L = gl.load(network)
    pdir = gl.param_dir(dataset, network)
    L.load_parameters(pdir.encode())
    n, flips = 240, 60
    rng = np.random.default_rng(17)
    if network.startswith("cnv"):
        imgs = rng.integers(0, 256, (n, 3072), dtype=np.uint8)
        path = tmp_path / "imgs.bin"
        np.concatenate([np.ones((n, 1), np.uint8), imgs], axis=1).tofile(path)
    else:
        imgs = rng.integers(0, 256, (n, 784), dtype=np.uint8)
        path = tmp_path / "imgs-idx3-ubyte"
        with open(path, "wb") as f:
            f.write(struct.pack(">4I", 0x803, n, 28, 28) + imgs.tobytes())
    assert L.bnn_mi355x_set_fault_seed(1234 + word_size) == 0
    cnt, usec = C.c_int(0), C.c_float(0)
    p = L.inference_multiple_with_faults(str(path).encode(), 10, C.byref(cnt), C.byref(usec), flips, word_size, target, None, 0)
    assert p and cnt.value == n and usec.value > 0
    got = np.ctypeslib.as_array(p, shape=(n,)).copy()
    L.free_results(p)
    rec = (C.c_int * (8 * flips))()
    assert L.bnn_mi355x_last_faults(rec, flips) == flips
    recs = np.array(rec[:], np.int32).reshape(flips, 8)
    # replay: faults drawn for image i are applied before image i is classified
    o = ol.Oracle(network, pdir)
    want = np.zeros(n, np.int32)
    k, start = 0, 0
    while start < n:
        while k < flips and recs[k, 0] <= start:
            assert o.apply_fault(recs[k]) >= 0
            k += 1
        end = int(recs[k, 0]) if k < flips else n
        want[start:end] = o.classes_batched(imgs[start:end], 10)
        start = end
    assert got.tolist() == want.tolist()
    clean = ol.Oracle(network, pdir).classes_batched(imgs, 10)
    assert target == 1 or (want != clean).any() or flips < 10   # the campaign did change something (weights/any)
    # the faults stay in the loaded parameters ...
    p = L.inference_multiple(str(path).encode(), 10, C.byref(cnt), None, 0)
    again = np.ctypeslib.as_array(p, shape=(n,)).copy()
    L.free_results(p)
    assert again.tolist() == o.classes_batched(imgs, 10).tolist()
    # ... until the next load_parameters
    L.load_parameters(pdir.encode())
    p = L.inference_multiple(str(path).encode(), 10, C.byref(cnt), None, 0)
    fresh = np.ctypeslib.as_array(p, shape=(n,)).copy()
    L.free_results(p)
    assert fresh.tolist() == clean.tolist()


@pytest.mark.gpu
def test_campaign_over_a_multi_chunk_file(tmp_path):
    """70 000 MNIST images: the resident load streams the file in several chunks (pageable host chunks, raw pixels), runs of images between
    fault times cross chunk borders; replayed in the oracle like the small campaigns"""
    network, dataset, n, flips = "lfcW1A1", "mnist", 70000, 40
    L = gl.load(network)
    pdir = gl.param_dir(dataset, network)
    L.load_parameters(pdir.encode())
    imgs = np.random.default_rng(23).integers(0, 256, (n, 784), dtype=np.uint8)
    path = tmp_path / "imgs-idx3-ubyte"
    with open(path, "wb") as f:
        f.write(struct.pack(">4I", 0x803, n, 28, 28) + imgs.tobytes())
    assert L.bnn_mi355x_set_fault_seed(99) == 0
    cnt = C.c_int(0)
    p = L.inference_multiple_with_faults(str(path).encode(), 10, C.byref(cnt), None, flips, 8, 0, None, 0)
    assert p and cnt.value == n
    got = np.ctypeslib.as_array(p, shape=(n,)).copy()
    L.free_results(p)
    rec = (C.c_int * (8 * flips))()
    assert L.bnn_mi355x_last_faults(rec, flips) == flips
    recs = np.array(rec[:], np.int32).reshape(flips, 8)
    o = ol.Oracle(network, pdir)
    want = np.zeros(n, np.int32)
    k, start = 0, 0
    while start < n:
        while k < flips and recs[k, 0] <= start:
            assert o.apply_fault(recs[k]) >= 0
            k += 1
        end = int(recs[k, 0]) if k < flips else n
        want[start:end] = o.classes_batched(imgs[start:end], 10)
        start = end
    assert got.tolist() == want.tolist()
    L.load_parameters(pdir.encode())


def test_planner_refuses_requests_it_cannot_honour():
    """thresholds-only faults on layers without threshold memory, or a layer outside the network: an error,
    not a campaign on something else under the same label (host-only entry point, no GPU needed)"""
    import ctypes as C
    L = gl.load("cnvW1A1")
    rec = (C.c_int * 80)()
    layers = (C.c_int * 1)(8)                                   # CNV layer 8: PassThroughActivation, no thresholds
    assert L.bnn_mi355x_plan_faults(7, 100, 10, 1, 1, layers, 1, rec, 10) < 0
    assert b"no threshold memory" in L.bnn_mi355x_last_error()
    assert L.bnn_mi355x_plan_faults(7, 100, 10, 1, 0, layers, 1, rec, 10) == 10       # weights of layer 8: fine
    assert all(rec[8 * i + 2] == 8 and rec[8 * i + 1] == 0 for i in range(10))
    assert L.bnn_mi355x_plan_faults(7, 100, 10, 1, -1, layers, 1, rec, 10) == 10      # "any": only weights exist there
    assert all(rec[8 * i + 1] == 0 for i in range(10))
    for bad in (9, -1, 100):
        layers[0] = bad
        assert L.bnn_mi355x_plan_faults(7, 100, 10, 1, -1, layers, 1, rec, 10) < 0
        assert b"out of range" in L.bnn_mi355x_last_error()
