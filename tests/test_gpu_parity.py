"""GPU parity tests proper: the HIP path (through the C ABI of
include/bnn_mi355x.h) against the CPU restatement (oracle/) on the same seeded
inputs, bit-exact on the RAW outputs (all 64 16-bit scores per CNV image, the
full 64-bit word per LFC image), not only on the argmax."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import gpu_lib as gl
import oracle_lib as ol

pytestmark = pytest.mark.gpu

NETS = [("cnvW1A1", "cifar10"), ("cnvW1A2", "cifar10"), ("cnvW2A2", "cifar10"),
        ("lfcW1A1", "mnist"), ("lfcW1A2", "mnist"),
        ("cnvW1A1", "streetview"), ("cnvW1A1", "road-signs"), ("lfcW1A1", "chars_merged")]

_nets, _oracles = {}, {}


def gpu_net(network, dataset):
    # one .so holds one parameter set at a time (like the reference): reload on switch
    key = (network, dataset)
    if _nets.get(network, (None, None))[0] != dataset:
        _nets[network] = (dataset, gl.Net(network, dataset))
    return _nets[network][1]


def oracle(network, dataset):
    key = (network, dataset)
    if key not in _oracles:
        _oracles[key] = ol.Oracle(network, ol.param_dir(dataset, network))
    return _oracles[key]


def rand_images(network, n, seed, kind="uniform"):
    rng = np.random.default_rng(seed)
    isz = 3072 if network.startswith("cnv") else 784
    if kind == "uniform":
        return rng.integers(0, 256, size=(n, isz), dtype=np.uint8)
    if kind == "sparse":  # MNIST-like: 10 % bright pixels
        return np.where(rng.random((n, isz)) < 0.1, 255, 0).astype(np.uint8)
    if kind == "edges":  # quantiser edge values
        return rng.choice(np.array([0, 1, 126, 127, 128, 129, 254, 255], np.uint8), size=(n, isz))
    raise ValueError(kind)


@pytest.mark.parametrize("network,dataset", NETS, ids=lambda x: x)
@pytest.mark.parametrize("kind", ["uniform", "sparse", "edges"])
def test_raw_outputs_bit_exact(network, dataset, kind):
    n = 333 if network.startswith("cnv") else 2000  # ragged: not a multiple of the 256-lane blocks
    imgs = rand_images(network, n, 7, kind)
    g = gpu_net(network, dataset).raw(imgs)
    o = oracle(network, dataset)
    ref = o.scores_fast(imgs) if o.is_cnv else o.words_fast(imgs)
    assert g.shape == ref.shape
    bad = np.nonzero((g != ref).reshape(n, -1).any(axis=1))[0]
    assert bad.size == 0, "first mismatching images: %s" % bad[:8]


@pytest.mark.parametrize("network,dataset", NETS[:5], ids=lambda x: x)
def test_gpu_matches_faithful_scalar_oracle(network, dataset):
    """a few images against the one-MAC-at-a-time restatement (not the popcount one)"""
    imgs = rand_images(network, 3, 11)
    g = gpu_net(network, dataset).raw(imgs)
    o = oracle(network, dataset)
    for i in range(3):
        if o.is_cnv:
            assert g[i].tolist() == o.scores_ref(imgs[i]).tolist()
        else:
            assert int(g[i]) == o.word_ref(imgs[i])


@pytest.mark.parametrize("network,dataset", NETS, ids=lambda x: x)
def test_batched_classes(network, dataset):
    ncls = ol.num_classes(dataset, network)
    n = 1000
    imgs = rand_images(network, n, 5)
    got = gpu_net(network, dataset).classify(imgs, ncls)
    want = oracle(network, dataset).classes_batched(imgs, ncls)
    assert got.tolist() == want.tolist()


@pytest.mark.parametrize("network", ["cnvW1A1", "cnvW1A2", "cnvW2A2", "lfcW1A1", "lfcW1A2"])
@pytest.mark.parametrize("seed", [3, 4])
def test_random_params_bit_exact(network, seed, tmp_path):
    """random weights/thresholds in the reference file format (unsorted threshold pairs, extremes):
    every compare outcome of every layer gets exercised, not only what the trained nets hit"""
    import random_params
    random_params.make(str(tmp_path), network, seed)
    L = gl.load(network)
    L.load_parameters(str(tmp_path).encode())
    assert L.bnn_mi355x_last_error() == b""
    _nets.pop(network, None)                      # the cached Net of this library now holds other params
    net = gl.Net.__new__(gl.Net)
    net.L, net.network, net.is_cnv, net.isz = L, network, network.startswith("cnv"), L.bnn_mi355x_image_bytes()
    n = 300 if net.is_cnv else 1500
    imgs = rand_images(network, n, seed)
    o = ol.Oracle(network, str(tmp_path))
    ref = o.scores_fast(imgs) if o.is_cnv else o.words_fast(imgs)
    got = net.raw(imgs)
    assert (got == ref).all()
    assert np.unique(ref).size > 8


def test_weights_of_minus_two_all_paths(tmp_path):
    """cnvW2A2 rows holding the weight -2 (fault-injection territory): small batch (pixel lanes + one-launch
    tail), mid batch (quad kernels, 8-neuron form) and a wide batch, raw scores against the oracle -- whose
    fast path is itself checked against the faithful one in test_oracle_selfcheck.py"""
    import random_params
    random_params.make(str(tmp_path), "cnvW2A2", 5, neg2=0.03)
    L = gl.load("cnvW2A2")
    L.load_parameters(str(tmp_path).encode())
    assert L.bnn_mi355x_last_error() == b""
    _nets.pop("cnvW2A2", None)
    net = gl.Net.__new__(gl.Net)
    net.L, net.network, net.is_cnv, net.isz = L, "cnvW2A2", True, 3072
    o = ol.Oracle("cnvW2A2", str(tmp_path))
    for n in (3, 700, 1500, 20000):
        imgs = rand_images("cnvW2A2", n, n)
        got = net.raw(imgs)
        pick = np.arange(n) if n <= 1500 else np.random.default_rng(1).choice(n, 400, replace=False)
        assert (got[pick] == o.scores_fast(imgs[pick])).all(), n
    assert (net.raw(imgs[:2])[0] == o.scores_ref(imgs[0])).all()          # and the faithful scalar path
    L.load_parameters(gl.param_dir("cifar10", "cnvW2A2").encode())


def test_detail_scores():
    ncls = 10
    imgs = rand_images("cnvW1A1", 77, 3)
    got = gpu_net("cnvW1A1", "cifar10").classify(imgs, ncls, detail=True).reshape(77, ncls)
    want = oracle("cnvW1A1", "cifar10").scores_fast(imgs)[:, :ncls]
    assert (got == want).all()


def test_empty_and_single():
    net = gpu_net("cnvW1A1", "cifar10")
    assert net.classify(np.zeros((0, 3072), np.uint8), 10).size == 0
    one = rand_images("cnvW1A1", 1, 1)
    assert net.raw(one)[0].tolist() == oracle("cnvW1A1", "cifar10").scores_ref(one[0]).tolist()


EXP = json.load(open(os.path.join(ol.GOLDEN, "expected.json")))


@pytest.mark.parametrize("e", EXP["scores"], ids=lambda e: e["input"] + "-" + e["network"])
def test_recorded_scores_through_file_abi(e):
    """the reference's recorded score vectors, through inference(path, results, ...)"""
    net = gpu_net(e["network"], e["params"])
    res = (C.c_int * 64)()
    usec = C.c_float(0)
    cls = net.L.inference(os.path.join(ol.GOLDEN, e["input"]).encode(), res, 10, C.byref(usec))
    assert list(res[:10]) == e["scores"], e["source"]
    assert cls == int(np.argmax(e["scores"]))
    assert usec.value > 0


@pytest.mark.parametrize("e", EXP["classes"], ids=lambda e: e["input"] + "-" + e["network"] + "-" + e["params"])
def test_recorded_classes_through_file_abi(e):
    net = gpu_net(e["network"], e["params"])
    ncls = ol.num_classes(e["params"], e["network"])
    path = os.path.join(ol.GOLDEN, e["input"]).encode()
    assert net.L.inference(path, None, ncls, None) == e["class"], e["source"]
    n = C.c_int(0)
    usec = C.c_float(0)
    p = net.L.inference_multiple(path, ncls, C.byref(n), C.byref(usec), 0)
    assert n.value == 1 and p[0] == e["class"], e["source"]
    net.L.free_results(p)


def test_lfc_single_one_hot():
    net = gpu_net("lfcW1A1", "mnist")
    res = (C.c_int * 64)()
    cls = net.L.inference(os.path.join(ol.GOLDEN, "3.image-idx3-ubyte").encode(), res, 10, None)
    assert cls == 3 and list(res) == [1 if i == 3 else 0 for i in range(64)]


def test_full_size_batch_properties():
    """BASELINE-size batch (10 000 CIFAR images): oracle on a seeded sample, plus
    size-independent properties -- batch-order equivariance and chunk independence."""
    n = 10000
    imgs = rand_images("cnvW1A1", n, 1)
    net = gpu_net("cnvW1A1", "cifar10")
    s = net.raw(imgs)
    perm = np.random.default_rng(0).permutation(n)
    assert (net.raw(imgs[perm]) == s[perm]).all()
    pick = np.random.default_rng(1).choice(n, 400, replace=False)
    assert (oracle("cnvW1A1", "cifar10").scores_fast(imgs[pick]) == s[pick]).all()
    # the same image repeated gives the same scores wherever it sits in the batch
    rep = np.repeat(imgs[:1], 700, axis=0)
    assert (net.raw(rep) == s[0]).all()


@pytest.mark.parametrize("network", ["cnvW1A1", "cnvW1A2", "cnvW2A2"])
def test_large_batch_takes_the_wide_paths(network):
    """60 000 images through the host-buffer entry point = chunks of up to 16 384: the wider conv stages run in their
    'all neuron groups in one block' form (gpb_for in kernels.hip), the 500-image call below in the 8-neuron
    form; both must agree with each other and with a seeded oracle sample (the full-size device-API test
    covers the wide form of the remaining stages)"""
    n = 60000
    imgs = rand_images(network, n, 21)
    net = gpu_net(network, "cifar10")
    big = net.raw(imgs)
    pick = np.random.default_rng(2).choice(n, 256, replace=False)
    assert (oracle(network, "cifar10").scores_fast(imgs[pick]) == big[pick]).all()
    assert (net.raw(imgs[:500]) == big[:500]).all()       # narrow path (500 images) == wide path


def test_lfc_full_size_batch():
    n = 10000
    for network in ("lfcW1A1", "lfcW1A2"):
        imgs = rand_images(network, n, 0)
        got = gpu_net(network, "mnist").raw(imgs)
        assert (got == oracle(network, "mnist").words_fast(imgs)).all()


def test_device_api_and_blob_roundtrip():
    """images resident in HBM (torch is only the allocator here) + export/import of the packed blob"""
    import torch
    net = gpu_net("cnvW1A1", "cifar10")
    n = 5000
    imgs = rand_images("cnvW1A1", n, 9)
    d = torch.from_numpy(imgs).cuda()
    cls = torch.zeros(n, dtype=torch.int32, device="cuda")
    sc = torch.zeros(n, 64, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    assert net.L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, cls.data_ptr(), sc.data_ptr(), None, st) == 0
    torch.cuda.synchronize()
    o = oracle("cnvW1A1", "cifar10")
    want = o.scores_fast(imgs)
    assert (sc.cpu().numpy() == want).all()
    assert cls.cpu().numpy().tolist() == o.classes_batched(imgs, 10).tolist()
    # blob export -> import gives the same network
    size = net.L.bnn_mi355x_export_params(None, 0)
    blob = np.zeros(size, np.uint8)
    assert net.L.bnn_mi355x_export_params(blob.ctypes.data, size) == size
    assert (blob == gl.pack_params("cnvW1A1", gl.param_dir("cifar10", "cnvW1A1"))).all()
    assert net.L.bnn_mi355x_import_params(blob.ctypes.data, size) == 0
    assert (net.raw(imgs[:64]) == want[:64]).all()


def test_device_api_crosses_the_chunk_boundary():
    """more images than one pass holds (131 072): the device entry point walks the batch in chunks;
    results must not depend on where the boundary falls"""
    import torch
    net = gpu_net("cnvW1A1", "cifar10")
    n = 131072 + 777
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    d = torch.randint(0, 256, (n, 3072), dtype=torch.uint8, device="cuda", generator=g)
    cls = torch.zeros(n, dtype=torch.int32, device="cuda")
    sc = torch.zeros(n, 64, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    assert net.L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, cls.data_ptr(), sc.data_ptr(), None, None) == 0
    torch.cuda.synchronize()
    pick = np.concatenate([np.arange(131072 - 150, 131072 + 150), np.array([0, n - 1]), np.random.default_rng(3).choice(n, 100)])
    host = d[torch.from_numpy(pick).cuda()].cpu().numpy()
    o = oracle("cnvW1A1", "cifar10")
    assert (sc.cpu().numpy()[pick] == o.scores_fast(host)).all()
    assert cls.cpu().numpy()[pick].tolist() == o.classes_batched(host, 10).tolist()
    # the tail chunk alone gives the same answers
    tail = d[131072:].contiguous()
    cls2 = torch.zeros(n - 131072, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert net.L.bnn_mi355x_inference_device(tail.data_ptr(), n - 131072, 10, cls2.data_ptr(), None, None, None) == 0
    torch.cuda.synchronize()
    assert (cls2 == cls[131072:]).all()


@pytest.mark.parametrize("network", ["lfcW1A1", "lfcW1A2"])
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 128, 129, 256, 257, 511, 512, 513, 1023, 1025, 2047, 2048, 2049, 4093, 4096, 4097])
def test_lfc_small_batches_take_the_fused_kernel(network, n):
    """<= 1024 (lfcW1A2: 2048) images: the LFC nets run as one launch, a block per group of 1/2/4/8 images (k_lfc_fused<IPB>,
    k_lfc_fused_a2<IPB>); sizes either side of every policy edge, ragged last groups included; beyond, lfcW1A1 takes
    k_lfc_block_s (1025 images: three per block) and lfcW1A2 the staged path"""
    import torch
    net = gpu_net(network, "mnist")
    o = oracle(network, "mnist")
    for kind in ("uniform", "sparse"):
        imgs = rand_images(network, n, 30 + n, kind)
        assert (net.raw(imgs) == o.words_fast(imgs)).all()
        d = torch.from_numpy(imgs).cuda()
        cls = torch.zeros(n, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        assert net.L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, cls.data_ptr(), None, None, None) == 0
        torch.cuda.synchronize()
        assert cls.cpu().numpy().tolist() == o.classes_batched(imgs, 10).tolist()


def test_lfc_device_decode_refuses_label_sets_it_cannot_decode_exactly():
    """the device-side LFC decode is an exact floor(log2); the reference's (unsigned) log2((double) word) rounds up
    for some words of 48 and more bits (e.g. 2^49 - 1): beyond 47 classes the device entry point refuses classes
    (words are still available) instead of returning a decode that could differ"""
    import torch
    net = gpu_net("lfcW1A1", "mnist")
    d = torch.zeros((4, 784), dtype=torch.uint8, device="cuda")
    cls = torch.zeros(4, dtype=torch.int32, device="cuda")
    words = torch.zeros(4, dtype=torch.int64, device="cuda")
    assert net.L.bnn_mi355x_inference_device(d.data_ptr(), 4, 48, cls.data_ptr(), None, None, None) != 0
    assert b"number_class <= 47" in net.L.bnn_mi355x_last_error()
    assert net.L.bnn_mi355x_inference_device(d.data_ptr(), 4, 48, None, None, words.data_ptr(), None) == 0
    assert net.L.bnn_mi355x_inference_device(d.data_ptr(), 4, 47, cls.data_ptr(), None, None, None) == 0
    torch.cuda.synchronize()
    o = oracle("lfcW1A1", "mnist")
    assert (words.cpu().numpy().view(np.uint64) == o.words_fast(np.zeros((4, 784), np.uint8))).all()


def test_lfc_device_decode():
    import torch
    net = gpu_net("lfcW1A1", "mnist")
    n = 3000
    imgs = rand_images("lfcW1A1", n, 4, "sparse")
    d = torch.from_numpy(imgs).cuda()
    cls = torch.zeros(n, dtype=torch.int32, device="cuda")
    words = torch.zeros(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert net.L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, cls.data_ptr(), None, words.data_ptr(), None) == 0
    torch.cuda.synchronize()
    o = oracle("lfcW1A1", "mnist")
    assert (words.cpu().numpy().view(np.uint64) == o.words_fast(imgs)).all()
    assert cls.cpu().numpy().tolist() == o.classes_batched(imgs, 10).tolist()


def test_abi_edge_behaviour(tmp_path):
    """NULL out-params, deinit() between calls, bad inputs: errors are reported, never thrown across the ABI"""
    net = gpu_net("cnvW1A1", "cifar10")
    L = net.L
    deer = os.path.join(ol.GOLDEN, "deer.cifar").encode()
    p = L.inference_multiple(deer, 10, None, None, 0)            # both out-params NULL (bnn.py passes pointers, C callers may not)
    assert p and p[0] == 4
    L.free_results(p)
    L.deinit()                                                   # frees the I/O workspace only (FoldedMVDeinit): weights stay
    assert L.inference(deer, None, 10, None) == 4
    L.deinit()
    L.deinit()                                                   # idempotent
    n = C.c_int(-1)
    assert not L.inference_multiple(b"/nonexistent/file.bin", 10, C.byref(n), None, 0)
    assert b"Could not open file" in L.bnn_mi355x_last_error()
    assert L.inference(b"/nonexistent/file.bin", None, 10, None) == -1
    assert not L.bnn_mi355x_inference_buffer(None, 5, 10, None, 0)
    imgs = rand_images("cnvW1A1", 4, 2)
    assert not L.bnn_mi355x_inference_buffer(imgs.ctypes.data, 4, 0, None, 0)      # number_class out of 1..64
    assert not L.bnn_mi355x_inference_buffer(imgs.ctypes.data, 4, 65, None, 0)
    empty = tmp_path / "empty.bin"
    empty.write_bytes(b"")
    p = L.inference_multiple(str(empty).encode(), 10, C.byref(n), None, 0)          # zero records: an empty result, not an error
    assert p and n.value == 0
    L.free_results(p)
    assert L.inference(str(empty).encode(), None, 10, None) == -1
    short = tmp_path / "short.bin"
    short.write_bytes(open(os.path.join(ol.GOLDEN, "deer.cifar"), "rb").read() + b"\x01" * 100)   # trailing partial record ignored
    p = L.inference_multiple(str(short).encode(), 10, C.byref(n), None, 0)
    assert n.value == 1 and p[0] == 4
    L.free_results(p)
    assert net.raw(imgs).shape == (4, 64)                        # still healthy afterwards


@pytest.mark.parametrize("network,dataset,n", [("cnvW1A1", "cifar10", 70001), ("cnvW2A2", "cifar10", 33000),
                                               ("lfcW1A1", "mnist", 100003), ("lfcW1A2", "mnist", 40000)])
def test_file_abi_streams_multi_chunk_files(network, dataset, n, tmp_path):
    """inference_multiple on files of many chunks (512 ... 16 384 records; the LFC nets: binarised on the host, 8 192 ...
    32 768 images): reader threads fill a ring of pinned pieces, dropping the label byte of every record on the way; same
    classes as the in-memory entry point, detail scores included; inference() classifies the first record only"""
    net = gpu_net(network, dataset)
    L = net.L
    cnv = network.startswith("cnv")
    imgs = rand_images(network, n, 77, "uniform")
    path = tmp_path / "big.bin"
    with open(path, "wb") as f:
        if cnv:
            rec = np.empty((n, 3073), np.uint8)
            rec[:, 0] = np.arange(n) % 10
            rec[:, 1:] = imgs
            f.write(rec.tobytes())
        else:
            f.write((0x803).to_bytes(4, "big") + n.to_bytes(4, "big") + (28).to_bytes(4, "big") * 2)
            f.write(imgs.tobytes())
    want = net.classify(imgs, 10)
    cnt, usec = C.c_int(0), C.c_float(0)
    p = L.inference_multiple(str(path).encode(), 10, C.byref(cnt), C.byref(usec), 0)
    assert p and cnt.value == n and usec.value > 0
    got = np.ctypeslib.as_array(p, (n,)).copy()
    L.free_results(p)
    assert (got == want).all()
    # both entry points cut the call by the same plan (ramped chunks): the images either side of every chunk edge,
    # against the oracle, which knows nothing of chunks
    bases = (C.c_int * 64)()
    edges = set()
    for from_file in (0, 1):      # `want` came through the host-buffer plan, `got` through the file plan
        k = L.bnn_mi355x_chunk_plan(n, from_file, bases, 64)
        e = [bases[i] for i in range(k)]
        assert 3 <= k <= 64 and e[0] == 0 and e[-1] == n and all(0 < b - a <= 32768 for a, b in zip(e, e[1:]))
        edges |= set(e)
    near = sorted({min(max(e + d, 0), n - 1) for e in edges for d in (-2, -1, 0, 1)})
    assert (got[near] == oracle(network, dataset).classes_batched(imgs[near], 10)).all()
    if cnv:
        p = L.inference_multiple(str(path).encode(), 10, C.byref(cnt), None, 1)
        det = np.ctypeslib.as_array(p, (n * 10,)).copy().reshape(n, 10)
        L.free_results(p)
        assert (det == net.raw(imgs)[:, :10]).all()
        res = (C.c_int * 64)()
        cls = L.inference(str(path).encode(), res, 10, None)
        assert list(res[:10]) == det[0].tolist() and cls == int(np.argmax(det[0]))
    else:
        res = (C.c_int * 64)()
        cls = L.inference(str(path).encode(), res, 10, None)
        assert sum(res) <= 1 and (sum(res) == 0 or res[cls] == 1)
    # a second, smaller file right after a big one (buffers are reused)
    small = tmp_path / "small.bin"
    with open(small, "wb") as f:
        if cnv:
            f.write(rec[:5].tobytes())
        else:
            f.write((0x803).to_bytes(4, "big") + (5).to_bytes(4, "big") + (28).to_bytes(4, "big") * 2)
            f.write(imgs[:5].tobytes())
    p = L.inference_multiple(str(small).encode(), 10, C.byref(cnt), None, 0)
    assert cnt.value == 5 and list(p[:5]) == want[:5].tolist()
    L.free_results(p)


@pytest.mark.parametrize("variant,dataset", [("cnvW1A1-TMR", "cifar10"), ("cnvW2A2-resilient-interleaved", "cifar10"),
                                             ("lfcW1A2-interleaved", "mnist")])
def test_hardened_variants_are_the_base_network(variant, dataset, tmp_path, variant_libs):
    """cnvW1A1-TMR & co (bnn.py:41-53): same classes as the base network through the Python API; fault
    injection is refused for them (their memory organisation is not modelled), not silently approximated"""
    import sys
    sys.path.insert(0, os.path.join(gl.ROOT, "bnn-pynq_amd"))
    import bnn
    base = variant.split("-")[0]
    cnv = base.startswith("cnv")
    imgs = rand_images(base, 300, 5)
    path = tmp_path / "in.bin"
    with open(path, "wb") as f:
        if cnv:
            rec = np.zeros((300, 3073), np.uint8)
            rec[:, 1:] = imgs
            f.write(rec.tobytes())
        else:
            f.write((0x803).to_bytes(4, "big") + (300).to_bytes(4, "big") + (28).to_bytes(4, "big") * 2 + imgs.tobytes())
    Clf = bnn.CnvClassifier if cnv else bnn.LfcClassifier
    v = Clf(variant, dataset, bnn.RUNTIME_SW)
    got = v.classify_cifars(str(path)) if cnv else v.classify_mnists(str(path))
    assert list(got) == oracle(base, dataset).classes_batched(imgs, len(v.classes)).tolist()
    assert v.bnn.interface.bnn_mi355x_network().decode() == variant
    n = C.c_int(0)
    p = v.bnn.interface.inference_multiple_with_faults(str(path).encode(), 10, C.byref(n), None, 5, 1, -1, None, 0)
    assert not p and b"not modelled" in v.bnn.interface.bnn_mi355x_last_error()
    p = v.bnn.interface.inference_multiple_with_faults(str(path).encode(), 10, C.byref(n), None, 0, 1, -1, None, 0)   # no flips: plain inference
    assert p and n.value == 300
    v.bnn.interface.free_results(p)


def test_layer0_integer_pipe_kernel_agrees():
    """BNN_MI355X_L0=valu selects k_conv0 (v_dot4c) instead of the MFMA first layer: same bits"""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); import torch, gpu_lib as gl, oracle_lib as ol\n"
        "for net in ('cnvW1A1', 'cnvW2A2'):\n"
        "    imgs = np.random.default_rng(5).integers(0, 256, (700, 3072), dtype=np.uint8)\n"
        "    got = gl.Net(net, 'cifar10').raw(imgs)\n"
        "    assert (got == ol.Oracle(net, ol.param_dir('cifar10', net)).scores_fast(imgs)).all(), net\n"
        "print('valu-l0-ok')\n" % os.path.join(gl.ROOT, "tests"))
    env = dict(os.environ, BNN_MI355X_L0="valu")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "valu-l0-ok" in out.stdout, out.stdout + out.stderr


def test_layer0_threshold_above_the_trained_range_with_weights_of_minus_two(tmp_path):
    """cnvW2A2 layer 0 under fault injection: a row of -2 weights reaches |dot| up to 2 * 27 * 128, and a
    faulted threshold can lie above 27 * 128.  The matrix-pipe table clamps thresholds to the REACHABLE range
    of the dot product, which depends on the weight range: rows 0..2 hold 14 x (-2) + 1 x (-1) taps, i.e.
    dot = 3712 on an all-black picture (q = -128), against thresholds 4000 (must not fire), 3711 (fires) and
    3712 (strict compare: does not).  Both forms of layer 0 against the faithful scalar restatement."""
    import subprocess
    import sys

    import random_params
    from bnn import params_io
    from test_gpu_layers import stage_output, unpack
    W, T = random_params.make(str(tmp_path), "cnvW2A2", 17)
    for row, t in enumerate((4000, 3711, 3712)):
        W[0][row, :] = 0
        W[0][row, :14] = -2
        W[0][row, 14] = -1
        T[0][row, :] = (2 * t, 2 * t + 1)        # file units: 2^-8, accumulator = 2 * dot; blob threshold = floor(T / 2)
    params_io.write_params(str(tmp_path), "cnvW2A2", W, T)
    imgs = np.zeros((2, 3072), np.uint8)
    imgs[1] = np.random.default_rng(3).integers(0, 256, 3072)
    o = ol.Oracle("cnvW2A2", str(tmp_path))
    want = [o.layer_ref(imgs[i], 0) for i in range(2)]
    assert want[0].reshape(900, 64)[0, :3].tolist() == [-1, 1, -1]   # the crafted rows: no threshold, both, none
    L = gl.load("cnvW2A2")
    L.load_parameters(str(tmp_path).encode())
    _nets.pop("cnvW2A2", None)
    raw = stage_output(L, imgs, 0)
    for i in range(2):
        assert (unpack(raw[i], 900, 64, 2) == want[i]).all()
    # the integer-pipe form of layer 0 (selected by the environment at load time): same bits
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r]\n"
            "import torch, gpu_lib as gl\nfrom test_gpu_layers import stage_output\n"
            "L = gl.load('cnvW2A2'); L.load_parameters(%r.encode())\n"
            "imgs = np.load(%r)\nnp.save(%r, stage_output(L, imgs, 0))\n"
            % (os.path.join(gl.ROOT, "tests"), os.path.join(gl.ROOT, "bnn-pynq_amd"), str(tmp_path),
               str(tmp_path / "imgs.npy"), str(tmp_path / "valu.npy")))
    np.save(tmp_path / "imgs.npy", imgs)
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, BNN_MI355X_L0="valu"))
    assert (np.load(tmp_path / "valu.npy") == raw).all()
    L.load_parameters(gl.param_dir("cifar10", "cnvW2A2").encode())


@pytest.mark.parametrize("network,dataset", [("lfcW1A1", "mnist"), ("lfcW1A2", "mnist"), ("lfcW1A1", "chars_merged")])
def test_lfc_single_image_decode_on_words_with_several_bits(network, dataset, tmp_path):
    """inference() on an LFC net decodes with round(log2(word)) and writes a 64-entry one-hot
    (foldedmv-offload.cpp:152-165); the batched entry points use floor(log2) (:213-220).  The two differ exactly
    when the word has more than one bit inside number_class, e.g. 0b1100 -> 4 vs 3 -- and the one-hot index can
    then be a class that no neuron voted for.  Images are searched with the oracle until their output words
    have >= 2 bits set and the two decodes disagree; both entry points are compared on them."""
    net = gpu_net(network, dataset)
    o = oracle(network, dataset)
    ncls = ol.num_classes(dataset, network)
    rng = np.random.default_rng(77)
    found, agreeing, tries = [], [], 0
    while len(found) < 6 and tries < 12:
        tries += 1
        imgs = np.where(rng.random((4096, 784)) < rng.uniform(0.02, 0.6), rng.integers(128, 256, (4096, 784)), rng.integers(0, 128, (4096, 784))).astype(np.uint8)
        words = o.words_fast(imgs)
        mask = np.uint64((1 << ncls) - 1)
        for i in np.flatnonzero([bin(int(w & mask)).count("1") >= 2 for w in words]):
            w = int(words[i])
            if o.L.bnn_oracle_decode_lfc_single(w, ncls) != o.L.bnn_oracle_decode_lfc_batched(w, ncls):
                found.append((imgs[i].copy(), w))
                if len(found) == 6:
                    break
            elif len(agreeing) < 4 and w not in [x[1] for x in agreeing]:
                agreeing.append((imgs[i].copy(), w))
    if dataset == "mnist":
        assert len(found) >= 3, "no output words on which the two decodes differ"
    n_differ = len(found)
    found += agreeing            # chars_merged (47 classes): random pictures reach few words; multi-bit ones still
    assert len(found) >= 3, "no multi-bit output words found"
    res = (C.c_int * 64)()
    usec = C.c_float(0)
    for k, (img, w) in enumerate(found):
        path = tmp_path / ("m%d.idx3" % k)
        with open(path, "wb") as f:       # two images: inference() must classify the FIRST only
            f.write((0x803).to_bytes(4, "big") + (2).to_bytes(4, "big") + (28).to_bytes(4, "big") * 2 + img.tobytes() + bytes(784))
        cls = net.L.inference(str(path).encode(), res, ncls, C.byref(usec))
        hot = o.L.bnn_oracle_lfc_single_hot(w, ncls)
        assert cls == o.L.bnn_oracle_decode_lfc_single(w, ncls) == (hot if hot < 64 else 0)
        assert list(res) == [1 if i == hot else 0 for i in range(64)]
        cnt = C.c_int(0)
        p = net.L.inference_multiple(str(path).encode(), ncls, C.byref(cnt), None, 0)
        assert cnt.value == 2 and p[0] == o.L.bnn_oracle_decode_lfc_batched(w, ncls)
        assert (p[0] != cls) == (k < n_differ)
        net.L.free_results(p)
        assert net.L.inference(str(path).encode(), None, ncls, None) == cls      # results may be NULL (bnn.py:133)


def test_device_calls_on_two_streams_share_the_workspace_safely():
    """bnn_mi355x_inference_device on alternating streams: one activation workspace per library, so a call must
    first wait for the previous call's kernels on the other stream (include/bnn_mi355x.h); results equal the
    single-stream ones and the oracle's.  Also: params_crc of what the GPU holds, import straight from HBM."""
    import torch
    net = gpu_net("cnvW1A1", "cifar10")
    L = net.L
    o = oracle("cnvW1A1", "cifar10")
    n = 3000
    batches = [rand_images("cnvW1A1", n, 900 + k) for k in range(4)]
    dev = [torch.from_numpy(b).cuda() for b in batches]
    out = [torch.full((n,), -1, dtype=torch.int32, device="cuda") for _ in range(4)]
    streams = [torch.cuda.Stream().cuda_stream, None, torch.cuda.Stream().cuda_stream]   # None: the null stream
    keep = streams  # (the Stream objects' handles stay valid while the process lives; torch caches its streams)
    assert L.bnn_mi355x_reserve(n) == 0
    torch.cuda.synchronize()
    for rep in range(3):
        for k in range(4):
            s = streams[(rep * 4 + k) % 3]
            assert L.bnn_mi355x_inference_device(dev[k].data_ptr(), n, 10, out[k].data_ptr(), None, None, s) == 0
    torch.cuda.synchronize()
    for k in range(4):
        assert (out[k].cpu().numpy() == o.classes_batched(batches[k], 10)).all(), k
    # the blob as the GPU holds it: CRC equal to the host blob's, and importable from a device tensor
    import zlib
    size = L.bnn_mi355x_params_bytes()
    host = np.zeros(size, np.uint8)
    assert L.bnn_mi355x_export_params(host.ctypes.data, size) == size
    assert L.bnn_mi355x_params_crc() == zlib.crc32(host.tobytes())
    d_blob = torch.from_numpy(host).cuda()
    assert L.bnn_mi355x_import_params_device(d_blob.data_ptr(), size, torch.cuda.current_stream().cuda_stream) == 0
    assert L.bnn_mi355x_params_crc() == zlib.crc32(host.tobytes())
    assert (net.classify(batches[0][:500], 10) == o.classes_batched(batches[0][:500], 10)).all()
    bad = host.copy()
    bad[24] ^= 0xFF                                                       # l0_mfma_offset
    d_bad = torch.from_numpy(bad).cuda()
    assert L.bnn_mi355x_import_params_device(d_bad.data_ptr(), size, None) != 0 and b"mismatch" in L.bnn_mi355x_last_error()
    assert L.bnn_mi355x_import_params_device(d_blob.data_ptr(), size - 1, None) != 0
    assert (net.classify(batches[1][:200], 10) == o.classes_batched(batches[1][:200], 10)).all()   # still loaded


@pytest.mark.parametrize("n", [1025, 1537, 2048, 3001, 4097, 4351, 4352, 5000, 10000, 16383, 32768, 32769, 70000])
def test_lfc_mid_batches_take_the_block_kernel(n):
    """lfcW1A1 beyond the one-launch small-batch kernel (<= 1024 images), up to one pass of 131 072 images:
    one k_lfc_block_s launch, a 1024-thread block per ceil(n / 512) images walking all four layers with the
    activations fed through SGPRs (BASELINE config 2 is 10 000 images).  Sizes either side of the policy edges,
    ragged last blocks, raw words through the host path and classes through the device path."""
    import torch
    net = gpu_net("lfcW1A1", "mnist")
    o = oracle("lfcW1A1", "mnist")
    imgs = rand_images("lfcW1A1", n, 50 + n, "uniform" if n % 2 else "sparse")
    want = o.words_fast(imgs)
    assert (net.raw(imgs) == want).all()
    d = torch.from_numpy(imgs).cuda()
    cls = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    words = torch.zeros(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert net.L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, cls.data_ptr(), None, words.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert (words.cpu().numpy().view(np.uint64) == want).all()
    assert cls.cpu().numpy().tolist() == o.classes_batched(imgs, 10).tolist()


def test_lfc_block_kernel_beyond_its_policy_range():
    """k_lfc_block_s with more than 64 images per block (several chunks of 64 per layer, ragged last chunk and last
    block), which the shipped policy never asks of it: forced with BNN_MI355X_LFC_BLOCK_MAX"""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path[:0] = [%r, %r]\n"
        "import torch, gpu_lib as gl, oracle_lib as ol\n"
        "net = gl.Net('lfcW1A1', 'mnist'); o = ol.Oracle('lfcW1A1', ol.param_dir('mnist', 'lfcW1A1'))\n"
        "for n in (32769, 70001, 131072):\n"      # through the device entry point: the host path feeds chunks of 32 768
        "    imgs = np.random.default_rng(n).integers(0, 256, (n, 784), dtype=np.uint8)\n"
        "    d = torch.from_numpy(imgs).cuda(); w = torch.zeros(n, dtype=torch.int64, device='cuda'); c = torch.zeros(n, dtype=torch.int32, device='cuda')\n"
        "    torch.cuda.synchronize()\n"
        "    assert net.L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, c.data_ptr(), None, w.data_ptr(), None) == 0\n"
        "    torch.cuda.synchronize()\n"
        "    assert (w.cpu().numpy().view(np.uint64) == o.words_fast(imgs)).all(), n\n"
        "    assert (c.cpu().numpy() == o.classes_batched(imgs, 10)).all(), n\n"
        "print('block-ok')\n" % (os.path.join(gl.ROOT, "tests"), os.path.join(gl.ROOT, "bnn-pynq_amd")))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BNN_MI355X_LFC_BLOCK_MAX="1000000"), capture_output=True, text=True, timeout=600)
    assert "block-ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
    # ... and the staged kernels (one launch per layer, a lane per image), which the policy now only runs for per-stage
    # profiling: the same batches with the block kernel switched off
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BNN_MI355X_LFC_BLOCK_MAX="0"), capture_output=True, text=True, timeout=600)
    assert "block-ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_set_device_is_idempotent_once_bound():
    """a second set_device to the ordinal the library is bound to answers 0 (bench.py's other_configs reuse a loaded
    library); any other ordinal is an error"""
    net = gpu_net("lfcW1A1", "mnist")
    L = net.L
    assert L.bnn_mi355x_set_device(0) == 0
    assert L.bnn_mi355x_set_device(1) != 0 and b"bound to device 0" in L.bnn_mi355x_last_error()
    imgs = rand_images("lfcW1A1", 100, 1)
    assert (net.raw(imgs) == oracle("lfcW1A1", "mnist").words_fast(imgs)).all()


def test_device_entry_point_refuses_misaligned_pointers():
    """the kernels read images with 128-bit loads: d_images must be 16-byte aligned (include/bnn_mi355x.h); a
    misaligned pointer is refused before anything is launched, an aligned view of the same data is classified"""
    import torch
    for network, dataset, isz in (("cnvW1A1", "cifar10", 3072), ("lfcW1A1", "mnist", 784)):
        net = gpu_net(network, dataset)
        n = 300
        imgs = rand_images(network, n, 8)
        raw = torch.zeros(n * isz + 64, dtype=torch.uint8, device="cuda")
        cls = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
        for off in (1, 4, 8):
            raw[off:off + n * isz] = torch.from_numpy(imgs.reshape(-1)).cuda()
            assert net.L.bnn_mi355x_inference_device(raw.data_ptr() + off, n, 10, cls.data_ptr(), None, None, None) != 0
            assert b"16-byte aligned" in net.L.bnn_mi355x_last_error()
        raw[16:16 + n * isz] = torch.from_numpy(imgs.reshape(-1)).cuda()
        assert net.L.bnn_mi355x_inference_device(raw.data_ptr() + 16, n, 10, cls.data_ptr() + 2, None, None, None) != 0   # classes: 4 bytes
        assert net.L.bnn_mi355x_inference_device(raw.data_ptr() + 16, n, 10, cls.data_ptr() + 4, None, None, None) == 0
        torch.cuda.synchronize()
        assert int(cls[0]) == -1 and cls[1:].cpu().numpy().tolist() == oracle(network, dataset).classes_batched(imgs, 10).tolist()


def test_device_call_captured_into_a_graph_after_a_call_on_another_stream():
    """the header promises bnn_mi355x_inference_device is graph-capturable once the workspace is reserved.  The hard
    case: an earlier call on stream A is still in flight (it owns the shared workspace) when a call on stream B is
    CAPTURED -- the hand-over must be resolved before the capture starts recording (host-side wait), not by a
    stream-wait on an event from outside the capture.  The graph is then replayed on fresh inputs."""
    import torch
    net = gpu_net("cnvW1A1", "cifar10")
    L = net.L
    o = oracle("cnvW1A1", "cifar10")
    n = 2000
    a_host, b_host, c_host = (rand_images("cnvW1A1", n, 700 + k) for k in range(3))
    d_a = torch.from_numpy(a_host).cuda()
    d_b = torch.from_numpy(b_host).cuda()
    out_a = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    out_b = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    assert L.bnn_mi355x_reserve(n) == 0
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(4):   # keep stream A busy with the workspace
        assert L.bnn_mi355x_inference_device(d_a.data_ptr(), n, 10, out_a.data_ptr(), None, None, sa.cuda_stream) == 0
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=sb):
        rc = L.bnn_mi355x_inference_device(d_b.data_ptr(), n, 10, out_b.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L.bnn_mi355x_last_error()
    torch.cuda.synchronize()
    assert (out_a.cpu().numpy() == o.classes_batched(a_host, 10)).all()
    graph.replay()
    torch.cuda.synchronize()
    assert (out_b.cpu().numpy() == o.classes_batched(b_host, 10)).all()
    d_b.copy_(torch.from_numpy(c_host).cuda())          # same graph, new images in the captured buffer
    graph.replay()
    torch.cuda.synchronize()
    assert (out_b.cpu().numpy() == o.classes_batched(c_host, 10)).all()
    # and an ordinary call afterwards still works
    assert L.bnn_mi355x_inference_device(d_a.data_ptr(), n, 10, out_a.data_ptr(), None, None, None) == 0
    torch.cuda.synchronize()
    assert (out_a.cpu().numpy() == o.classes_batched(a_host, 10)).all()


@pytest.mark.parametrize("n", [681, 683, 1025, 7811, 9999])
def test_file_abi_either_side_of_the_pinned_ring_threshold(n, tmp_path):
    """files below 2 MB (682 CIFAR records) go through a pageable host chunk, larger ones through the ring of pinned
    pieces that worker threads fill with pread() (feed_chunks: a first chunk in 256 KB pieces, the calling thread filling
    the first of them itself): every class of an odd-sized file on both sides, against the oracle; twice in a row (the
    ring's slots and the workers are reused)"""
    net = gpu_net("cnvW1A1", "cifar10")
    imgs = rand_images("cnvW1A1", n, 4000 + n)
    rec = np.empty((n, 3073), np.uint8)
    rec[:, 0] = 3
    rec[:, 1:] = imgs
    path = tmp_path / "f.bin"
    path.write_bytes(rec.tobytes())
    want = oracle("cnvW1A1", "cifar10").classes_batched(imgs, 10)
    for _ in range(2):
        cnt = C.c_int(0)
        p = net.L.inference_multiple(str(path).encode(), 10, C.byref(cnt), None, 0)
        assert p and cnt.value == n
        got = np.ctypeslib.as_array(p, (n,)).copy()
        net.L.free_results(p)
        assert (got == want).all()


def _one_lane_reference(net, imgs):
    """raw outputs (CNV: scores [n,64], LFC: words [n]) and classes of `imgs` through bnn_mi355x_inference_device in slices
    of 8 000 images on the null stream: below every fork / lane limit, i.e. the plain one-stream path that the other tests
    compare with the restatement image by image"""
    import torch
    n = imgs.shape[0]
    d = torch.from_numpy(imgs).cuda()
    cls = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    raw = torch.zeros((n, 64), dtype=torch.int16, device="cuda") if net.is_cnv else torch.zeros(n, dtype=torch.int64, device="cuda")
    for b in range(0, n, 8000):
        m = min(8000, n - b)
        rc = net.L.bnn_mi355x_inference_device(d.data_ptr() + b * net.isz, m, 10, cls.data_ptr() + 4 * b,
                                               raw.data_ptr() + 128 * b if net.is_cnv else None,
                                               None if net.is_cnv else raw.data_ptr() + 8 * b, None)
        assert rc == 0
    torch.cuda.synchronize()
    r = raw.cpu().numpy()
    return (r if net.is_cnv else r.view(np.uint64)), cls.cpu().numpy()


def _oracle_sample(o, imgs, classes, edges, seed):
    """the restatement on the images either side of `edges` and on 200 random ones"""
    n = imgs.shape[0]
    pick = sorted({min(max(e + k, 0), n - 1) for e in edges for k in (-2, -1, 0, 1)} | set(np.random.default_rng(seed).choice(n, 200, replace=False).tolist()))
    assert (classes[pick] == o.classes_batched(imgs[pick], 10)).all()


def test_host_paths_on_two_compute_lanes(tmp_path):
    """a host-path call of three or more chunks alternates them over two streams, each with its own activation workspace and
    pair of staging buffers (runtime.hip, "Two compute lanes"): classes / raw outputs of EVERY image equal to the one-stream device
    path's (slices of 8 000 images), and the restatement's on the images at every chunk edge and a random sample, for the
    file and the buffer entry points -- CNV (label bytes stripped on the device) and LFC, ragged last chunk, repeated calls
    (the lanes' buffers are reused) --, `usecPerImage` the union of the chunks' device intervals (positive, below the wall
    time); the same with one lane forced (BNN_MI355X_LANES=1) and with the pinned ring off"""
    import subprocess
    import sys
    import time
    for network, dataset, n in (("cnvW1A1", "cifar10", 23001), ("cnvW2A2", "cifar10", 20011), ("lfcW1A1", "mnist", 70003)):
        net, o = gpu_net(network, dataset), oracle(network, dataset)
        k = C.c_int(0)
        bases = (C.c_int * 64)()
        imgs = rand_images(network, n, 77)
        want_raw, want = _one_lane_reference(net, imgs)
        edges = set()
        for from_file in (0, 1):
            nb = net.L.bnn_mi355x_chunk_plan(n, from_file, bases, 64)
            assert nb >= 4                                            # three or more chunks: two lanes (files: ring-fed)
            edges |= {bases[i] for i in range(nb)}
        _oracle_sample(o, imgs, want, edges, n)
        path = str(tmp_path / (network + ".bin"))
        with open(path, "wb") as f:
            if net.is_cnv:
                r = np.empty((n, 3073), np.uint8); r[:, 0] = 3; r[:, 1:] = imgs; f.write(r.tobytes())
            else:
                f.write((0x803).to_bytes(4, "big") + n.to_bytes(4, "big") + (28).to_bytes(4, "big") * 2 + imgs.tobytes())
        for rep in range(3):
            usec = C.c_float(0)
            t0 = time.perf_counter()
            p = net.L.inference_multiple(path.encode(), 10, C.byref(k), C.byref(usec), 0)
            wall_us = (time.perf_counter() - t0) * 1e6
            assert p and k.value == n
            got = np.ctypeslib.as_array(p, (n,)).copy()
            net.L.free_results(p)
            assert (got == want).all(), (network, "file", rep)
            assert 0 < usec.value * n < wall_us, (network, usec.value * n, wall_us)
        for rep in range(2):
            assert (net.raw(imgs) == want_raw).all(), (network, "buffer", rep)
    code = (
        "import sys, ctypes as C, numpy as np; sys.path[:0] = [%r, %r]\n"
        "import gpu_lib as gl, oracle_lib as ol\n"
        "net = gl.Net('cnvW1A1', 'cifar10'); o = ol.Oracle('cnvW1A1', ol.param_dir('cifar10', 'cnvW1A1'))\n"
        "imgs = np.random.default_rng(5).integers(0, 256, (12001, 3072), dtype=np.uint8)\n"
        "want = o.scores_fast(imgs)\n"
        "for rep in range(2): assert (net.raw(imgs) == want).all()\n"
        "r = np.empty((12001, 3073), np.uint8); r[:, 0] = 1; r[:, 1:] = imgs; open(%r, 'wb').write(r.tobytes())\n"
        "k = C.c_int(0); p = net.L.inference_multiple(%r, 10, C.byref(k), None, 0)\n"
        "assert p and (np.ctypeslib.as_array(p, (12001,)) == o.classes_batched(imgs, 10)).all()\n"
        "print('lanes-ok')\n" % (os.path.join(gl.ROOT, "tests"), os.path.join(gl.ROOT, "bnn-pynq_amd"), str(tmp_path / "f.bin"), str(tmp_path / "f.bin").encode()))
    # one lane forced; the pinned ring switched off (the file then streams through pageable host chunks, on two lanes)
    for knob in ({"BNN_MI355X_LANES": "1"}, {"BNN_MI355X_NO_FEEDER": "1"}):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **knob), capture_output=True, text=True, timeout=600)
        assert "lanes-ok" in out.stdout, str(knob) + out.stdout[-1500:] + out.stderr[-3000:]


@pytest.mark.parametrize("network", ["cnvW1A1", "cnvW2A2"])
def test_device_call_forks_over_two_lanes(network):
    """a device-pointer pass of 16 384 CNV images and more runs its halves on two streams with two activation workspaces
    and joins them before the caller's stream goes on (bnn_mi355x_inference_device): scores and classes of EVERY image
    equal to the one-stream path's (slices of 8 000 images), and the restatement's around the split and on a random
    sample, either side of the limit and at ragged sizes; calls alternating over three caller streams without host
    synchronisation between them (each must wait for BOTH lanes of the one before); a host-buffer call issued while
    forked device work is still in flight (its second lane shares the second workspace); the same batch through a
    captured graph (one lane) gives the same classes"""
    import torch
    net, o = gpu_net(network, "cifar10"), oracle(network, "cifar10")
    L = net.L
    for n in (16383, 16384, 16385, 20011, 33001):
        imgs = rand_images(network, n, 500 + n % 7)
        want_sc, want = _one_lane_reference(net, imgs)
        d = torch.from_numpy(imgs).cuda()
        cls = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        sc = torch.zeros((n, 64), dtype=torch.int16, device="cuda")
        assert L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, cls.data_ptr(), sc.data_ptr(), None, None) == 0
        torch.cuda.synchronize()
        assert (sc.cpu().numpy() == want_sc).all(), (network, n)
        assert (cls.cpu().numpy() == want).all(), (network, n)
        _oracle_sample(o, imgs, want, {((n // 2) + 255) & ~255}, n)
    n = 20000
    batches = [rand_images(network, n, 40 + k) for k in range(3)]
    want = [_one_lane_reference(net, b)[1] for b in batches]
    dev = [torch.from_numpy(b).cuda() for b in batches]
    out = [torch.full((n,), -1, dtype=torch.int32, device="cuda") for _ in range(3)]
    streams = [torch.cuda.Stream().cuda_stream, None, torch.cuda.Stream().cuda_stream]
    torch.cuda.synchronize()
    for rep in range(4):
        for k in range(3):
            s = streams[(rep + k) % 3]
            assert L.bnn_mi355x_inference_device(dev[k].data_ptr(), n, 10, out[k].data_ptr(), None, None, s) == 0
    # no synchronisation: the host-buffer call below must itself wait for the forked work on both of its lanes
    host = rand_images(network, 9001, 77)
    assert (net.raw(host) == o.scores_fast(host)).all()
    torch.cuda.synchronize()
    for k in range(3):
        assert (out[k].cpu().numpy() == want[k]).all(), (network, k)
    # captured into a graph (one lane while capturing) and replayed
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    cap = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    assert L.bnn_mi355x_reserve(n) == 0              # (the forked calls above sized the first workspace for a half)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        assert L.bnn_mi355x_inference_device(dev[0].data_ptr(), n, 10, cap.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream) == 0
    g.replay()
    torch.cuda.synchronize()
    assert (cap.cpu().numpy() == want[0]).all()
