"""GPU: the entry points that take HOST data, round 4 -- the LFC nets' inputs binarised on the host side of the copy
(104 bytes per image over PCIe, like the reference's binarizeAndPack), small calls with no transfer at all (pinned,
device-mapped I/O), results landing in pinned memory, the copies of a host buffer on a helper thread, the file readers
dropping the label bytes.  Everything against the CPU restatement, through the C ABI."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import gpu_lib as gl
import oracle_lib as ol
from test_gpu_parity import gpu_net, oracle, rand_images

pytestmark = pytest.mark.gpu


def write_file(path, imgs, cnv, labels=None):
    n = imgs.shape[0]
    with open(path, "wb") as f:
        if cnv:
            rec = np.empty((n, 3073), np.uint8)
            rec[:, 0] = (np.arange(n) * 7 + 3) % 251 if labels is None else labels   # label bytes that would show in the pixels
            rec[:, 1:] = imgs
            f.write(rec.tobytes())
        else:
            f.write((0x803).to_bytes(4, "big") + n.to_bytes(4, "big") + (28).to_bytes(4, "big") * 2 + imgs.tobytes())


def classify_file(L, path, n, ncls=10, detail=0):
    cnt, usec = C.c_int(0), C.c_float(0)
    p = L.inference_multiple(str(path).encode(), ncls, C.byref(cnt), C.byref(usec), detail)
    assert p and cnt.value == n, L.bnn_mi355x_last_error()
    out = np.ctypeslib.as_array(p, (n * (ncls if detail else 1),)).copy()
    L.free_results(p)
    return out, usec.value


@pytest.mark.parametrize("network", ["lfcW1A1", "lfcW1A2"])
def test_lfc_host_paths_ship_binarised_words(network, tmp_path):
    """inference_multiple(path), inference_buffer and inference_raw binarise on the host (csrc/pack_inputs.cpp on worker
    threads; up to 1 024 images: on the calling thread, the kernels reading the words in pinned memory) -- the raw output
    WORD of every image equal to the restatement's and to the device-pointer path's, which binarises on the GPU: sizes either
    side of the direct / ring / chunk limits, pixel values at the decision level"""
    import torch
    net, o = gpu_net(network, "mnist"), oracle(network, "mnist")
    L = net.L
    for n, kind in ((1, "edges"), (2, "edges"), (1023, "uniform"), (1024, "edges"), (1025, "edges"), (4097, "sparse"), (16384, "uniform"),
                    (16385, "edges"), (40001, "uniform")):
        imgs = rand_images(network, n, 900 + n % 13, kind)
        want = o.words_fast(imgs)
        assert (net.raw(imgs) == want).all(), (network, n, "buffer")
        d = torch.from_numpy(imgs).cuda()
        w = torch.zeros(n, dtype=torch.int64, device="cuda")
        assert L.bnn_mi355x_inference_device(d.data_ptr(), n, 10, None, None, w.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert (w.cpu().numpy().view(np.uint64) == want).all(), (network, n, "device")
        path = tmp_path / "in.idx3"
        write_file(path, imgs, False)
        got, usec = classify_file(L, path, n)
        assert (got == o.classes_batched(imgs, 10)).all() and usec > 0, (network, n, "file")


def test_single_image_calls_need_no_transfer(tmp_path):
    """inference(path) -- what classify_image and the webcam loops call per frame -- places the record in pinned memory the
    GPU addresses and reads scores / the output word back from pinned memory: every image of a file classified one by one
    equals the batched call's scores (CNV) / the restatement's word (LFC); usecPerImage is positive and below the wall time"""
    import time
    for network, dataset in (("cnvW1A1", "cifar10"), ("cnvW2A2", "cifar10"), ("lfcW1A1", "mnist"), ("lfcW1A2", "mnist")):
        net, o = gpu_net(network, dataset), oracle(network, dataset)
        cnv = net.is_cnv
        imgs = rand_images(network, 12, 31, "edges" if not cnv else "uniform")
        ref = o.scores_fast(imgs) if cnv else o.words_fast(imgs)
        for i in range(12):
            path = tmp_path / ("one%d.bin" % i)
            write_file(path, imgs[i:i + 1], cnv, labels=np.array([200 + i], np.uint8))
            res = (C.c_int * 64)()
            usec = C.c_float(0)
            t = time.perf_counter()
            cls = net.L.inference(str(path).encode(), res, 10, C.byref(usec))
            wall = (time.perf_counter() - t) * 1e6
            assert 0 < usec.value < wall
            if cnv:
                assert list(res[:10]) == ref[i, :10].tolist() and cls == int(np.argmax(ref[i, :10]))
            else:
                hot = ol.lib().bnn_oracle_lfc_single_hot(int(ref[i]), 10)
                assert cls == hot and [int(x) for x in res] == [1 if j == hot else 0 for j in range(64)]
        # single images through the host-buffer entry points take the same way
        for i in range(3):
            assert (net.raw(imgs[i:i + 1]) == ref[i:i + 1]).all()


SWITCH_CODE = """
import sys, ctypes as C, numpy as np
sys.path[:0] = [%(tests)r, %(pkg)r]
import gpu_lib as gl, oracle_lib as ol
for network, dataset, n in (("cnvW1A1", "cifar10", 9001), ("lfcW1A1", "mnist", 33003), ("lfcW1A1", "mnist", 700), ("cnvW1A1", "cifar10", 1)):
    net = gl.Net(network, dataset); o = ol.Oracle(network, ol.param_dir(dataset, network))
    imgs = np.random.default_rng(n).integers(0, 256, (n, net.isz), dtype=np.uint8)
    want = o.scores_fast(imgs) if net.is_cnv else o.words_fast(imgs)
    for rep in range(2):
        assert (net.raw(imgs) == want).all(), (network, n, "buffer")
    path = %(dir)r + "/" + network + str(n)
    with open(path, "wb") as f:
        if net.is_cnv:
            r = np.empty((n, 3073), np.uint8); r[:, 0] = 9; r[:, 1:] = imgs; f.write(r.tobytes())
        else:
            f.write((0x803).to_bytes(4, "big") + n.to_bytes(4, "big") + (28).to_bytes(4, "big") * 2 + imgs.tobytes())
    k = C.c_int(0); p = net.L.inference_multiple(path.encode(), 10, C.byref(k), None, 0)
    assert p and k.value == n and (np.ctypeslib.as_array(p, (n,)) == o.classes_batched(imgs, 10)).all(), (network, n, "file")
    net.L.free_results(p)
print("switch-ok")
"""


@pytest.mark.parametrize("knob", ["BNN_MI355X_NO_HOST_PACK=1", "BNN_MI355X_NO_DIRECT=1", "BNN_MI355X_NO_COPIER=1", "BNN_MI355X_NO_MAPPED_RESULTS=1",
                                  "BNN_MI355X_NO_FEEDER=1", "BNN_MI355X_DIRECT_TIMING=host"])
def test_host_path_switches_change_nothing_but_the_route(knob, tmp_path):
    """the A/B switches of the round-4 host paths (raw pixels to HBM instead of host-binarised words; no direct small calls; copies
    and launches on one thread; results through HBM and a copy; no pinned ring; single images timed by the host and waited for
    on a completion word in pinned memory instead of events): same bits either way"""
    code = SWITCH_CODE % {"tests": os.path.join(gl.ROOT, "tests"), "pkg": os.path.join(gl.ROOT, "bnn-pynq_amd"), "dir": str(tmp_path)}
    k, v = knob.split("=")
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{k: v}), capture_output=True, text=True, timeout=900)
    assert "switch-ok" in out.stdout, knob + out.stdout[-1500:] + out.stderr[-3000:]


def test_single_images_timed_by_the_host(tmp_path):
    """BNN_MI355X_DIRECT_TIMING=host: inference(path) of every network with the completion word instead of events -- recorded
    fixtures (deer -> 4 with its recorded scores, 3.image -> 3), usecPerImage positive and below the call's wall time, raw
    scores of a random image through the host-buffer entry point"""
    code = """
import sys, os, time, ctypes as C, numpy as np
sys.path[:0] = [%r, %r]
import gpu_lib as gl, oracle_lib as ol
G = ol.GOLDEN
for network, dataset, f, want in (("cnvW1A1", "cifar10", "deer.cifar", 4), ("cnvW1A2", "cifar10", "deer.cifar", 4), ("cnvW2A2", "cifar10", "deer.cifar", 4),
                                  ("lfcW1A1", "mnist", "3.image-idx3-ubyte", 3), ("lfcW1A2", "mnist", "3.image-idx3-ubyte", 3)):
    net = gl.Net(network, dataset)
    res, usec = (C.c_int * 64)(), C.c_float(0)
    for rep in range(5):
        t = time.perf_counter()
        cls = net.L.inference(os.path.join(G, f).encode(), res, 10, C.byref(usec))
        wall = (time.perf_counter() - t) * 1e6
        assert cls == want and 0 < usec.value < wall, (network, cls, usec.value, wall)
    if network == "cnvW1A1":
        assert list(res[:10]) == [234, 231, 265, 248, 410, 257, 224, 262, 226, 233]
net = gl.Net("cnvW2A2", "cifar10")
o = ol.Oracle("cnvW2A2", ol.param_dir("cifar10", "cnvW2A2"))
img = np.random.default_rng(3).integers(0, 256, (1, 3072), dtype=np.uint8)
assert (net.raw(img) == o.scores_fast(img)).all()
print("host-timed-ok")
""" % (os.path.join(gl.ROOT, "tests"), os.path.join(gl.ROOT, "bnn-pynq_amd"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BNN_MI355X_DIRECT_TIMING="host"), capture_output=True, text=True, timeout=900)
    assert "host-timed-ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_chunk_override_above_the_workspace_is_cut_on_the_gpu(tmp_path):
    """BNN_MI355X_CHUNKS with a head of 131 072 on a call of 140 000 images used to give ONE chunk of 140 000 -- past the
    activation workspace (round-3 advisor finding).  Now two chunks; classes of a sample either side of the cut against the
    restatement.  And a 3-chunk lfcW1A1 call on one lane reports a usecPerImage that is the SUM of its chunks' device
    times: within 35 % of three single-chunk calls of the same size (the events around a launch must belong to that launch)."""
    code = """
import sys, os, ctypes as C, numpy as np
sys.path[:0] = [%r, %r]
import gpu_lib as gl, oracle_lib as ol
os.environ["BNN_MI355X_CHUNKS"] = "131072:0:131072"
net = gl.Net("cnvW1A1", "cifar10"); o = ol.Oracle("cnvW1A1", ol.param_dir("cifar10", "cnvW1A1"))
n = 140000
imgs = np.random.default_rng(1).integers(0, 256, (n, 3072), dtype=np.uint8)
b = (C.c_int * 16)(); k = net.L.bnn_mi355x_chunk_plan(n, 0, b, 16)
assert [b[i] for i in range(k)] == [0, 131072, 140000]
got = net.classify(imgs, 10)
pick = sorted(set(range(0, 64)) | set(range(131072 - 64, 131072 + 64)) | set(range(n - 64, n)))
assert (got[pick] == o.classes_batched(imgs[pick], 10)).all()
os.environ["BNN_MI355X_CHUNKS"] = "16384:0:16384"
lfc = gl.Net("lfcW1A1", "mnist")
px = np.random.default_rng(2).integers(0, 256, (49152, 784), dtype=np.uint8)
for _ in range(3): lfc.raw(px[:16384])
one = []
for _ in range(5):
    lfc.raw(px[:16384]); one.append(lfc.usec * 16384)
three = []
for _ in range(5):
    lfc.raw(px); three.append(lfc.usec * 49152)
one, three = sorted(one)[2], sorted(three)[2]
assert 0.65 * 3 * one < three < 1.35 * 3 * one, (one, three)
print("override-ok", one, three)
""" % (os.path.join(gl.ROOT, "tests"), os.path.join(gl.ROOT, "bnn-pynq_amd"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BNN_MI355X_LANES="1"), capture_output=True, text=True, timeout=900)
    assert "override-ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_file_records_with_every_label_byte(tmp_path):
    """the file readers scatter the records (preadv) so that only the image bodies reach the pinned ring: a file whose label
    bytes take every value, ragged sizes around the readers' batches of 512 records and the pieces' 85 / 170 records, detail
    scores of every image equal to the host-buffer path's"""
    net = gpu_net("cnvW1A1", "cifar10")
    for n in (683, 1024 + 511, 5000):
        imgs = rand_images("cnvW1A1", n, 5000 + n)
        path = tmp_path / ("l%d.bin" % n)
        write_file(path, imgs, True, labels=(np.arange(n) % 256).astype(np.uint8))
        det, _ = classify_file(net.L, path, n, detail=1)
        assert (det.reshape(n, 10) == net.raw(imgs)[:, :10]).all(), n


@pytest.mark.parametrize("network", ["cnvW1A1", "cnvW1A2", "cnvW2A2"])
def test_a_few_cifar_images_from_a_host_buffer_are_read_in_place(network):
    """up to 32 CIFAR images handed over in host memory take the direct way (copied into the pinned I/O block, read there by the
    kernels, scores / classes written back into it): raw scores, batched classes and detail scores either side of the limit,
    against the restatement; repeated calls (the block is reused)"""
    net, o = gpu_net(network, "cifar10"), oracle(network, "cifar10")
    for n in (2, 3, 7, 31, 32, 33):
        imgs = rand_images(network, n, 70 + n, "edges" if n % 2 else "uniform")
        want = o.scores_fast(imgs)
        for rep in range(2):
            assert (net.raw(imgs) == want).all(), (network, n)
        assert (net.classify(imgs, 10) == o.classes_batched(imgs, 10)).all(), (network, n)
        assert (net.classify(imgs, 10, detail=True).reshape(n, 10) == want[:, :10]).all(), (network, n)
