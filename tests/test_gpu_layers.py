"""GPU, per-kernel parity: the output of EVERY stage -- sliding-window + MVAU +
threshold (+ max-pool) -- as it sits bit-packed in HBM, against the faithful scalar restatement's
activations after the same layer, for all five networks, shipped and random parameters."""
import numpy as np
import pytest

import gpu_lib as gl
import oracle_lib as ol

pytestmark = pytest.mark.gpu

NETS = [("cnvW1A1", "cifar10"), ("cnvW1A2", "cifar10"), ("cnvW2A2", "cifar10"), ("lfcW1A1", "mnist"), ("lfcW1A2", "mnist")]
CNV_SHAPE = [(900, 64), (196, 64), (144, 128), (25, 128), (9, 256), (1, 256), (1, 512), (1, 512)]  # pixels, channels


def unpack(raw, pixels, channels, planes):
    """HBM layout -> value-domain array [pixels * channels] (pixel-major, like oracle.layer_ref)"""
    if planes == 1:   # [pixel][C/32] dwords, bit c = fired (+1)
        bits = np.unpackbits(raw.view(np.uint8), bitorder="little").reshape(pixels, channels)
        return np.where(bits == 1, 1, -1).astype(np.int8).reshape(-1)
    w = raw.view(np.uint64).reshape(pixels, channels // 64, 2)      # [pixel][C/64][sign, non-zero]
    sign = np.unpackbits(np.ascontiguousarray(w[:, :, 0]).view(np.uint8), bitorder="little").reshape(pixels, channels)
    nz = np.unpackbits(np.ascontiguousarray(w[:, :, 1]).view(np.uint8), bitorder="little").reshape(pixels, channels)
    return np.where(nz == 1, np.where(sign == 1, -1, 1), 0).astype(np.int8).reshape(-1)


def stage_output(L, imgs, stage):
    n = imgs.shape[0]
    buf = np.zeros(n * 16384, np.uint8)
    per = L.bnn_mi355x_debug_stage_output(imgs.ctypes.data, n, stage, buf.ctypes.data, buf.size)
    assert per > 0, L.bnn_mi355x_last_error()
    return buf[: n * per].reshape(n, per)


def check_all_stages(network, pdir, seed):
    L = gl.load(network)
    L.load_parameters(pdir.encode())
    assert L.bnn_mi355x_last_error() == b""
    o = ol.Oracle(network, pdir)
    planes = 2 if network.endswith("A2") else 1
    rng = np.random.default_rng(seed)
    if o.is_cnv:
        imgs = rng.integers(0, 256, (3, 3072), dtype=np.uint8)
        for stage, (pixels, channels) in enumerate(CNV_SHAPE):
            raw = stage_output(L, imgs, stage)
            for i in range(len(imgs)):
                want = o.layer_ref(imgs[i], stage)
                got = unpack(raw[i], pixels, channels, planes)
                assert got.size == want.size and (got == want).all(), "stage %d image %d" % (stage, i)
    else:
        imgs = rng.integers(0, 256, (40, 784), dtype=np.uint8)
        raw = stage_output(L, imgs, 0)                              # binarizeAndPack
        bits = np.unpackbits(raw, axis=1, bitorder="little")
        assert (bits[:, :784] == (imgs >= 128)).all() and (bits[:, 784:] == 0).all()
        for layer in range(3):
            raw = stage_output(L, imgs, layer + 1)
            for i in range(len(imgs)):
                assert (unpack(raw[i], 1, 1024, planes) == o.layer_ref(imgs[i], layer)).all(), "layer %d image %d" % (layer, i)


@pytest.mark.parametrize("network,dataset", NETS, ids=lambda x: x)
def test_every_stage_shipped_params(network, dataset):
    check_all_stages(network, gl.param_dir(dataset, network), 41)


@pytest.mark.parametrize("network,dataset", NETS, ids=lambda x: x)
def test_every_stage_random_params(network, dataset, tmp_path):
    import random_params
    random_params.make(str(tmp_path), network, 9)
    check_all_stages(network, str(tmp_path), 42)
    gl.load(network).load_parameters(gl.param_dir(dataset, network).encode())   # leave the library as found


def test_every_stage_with_weights_of_minus_two(tmp_path):
    """cnvW2A2 with 4 % of the 2-bit weight fields = 0b10 (-2): the value a bit flip makes out of a 0 or a -1;
    the reference multiplies by it (ap_int<2>), so must every stage -- checked against the FAITHFUL scalar
    restatement, which uses the integer weights"""
    import random_params
    W, _ = random_params.make(str(tmp_path), "cnvW2A2", 12, neg2=0.04)
    assert all((w == -2).any() for w in W)
    check_all_stages("cnvW2A2", str(tmp_path), 43)
    gl.load("cnvW2A2").load_parameters(gl.param_dir("cifar10", "cnvW2A2").encode())


@pytest.mark.parametrize("mode", ["mfma", "lds"])
def test_layer1_matrix_pipe_experiment_is_bit_exact(tmp_path, mode):
    """BNN_MI355X_L1=mfma (side experiment): cnvW1A1 layer 1 as an FP4 implicit GEMM on the matrix
    cores; BNN_MI355X_L1=lds (comparison figure): the same layer in the north-star's literal wording (LDS-staged
    weights, __popcll, shuffle pooling).  Same bits as the XNOR-popcount kernel: the stage's HBM output against the faithful scalar restatement,
    shipped and random parameters (incl. never / always firing thresholds), odd and even image counts, and the
    whole network's raw scores on a batch that spans several blocks."""
    import os
    import subprocess
    import sys

    import random_params
    random_params.make(str(tmp_path), "cnvW1A1", 23)
    code = (
        "import sys, numpy as np; sys.path[:0] = [%r, %r]\n"
        "import torch, gpu_lib as gl, oracle_lib as ol\n"
        "from test_gpu_layers import stage_output, unpack\n"
        "L = gl.load('cnvW1A1')\n"
        "for pdir in (gl.param_dir('cifar10', 'cnvW1A1'), %r):\n"
        "    L.load_parameters(pdir.encode()); assert L.bnn_mi355x_last_error() == b''\n"
        "    o = ol.Oracle('cnvW1A1', pdir)\n"
        "    for n in (1, 2, 5):\n"
        "        imgs = np.random.default_rng(60 + n).integers(0, 256, (n, 3072), dtype=np.uint8)\n"
        "        raw = stage_output(L, imgs, 1)\n"
        "        for i in range(n):\n"
        "            assert (unpack(raw[i], 196, 64, 1) == o.layer_ref(imgs[i], 1)).all(), (pdir, n, i)\n"
        "    imgs = np.random.default_rng(8).integers(0, 256, (3001, 3072), dtype=np.uint8)\n"
        "    net = gl.Net.__new__(gl.Net); net.L, net.network, net.is_cnv, net.isz = L, 'cnvW1A1', True, 3072\n"
        "    assert (net.raw(imgs) == o.scores_fast(imgs)).all(), pdir\n"
        "print('alt-l1-ok')\n" % (os.path.join(gl.ROOT, "tests"), os.path.join(gl.ROOT, "bnn-pynq_amd"), str(tmp_path)))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BNN_MI355X_L1=mode), capture_output=True, text=True, timeout=900)
    assert "alt-l1-ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_layer0_tile_form_is_bit_exact(tmp_path):
    """k_conv0_tile (layer 0 from LDS-staged, once-quantised images: the form used from 2048 images on), forced
    onto small batches with BNN_MI355X_L0_TILE_MIN=1 so that the faithful scalar restatement can check every bit
    of the stage's output: three CNV nets, shipped and random parameters, weights of -2, ragged blocks (1..17
    images against blocks of 8), and the whole network's scores."""
    import os
    import subprocess
    import sys

    import random_params
    dirs = {}
    for k, net in enumerate(("cnvW1A1", "cnvW1A2", "cnvW2A2")):
        d = tmp_path / net
        d.mkdir()
        random_params.make(str(d), net, 30 + k, **({"neg2": 0.04} if net == "cnvW2A2" else {}))
        dirs[net] = str(d)
    code = (
        "import sys, numpy as np; sys.path[:0] = [%r, %r]\n"
        "import torch, gpu_lib as gl, oracle_lib as ol\n"
        "from test_gpu_layers import stage_output, unpack\n"
        "dirs = %r\n"
        "for net in ('cnvW1A1', 'cnvW1A2', 'cnvW2A2'):\n"
        "    L = gl.load(net); planes = 2 if net.endswith('A2') else 1\n"
        "    for pdir in (gl.param_dir('cifar10', net), dirs[net]):\n"
        "        L.load_parameters(pdir.encode()); assert L.bnn_mi355x_last_error() == b''\n"
        "        o = ol.Oracle(net, pdir)\n"
        "        for n in (1, 7, 8, 9, 17):\n"
        "            imgs = np.random.default_rng(70 + n).integers(0, 256, (n, 3072), dtype=np.uint8)\n"
        "            imgs[0, :] = np.random.default_rng(n).choice(np.array([0, 1, 127, 128, 254, 255], np.uint8), 3072)\n"
        "            raw = stage_output(L, imgs, 0)\n"
        "            for i in sorted({0, n // 2, n - 1}):\n"
        "                assert (unpack(raw[i], 900, 64, planes) == o.layer_ref(imgs[i], 0)).all(), (net, pdir, n, i)\n"
        "        imgs = np.random.default_rng(9).integers(0, 256, (1003, 3072), dtype=np.uint8)\n"
        "        net_ = gl.Net.__new__(gl.Net); net_.L, net_.network, net_.is_cnv, net_.isz = L, net, True, 3072\n"
        "        assert (net_.raw(imgs) == o.scores_fast(imgs)).all(), (net, pdir)\n"
        "print('tile-l0-ok')\n" % (os.path.join(gl.ROOT, "tests"), os.path.join(gl.ROOT, "bnn-pynq_amd"), dirs))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BNN_MI355X_L0_TILE_MIN="1"), capture_output=True, text=True, timeout=900)
    assert "tile-l0-ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("network", ["cnvW1A1", "cnvW1A2", "cnvW2A2"])
def test_layer0_forms_agree_across_the_policy_edge(network):
    """8191 images run layer 0 with a lane per pixel (k_conv0_mfma), 8192 and more from LDS-staged images
    (k_conv0_tile, blocks of 8 images; 8195 leaves a block of 3): stage-0 bits of the same images equal in both"""
    L = gl.load(network)
    L.load_parameters(gl.param_dir("cifar10", network).encode())
    imgs = np.random.default_rng(77).integers(0, 256, (8195, 3072), dtype=np.uint8)
    tile = stage_output(L, imgs, 0)
    pixel = stage_output(L, imgs[:8191], 0)
    assert (tile[:8191] == pixel).all()
    assert (stage_output(L, imgs[8184:], 0) == tile[8184:]).all()       # the ragged block's images, as a small batch
