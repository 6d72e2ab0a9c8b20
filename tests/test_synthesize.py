"""bnn/params_io.py, float model -> thresholds (the second half of finnthesizer.py, "parity unpinned": the
trained .npz archives are missing blobs in the reference, the tool is Python 2).  Tested by construction:
a random float network -- binarised / ternarised weights, bias, batch norm, sign or 3-level activation,
max-pool BEFORE the batch norm as in the training graph (bnn/src/training/cnv.py:37-239, lfc.py:38-104) --
is evaluated in floating point and must make exactly the decisions the integer runtime makes on the
parameter files derived from it.  CPU: against the oracle, all five topologies.  GPU: the product."""
import numpy as np
import pytest

import oracle_lib as ol
from bnn import params_io

NETS = ["lfcW1A1", "lfcW1A2", "cnvW1A1", "cnvW1A2", "cnvW2A2"]
CONV = [(3, 64, False), (64, 64, True), (64, 128, False), (128, 128, True), (128, 256, False), (256, 256, False)]  # in, out, pool
CNV_FC = [(256, 512), (512, 512), (512, 10)]
LFC_FC = [(784, 1024), (1024, 1024), (1024, 1024), (1024, 10)]


def random_float_net(network, seed, negative_scales="exact"):
    """layer dicts as the training scripts store them.  negative_scales: "exact" -- a third of the neurons get
    gamma < 0 in the layers whose conversion is exact for them (the all-1-bit FC layers: makeFCBNComplex's
    popcount form); "all" -- in every unpooled layer (see test_negative_scale_is_off_by_one_at_the_tie)."""
    rng = np.random.default_rng(seed)
    cnv = network.startswith("cnv")
    shapes = [("conv", i, o) for i, o, _ in CONV] + [("fc", i, o) for i, o in CNV_FC] if cnv else [("fc", i, o) for i, o in LFC_FC]
    layers = []
    for l, (kind, cin, cout) in enumerate(shapes):
        fanin = cin * 9 if kind == "conv" else cin
        W = rng.uniform(-1, 1, size=(cout, cin, 3, 3) if kind == "conv" else (cin, cout))
        d = {"W": W}
        if not (cnv and l == len(shapes) - 1):
            sigma = np.sqrt(fanin) * (0.6 if (cnv and l == 0) else 1.0) * (0.8 if "W2" in network else 1.0)
            d["bias"] = rng.normal(0, 0.1, cout)
            d["mean"] = rng.normal(0, 0.4 * sigma, cout)
            d["invstd"] = rng.uniform(0.6, 1.6, cout) / sigma
            d["gamma"] = rng.uniform(0.5, 1.5, cout) * np.where(rng.random(cout) < 0.33, -1.0, 1.0)
            if negative_scales == "exact" and not (kind == "fc" and network.endswith("W1A1")):
                # conv and multi-threshold layers: the reference rounds a flipped neuron's threshold with
                # ceil(-t) (finnthesizer.py:231,287,302) where the strict compare T < acc would need floor(-t):
                # one accumulator step off, see the tie test below
                d["gamma"] = np.abs(d["gamma"])
            if kind == "conv" and CONV[l][2]:
                # The accelerator thresholds first and pools the decisions (top.cpp:214-223); the training graph
                # pools first.  The two agree for a neuron whose batch-norm scale is positive (thresholding is then
                # monotone increasing); a negative scale in a pooled layer would turn the max into a min, which no
                # threshold file can express -- the reference's conversion has the same limit.
                d["gamma"] = np.abs(d["gamma"])
            d["beta"] = rng.normal(0, 0.4, cout)
            if negative_scales == "exact" and kind == "fc" and network.endswith("A2") and l == len(shapes) - 1:
                # single-threshold FC layer on signed sums (lfcW1A2's last): the reference stores int(t), which
                # truncates toward zero (finnthesizer.py:193) -- exact for the strict compare only while t >= 0
                d["mean"], d["beta"], d["bias"] = np.abs(d["mean"]) + 1.0, -np.abs(d["beta"]), -np.abs(d["bias"])
        layers.append(d)
    return layers


def quant_w(W, network):
    return np.where(W >= 0, 1.0, -1.0) if "W1" in network else np.floor(W + 0.5)


def act(y, a2):
    """sign (1-bit) or the 3-level quantiser with decision levels -0.5 / +0.5; also the distance of y to the
    nearest decision level (a float evaluation is only a judge away from ties)"""
    if not a2:
        return np.where(y > 0, 1.0, -1.0), np.abs(y).min()
    return -1.0 + (y > -0.5) + (y > 0.5), np.minimum(np.abs(y + 0.5), np.abs(y - 0.5)).min()


def float_forward(network, layers, imgs):
    """(final outputs [n, 10], smallest distance to a decision level seen anywhere)"""
    cnv, a2 = network.startswith("cnv"), network.endswith("A2")
    margin = np.inf
    n = imgs.shape[0]

    def bn(y, d):  # y [..., C]
        return (y + d["bias"] - d["mean"]) * d["invstd"] * d["gamma"] + d["beta"]

    if cnv:
        q = np.clip(np.floor(256.0 * imgs.astype(np.float64) / 255.0 - 128 + 0.5), -128, 127) / 128.0  # ap_fixed<8,1>
        x = q.reshape(n, 3, 32, 32).transpose(0, 2, 3, 1)  # NHWC
        for l, (cin, cout, pool) in enumerate(CONV):
            d = layers[l]
            Wq = quant_w(d["W"], network).transpose(2, 3, 1, 0).reshape(9 * cin, cout)  # (ky, kx, c) x out
            H = x.shape[1] - 2
            cols = np.concatenate([x[:, ky:ky + H, kx:kx + H, :] for ky in range(3) for kx in range(3)], axis=3)
            y = cols.reshape(-1, 9 * cin) @ Wq
            y = y.reshape(n, H, H, cout)
            if pool:  # training graph: conv -> max-pool -> batch norm -> activation
                y = y.reshape(n, H // 2, 2, H // 2, 2, cout).max(axis=(2, 4))
            x, m = act(bn(y, d), a2)
            margin = min(margin, m)
        x = x.reshape(n, 256)
        fcs = layers[6:]
    else:
        x = np.where(imgs >= 128, 1.0, -1.0)
        fcs = layers
    for l, d in enumerate(fcs):
        y = x @ quant_w(d["W"], network)
        if "gamma" not in d:
            return y, margin  # CNV layer 8: raw sums
        last = l == len(fcs) - 1
        x, m = act(bn(y, d), a2 and not last)
        margin = min(margin, m)
    return x, margin


def expected_outputs(network, layers, imgs):
    out, margin = float_forward(network, layers, imgs)
    assert margin > 1e-7, "float model too close to a decision level for an exact comparison"
    return out


def integer_outputs(network, raw, n_real=10):
    """runtime / oracle raw outputs -> the float model's units, real neurons only"""
    if network.startswith("cnv"):
        s = raw[:, :n_real].astype(np.float64)
        return 2 * s - 512 if network == "cnvW1A1" else s  # popcount of matches -> signed sum
    bits = (raw[:, None] >> np.arange(n_real, dtype=np.uint64)) & np.uint64(1)
    return 2.0 * bits - 1.0


def make_params(network, tmp_path, seed):
    layers = random_float_net(network, seed)
    W, T = params_io.synthesize(network, layers)
    params_io.write_params(str(tmp_path), network, W, T, classes=[str(i) for i in range(10)])
    return layers, W, T


@pytest.mark.parametrize("network", NETS)
def test_float_network_equals_oracle_on_synthesized_params(network, tmp_path):
    layers, W, T = make_params(network, tmp_path, seed=11)
    cnv = network.startswith("cnv")
    imgs = np.random.default_rng(12).integers(0, 256, (24 if cnv else 1500, 3072 if cnv else 784), dtype=np.uint8)
    want = expected_outputs(network, layers, imgs)
    o = ol.Oracle(network, str(tmp_path))
    raw = o.scores_fast(imgs) if cnv else o.words_fast(imgs)
    got = integer_outputs(network, raw)
    assert (got == want).all()
    assert np.unique(want).size > (8 if cnv else 1)
    # shape and range facts of the derived files
    lay = params_io.layout(network)
    for l, L in enumerate(lay):
        assert W[l].shape == (L["mh"], L["mw"]) and T[l].min() >= -32768 and T[l].max() <= 32767
    assert (T[-1][10:] == 32767).all() or lay[-1]["nthr"] == 0          # padding neurons never fire
    if not cnv:
        assert (W[0][:, 784:] == 1).all()                                 # padding synapses: weight bit 1


def test_negative_scale_is_off_by_one_at_the_tie(tmp_path):
    """What the port inherits from the reference: for a neuron with gamma*invstd < 0 in a multi-threshold (or
    conv) layer the stored threshold is ceil(-t); with the accelerator's strict compare T < acc the neuron then
    misses exactly the inputs whose accumulator EQUALS ceil(-t).  Everywhere else float model and integer
    path agree.  (lfcW1A2 layer 0: 784 binarised inputs, two thresholds per neuron, 48 padding columns.)"""
    net = "lfcW1A2"
    layers = random_float_net(net, 31, negative_scales="all")
    W, T = params_io.synthesize(net, layers)
    params_io.write_params(str(tmp_path), net, W, T)
    imgs = np.random.default_rng(32).integers(0, 256, (64, 784), dtype=np.uint8)
    o = ol.Oracle(net, str(tmp_path))
    hw = np.array([o.layer_ref(im, 0) for im in imgs])[:, :1024].astype(np.float64)
    d = layers[0]
    x = np.where(imgs >= 128, 1.0, -1.0)
    z = (x @ quant_w(d["W"], net) + d["bias"] - d["mean"]) * d["invstd"] * d["gamma"] + d["beta"]
    want = -1.0 + (z > -0.5) + (z > 0.5)
    acc = x @ W[0][:, :784].T.astype(np.float64) - 48          # what the accelerator accumulates (padding: -1 each)
    tie = (acc[:, :, None] == T[0][None, :, :]).any(axis=2)
    flipped = (d["gamma"] * d["invstd"] < 0)[None, :]
    differs = hw != want
    assert differs.any() and not (differs & ~(tie & flipped)).any()
    assert (hw[differs] == want[differs] - 1).all()              # one level short, never more


def test_threshold_formulas_by_hand():
    one = lambda v: np.array([v], np.float64)  # noqa: E731
    # popcount form: t = mean - bias - beta/(gamma*invstd) = 3 - 1 - 2/(2*0.5) = 0 ; (100 + 0)/2 = 50
    T, flip = params_io.bn_thresholds(100, one(1), one(2), one(2), one(3), one(0.5), conv=False)
    assert T.tolist() == [[50.0]] and not flip[0]
    # negative scale: t = 0 - 0 - 15/(-2) = 7.5, flipped -> -7.5 ; int() truncates: (100 - 7.5)/2 = 46.25 -> 46
    T, flip = params_io.bn_thresholds(100, one(0), one(15), one(-2), one(0), one(1), conv=False)
    assert T.tolist() == [[46.0]] and flip[0]
    # truncation toward zero, not floor: (10 - 13.5)/2 = -1.75 -> -1
    T, _ = params_io.bn_thresholds(10, one(0), one(0), one(1), one(-13.5), one(1), conv=False)
    assert T.tolist() == [[-1.0]]
    # conv layer 0: signed, 2^-8 units: floor(256 * 0.3) = 76 ; flipped: ceil(-256 * 0.3) = -76
    T, _ = params_io.bn_thresholds(27, one(0), one(0), one(1), one(0.3), one(1), use_popcount=False, frac_bits=8)
    assert T.tolist() == [[76.0]]
    T, flip = params_io.bn_thresholds(27, one(0), one(0), one(-1), one(0.3), one(1), use_popcount=False, frac_bits=8)
    assert T.tolist() == [[-76.0]] and flip[0]
    # 2-bit activations: levels -0.5 / +0.5 -> t = mean - bias + (step - beta)/scale = 1 + (-+0.5 - 0.25)/0.5
    T, _ = params_io.bn_thresholds(64, one(0), one(0.25), one(1), one(1), one(0.5), abits=2)
    assert T.tolist() == [[np.floor(1 - 1.5), np.floor(1 + 0.5)]]
    T, flip = params_io.bn_thresholds(64, one(0), one(0.25), one(-1), one(1), one(0.5), abits=2)
    assert T.tolist() == [[np.ceil(-(1 + 1.5)), np.ceil(-(1 - 0.5))]] and flip[0]  # descending: kept as is
    # layer-0 thresholds saturate to 16 bits because the memory object is built with its defaults
    W, Tp = params_io.pad_layer(np.ones((2, 27), np.int64), np.array([[40000.0], [-40000.0]]), 4, 27, 1, 1, 8)
    assert Tp[:, 0].tolist() == [32767, -32768, 32767, 32767]
    # AccuOffset: +-1 arithmetic over 48 padding columns (weight +1 x input -1) is taken off the thresholds
    W, Tp = params_io.pad_layer(np.ones((1, 784), np.int64), np.array([[5.0, 9.0]]), 1, 832, 1, 2, 1)
    assert Tp.tolist() == [[5 - 48, 9 - 48]] and (W[0, 784:] == 1).all()
    # column interleave of the first FC layer behind a conv stack: chan*P + pix -> pix*C + chan
    Wq = np.arange(12).reshape(1, 12)
    assert params_io.interleave_fc_columns(Wq, 3).tolist() == [[0, 4, 8, 1, 5, 9, 2, 6, 10, 3, 7, 11]]


def test_npz_reader_round_trip(tmp_path):
    layers = random_float_net("cnvW1A1", 3)
    flat = []
    for d in layers:
        flat += [d[k] for k in ("W", "bias", "beta", "gamma", "mean", "invstd") if k in d]
    np.savez(tmp_path / "net.npz", *flat)
    back = params_io.read_npz_layers(str(tmp_path / "net.npz"), "cnvW1A1")
    assert len(back) == 9 and "gamma" not in back[8]
    assert all((back[l][k] == layers[l][k]).all() for l in range(9) for k in layers[l])


@pytest.mark.gpu
@pytest.mark.parametrize("network,count", [("lfcW1A1", 10000), ("cnvW1A1", 192), ("lfcW1A2", 2000), ("cnvW2A2", 64)])
def test_float_network_equals_gpu_runtime_on_synthesized_params(network, count, tmp_path):
    import ctypes as C

    import gpu_lib as gl
    layers, _, _ = make_params(network, tmp_path, seed=21)
    cnv = network.startswith("cnv")
    imgs = np.random.default_rng(22).integers(0, 256, (count, 3072 if cnv else 784), dtype=np.uint8)
    want = expected_outputs(network, layers, imgs)
    L = gl.load(network)
    L.load_parameters(str(tmp_path).encode())
    assert not L.bnn_mi355x_last_error()
    usec = C.c_float(0)
    if cnv:
        raw = np.zeros((count, 64), np.int16)
        assert L.bnn_mi355x_inference_raw(imgs.ctypes.data, count, raw.ctypes.data, None, C.byref(usec)) == 0
    else:
        raw = np.zeros(count, np.uint64)
        assert L.bnn_mi355x_inference_raw(imgs.ctypes.data, count, None, raw.ctypes.data, C.byref(usec)) == 0
    assert (integer_outputs(network, raw) == want).all()
