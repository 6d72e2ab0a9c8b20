"""Reading and writing parameter sets in the reference's on-disk format
(``bnn/params/<dataset>/<network>/<layer>-<pe>-{weights,thres}.bin``).

This is the packing half of ``bnn/src/training/finnthesizer.py`` (``BNNProcElemMem.
__updatePEMapping / __wmem2bin / __tmem2bin``, finnthesizer.py:557-588,616-690; Python 2 in the
reference) restated for Python 3 on plain integer matrices, plus the inverse (the loader the runtime
uses in C++, ``csrc/packed_params.cpp``).  It lets users write their own trained networks into the
format both the reference and this runtime read, and lets the tests build random parameter sets.

Format: per layer and PE one file of WMEM little-endian 64-bit words; word
``nf*SF + sf`` (SF = MW/SIMD) of PE ``p`` holds columns ``sf*SIMD .. +SIMD-1`` of matrix row
``n = nf*PE + p``, column ``s`` in bits ``[WPI*s, WPI*(s+1))``: 1-bit 1 <=> +1, 0 <=> -1; 2-bit
two's complement {0, 1, 3} <=> {0, +1, -1}.  Threshold files: TMEM x nThr little-endian int64,
entry ``nf*nThr + i`` for row ``nf*PE + p``.
"""
import os

import numpy as np

# (PE, SIMD, WMEM, TMEM) per layer: bnn/src/network/<net>/hw/config.h
_CNV_W1 = [(16, 3, 36, 4), (32, 32, 36, 2), (16, 32, 144, 8), (16, 32, 288, 8), (4, 32, 2304, 64),
           (1, 32, 18432, 256), (1, 4, 32768, 512), (1, 8, 32768, 512), (4, 1, 8192, 16)]
_CNV_W2 = [(8, 3, 72, 8), (16, 16, 144, 4), (8, 16, 576, 16), (8, 16, 1152, 16), (4, 8, 9216, 64),
           (1, 8, 73728, 256), (1, 2, 65536, 512), (2, 2, 65536, 256), (4, 1, 8192, 16)]
_LFC = [(32, 64, 416, 32), (64, 32, 512, 16), (32, 64, 512, 32), (16, 8, 512, 4)]


def layout(network):
    """list of dicts per layer: pe, simd, wmem, tmem, wbits, nthr, mh, mw"""
    if network.startswith("cnv"):
        wbits = 2 if "W2" in network else 1
        abits = 2 if network.endswith("A2") else 1
        fold = _CNV_W2 if wbits == 2 else _CNV_W1
        nthr = [abits] * 8 + [0]
    elif network.startswith("lfc"):
        wbits, fold = 1, _LFC
        nthr = [2, 2, 2, 1] if network.endswith("A2") else [1, 1, 1, 1]
    else:
        raise ValueError(network)
    out = []
    for (pe, simd, wmem, tmem), nt in zip(fold, nthr):
        out.append(dict(pe=pe, simd=simd, wmem=wmem, tmem=tmem, wbits=wbits, nthr=nt,
                        mh=tmem * pe, mw=(wmem // tmem) * simd))
    return out


def write_params(directory, network, weights, thresholds, classes=None):
    """weights[l]: int array [MH, MW] with values in {-1,+1} (1-bit) or {-1,0,+1} (2-bit; -2, the fourth
    ap_int<2> value, is written too: trained sets never hold it, fault-injected ones do);
    thresholds[l]: int array [MH, nThr] (ignored where nThr == 0), raw integers as stored
    (layer 0 of the CNV nets: units of 2^-8).  Writes the reference's file set."""
    os.makedirs(directory, exist_ok=True)
    for l, L in enumerate(layout(network)):
        W = np.asarray(weights[l]).reshape(L["mh"], L["mw"])
        sf = L["mw"] // L["simd"]
        if L["wbits"] == 1:
            fields = (W > 0).astype(np.uint64)
        else:
            fields = (W.astype(np.int64) & 3).astype(np.uint64)  # ap_int<2> two's complement: -1 -> 0b11, -2 -> 0b10
        f = fields.reshape(L["mh"], sf, L["simd"])
        shifts = (np.arange(L["simd"], dtype=np.uint64) * np.uint64(L["wbits"]))
        words = np.bitwise_or.reduce(f << shifts, axis=2)  # [MH, SF]
        for p in range(L["pe"]):
            rows = words[p::L["pe"]]  # rows n = nf*PE + p, nf ascending
            rows.astype("<u8").tofile(os.path.join(directory, "%d-%d-weights.bin" % (l, p)))
            if L["nthr"]:
                T = np.asarray(thresholds[l]).reshape(L["mh"], L["nthr"])
                T[p::L["pe"]].astype("<i8").tofile(os.path.join(directory, "%d-%d-thres.bin" % (l, p)))
    if classes is not None:
        with open(os.path.join(directory, "classes.txt"), "w") as fp:
            fp.write("\n".join(classes))


def read_params(directory, network):
    """inverse of write_params: (weights, thresholds) as int arrays.  Short files are
    zero-filled like the reference's loader (foldedmv-offload.cpp:283-284)."""
    weights, thresholds = [], []
    for l, L in enumerate(layout(network)):
        sf = L["mw"] // L["simd"]
        W = np.zeros((L["mh"], L["mw"]), np.int8)
        T = np.zeros((L["mh"], max(L["nthr"], 1)), np.int64)
        mask = np.uint64((1 << L["wbits"]) - 1)
        shifts = (np.arange(L["simd"], dtype=np.uint64) * np.uint64(L["wbits"]))
        for p in range(L["pe"]):
            raw = np.fromfile(os.path.join(directory, "%d-%d-weights.bin" % (l, p)), dtype=np.uint8)
            buf = np.zeros(L["wmem"] * 8, np.uint8)
            buf[: min(raw.size, buf.size)] = raw[: buf.size]
            words = buf.view("<u8").reshape(L["tmem"], sf)
            f = ((words[:, :, None] >> shifts) & mask).astype(np.int64)
            vals = np.where(f > 0, 1, -1) if L["wbits"] == 1 else np.where(f >= 2, f - 4, f)
            W[p::L["pe"]] = vals.reshape(L["tmem"], L["mw"])
            if L["nthr"]:
                raw = np.fromfile(os.path.join(directory, "%d-%d-thres.bin" % (l, p)), dtype=np.uint8)
                buf = np.zeros(L["tmem"] * L["nthr"] * 8, np.uint8)
                buf[: min(raw.size, buf.size)] = raw[: buf.size]
                T[p::L["pe"]] = buf.view("<i8").reshape(L["tmem"], L["nthr"])
        weights.append(W)
        thresholds.append(T)
    return weights, thresholds


# ======================================================================================================
# Float model -> integer matrices: the other half of finnthesizer.py (makeFCBNComplex, makeFCBNComplex_QNN,
# makeConvBNComplex, finnthesizer.py:169-330; BNNWeightReader's FC column interleave :374-387; the padding
# and saturation of BNNProcElemMem :528-555,616-690; the per-network drivers cifar10-gen-weights-*.py,
# mnist-gen-weights-*.py / convertFCNetwork :40-100), restated for Python 3 on whole arrays.
#
# "parity unpinned": the trained .npz archives these functions were written for are missing blobs in the
# reference checkout and the tool itself is Python 2, so no byte comparison with shipped parameter files
# is possible.  What is tested instead (tests/test_synthesize.py): a random float network evaluated in
# floating point, layer by layer, makes the same decisions as the oracle / the GPU runtime on the derived
# parameter files, and the hand-checkable properties of each formula.
#
# A layer of the float model is a dict of numpy arrays, in the order the training scripts store them
# (finnthesizer.py BNNWeightReader: W, then bias, beta, gamma, mean, invstd):
#   W        FC: [ins, outs] (a Lasagne DenseLayer: neurons in COLUMNS); conv: [out_ch, in_ch, k, k]
#   bias, beta, gamma, mean, invstd   [outs]: layer bias and batch-norm parameters,
#            y = (W.x + bias - mean) * invstd * gamma + beta ; activation = quantiser of y
# ======================================================================================================
def _quantize_weights(w, wbits):
    """finnthesizer.py quantize()/binarize(): 1 bit: +1 where w >= 0 else -1 (the file bit is 1 / 0);
    more bits: floor(w + 0.5) (round half up, no clipping: training keeps the weights inside [-1, 1])"""
    if wbits == 1:
        return np.where(w >= 0, 1, -1).astype(np.int64)
    return np.floor(w + 0.5).astype(np.int64)


def _steps(abits):
    """decision levels of the activation quantiser on the batch-norm output: none but 0 for 1 bit; for A2
    {-0.5, +0.5} (np.linspace(-1, 1, 2^(A-1), endpoint=False) + 1/2, finnthesizer.py:223,270)"""
    if abits == 1:
        return np.zeros(1)
    return np.linspace(-1, 1, num=2 ** (abits - 1), endpoint=False) + 0.5


def bn_thresholds(fanin, bias, beta, gamma, mean, invstd, abits=1, use_popcount=True, frac_bits=0, conv=True):
    """Batch-norm + quantiser of one layer as integer thresholds on the accumulator.
    Returns (T [outs, nthr] float64 holding integers, flip [outs] bool); rows with flip have gamma*invstd < 0:
    their weights must be negated so that every neuron fires on the positive side.

    1-bit activations: t = mean - bias - beta / (gamma * invstd) on the signed sum; conv layers round it
    floor(f*t) / ceil(-f*t) with f = 2^frac_bits (finnthesizer.py:285-293), FC layers do not (:183-193);
    with use_popcount the compare moves to the number of matches: int((fanin + t) / 2), int() truncating
    toward zero as in Python 2.  2-bit activations: one threshold per level,
    t_i = mean - bias + (step_i - beta) / (gamma * invstd), floor(f*t) / ceil(-f*t) (:230-240,:300-306).
    """
    bias, beta, gamma, mean, invstd = (np.asarray(a, np.float64) for a in (bias, beta, gamma, mean, invstd))
    scale = gamma * invstd
    flip = scale < 0
    f = float(2 ** frac_bits)
    if abits == 1:
        t = (mean - bias) - beta / scale
        if conv:
            t = np.where(flip, np.ceil(-f * t), np.floor(f * t))
        else:
            t = np.where(flip, -t, t)
        if use_popcount:
            t = np.trunc((fanin + t) / 2.0)
        elif not conv:
            t = np.trunc(t)
        return t[:, None], flip
    step = _steps(abits)
    t = (mean - bias)[:, None] + (step[None, :] - beta[:, None]) / scale[:, None]
    t = np.where(flip[:, None], np.ceil(-f * t), np.floor(f * t))
    return t, flip


def fc_layer(layer, wbits=1, abits=1, ibits=1):
    """makeFCBNComplex / makeFCBNComplex_QNN as called by readFCBNComplex (finnthesizer.py:361-372): the
    popcount form only when weights, activations AND inputs are 1-bit.  layer["W"] is [ins, outs].
    A layer without batch-norm entries (CNV layer 8, readFCBNComplex_no_thresholds :395-412) gets the
    reference's stand-in parameters.  Returns (Wq [outs, ins], T [outs, nthr])."""
    W = np.asarray(layer["W"], np.float64)
    ins, outs = W.shape
    if "gamma" in layer:
        bn = [layer[k] for k in ("bias", "beta", "gamma", "mean", "invstd")]
    else:
        bn = [np.zeros(outs), np.zeros(outs), np.ones(outs), np.ones(outs), np.ones(outs)]
    T, flip = bn_thresholds(ins, *bn, abits=abits, use_popcount=(wbits == 1 and abits == 1 and ibits == 1), conv=False)
    Wq = _quantize_weights(np.where(flip[None, :], -W, W), wbits).T
    return Wq, T


def conv_layer(layer, wbits=1, abits=1, use_popcount=True, frac_bits=0, interleave_channels=True):
    """makeConvBNComplex (finnthesizer.py:258-330).  layer["W"] is [out_ch, in_ch, k, k]; the matrix column
    order is (ky, kx, in_ch) with interleave_channels (what every shipped network uses), else (in_ch, ky, kx).
    Returns (Wq [out_ch, in_ch*k*k], T [out_ch, nthr])."""
    W = np.asarray(layer["W"], np.float64)
    out_ch, in_ch, k, k2 = W.shape
    if k != k2:
        raise ValueError("Nonsymmetric conv kernels are not yet supported")
    T, flip = bn_thresholds(in_ch * k * k, *[layer[n] for n in ("bias", "beta", "gamma", "mean", "invstd")], abits=abits,
                            use_popcount=use_popcount and abits == 1, frac_bits=frac_bits, conv=True)
    Wf = np.where(flip[:, None, None, None], -W, W)
    if interleave_channels:
        Wf = Wf.transpose(0, 2, 3, 1)
    return _quantize_weights(Wf.reshape(out_ch, in_ch * k * k), wbits), T


def interleave_fc_columns(Wq, channels):
    """first FC layer behind a conv stack: the conv output reaches it pixel-major / channel-minor, the trained
    matrix has its columns channel-major (finnthesizer.py:374-387): column chan*P + pix -> pix*C + chan"""
    outs, ins = Wq.shape
    pix = ins // channels
    return Wq.reshape(outs, channels, pix).transpose(0, 2, 1).reshape(outs, ins)


def pad_layer(Wq, T, mh, mw, wbits, abits, ibits, thres_bits=16):
    """BNNProcElemMem.__padMatrix + the saturation of __tmem2bin (finnthesizer.py:528-555,660-674): pad the
    matrix to [mh, mw] -- padding weights +1 (1-bit; the padding INPUT bits are 0) resp. 0 (2-bit) --, padding
    neurons get the largest threshold; where a padding column is not transparent (+-1 arithmetic on a padded
    input: not the all-1-bit popcount form and not 2-bit weights) its contribution of -1 per column is taken
    off the thresholds (AccuOffset); thresholds saturate to thres_bits-bit integers."""
    n, s = Wq.shape
    W = np.full((mh, mw), 1 if wbits == 1 else 0, np.int64)
    W[:n, :s] = Wq
    Tp = np.full((mh, T.shape[1]), float(2 ** thres_bits - 1))
    Tp[:n] = T
    transparent = (wbits == 1 and abits == 1 and ibits == 1) or wbits >= 2
    if not transparent:
        Tp = Tp - (mw - s)
    lo, hi = -(2 ** (thres_bits - 1)), 2 ** (thres_bits - 1) - 1
    return W, np.trunc(np.clip(Tp, lo, hi)).astype(np.int64)


def synthesize(network, layers):
    """A trained float network -> the integer matrices write_params() takes, for one of the five shipped
    topologies, following the per-network drivers (cifar10-gen-weights-W1A1.py:75-178 and its W1A2 / W2A2
    siblings; mnist-gen-weights-*.py + convertFCNetwork).  `layers`: one dict per layer (see above).
    CNV layer 0 takes 8-bit fixed-point inputs: thresholds in units of 2^-8 (numThresBits=24,
    numThresIntBits=16), then saturated to 16 bits like every other layer, because the memory object is
    built with its defaults (cifar10-gen-weights-W1A1.py:88,99).  Returns (weights, thresholds)."""
    lay = layout(network)
    cnv = network.startswith("cnv")
    wbits = 2 if "W2" in network else 1
    a2 = network.endswith("A2")
    nl = len(lay)
    weights, thresholds = [], []
    for l, (L, P) in enumerate(zip(lay, layers)):
        abits = 1 if (l == nl - 1 or not a2) else 2        # ActivationPrecisions_integer: last layer 1
        ibits = (8 if cnv else 1) if l == 0 else (2 if a2 else 1)
        if cnv and l < 6:
            Wq, T = conv_layer(P, wbits, abits, use_popcount=(l > 0 and not a2 and wbits == 1), frac_bits=8 if l == 0 else 0)
        else:
            Wq, T = fc_layer(P, wbits, abits, ibits)
            if cnv and l == 6:
                Wq = interleave_fc_columns(Wq, 256)          # conv layer 5 has 256 channels of one pixel: the identity
        W, Tp = pad_layer(Wq, T, L["mh"], L["mw"], wbits, abits, ibits)
        weights.append(W)
        thresholds.append(Tp if L["nthr"] else np.zeros((L["mh"], 1), np.int64))
    return weights, thresholds


def read_npz_layers(path, network):
    """the training scripts' archive (arr_0, arr_1, ...: W, bias, beta, gamma, mean, invstd per layer; the last
    CNV layer has no batch norm) -> the list of layer dicts synthesize() takes"""
    z = np.load(path, allow_pickle=False)
    nl = len(layout(network))
    out, i = [], 0
    for l in range(nl):
        d = {"W": z["arr_%d" % i]}
        i += 1
        if not (network.startswith("cnv") and l == nl - 1):
            for k in ("bias", "beta", "gamma", "mean", "invstd"):
                d[k] = z["arr_%d" % i]
                i += 1
        out.append(d)
    return out
