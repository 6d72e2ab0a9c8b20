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
