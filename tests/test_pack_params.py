"""CPU: the host-side repacker of the product (csrc/packed_params.cpp, reached
through bnn_mi355x_pack_params -- pure host code) against the oracle's unpacked
weights/thresholds for every layer of every shipped parameter set."""
import struct

import numpy as np
import pytest

import gpu_lib as gl
import oracle_lib as ol

SETS = [("cnvW1A1", "cifar10"), ("cnvW1A2", "cifar10"), ("cnvW2A2", "cifar10"), ("lfcW1A1", "mnist"),
        ("lfcW1A2", "mnist"), ("cnvW1A1", "streetview"), ("cnvW1A1", "road-signs"), ("lfcW1A1", "chars_merged")]
AR_INT8, AR_XNOR, AR_TB, AR_TT = 0, 1, 2, 3


def arith_of(network, layer):
    cnv = network.startswith("cnv")
    if cnv and layer == 0:
        return AR_INT8
    if network.endswith("A1"):
        return AR_XNOR
    if network == "lfcW1A2" and layer == 0:
        return AR_XNOR
    return AR_TT if "W2" in network else AR_TB


def bits(words):
    """u64 array [rows, kw] -> 0/1 array [rows, kw*64], LSB first"""
    b = np.unpackbits(words.view(np.uint8).reshape(words.shape[0], -1), axis=1, bitorder="little")
    return b.astype(np.int8)


@pytest.mark.parametrize("network,dataset", SETS, ids=lambda x: x)
def test_blob_matches_oracle_weights(network, dataset):
    blob = gl.pack_params(network, gl.param_dir(dataset, network))
    o = ol.Oracle(network, ol.param_dir(dataset, network))
    magic0, magic1, version, net_id, nlayers, total, _, _ = struct.unpack_from("<8I", blob, 0)
    assert (magic0, magic1, version) == (0x4D4E4E42, 0x35353349, 4)
    assert total == blob.size and nlayers == o.nl
    for l in range(nlayers):
        off, rd, rows, kw = struct.unpack_from("<4I", blob, 32 + 16 * l)
        assert off % 256 == 0
        W = o.weights(l)
        mh, mw = W.shape
        assert rows == mh
        R = blob[off:off + rows * rd * 4].view(np.uint32).reshape(rows, rd)
        ar = arith_of(network, l)
        if ar == AR_INT8:
            assert rd == 12 and kw == 0
            taps = R[:, 2:9].copy().view(np.int8).reshape(rows, 28)   # tap tau = 3*(c*3+ky)+kx
            assert (taps[:, 27] == 0).all() and (R[:, 9:] == 0).all()
            want = W.reshape(rows, 3, 3, 3).transpose(0, 3, 1, 2).reshape(rows, 27)  # [n][ky][kx][c] -> [n][c][ky][kx]
            assert (taps[:, :27] == want).all()
        else:
            assert kw == mw // 64
            wq = R[:, 2:].copy().view(np.uint64)
            if ar == AR_XNOR:
                assert (bits(wq) == (W > 0)).all()
            elif ar == AR_TB:
                assert (W != 0).all() and (bits(wq) == (W < 0)).all()
            else:   # {sign, non-zero} plane pairs, the "weight is -2" plane (empty in trained sets), flag, pad
                assert rd == 4 + 6 * kw
                pairs = wq[:, : 2 * kw]
                assert (bits(np.ascontiguousarray(pairs[:, 0::2])) == (W < 0)).all()
                assert (bits(np.ascontiguousarray(pairs[:, 1::2])) == (W != 0)).all()
                assert (wq[:, 2 * kw:] == 0).all() and not (W == -2).any()
        # thresholds: check the pre-transformed form against its definition
        L = o.L
        t = R[:, :2].astype(np.uint32).view(np.int32)
        last_cnv = network.startswith("cnv") and l == 8
        if last_cnv:
            continue
        nthr = 2 if (network.endswith("A2") and not (network == "lfcW1A2" and l == 3)) else 1
        for n in range(0, mh, max(1, mh // 37)):
            for i in range(nthr):
                T = L.bnn_oracle_threshold(o.h, l, n, i)
                if ar == AR_INT8:
                    want_t = T >> 1
                elif ar == AR_XNOR and network == "lfcW1A2":
                    want_t = (mw - T + 1) >> 1
                elif ar == AR_XNOR:
                    want_t = mw - T
                else:
                    want_t = T
                assert t[n, i] == want_t
            if nthr == 1:
                assert t[n, 1] == t[n, 0]


@pytest.mark.parametrize("network,dataset", [s for s in SETS if s[0].startswith("cnv")], ids=lambda x: x)
def test_layer0_mfma_table(network, dataset):
    """the matrix-pipe copy of layer 0: two tables (first / second threshold), taps in the AR_INT8 order,
    the threshold folded into K slots 27/28 (a0 + 64*a1 = -clamp(t) - 1 with |a0| <= 32)"""
    blob = gl.pack_params(network, gl.param_dir(dataset, network))
    o = ol.Oracle(network, ol.param_dir(dataset, network))
    off = struct.unpack_from("<I", blob, 24)[0]
    assert off and off % 256 == 0
    W = o.weights(0)
    nthr = 2 if network.endswith("A2") else 1
    lim = 2 * 3456 if "W2" in network else 3456
    for which in range(2):
        A = blob[off + which * 2048: off + (which + 1) * 2048].copy().view(np.int8).reshape(64, 32)
        assert (A[:, :27] == W.reshape(64, 3, 3, 3).transpose(0, 3, 1, 2).reshape(64, 27)).all()
        assert (A[:, 29:] == 0).all()
        for n in range(64):
            t = o.L.bnn_oracle_threshold(o.h, 0, n, which if nthr == 2 else 0) >> 1
            tc = min(lim, max(-lim - 1, t))
            assert int(A[n, 27]) + 64 * int(A[n, 28]) == -tc - 1 and abs(int(A[n, 27])) <= 32
    # the same numbers once more, in the operand form of k_conv0_tile (packed_params.h): K = 32 part "big"
    # [ct][row i][h][16], K = 16 part "small" [threshold][ct][row i][h][8]; row i = 8g + 4h' + q of tile ct is neuron
    # 32ct + 16h' + 4g + q; a run (c, ky) = its three kx taps + a zero-weight byte
    P = [blob[off + w * 2048: off + (w + 1) * 2048].copy().view(np.int8).reshape(64, 32) for w in range(2)]
    big = blob[off + 4096: off + 4096 + 2048].copy().view(np.int8).reshape(2, 32, 2, 4, 4)
    small = blob[off + 6144: off + 6144 + 2048].copy().view(np.int8).reshape(2, 2, 32, 2, 8)
    runs = [[0, 1, 2, 6], [3, 4, 5, 8]]
    for ct in range(2):
        for i in range(32):
            n = 32 * ct + 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3)
            for h in range(2):
                for s_ in range(4):
                    assert big[ct, i, h, s_, :3].tolist() == P[0][n, 3 * runs[h][s_]: 3 * runs[h][s_] + 3].tolist() and big[ct, i, h, s_, 3] == 0
            for w in range(2):
                assert small[w, ct, i, 0].tolist() == P[w][n, 21:24].tolist() + [0] * 5
                assert small[w, ct, i, 1].tolist() == P[w][n, 27:29].tolist() + [0] * 6
    assert sorted(32 * ct + 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3) for ct in range(2) for i in range(32)) == list(range(64))
    # clamping never changes a decision: |dot| <= 27 * 128 * max|w|  (ap_int<2> weights reach -2 under faults)
    d = np.arange(-lim, lim + 1)
    for t in (-(1 << 22), -lim - 2, -lim - 1, -lim, 0, lim - 1, lim, lim + 1, 1 << 22):
        tc = min(lim, max(-lim - 1, t))
        assert ((t < d) == (tc < d)).all()


def test_threshold_transforms_are_equivalences():
    """the three pre-transformed compares used by the kernels are exact rewrites
    of the reference's strict `T < acc` over the whole reachable range"""
    rng = np.random.default_rng(0)
    # XNOR: T < MW - m  <=>  m < MW - T
    for MW in (576, 832, 1152, 2304):
        m = rng.integers(0, MW + 1, 20000)
        T = rng.integers(-32768, 32768, 20000)
        assert ((T < MW - m) == (m < MW - T)).all()
        # signed form: T < MW - 2m  <=>  m < floor((MW - T + 1) / 2)
        assert ((T < MW - 2 * m) == (m < ((MW - T + 1) >> 1))).all()
    # layer 0: T < 2*dot  <=>  floor(T/2) < dot
    d = rng.integers(-3456, 3457, 50000)
    T = rng.integers(-2 ** 23, 2 ** 23, 50000)
    assert ((T < 2 * d) == ((T >> 1) < d)).all()
    T = rng.integers(-7000, 7000, 50000)
    assert ((T < 2 * d) == ((T >> 1) < d)).all()


def test_missing_file_is_reported():
    L = gl.load("cnvW1A1")
    assert L.bnn_mi355x_pack_params(b"/nonexistent/dir", None, 0) == 0
    assert b"Could not open file" in L.bnn_mi355x_last_error()


def test_import_rejects_headers_that_are_not_this_networks_layout():
    """bnn_mi355x_import_params takes bytes from outside (broadcast, file): every header field is later used
    as an offset or stride on the host and on the device, so each must be exactly what this library writes.
    (On a GPU-less box a VALID blob gets as far as "no HIP device"; a corrupted one must fail before that.)"""
    import ctypes as C
    L = gl.load("cnvW2A2")
    good = gl.pack_params("cnvW2A2", gl.param_dir("cifar10", "cnvW2A2"))
    assert L.bnn_mi355x_params_bytes() == good.size

    def error_for(blob):
        rc = L.bnn_mi355x_import_params(blob.ctypes.data, blob.size)
        return rc, L.bnn_mi355x_last_error().decode()

    rc, err = error_for(good)
    assert (rc == 0 and err == "") or "no HIP device" in err
    hdr = 32  # magic0, magic1, version, net, nlayers, total_bytes, l0_mfma_offset, reserved; then 9 x {offset, row_dwords, rows, kw}
    cases = {"l0_mfma_offset": (24, good.size), "kw of layer 1": (hdr + 16 * 1 + 12, 4096), "offset of layer 3": (hdr + 16 * 3, 4),
             "rows of layer 8": (hdr + 16 * 8 + 8, 1 << 20), "row_dwords of layer 2": (hdr + 16 * 2 + 4, 3), "reserved": (28, 1)}
    for what, (at, value) in cases.items():
        bad = good.copy()
        bad[at:at + 4] = np.frombuffer(struct.pack("<I", value), np.uint8)
        rc, err = error_for(bad)
        assert rc != 0 and "packed params" in err and "mismatch" in err, what
    rc, err = error_for(good[:-256].copy())
    assert rc != 0 and "size mismatch" in err
