"""CnvClassifier.image_to_cifar (reference bnn/bnn.py:226-242) on the device.

The oracle of this step is Pillow itself (the reference calls it; it is installed here and on the
GPU box): the device-made record must equal, byte for byte, the record the reference's procedure
(PIL thumbnail with LANCZOS + paste on a white canvas) writes for the same picture.
"""
import ctypes as C
import io
import os
import sys

import numpy as np
import pytest

import gpu_lib as gl

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

sys.path.insert(0, os.path.join(gl.ROOT, "bnn-pynq_amd"))


def pil_record(img):
    """the reference's procedure, on the host (bnn.py:226-242)"""
    img = img.copy()
    img.thumbnail((32, 32), Image.LANCZOS, reducing_gap=None)
    canvas = Image.new("RGBA", (32, 32), (255, 255, 255, 0))
    canvas.paste(img, (int((32 - img.size[0]) / 2), int((32 - img.size[1]) / 2)))
    px = np.array(canvas)
    return np.concatenate([np.array([1], np.uint8)] + [px[:, :, c].flatten() for c in range(3)])


def picture(w, h, mode, seed, kind):
    rng = np.random.default_rng(seed)
    shape = (h, w) if mode == "L" else (h, w, 3)
    if kind == "noise":
        a = rng.integers(0, 256, shape, dtype=np.uint8)
    elif kind == "blocks":  # saturated blocks: the Lanczos lobes overshoot, clip8 works at both ends
        by, bx = max(h // 7, 1), max(w // 5, 1)
        yy, xx = np.mgrid[0:h, 0:w]
        base = (((yy // by) + (xx // bx)) % 2 * 255).astype(np.uint8)
        a = base if mode == "L" else np.stack([base, 255 - base, np.roll(base, bx // 2 + 1, axis=1)], axis=2)
    else:  # smooth gradient + a little noise
        yy, xx = np.mgrid[0:h, 0:w]
        g = (yy * 255.0 / max(h - 1, 1) * 0.5 + xx * 255.0 / max(w - 1, 1) * 0.5)
        a = np.clip(g + rng.normal(0, 3, (h, w)), 0, 255).astype(np.uint8)
        if mode != "L":
            a = np.stack([a, a[::-1], a[:, ::-1]], axis=2)
    if mode == "RGBA":  # colour planes as for RGB, alpha: every regime of the premultiply / unpremultiply pair
        alpha = rng.choice(np.array([0, 1, 2, 17, 128, 200, 254, 255], np.uint8), size=(h, w), p=[.15, .05, .05, .1, .15, .15, .1, .25])
        if kind == "smooth":
            alpha = np.clip(np.mgrid[0:h, 0:w][1] * 255.0 / max(w - 1, 1) + rng.normal(0, 2, (h, w)), 0, 255).astype(np.uint8)
        a = np.dstack([a, alpha])
    return Image.fromarray(np.ascontiguousarray(a), mode)


SIZES = [(1, 1), (5, 7), (32, 32), (31, 32), (32, 33), (33, 32), (33, 33), (40, 20), (20, 40), (64, 64), (100, 75),
         (75, 100), (640, 480), (333, 1), (1, 333), (500, 3), (3, 500), (1023, 769), (2000, 37), (41, 1777),
         (1920, 1080), (3001, 1999),
         (2, 201), (2, 200), (7, 701), (7, 700), (30, 3001), (33, 3301), (40, 4001), (1, 101), (600, 5)]  # Pillow: vertical pass first when h > 100 w


def test_thumbnail_size_matches_pillow():
    """host arithmetic, no GPU: the size rule of Image.thumbnail((32, 32))"""
    L = gl.load("cnvW1A1")
    rng = np.random.default_rng(5)
    cases = list(SIZES) + [(int(a), int(b)) for a, b in rng.integers(1, 5000, (3000, 2))]
    cases += [(w, h) for w in range(1, 70) for h in range(1, 70)]
    ow, oh = C.c_int(0), C.c_int(0)
    for w, h in cases:
        im = Image.new("L", (w, h))
        im.thumbnail((32, 32), Image.NEAREST, reducing_gap=None)
        resized = L.bnn_mi355x_thumbnail_size(w, h, C.byref(ow), C.byref(oh))
        assert (ow.value, oh.value) == im.size, (w, h)
        assert resized == int(not (32 >= w and 32 >= h))


@pytest.fixture(scope="module")
def clf():
    import bnn
    return bnn.CnvClassifier(bnn.NETWORK_CNVW1A1, "cifar10", bnn.RUNTIME_SW)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["RGB", "L", "RGBA"])
@pytest.mark.parametrize("kind", ["noise", "blocks", "smooth"])
def test_records_equal_pillow(clf, mode, kind):
    imgs = [picture(w, h, mode, 11 * i + len(kind), kind) for i, (w, h) in enumerate(SIZES)]
    recs = clf.images_to_cifar(imgs)
    assert recs.shape == (len(imgs), 3073)
    for im, (w, h), r in zip(imgs, SIZES, recs):
        assert im.size == (w, h)  # the caller's image is not shrunk in place
        want = pil_record(im)
        assert (r == want).all(), ("size", (w, h), "first differing byte", int(np.argmax(r != want)))


@pytest.mark.gpu
def test_mixed_batch_modes_and_repeated_sizes(clf):
    """one call, pictures of changing and repeating sizes (coefficient tables are reused while the size
    repeats), LA and palette pictures taking the host route in between"""
    imgs = []
    for i in range(24):
        w, h = [(200, 100), (200, 100), (64, 48), (200, 100)][i % 4]
        imgs.append(picture(w, h, "RGB" if i % 3 else "L", 100 + i, "noise"))
    imgs.insert(5, picture(80, 60, "RGB", 1, "smooth").convert("LA"))
    imgs.insert(9, picture(90, 50, "RGB", 2, "blocks").convert("P"))
    recs = clf.images_to_cifar(imgs)
    for im, r in zip(imgs, recs):
        assert (r == pil_record(im)).all()
    # decoded pictures as arrays take the same route
    as_arrays = [np.asarray(im) for im in imgs[:5]]
    assert (clf.images_to_cifar(as_arrays) == recs[:5]).all()
    with pytest.raises(ValueError):
        clf.images_to_cifar([np.zeros((4, 4, 2), np.uint8)])


@pytest.mark.gpu
def test_row_stride_and_errors(clf):
    L = clf.bnn.interface
    big = np.random.default_rng(3).integers(0, 256, (300, 500, 3), dtype=np.uint8)
    view = big[10:210, 50:350]  # 200 x 300 window of a larger picture: rows 1500 bytes apart
    ptrs = (C.c_void_p * 1)(view.ctypes.data)
    one = lambda v: (C.c_int * 1)(v)  # noqa: E731
    out = np.zeros((1, 3073), np.uint8)
    assert L.bnn_mi355x_images_to_cifar(ptrs, one(300), one(200), one(3), (C.c_long * 1)(1500), 1, out.ctypes.data) == 0
    assert (out[0] == pil_record(Image.fromarray(np.ascontiguousarray(view), "RGB"))).all()
    assert L.bnn_mi355x_images_to_cifar(ptrs, one(300), one(200), one(2), None, 1, out.ctypes.data) == -1
    assert b"bytes" in L.bnn_mi355x_last_error()
    assert L.bnn_mi355x_images_to_cifar(ptrs, one(300), one(200), one(3), (C.c_long * 1)(100), 1, out.ctypes.data) == -1
    assert L.bnn_mi355x_images_to_cifar(None, None, None, None, None, 0, None) == 0
    lfc = gl.load("lfcW1A1")
    assert lfc.bnn_mi355x_images_to_cifar(ptrs, one(300), one(200), one(3), None, 1, out.ctypes.data) == -1


@pytest.mark.gpu
def test_classify_images_end_to_end(clf):
    """classify_images / classify_image on pictures == the file ABI on PIL-made records"""
    import tempfile
    imgs = [picture(w, h, "RGB", 7 * i, "smooth" if i % 2 else "noise") for i, (w, h) in enumerate(SIZES[:12])]
    got = clf.classify_images(imgs)
    with tempfile.NamedTemporaryFile() as tmp:
        for im in imgs:
            tmp.write(pil_record(im).tobytes())
        tmp.flush()
        want = clf.classify_cifars(tmp.name)
    assert list(got) == list(want)
    d = clf.classify_image_details(imgs[3])
    assert len(d) == len(clf.classes)
    assert clf.classify_image(imgs[3]) == int(np.argmax(d))  # single-image decode: first maximum
