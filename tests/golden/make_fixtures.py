#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's own
test images and recorded outputs.  Run once in the build container (needs
/root/reference and Pillow); the GPU box only ever sees the committed outputs.

What is produced (all DATA, no reference source text):
  * <name>.cifar   3073-byte CIFAR-10 records made by the image_to_cifar
                   procedure of bnn/bnn.py:226-242 from the reference's demo
                   images (tests/Test_image/*, notebooks/pictures/*);
  * deer.bin, 3.image-idx3-ubyte   byte copies of the reference's own test
                   fixtures (tests/Test_image/);
  * expected.json  the class scores / class indices RECORDED in the
                   reference's notebooks and tests, with the file:line of each.

The recorded numbers are typed in below from the notebook outputs; the script
re-reads the notebooks and asserts they are really there, so a typo cannot pin
the oracle to a wrong value.
"""
import json
import os
import shutil
import sys

import numpy as np
from PIL import Image

REF = os.environ.get("BNN_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def image_to_cifar(img):
    """bnn/bnn.py:226-242.  Image.ANTIALIAS == LANCZOS resampled from the full
    resolution image; on Pillow >= 7 that is reducing_gap=None (the default
    for thumbnail() is 2.0, which pre-shrinks JPEGs with draft mode and box
    filtering and does not reproduce tests/Test_image/deer.bin)."""
    img = img.copy()
    img.thumbnail((32, 32), Image.LANCZOS, reducing_gap=None)
    background = Image.new("RGBA", (32, 32), (255, 255, 255, 0))
    background.paste(img, (int((32 - img.size[0]) / 2), int((32 - img.size[1]) / 2)))
    a = np.array(background)
    rec = bytes([1]) + a[:, :, 0].tobytes() + a[:, :, 1].tobytes() + a[:, :, 2].tobytes()
    assert len(rec) == 3073
    return rec


def notebook_text(name):
    nb = json.load(open(os.path.join(REF, "notebooks", name)))
    chunks = []
    for c in nb["cells"]:
        for o in c.get("outputs", []):
            t = o.get("text") or o.get("data", {}).get("text/plain")
            if t:
                chunks.append("".join(t))
    return "\n".join(chunks)


def assert_recorded(nbname, scores):
    """every score must appear, in order, in the notebook's recorded outputs"""
    txt = notebook_text(nbname)
    pos = 0
    for s in scores:
        k = txt.find(str(s), pos)
        assert k >= 0, (nbname, s)
        pos = k
    # and as a contiguous ranking block: ten lines each ending in the score
    import re
    nums = [int(x) for x in re.findall(r"^\s*\S+\s+(-?\d+)\s*$", txt, re.M)]
    joined = ",".join(map(str, nums))
    assert ",".join(map(str, scores)) in joined, (nbname, scores)


def main():
    exp = {"scores": [], "classes": []}

    # --- byte copies of the reference's own binary fixtures -----------------
    for f in ("deer.bin", "3.image-idx3-ubyte"):
        shutil.copyfile(os.path.join(REF, "tests", "Test_image", f), os.path.join(OUT, f))
        os.chmod(os.path.join(OUT, f), 0o644)

    # --- CIFAR records from the demo images ---------------------------------
    def emit(name, img):
        with open(os.path.join(OUT, name + ".cifar"), "wb") as fp:
            fp.write(image_to_cifar(img))

    deer = Image.open(os.path.join(REF, "tests", "Test_image", "deer.jpg"))
    emit("deer", deer)
    # the notebooks use notebooks/pictures/deer.jpg: same bytes as the test image?
    deer_nb = open(os.path.join(REF, "notebooks", "pictures", "deer.jpg"), "rb").read()
    assert deer_nb == open(os.path.join(REF, "tests", "Test_image", "deer.jpg"), "rb").read()

    car = Image.open(os.path.join(REF, "notebooks", "pictures", "car.png"))
    car.thumbnail((64, 64), Image.LANCZOS, reducing_gap=None)  # CNV-BNN_Cifar10.ipynb cell
    emit("car", car)

    emit("six", Image.open(os.path.join(REF, "tests", "Test_image", "6.png")))
    emit("stop", Image.open(os.path.join(REF, "tests", "Test_image", "stop.jpg")))
    for f in ("cross.jpg", "end_no_overtaking.png", "stop.jpg"):
        emit("road_" + os.path.splitext(f)[0],
             Image.open(os.path.join(REF, "notebooks", "pictures", "road_signs", f)))

    # --- recorded score vectors (40 numbers) --------------------------------
    rec = [
        ("deer.cifar", "cnvW1A1", "cifar10", [234, 231, 265, 248, 410, 257, 224, 262, 226, 233],
         "notebooks/CNV-QNN_Cifar10.ipynb:164-173", "CNV-QNN_Cifar10.ipynb"),
        ("deer.cifar", "cnvW1A2", "cifar10", [-20, -46, -38, -6, 268, 6, -14, -28, -38, -30],
         "notebooks/CNV-QNN_Cifar10.ipynb:252-261", "CNV-QNN_Cifar10.ipynb"),
        ("deer.cifar", "cnvW2A2", "cifar10", [-24, -34, -21, -13, 244, 4, -7, -20, -27, -13],
         "notebooks/CNV-QNN_Cifar10.ipynb:326-335", "CNV-QNN_Cifar10.ipynb"),
        ("car.cifar", "cnvW1A1", "cifar10", [258, 417, 233, 206, 238, 215, 222, 238, 236, 249],
         "notebooks/CNV-BNN_Cifar10.ipynb:246-255", "CNV-BNN_Cifar10.ipynb"),
    ]
    for img, net, ds, scores, src, nb in rec:
        assert_recorded(nb, scores)
        exp["scores"].append({"input": img, "network": net, "params": ds,
                              "scores": scores, "source": src})

    # --- recorded class indices ----------------------------------------------
    exp["classes"] = [
        {"input": "3.image-idx3-ubyte", "network": "lfcW1A1", "params": "mnist", "class": 3,
         "source": "tests/test_qnn.py:44-73; bnn/src/network/make-hw.sh:127-129"},
        {"input": "3.image-idx3-ubyte", "network": "lfcW1A2", "params": "mnist", "class": 3,
         "source": "tests/test_qnn.py:44-73"},
        {"input": "deer.cifar", "network": "cnvW1A1", "params": "cifar10", "class": 4,
         "source": "tests/test_qnn.py:87-134"},
        {"input": "deer.cifar", "network": "cnvW1A2", "params": "cifar10", "class": 4,
         "source": "tests/test_qnn.py:87-134"},
        {"input": "deer.cifar", "network": "cnvW2A2", "params": "cifar10", "class": 4,
         "source": "tests/test_qnn.py:87-134"},
        {"input": "deer.bin", "network": "cnvW1A1", "params": "cifar10", "class": 4,
         "source": "bnn/src/network/make-hw.sh:123-125"},
        {"input": "deer.bin", "network": "cnvW1A2", "params": "cifar10", "class": 4,
         "source": "bnn/src/network/make-hw.sh:123-125"},
        {"input": "deer.bin", "network": "cnvW2A2", "params": "cifar10", "class": 4,
         "source": "bnn/src/network/make-hw.sh:123-125"},
        {"input": "six.cifar", "network": "cnvW1A1", "params": "streetview", "class": 5,
         "source": "tests/test_qnn.py:148-161"},
        {"input": "stop.cifar", "network": "cnvW1A1", "params": "road-signs", "class": 14,
         "source": "tests/test_qnn.py:176-189"},
        {"input": "road_cross.cifar", "network": "cnvW1A1", "params": "road-signs", "class": 27,
         "source": "notebooks/CNV-BNN_Road-Signs.ipynb:162,195"},
        {"input": "road_end_no_overtaking.cifar", "network": "cnvW1A1", "params": "road-signs",
         "class": 41, "source": "notebooks/CNV-BNN_Road-Signs.ipynb:162,195"},
        {"input": "road_stop.cifar", "network": "cnvW1A1", "params": "road-signs", "class": 14,
         "source": "notebooks/CNV-BNN_Road-Signs.ipynb:162,195"},
    ]
    # the interior of the deer.jpg record equals the reference's deer.bin (which
    # was made by an older bnn.py revision with 0 instead of 255 padding)
    a = np.frombuffer(open(os.path.join(OUT, "deer.cifar"), "rb").read(), np.uint8)[1:].reshape(3, 32, 32)
    b = np.frombuffer(open(os.path.join(OUT, "deer.bin"), "rb").read(), np.uint8)[1:].reshape(3, 32, 32)
    cols = np.where((b != 0).any(axis=(0, 1)))[0]
    assert (a[:, :, cols.min():cols.max() + 1] == b[:, :, cols.min():cols.max() + 1]).all()
    exp["notes"] = ("deer.cifar interior == deer.bin interior (checked at generation); "
                    "airplane.jpg / bird.jpg recorded scores are not reproducible (old-Pillow JPEG "
                    "draft scaling on the host) and are not used.")
    with open(os.path.join(OUT, "expected.json"), "w") as fp:
        json.dump(exp, fp, indent=1)
    print("wrote", len(exp["scores"]), "score vectors and", len(exp["classes"]), "class indices")


if __name__ == "__main__":
    sys.exit(main())
