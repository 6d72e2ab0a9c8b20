#!/usr/bin/env python3
"""The flow of the reference's notebooks (notebooks/CNV-QNN_Cifar10.ipynb, LFC-QNN_MNIST.ipynb) on an
MI355X: same `bnn` API, parameters and inputs; run from the repository root on a GPU host.

    python examples/classify_demo.py
"""
import os
import sys

import numpy as np
import torch  # noqa: F401  first: one HIP runtime per process

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bnn-pynq_amd"))
import bnn  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

print(bnn.available_params(bnn.NETWORK_CNVW1A1))
for net in (bnn.NETWORK_CNVW1A1, bnn.NETWORK_CNVW1A2, bnn.NETWORK_CNVW2A2):
    clf = bnn.CnvClassifier(net, "cifar10", bnn.RUNTIME_HW)
    deer = os.path.join(GOLDEN, "deer.cifar")              # image_to_cifar(deer.jpg) of the reference's test image
    ranking = clf.classify_cifar_details(deer)
    print("%s: %s -> %s" % (net, ranking.tolist(), clf.class_name(int(np.argmax(ranking)))))

# a CIFAR-10-sized batch of synthetic images, from memory (extension) and through the file API (reference)
clf = bnn.CnvClassifier(bnn.NETWORK_CNVW1A1, "cifar10", bnn.RUNTIME_HW)
imgs = np.random.default_rng(0).integers(0, 256, (10000, 3072), dtype=np.uint8)
classes = clf.classify_array(imgs)
print("10 000 images: %.2f us/image on the GPU, class histogram %s" % (clf.usecPerImage, np.bincount(classes, minlength=10).tolist()))

lfc = bnn.LfcClassifier(bnn.NETWORK_LFCW1A1, "mnist", bnn.RUNTIME_HW)
print("MNIST digit:", lfc.classify_mnist(os.path.join(GOLDEN, "3.image-idx3-ubyte")))
