#!/usr/bin/env python3
"""A fault-injection campaign with the reference's `bnn.faults` drivers (the flow of the fork's fault
notebooks) on an MI355X: 2 000 synthetic CIFAR-10-shaped images whose "labels" are the fault-free
classes (control accuracy 100 %), three runs for each of {50, 500} upsets x {weight bit, threshold word}.

    python examples/fault_campaign.py [output_dir]
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch  # noqa: F401  first: one HIP runtime per process

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bnn-pynq_amd"))
import bnn  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="bnn_faults_")
rng = np.random.default_rng(0)
rec = rng.integers(0, 256, (2000, 3073), dtype=np.uint8)
path = os.path.join(out, "set.bin")
os.makedirs(out, exist_ok=True)
rec.tofile(path)
labels = bnn.CnvClassifier(bnn.NETWORK_CNVW1A1, "cifar10").classify_cifars(path).tolist()

test = bnn.faults.CNVFaultTest.CIFARTest(bnn.NETWORK_CNVW1A1, path, labels)
T = bnn.faults.NetworkTest.TestType
bnn.faults.NetworkTest(test).test_network(out, 3, [50, 500], [T.weight_bit(), T.threshold_word()])
for flips in (50, 500):
    stats = json.load(open(os.path.join(out, "cnvW1A1", "cifar10", "%dflips" % flips, "cnvW1A1_cifar10_stats.json")))
    for name, e in stats["results"].items():
        print("%4d x %-14s accuracy min %.2f avg %.2f max %.2f (%d of 3 runs changed something)"
              % (flips, name, e["min accuracy"], e["avg accuracy"], e["max accuracy"], e["effective count"]))
print("results under", out)
