#!/usr/bin/env python3
"""A long fault campaign replayed in the CPU restatement (as tests/test_faults.py does for short ones):
usage: fuzz_campaign.py [network [n_images [flips [word_size [target]]]]]"""
import ctypes as C, os, struct, sys, tempfile
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_lib as gl
import oracle_lib as ol
net = sys.argv[1] if len(sys.argv) > 1 else "cnvW2A2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
flips = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
ws = int(sys.argv[4]) if len(sys.argv) > 4 else 1
target = int(sys.argv[5]) if len(sys.argv) > 5 else 0
cnv = net.startswith("cnv")
ds = "cifar10" if cnv else "mnist"
L = gl.load(net); pdir = gl.param_dir(ds, net); L.load_parameters(pdir.encode())
rng = np.random.default_rng(4)
imgs = rng.integers(0, 256, (n, 3072 if cnv else 784), dtype=np.uint8)
with tempfile.NamedTemporaryFile(dir="/tmp") as f:
    if cnv:
        f.write(np.concatenate([np.ones((n, 1), np.uint8), imgs], axis=1).tobytes())
    else:
        f.write(struct.pack(">4I", 0x803, n, 28, 28) + imgs.tobytes())
    f.flush()
    L.bnn_mi355x_set_fault_seed(4242)
    cnt = C.c_int(0)
    p = L.inference_multiple_with_faults(f.name.encode(), 10, C.byref(cnt), None, flips, ws, target, None, 0)
    assert p and cnt.value == n
    got = np.ctypeslib.as_array(p, (n,)).copy(); L.free_results(p)
rec = (C.c_int * (8 * flips))(); assert L.bnn_mi355x_last_faults(rec, flips) == flips
recs = np.array(rec[:], np.int32).reshape(flips, 8)
o = ol.Oracle(net, pdir)
want = np.zeros(n, np.int32); k = start = 0
while start < n:
    while k < flips and recs[k, 0] <= start:
        assert o.apply_fault(recs[k]) >= 0; k += 1
    end = int(recs[k, 0]) if k < flips else n
    want[start:end] = o.classes_batched(imgs[start:end], 10); start = end
W = o.weights(1)
print("%s: %d images, %d flips (word %d, target %d): replay %s; %d of the layer-1 weights are now -2; accuracy vs clean %.1f %%"
      % (net, n, flips, ws, target, "identical" if (got == want).all() else "DIFFERENT", int((W == -2).sum()),
         100.0 * (got == ol.Oracle(net, pdir).classes_batched(imgs, 10)).mean()))
assert (got == want).all()
