"""CPU: the chunk plan of the host-data entry points (bnn_mi355x_chunk_plan is host-only arithmetic)."""
import ctypes as C

import gpu_lib as gl


def test_chunk_plan_properties():
    """the plan the host-data entry points cut a call by: covers [0, n) in order, no chunk above 32 768 images, small
    chunks first (the first transfer is what nothing overlaps), growing by at most x1.5 (host memory) / x1.25 (file) per
    step so that a chunk's transfer fits behind the previous chunk's stages; a file's chunks ramp down again at the end"""
    for network, scale in (("cnvW1A1", 1), ("lfcW1A1", 4)):
        L = gl.load(network)
        for from_file in (0, 1):
            for n in (0, 1, 2048, 4096, 4097, 10000, 32768, 32769, 70001, 131072, 131072 + 777, 1048576):
                bases = (C.c_int * 256)()
                k = L.bnn_mi355x_chunk_plan(n, from_file, bases, 256)
                e = [bases[i] for i in range(k)]
                assert e[0] == 0 and e[-1] == n and 2 <= k <= 256
                sizes = [b - a for a, b in zip(e, e[1:])]
                assert all(0 < s <= 32768 for s in sizes) or n == 0
                if n >= 131072:
                    assert sizes[0] == 2048 * scale and (max(sizes) == 32768 or (from_file and n < 1048576))
                    grow = 1.25 if from_file else 1.5
                    up = sizes[:sizes.index(max(sizes)) + 1]
                    assert all(b <= a * grow + 256 for a, b in zip(up, up[1:]))
                    if from_file:
                        assert sizes[-1] == 2048 * scale                 # ramp down: what follows the last byte is short
                    else:
                        assert sizes[-1] > 2048 * scale or n % 32768     # none
                if n <= 4096 * scale:
                    assert k == 2
