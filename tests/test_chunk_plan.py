"""CPU: the chunk plan of the host-data entry points (bnn_mi355x_chunk_plan is host-only arithmetic)."""
import ctypes as C

import gpu_lib as gl


def test_chunk_plan_properties():
    """the plan the host-data entry points cut a call by: covers [0, n) in order, no chunk above 32 768 images, small
    chunks at both ends of a large call (first transfer / last stages are what nothing overlaps)"""
    for network, scale in (("cnvW1A1", 1), ("lfcW1A1", 4)):
        L = gl.load(network)
        for n in (0, 1, 2048, 4096, 4097, 10000, 32768, 32769, 70001, 131072, 131072 + 777, 1048576):
            bases = (C.c_int * 128)()
            k = L.bnn_mi355x_chunk_plan(n, bases, 128)
            e = [bases[i] for i in range(k)]
            assert e[0] == 0 and e[-1] == n and k >= 2
            sizes = [b - a for a, b in zip(e, e[1:])]
            assert all(0 < s <= 32768 for s in sizes) or n == 0
            if n >= 131072:
                assert sizes[0] == 2048 * scale and sizes[-1] == 4096 * scale and max(sizes) == 32768
            if n <= 4096 * scale:
                assert k == 2
