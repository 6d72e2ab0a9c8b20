"""CPU: the chunk plan of the host-data entry points (bnn_mi355x_chunk_plan is host-only arithmetic)."""
import ctypes as C

import gpu_lib as gl


def test_chunk_plan_properties():
    """the plan the host-data entry points cut a call by: covers [0, n) in order, no chunk above 16 384 (CNV) / 32 768 (LFC) images, small
    chunks first (the first transfer is what nothing overlaps: 2 048 images from a buffer, 4 096 from a file), growing by
    at most x1.5 per step so that a chunk's transfer fits behind the previous chunk's stages, no ramp down"""
    for network, scale, big in (("cnvW1A1", 1, 16384), ("lfcW1A1", 4, 32768)):
        L = gl.load(network)
        for from_file in (0, 1):
            head = (2 if from_file else 1) * 2048 * scale
            for n in (0, 1, 2048, 4096, 4097, 10000, 32768, 32769, 70001, 131072, 131072 + 777, 1048576):
                bases = (C.c_int * 256)()
                k = L.bnn_mi355x_chunk_plan(n, from_file, bases, 256)
                e = [bases[i] for i in range(k)]
                assert e[0] == 0 and e[-1] == n and 2 <= k <= 256
                sizes = [b - a for a, b in zip(e, e[1:])]
                assert all(0 < s <= big for s in sizes) or n == 0
                if n > 2 * head:
                    assert sizes[0] == head
                    up = sizes[:-1]                                   # (the last chunk is what is left, or has taken a small remainder in)
                    assert all(b <= a * 1.5 + 256 and b >= a for a, b in zip(up[:-1], up[1:-1])) and (len(sizes) < 2 or sizes[-1] * 2 >= sizes[-2] or sizes[-1] + sizes[-2] > big)
                    if n >= 131072 * scale:
                        assert max(sizes) == big
                else:
                    assert k == 2
