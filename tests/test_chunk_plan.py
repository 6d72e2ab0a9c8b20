"""CPU: the chunk plan of the host-data entry points (bnn_mi355x_chunk_plan is host-only arithmetic)."""
import ctypes as C
import os
import subprocess
import sys

import gpu_lib as gl


def plan(L, n, from_file=0):
    bases = (C.c_int * 512)()
    k = L.bnn_mi355x_chunk_plan(n, from_file, bases, 512)
    return [bases[i] for i in range(k)]


def test_chunk_plan_properties():
    """the plan the host-data entry points cut a call by: covers [0, n) in order, no chunk above 16 384 (CNV) / 32 768 (LFC) images, small
    chunks first (the first transfer is what nothing overlaps: 512 CIFAR images; the LFC nets, whose host paths ship binarised
    words: 8 192), doubling up to 4 096 (LFC: all the way) and growing by half from there on so that a chunk's transfer fits behind
    the previous chunk's stages; a CNV call of 8 192 ... 32 767 images ramps down again at its end (what follows the last byte is
    the last chunk's stages), smaller and larger calls and the LFC nets do not"""
    for network, scale, big, head in (("cnvW1A1", 1, 16384, 512), ("lfcW1A1", 4, 32768, 8192)):
        L = gl.load(network)
        for from_file in (0, 1):
            for n in (0, 1, 512, 1024, 1025, 1537, 2048, 4096, 4097, 10000, 16385, 32767, 32768, 32769, 70001, 131072, 131072 + 777, 1048576):
                e = plan(L, n, from_file)
                k = len(e)
                assert e[0] == 0 and e[-1] == n and 2 <= k <= 512
                sizes = [b - a for a, b in zip(e, e[1:])]
                assert all(0 < s <= big for s in sizes) or n == 0
                if n <= 2 * head:
                    assert k == 2
                    continue
                assert sizes[0] == head or (sizes[0] > head and len(sizes) == 2)     # (a tiny middle chunk joins the first)
                two_sided = scale == 1 and 8192 <= n < 32768
                if two_sided:
                    def ramp(seq):                                    # how far `seq` follows the growth rule from `head`
                        want, k = head, 0
                        while k < len(seq) and seq[k] == want:
                            want = min((want * (200 if want < 4096 else 150) // 100 + 255) & ~255, big)
                            k += 1
                        return k
                    a, b = ramp(sizes), ramp(sizes[::-1])
                    assert sizes[-1] == head and min(sizes) >= head and (a >= 1 or len(sizes) == 2) and b >= 1
                    assert a + b >= len(sizes) - 1 and abs(a - b) <= 1     # the two ramps and at most the remainder between them
                else:
                    up = sizes[:-1]                                   # (the last chunk is what is left, or has taken a small remainder in)
                    assert all(b >= a and b <= a * (2.0 if a < 4096 * scale or scale == 4 else 1.5) + 256 for a, b in zip(up[:-1], up[1:-1]))
                    assert len(sizes) < 2 or sizes[-1] * 2 >= sizes[-2] or sizes[-1] + sizes[-2] > big
                    if n >= 131072 * scale:
                        assert max(sizes) == big
    # the reference's own call size: a 10 000-record test-set file
    e = plan(gl.load("cnvW1A1"), 10000, 1)
    assert [b - a for a, b in zip(e, e[1:])] == [512, 1024, 2048, 2832, 2048, 1024, 512]
    e = plan(gl.load("lfcW1A1"), 10000, 1)
    assert e == [0, 10000]


def test_chunk_plan_override_cannot_exceed_the_workspace():
    """BNN_MI355X_CHUNKS (tuning / A-B switch): whatever its fields say, no chunk is larger than its `max` field, and that is
    at most 131 072 images -- the activation workspace of one pass.  (Round 3: head 131072 on a call of 200 000 images gave ONE
    chunk of 200 000.)  Read at every call, so a child process per setting is not needed -- but the variable must not leak
    into the other tests: run in a child."""
    code = r"""
import ctypes as C, os, sys
sys.path.insert(0, %r)
import gpu_lib as gl
L = gl.load("cnvW1A1")
def sizes(n):
    b = (C.c_int * 512)()
    k = L.bnn_mi355x_chunk_plan(n, 0, b, 512)
    e = [b[i] for i in range(k)]
    assert e[0] == 0 and e[-1] == n
    return [y - x for x, y in zip(e, e[1:])]
os.environ["BNN_MI355X_CHUNKS"] = "131072:0:131072"
assert sizes(131072) == [131072]
s = sizes(200000)
assert max(s) <= 131072 and sum(s) == 200000 and len(s) == 2, s
os.environ["BNN_MI355X_CHUNKS"] = "2000000000:2000000000:4096:400"          # 2 * head would overflow an int
s = sizes(1000000)
assert max(s) <= 4096 and sum(s) == 1000000, s[:4]
os.environ["BNN_MI355X_CHUNKS"] = "65536:0:8192"                             # head above max: clamped to max
s = sizes(20000)
assert max(s) <= 8192 and sum(s) == 20000, s
os.environ["BNN_MI355X_CHUNKS"] = "0:0:200000"                               # max above the workspace: the override is ignored
assert max(sizes(400000)) <= 16384
print("ok")
""" % os.path.join(gl.ROOT, "tests")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
