"""CPU build check: the instruction stream of the throughput kernels in the BUILT gfx950 code object has the
cadence their speed depends on -- every (logic op, v_bcnt) pair followed by one issue bubble, nothing but the
pairs and a handful of decision instructions per neuron (tools/check_cadence.py).  A toolchain
that stops producing that stream costs ~20 % with every parity test still green: this is where it shows."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_cadence  # noqa: E402

OBJ = os.path.join(ROOT, "bnn-pynq_amd", "build", "kernels.o")


def test_inner_loops_have_the_measured_cadence():
    if not os.path.exists(OBJ):  # a tree that arrived with prebuilt libraries only: rebuild the object (hipcc cross-compiles)
        import subprocess
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "bnn-pynq_amd"), "build/kernels.o"], check=True)
    ok, report = check_cadence.run(OBJ)
    bad = {k: v["problems"] for k, v in report.items() if v["problems"]}
    assert ok and not bad, bad
    l1 = report["k_quad_x<1, 30, true, 32>"]                 # the dominant kernel: CNV layer 1
    assert l1["pairs"] == 72 and l1["other_valu"] <= 4       # 9 words x 2 halves x 4 pixels; v_min, v_min3, v_alignbit (+ loop counter)


def test_checker_notices_a_broken_stream():
    """the checker itself: pairs back to back, a missing logic op, a v_cndmask are all reported"""
    good = [("v_xor_b32", "v1, s0, v2"), ("v_bcnt_u32_b32", "v3, v1, v3"), ("s_nop", "0")] * 4
    assert not check_cadence.check_loop(good, ("v_xor_b32",), 4, 0)["problems"]
    back_to_back = [x for x in good if x[0] != "s_nop"]
    assert any("back to back" in p for p in check_cadence.check_loop(back_to_back, ("v_xor_b32",), 4, 0)["problems"])
    assert check_cadence.check_loop(good[:-3], ("v_xor_b32",), 4, 0)["problems"]                       # a pair missing
    assert check_cadence.check_loop(good + [("v_cndmask_b32", "v0, v1, v2, vcc")], ("v_xor_b32",), 4, 1)["problems"]
    assert check_cadence.check_loop([("v_bcnt_u32_b32", "v3, v1, v3")] + good, ("v_xor_b32",), 4, 0)["problems"]
    two_nops = []
    for k in range(4):
        two_nops += good[:2] + [("s_nop", "0"), ("s_nop", "0")]
    assert any("more than one bubble" in p for p in check_cadence.check_loop(two_nops + good[:2], ("v_xor_b32",), 5, 0)["problems"])
