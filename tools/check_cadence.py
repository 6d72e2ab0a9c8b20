#!/usr/bin/env python3
"""tools/check_cadence.py [kernels.o]: verify, in the BUILT gfx950 code object, the instruction cadence the
throughput kernels' speed depends on (profiles/r01_microbench9_nop_cadence.txt):

  * the per-neuron loop of each checked kernel holds exactly the expected number of (logic op, v_bcnt_u32_b32)
    pairs, every v_bcnt directly preceded by its logic op (v_xor_b32 / v_bitop3_b32 / v_and_b32);
  * every pair is followed by exactly ONE issue bubble before the next VALU instruction: an `s_nop 0`, or a
    scalar instruction that the compiler scheduled there instead (the (op, bcnt, bubble) cadence issues in
    6.3 SIMD-cycles per pair, pairs back to back in 8.0, two bubbles in 7.9-8.3).  Never two pairs back to
    back; per loop iteration at most SLACK_VALU pairs may be followed directly by one of the iteration's few
    other VALU instructions (loop counter, decision tail) and at most SLACK_BUBBLES by more than one bubble
    (loop control);
  * no v_cmp / v_cndmask / v_mov in the loop body (decisions are sign bits shifted in by v_alignbit; chains
    start from a seed operand, not from a zeroed register).

The nops are inserted partly by the compiler (behind an inline-asm statement whose VGPR result the next
instruction touches) and partly written out in the asm; this check is what turns that habit from luck into a
build-time fact: it runs in `pytest -m "not gpu"` (tests/test_kernel_cadence.py) and fails when a toolchain
changes the stream.  Exit code 0 = all kernels conform; the report goes to stdout as JSON.
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel (demangled prefix) -> (logic ops of a pair, pairs per neuron-loop iteration, other VALU allowed per iteration)
#   k_quad_x<CW, ID, POOL, 32>: 9*CW words x 2 halves x 4 pixels; tail: v_min + v_min3 + v_alignbit (pool) / 4 v_alignbit
#   k_vec_x<KW, ...>: two neurons per iteration: 2 * KW words x 2 halves; tail: 2 v_alignbit
#   k_quad<ARITH=2 (W1A2) / 3 (W2A2), ...>: v_bitop3 pairs (W2A2: v_and pair + v_bitop3 pair per half word)
CHECKS = [
    ("k_quad_x<1, 30, true, 32>", ("v_xor_b32",), 72, 4),
    ("k_quad_x<1, 14, false, 32>", ("v_xor_b32",), 72, 6),
    ("k_quad_x<2, 12, true, 32>", ("v_xor_b32",), 144, 4),
    ("k_vec_x<18, true, 2, 5, 32, false>", ("v_xor_b32",), 72, 3),
    ("k_vec_x<16, false, 1, 1, 32, false>", ("v_xor_b32",), 64, 3),
    ("k_vec_x<13, false, 1, 1, 32, false>", ("v_xor_b32",), 52, 3),
    ("k_quad<2, 1, 30, true, true, 32, false>", ("v_bitop3_b32",), 72, 12),
    ("k_quad<2, 2, 12, true, true, 32, false>", ("v_bitop3_b32",), 144, 12),
    ("k_quad<3, 1, 30, true, true, 32, false>", ("v_and_b32", "v_bitop3_b32"), 144, 12),
    ("k_vec<2, 16, true, false, 1, 1, 32, false, false>", ("v_bitop3_b32",), 64, 10),
    # one-launch LFC kernel: per image 2 * 16 pairs (weights in VGPRs, the image in SGPRs), then the ballot (v_cmp)
    # and two v_writelane
    ("k_lfc_block_s<false>", ("v_xor_b32",), 32, 4, ("v_cmp",)),
    ("k_lfc_block_s<true>", ("v_xor_b32",), 32, 4, ("v_cmp",)),   # the same from host-binarised words (csrc/pack_inputs.h)
]
FORBIDDEN = ("v_cmp", "v_cndmask", "v_mov_b32")
SLACK_VALU, SLACK_BUBBLES = 4, 2


def disassemble(obj):
    """host object with an embedded HIP fat binary -> symbolised disassembly text of its gfx950 code object"""
    tmp = tempfile.mkdtemp(prefix="cadence_")
    try:
        local = os.path.join(tmp, "k.o")
        shutil.copy(obj, local)
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", local], check=True, cwd=tmp, capture_output=True)
        cos = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not cos:
            raise RuntimeError("no gfx950 code object in " + obj)
        out = subprocess.run([LLVM + "/llvm-objdump", "-d", "--symbolize-operands", "--no-show-raw-insn", os.path.join(tmp, cos[0])],
                             check=True, capture_output=True, text=True).stdout
        return subprocess.run([shutil.which("c++filt") or "c++filt"], input=out, capture_output=True, text=True, check=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def functions(text):
    """{demangled kernel name: [(label or None, mnemonic, operands)]}"""
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            name = m.group(1)
            if re.fullmatch(r"L\d+", name):
                if cur is not None:
                    cur.append((name, None, None))
                continue
            cur = out.setdefault(name, [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        parts = ins.split(None, 1)
        cur.append((None, re.sub(r"_e(32|64)$", "", parts[0]), parts[1] if len(parts) > 1 else ""))
    return out


def inner_loops(body):
    """instruction lists of the innermost loops: [label L ... backward branch to L] with no other label inside"""
    pos = {lab: i for i, (lab, _, _) in enumerate(body) if lab}
    loops = []
    for i, (lab, mn, ops) in enumerate(body):
        if mn and mn.startswith("s_cbranch") and ops.strip() in pos and pos[ops.strip()] < i:
            seg = body[pos[ops.strip()] + 1: i]
            if not any(l for l, _, _ in seg):
                loops.append([(m, o) for _, m, o in seg])
    return loops


def is_valu(mn):
    return mn.startswith("v_")


def check_loop(loop, logic_ops, pairs, other_valu, allowed=()):
    problems = []
    n_pairs = 0
    i = 0
    others = 0
    direct_valu = long_bubbles = 0
    while i < len(loop):
        mn, ops = loop[i]
        if any(mn.startswith(f) for f in FORBIDDEN) and not any(mn.startswith(a) for a in allowed):
            problems.append("forbidden instruction in the loop body: %s %s" % (mn, ops))
        if mn == "v_bcnt_u32_b32":
            problems.append("v_bcnt without its logic op directly in front (instruction %d)" % i)
        if mn in logic_ops and i + 1 < len(loop) and loop[i + 1][0] == "v_bcnt_u32_b32":
            dst = ops.split(",")[0].strip()
            src = [x.strip() for x in loop[i + 1][1].split(",")]
            if len(src) < 2 or src[1] != dst:
                problems.append("v_bcnt does not consume the preceding logic op's result (instruction %d)" % i)
            n_pairs += 1
            # bubbles between this pair and the next VALU instruction
            j = i + 2
            bubbles = 0
            while j < len(loop) and not is_valu(loop[j][0]):
                bubbles += 1
                j += 1
            if j < len(loop) and bubbles == 0:
                if loop[j][0] in logic_ops and j + 1 < len(loop) and loop[j + 1][0] == "v_bcnt_u32_b32":
                    problems.append("two pairs back to back at instruction %d (no issue bubble between them)" % i)
                else:
                    direct_valu += 1
            elif j < len(loop) and bubbles > 1:
                long_bubbles += 1
            i += 2
            continue
        if is_valu(mn):
            others += 1
        i += 1
    if n_pairs != pairs:
        problems.append("%d (logic op, v_bcnt) pairs in the loop body, expected %d" % (n_pairs, pairs))
    if direct_valu > SLACK_VALU:
        problems.append("%d pairs are followed directly by another VALU instruction, at most %d tolerated" % (direct_valu, SLACK_VALU))
    if long_bubbles > SLACK_BUBBLES:
        problems.append("%d pairs are followed by more than one bubble, at most %d tolerated" % (long_bubbles, SLACK_BUBBLES))
    if others > other_valu:
        problems.append("%d other VALU instructions per iteration, at most %d expected" % (others, other_valu))
    return {"pairs": n_pairs, "other_valu": others, "instructions": len(loop), "pairs_followed_by_valu": direct_valu,
            "pairs_followed_by_several_bubbles": long_bubbles, "problems": problems}


def run(obj):
    fns = functions(disassemble(obj))
    report, ok = {}, True
    for prefix, logic_ops, pairs, other, *rest in CHECKS:
        names = [n for n in fns if ("::" + prefix + "(") in n or n.startswith("void " + prefix + "(") or (prefix + "(") in n]
        if len(names) != 1:
            report[prefix] = {"problems": ["kernel not found in the code object (%d matches)" % len(names)]}
            ok = False
            continue
        loops = inner_loops(fns[names[0]])
        # the neuron loop is the innermost loop with the most v_bcnt
        loops.sort(key=lambda l: -sum(1 for m, _ in l if m == "v_bcnt_u32_b32"))
        if not loops:
            report[prefix] = {"problems": ["no loop found"]}
            ok = False
            continue
        r = check_loop(loops[0], logic_ops, pairs, other, rest[0] if rest else ())
        report[prefix] = r
        ok = ok and not r["problems"]
    return ok, report


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "bnn-pynq_amd", "build", "kernels.o")
    ok, report = run(obj)
    print(json.dumps(report, indent=1))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
