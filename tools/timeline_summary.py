#!/usr/bin/env python3
"""tools/timeline_summary.py DIR: the device-side timeline of the LAST call in a rocprofv3 --kernel-trace
--memory-copy-trace run of tools/host_timeline.py: first copy -> last kernel, time covered by copies, by kernels, by
both, by neither; the longest gaps in the kernel stream; the first and last events."""
import csv
import glob
import sys

d = sys.argv[1]
ev = []
for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0][-40:]))
for path in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "copy")))
ev.sort()
# calls are separated by >= 30 ms of silence
calls, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur) > 30e6:
        calls.append(cur)
        cur = []
    cur.append(e)
calls.append(cur)
last = calls[-1]
t0, t1 = last[0][0], max(e[1] for e in last)


def union(evs):
    tot, end = 0, None
    for s, e, *_ in sorted(evs):
        if end is None or s > end:
            tot += e - s
            end = e
        elif e > end:
            tot += e - end
            end = e
    return tot


K = [e for e in last if e[2] == "K"]
Cc = [e for e in last if e[2] == "C"]
print("last call: %d kernels, %d copies, first event -> last end %.2f ms" % (len(K), len(Cc), (t1 - t0) / 1e6))
print("  kernels busy %.2f ms, copies busy %.2f ms, either %.2f ms" % (union(K) / 1e6, union(Cc) / 1e6, union(last) / 1e6))
print("  first kernel starts %.2f ms after the first event; last copy ends at %.2f ms; last kernel ends at %.2f ms"
      % ((K[0][0] - t0) / 1e6, (max(e[1] for e in Cc) - t0) / 1e6, (max(e[1] for e in K) - t0) / 1e6))
gaps = sorted(((K[i + 1][0] - max(k[1] for k in K[:i + 1]), (K[i][1] - t0) / 1e6, K[i][3], K[i + 1][3]) for i in range(len(K) - 1)), reverse=True)
print("  largest gaps in the kernel stream (us, at ms, between):")
for g in gaps[:8]:
    print("    %8.1f us at %.2f ms  %s -> %s" % (g[0] / 1e3, g[1], g[2], g[3]))
print("  copies: " + " ".join("%s:%.2f-%.2f" % (c[3][:3], (c[0] - t0) / 1e6, (c[1] - t0) / 1e6) for c in Cc[:40]))

# chunk view: every k_strip_records launch closes a chunk's transfer; list when it ran and when the chunk's last stage ended
strips = [e for e in K if "strip" in e[3]]
if strips:
    print("  chunks (strip kernel start ms, next strip start ms, stage kernels busy between them ms):")
    for i, sk in enumerate(strips):
        nxt = strips[i + 1][0] if i + 1 < len(strips) else t1
        print("    chunk %2d: transfer complete at %6.2f ms" % (i, (sk[0] - t0) / 1e6))
    stage = [e for e in K if "strip" not in e[3]]
    # idle gaps of the stage stream longer than 50 us
    g = []
    for a, b in zip(stage, stage[1:]):
        if b[0] - a[1] > 50e3:
            g.append(((a[1] - t0) / 1e6, (b[0] - a[1]) / 1e3))
    print("  stage-stream gaps > 50 us (at ms: us): " + " ".join("%.2f:%.0f" % x for x in g))
    print("  sum of those gaps: %.2f ms; stages busy %.2f ms" % (sum(x[1] for x in g) / 1e3, union(stage) / 1e6))
