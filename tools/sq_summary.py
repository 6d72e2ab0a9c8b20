#!/usr/bin/env python3
"""tools/sq_summary.py DIR [DIR ...]: per-kernel SQ counter summary from rocprofv3 --pmc passes (one or more
output directories, each a separate pass).  Per kernel (product kernels only, averaged over dispatches):
every counter collected, the dispatch duration, and the derived figures the roofline argument needs:

  valu_busy      = SQ_ACTIVE_INST_VALU * 4 / SQ_BUSY_CU_CYCLES-equivalent   (see below)
  cyc_per_valu   = SIMD-cycles per wave64 VALU instruction while the kernel runs

Units (MI355X_MICROARCH.md, cycle-constants table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
QUAD-cycles (4 shader cycles) summed over waves; SQ_BUSY_CYCLES counts per SE... -- so nothing here relies on
an absolute unit that is not calibrated: the derived figures use (a) ratios between counters of the same
unit and (b) instruction counts against the dispatch duration and the clock measured in the same pass
(GRBM_GUI_ACTIVE / 8 XCDs / duration).
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"(k_\w+<[^>]*>|k_\w+)", name)
    return m.group(1) if m else None


def load(dirs):
    """per kernel, only the dispatches at its LARGEST grid: the library's load-time warm-up launches the same kernels
    on small batches, which must not be averaged into the figures of the measured batch"""
    rows = collections.defaultdict(list)
    for d in dirs:
        for path in glob.glob(d + "/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(path)):
                k = short(r["Kernel_Name"])
                if k:
                    rows[k].append((path, r))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    meta = {}
    for k, lst in rows.items():
        g = max(int(r["Grid_Size"]) for _, r in lst)
        seen = set()
        for path, r in lst:
            if int(r["Grid_Size"]) != g:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (path, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                meta[k] = {"grid": g, "wg": int(r["Workgroup_Size"]), "vgpr_granules": int(r["VGPR_Count"]),
                           "sgpr": int(r["SGPR_Count"]), "lds": int(r["LDS_Block_Size"])}
    return acc, dur, meta


def main():
    acc, dur, meta = load(sys.argv[1:])
    out = {}
    for k in sorted(acc, key=lambda k: -sum(dur[k]) / len(dur[k])):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        us = sum(dur[k]) / len(dur[k]) / 1e3
        row = {"dispatches": len(dur[k]), "avg_us": round(us, 2), **meta[k], "counters": {n: round(v, 1) for n, v in sorted(c.items())}}
        d = {}
        if "GRBM_GUI_ACTIVE" in c:
            d["clock_ghz"] = round(c["GRBM_GUI_ACTIVE"] / 8 / (us * 1e3), 3)
        clk = d.get("clock_ghz", 2.38)
        simd_cycles = us * 1e3 * clk * 1024          # SIMD-cycles available to the kernel (1024 SIMDs)
        if "SQ_INSTS_VALU" in c:
            d["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / max(c.get("SQ_WAVES", 1), 1), 1) if "SQ_WAVES" in c else None
            d["simd_cycles_per_valu_inst"] = round(simd_cycles / c["SQ_INSTS_VALU"], 3)
        if "SQ_ACTIVE_INST_VALU" in c and "SQ_INSTS_VALU" in c:
            # quad-cycles of VALU execution per VALU instruction, x4 = shader cycles the instruction holds the pipe
            d["valu_active_cycles_per_inst"] = round(4 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"], 3)
            d["valu_busy_frac_of_simd_time"] = round(4 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles, 4)
        if "SQ_WAVE_CYCLES" in c:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA"):
                if n in c:
                    d[n.lower() + "_frac_of_wave_cycles"] = round(c[n] / c["SQ_WAVE_CYCLES"], 4)
            d["avg_waves_per_simd"] = round(4 * c["SQ_WAVE_CYCLES"] / simd_cycles, 2)
        row["derived"] = d
        out[k] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
