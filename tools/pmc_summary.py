#!/usr/bin/env python3
"""tools/pmc_summary.py FETCH_DIR WRITE_DIR [batch]: per-kernel HBM traffic from two rocprofv3
--pmc passes (FETCH_SIZE and WRITE_SIZE, KB per dispatch, averaged over dispatches).
gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM); both the raw
and the doubled figure are printed, the doubled one is what bench.py's roofline.traffic uses."""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    """average over the dispatches of each kernel AT ITS LARGEST GRID: the library's load-time warm-up launches the
    same kernels on a few thousand images, which must not dilute the per-batch figures"""
    out = collections.defaultdict(list)
    for path in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(k_\w+<[^>]*>|k_\w+)", r["Kernel_Name"])
            if m:
                out[m.group(1)].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    res = {}
    for k, v in out.items():
        g = max(x[0] for x in v)
        vals = [x[1] for x in v if x[0] == g]
        res[k] = (sum(vals) / len(vals), g)
    return res


def first_start(d):
    """kernel -> first Start_Timestamp in the pass: the order of the stages"""
    out = {}
    for path in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            m = re.search(r"(k_\w+<[^>]*>|k_\w+)", r["Kernel_Name"])
            if m and "Start_Timestamp" in r:
                t = int(r["Start_Timestamp"])
                out[m.group(1)] = min(out.get(m.group(1), t), t)
    return out


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    order = first_start(sys.argv[1])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
    # kernels of the full-size batch only: those whose largest grid covers at least one thread per image
    fetch = {k: v[0] for k, v in fetch.items() if v[1] >= n}
    write = {k: v[0] for k, v in write.items() if v[1] >= n}
    tf = tw = 0.0
    print("%-30s %12s %12s %12s" % ("kernel", "fetch B/img", "x2 (gfx950)", "write B/img"))
    for k in fetch:
        f, w = fetch[k] * 1024 / n, write.get(k, 0.0) * 1024 / n
        tf += f
        tw += w
        print("%-30s %12.1f %12.1f %12.1f" % (k, f, 2 * f, w))
    print("%-30s %12.1f %12.1f %12.1f" % ("TOTAL per image", tf, 2 * tf, tw))
    stages = [{"kernel": k, "fetch_bytes_per_image_x2": round(2 * fetch[k] * 1024 / n, 1), "write_bytes_per_image": round(write.get(k, 0.0) * 1024 / n, 1)}
              for k in sorted(fetch, key=lambda k: order.get(k, 0))]          # in launch order = stage order
    print(json.dumps({"images_per_batch": n, "fetch_bytes_per_image_raw": round(tf, 1),
                      "fetch_bytes_per_image_x2": round(2 * tf, 1), "write_bytes_per_image": round(tw, 1),
                      "hbm_bytes_per_step": int((2 * tf + tw) * n), "stages": stages}))


if __name__ == "__main__":
    main()
