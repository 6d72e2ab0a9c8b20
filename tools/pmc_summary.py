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
    out = collections.defaultdict(list)
    for path in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(k_\w+<[^>]*>|k_\w+)", r["Kernel_Name"])
            if m:
                out[m.group(1)].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
    tf = tw = 0.0
    print("%-30s %12s %12s %12s" % ("kernel", "fetch B/img", "x2 (gfx950)", "write B/img"))
    for k in fetch:
        f, w = fetch[k] * 1024 / n, write.get(k, 0.0) * 1024 / n
        tf += f
        tw += w
        print("%-30s %12.1f %12.1f %12.1f" % (k, f, 2 * f, w))
    print("%-30s %12.1f %12.1f %12.1f" % ("TOTAL per image", tf, 2 * tf, tw))
    print(json.dumps({"images_per_batch": n, "fetch_bytes_per_image_raw": round(tf, 1),
                      "fetch_bytes_per_image_x2": round(2 * tf, 1), "write_bytes_per_image": round(tw, 1),
                      "hbm_bytes_per_step": int((2 * tf + tw) * n)}))


if __name__ == "__main__":
    main()
