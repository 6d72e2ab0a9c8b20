#!/usr/bin/env python3
"""tools/make_traffic_json.py NETWORK PMC_TRAFFIC_TXT SOURCE_NOTE [OUT ...]: the JSON line that ends tools/pmc_summary.py's
report -> the entry of profiles/traffic.json that bench.py quotes as roofline.traffic (+ the per-stage table)."""
import json
import sys

net, txt, note = sys.argv[1], sys.argv[2], sys.argv[3]
d = json.loads(open(txt).read().strip().splitlines()[-1])
entry = {"hbm_bytes_per_step": d["hbm_bytes_per_step"], "images_per_step": d["images_per_batch"],
         "fetch_bytes_per_image_x2": d["fetch_bytes_per_image_x2"], "write_bytes_per_image": d["write_bytes_per_image"],
         "source": note, "stages": d.get("stages", [])}
for out in sys.argv[4:]:
    try:
        cur = json.load(open(out))
    except Exception:
        cur = {}
    cur[net] = entry
    json.dump(cur, open(out, "w"), indent=1)
print(json.dumps(entry)[:300])
