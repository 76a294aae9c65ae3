#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from a rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ pass (counter_collection.csv),
corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for gfx950: requests × 64 B, fetches doubled for wide
coalesced streaming reads (the counter tallies their 128-B requests at 64 B), writes as counted.
    python scripts/pmc_traffic.py <counter_collection.csv> <kernel substring> <algorithmic bytes> [out.json]"""
import csv
import json
import sys
from collections import defaultdict

path, key, algo = sys.argv[1], sys.argv[2], float(sys.argv[3])
acc = defaultdict(list)
name = None
with open(path) as f:
    for row in csv.DictReader(f):
        if key in row["Kernel_Name"]:
            name = row["Kernel_Name"].split("(")[0]
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in acc.items()}
rd, wr = mean.get("TCC_EA0_RDREQ", 0.0) * 64, mean.get("TCC_EA0_WRREQ", 0.0) * 64
out = {"kernel": name, "dispatches": max(len(v) for v in acc.values()), "fetch_bytes_per_launch_raw": rd,
       "fetch_bytes_per_launch_x2_gfx950": 2 * rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": 2 * rd + wr,
       "algorithmic_bytes_per_launch": algo, "ratio_to_algorithmic": (2 * rd + wr) / algo,
       "source": "rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ (separate pass), x64 B, fetch doubled per the guide"}
print(json.dumps(out, indent=1))
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
