#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run (counter_collection.csv): per kernel the mean of every counter over its dispatches."""
import csv
import json
import sys
from collections import defaultdict

path = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
with open(path) as f:
    for row in csv.DictReader(f):
        name = row["Kernel_Name"].split("(")[0]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"dispatches": max(len(v) for v in d.values())} for k, d in acc.items()}
keep = sys.argv[2:] or None
for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if keep and not any(s in k for s in keep):
        continue
    print(k, json.dumps({c: round(v, 1) for c, v in d.items()}))
