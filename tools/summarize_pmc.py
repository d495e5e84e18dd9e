#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes: per-kernel average of every counter (per dispatch)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name") or row.get("Kernel-Name") or ""
            short = name.split("(")[0].replace("void ", "").replace("oddio_hip::", "")
            try:
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
            except (KeyError, ValueError):
                pass
out = {}
for k, counters in acc.items():
    if not any(t in k for t in ("spatial_", "reduce_", "mixer_", "ordered_", "buffered_")):
        continue
    out[k] = {c: sum(v) / len(v) for c, v in sorted(counters.items())}
    out[k]["_dispatches"] = max(len(v) for v in counters.values())
print(json.dumps(out, indent=1))
