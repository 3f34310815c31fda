#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_<tag>/*/p_counter_collection.csv: per kernel, per counter, the mean
value per dispatch.  FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide
coalesced reads; kilobyte unit), WRITE_SIZE taken as is (kilobytes)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "pmc_" + tag)
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
    per_dispatch = defaultdict(float)
    names, grids = {}, {}
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "brbd::" not in k:
            continue
        key = (row["Dispatch_Id"], row["Counter_Name"])
        per_dispatch[key] += float(row["Counter_Value"])
        names[row["Dispatch_Id"]] = k.split("(")[0].replace("void ", "")
        grids[row["Dispatch_Id"]] = int(row["Grid_Size"])
    # only the full-batch dispatches of each kernel (the host-memspace pipeline launches the same
    # kernels on column chunks: smaller grids, not comparable per launch)
    full = defaultdict(int)
    for d, k in names.items():
        full[k] = max(full[k], grids[d])
    for (d, c), v in per_dispatch.items():
        if grids[d] == full[names[d]]:
            acc[names[d]][c].append(v)
out = {}
for k in sorted(acc):
    out[k] = {c: sum(v) / len(v) for c, v in sorted(acc[k].items())}
    out[k]["dispatches"] = max(len(v) for v in acc[k].values())
    if "FETCH_SIZE" in out[k]:
        out[k]["hbm_read_bytes_corrected"] = out[k]["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in out[k]:
        out[k]["hbm_write_bytes"] = out[k]["WRITE_SIZE"] * 1024
print(json.dumps(out, indent=1))
