#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a PMC summary (tools/pmc_summary.py) of bench.py's default workload,
so that the static `traffic` / executed-flop figures bench.py quotes always come from the same script
run as the round's kernel stats (tools/gpu_profile.sh calls this).
Usage: tools/pmc_traffic.py gpurun_out/<tag>_pmc_summary.json [out.json]"""
import json
import os
import sys

src = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
d = json.load(open(src))
NW = 1024  # waves of a 65536-instance launch


def pick(sub):
    ks = [k for k in d if sub in k]
    return ks[0] if ks else None


def flops(x):
    return (2 * x.get("SQ_INSTS_VALU_FMA_F64", 0) + x.get("SQ_INSTS_VALU_ADD_F64", 0) + x.get("SQ_INSTS_VALU_MUL_F64", 0)) / NW


res = {"source": "rocprofv3 --pmc passes of `bench.py --steps 4 --warmup 2` (tools/pmc_passes.sh), summary %s; FETCH_SIZE doubled per "
                 "MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads), both in KiB; per launch = mean over the "
                 "full-batch dispatches of the kernel; executed flops = (2 FMA + ADD + MUL) fp64 VALU instructions per wave"
                 % os.path.basename(src),
       "algorithmic_bytes_per_launch": 73924608, "executed_flops_per_eval": {}}
for algo in ("rnea", "aba"):
    k = pick("k_%s<1, 1>" % algo)
    if not k:
        continue
    x = d[k]
    res[algo + "_kernel"] = k
    if "hbm_read_bytes_corrected" in x and "hbm_write_bytes" in x:
        res[algo + "_kernel_read_bytes"] = x["hbm_read_bytes_corrected"]
        res[algo + "_kernel_write_bytes"] = x["hbm_write_bytes"]
        res[algo + "_kernel_bytes_per_launch"] = x["hbm_read_bytes_corrected"] + x["hbm_write_bytes"]
    if "SQ_INSTS_VALU_FMA_F64" in x:
        res["executed_flops_per_eval"][algo] = round(flops(x))
    if "GRBM_GUI_ACTIVE" in x:
        res[algo + "_kernel_gui_active_cycles_all_xcds"] = x["GRBM_GUI_ACTIVE"]
res["instance_major_staged"] = {}
for algo in ("rnea", "aba"):
    k = pick("k_%s<2, 1>" % algo)
    if k and "hbm_read_bytes_corrected" in d[k]:
        res["instance_major_staged"][k] = {"read_bytes": d[k]["hbm_read_bytes_corrected"], "write_bytes": d[k].get("hbm_write_bytes")}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
