#!/usr/bin/env python3
"""Table of the register / scratch usage of every model-compiled kernel
(pinocchio_amd/csrc/gen/*.usage, written by the Makefile)."""
import glob, os, re, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pinocchio_amd", "csrc", "gen")
rows = []
for f in sorted(glob.glob(os.path.join(root, "*.usage"))):
    cur = None
    for line in open(f):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
bad = 0
for r in rows:
    if "k_" not in r["name"]:
        continue
    short = re.sub(r"_ZN4brbd\d+spec_", "", r["name"]).replace("EvNS_5KArgsE", "")
    flag = ""
    if int(r.get("ScratchSize", 0)) > 0 and int(r.get("SGPRs Spill", 0)) > 8:
        flag = "  <-- scratch + SGPR spills"
        bad += 1
    print("%-44s VGPR %3s AGPR %3s scratch %4s sgpr_spill %3s vgpr_spill %3s occ %s%s" % (
        short, r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("SGPRs Spill"), r.get("VGPRs Spill"),
        r.get("Occupancy"), flag))
sys.exit(1 if bad else 0)
