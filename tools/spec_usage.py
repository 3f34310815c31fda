#!/usr/bin/env python3
"""Table of the register / scratch usage of every model-compiled kernel
(pinocchio_amd/csrc/gen/*.usage, written by the Makefile; pinocchio_amd/codegen/usage.py)."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pinocchio_amd.codegen import usage
bad = 0
for r in usage.kernel_reports(os.path.join(ROOT, "pinocchio_amd", "csrc", "gen")):
    short = re.sub(r"_ZN4brbd\d+spec_", "", r["name"]).replace("EvNS_5KArgsE", "")
    flag = ""
    if usage.refused(r):
        flag = "  <-- scratch accesses + SGPR spills"
        bad += 1
    print("%-44s VGPR %3s AGPR %3s scratch %4s (%s insts) sgpr_spill %3s vgpr_spill %3s occ %s%s" % (
        short, r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("ScratchInsts"), r.get("SGPRs Spill"),
        r.get("VGPRs Spill"), r.get("Occupancy"), flag))
sys.exit(1 if bad else 0)
