#!/usr/bin/env python3
"""One line per brbd kernel from gpurun_out/pmc_<tag>_summary.json."""
import json, sys
d = json.load(open(sys.argv[1]))
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
for k, x in d.items():
    if "SQ_WAVE_CYCLES" not in x:
        continue
    w = x["SQ_WAVE_CYCLES"]
    f64 = x.get("SQ_INSTS_VALU_FMA_F64", 0) + x.get("SQ_INSTS_VALU_ADD_F64", 0) + x.get("SQ_INSTS_VALU_MUL_F64", 0)
    fl = 2 * x.get("SQ_INSTS_VALU_FMA_F64", 0) + x.get("SQ_INSTS_VALU_ADD_F64", 0) + x.get("SQ_INSTS_VALU_MUL_F64", 0)
    print(k, "cyc/wave %.0f" % (w / nw * 4), "wait %.2f" % (x["SQ_WAIT_ANY"] / w), "wait_inst %.2f" % (x["SQ_WAIT_INST_ANY"] / w),
          "active %.2f" % (x["SQ_ACTIVE_INST_ANY"] / w), "valu/wave %.0f" % (x["SQ_INSTS_VALU"] / nw),
          "salu/wave %.0f" % (x["SQ_INSTS_SALU"] / nw), "f64/wave %.0f" % (f64 / nw), "flops/eval %.0f" % (fl / nw),
          "vmem rd/wr %.0f %.0f" % (x.get("SQ_INSTS_VMEM_RD", 0) / nw, x.get("SQ_INSTS_VMEM_WR", 0) / nw),
          "smem %.0f" % (x.get("SQ_INSTS_SMEM", 0) / nw),
          "HBM rd MB %.0f wr MB %.0f" % (x.get("hbm_read_bytes_corrected", 0) / 1e6, x.get("hbm_write_bytes", 0) / 1e6))
