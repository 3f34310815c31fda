#!/usr/bin/env python3
"""Flatten the reference's URDF fixtures into committed model tables.

Reads (as data, at generation time only -- /root/reference does not exist on the GPU box):
  /root/reference/models/simple_humanoid.urdf  -> pinocchio_amd/models/simple_humanoid_ff.json
  /root/reference/models/baxter_simple.urdf    -> pinocchio_amd/models/baxter_simple.json
The outputs are derived tables (joint kinds, placements, merged inertias), not copies of
the URDF text.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pinocchio_amd.urdf import build_model_from_urdf  # noqa: E402

REF = os.environ.get("BRBD_REFERENCE", "/root/reference")


def main():
    out = os.path.join(ROOT, "pinocchio_amd", "models")
    os.makedirs(out, exist_ok=True)
    for src, dst, ff in (("simple_humanoid.urdf", "simple_humanoid_ff.json", True),
                         ("baxter_simple.urdf", "baxter_simple.json", False)):
        with open(os.path.join(REF, "models", src)) as f:
            m = build_model_from_urdf(f.read(), free_flyer=ff, name=dst[:-5])
        with open(os.path.join(out, dst), "w") as f:
            f.write(m.to_json())
        print(dst, "njoints", m.njoints, "nq", m.nq, "nv", m.nv, "depth-first", m.is_depth_first())


if __name__ == "__main__":
    main()
