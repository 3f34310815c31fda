#!/usr/bin/env python3
"""Algorithmic flops per eval, counted once with a counting scalar type over the reference-structured
CPU restatement (tools/flops/count_flops.cpp includes oracle/brbd_oracle.c with `double` replaced),
as SURVEY 8d prescribes.  Writes profiles/counted_flops.json; BASELINE.md section 3 and bench.py
(`roofline.algorithmic_flops_per_eval`) carry the same frozen numbers.

    python tools/flops/count_flops.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from pinocchio_amd import sample_models as S  # noqa: E402


def dump(model, path):
    t = model.tables()
    rng = np.random.default_rng(0)
    q = model.random_configuration(rng, 1)[:, 0]
    v, a = rng.uniform(-1, 1, model.nv), rng.uniform(-1, 1, model.nv)
    with open(path, "w") as f:
        f.write(f"{t['njoints']} {t['nq']} {t['nv']}\n")
        for k in ("parents", "jtype", "idx_q", "idx_v"):
            f.write(" ".join(str(int(x)) for x in t[k]) + "\n")
        for k in ("placement", "inertia", "axis", "armature", "gravity"):
            f.write(" ".join(repr(float(x)) for x in np.asarray(t[k]).ravel()) + "\n")
        for x in (q, v, a):
            f.write(" ".join(repr(float(y)) for y in x) + "\n")


def main():
    exe = os.path.join(tempfile.gettempdir(), "brbd_count_flops")
    subprocess.check_call(["g++", "-O1", "-fpermissive", "-w", "-o", exe, os.path.join(HERE, "count_flops.cpp")])
    out = {"method": "counting scalar over oracle/brbd_oracle.c (tools/flops/count_flops.cpp): add + sub + mul + div, "
                     "an fma would count 2; sincos / sqrt calls listed separately; one random instance "
                     "(the count does not depend on the values)",
           "models": {}}
    for name in ("simple_humanoid", "talos_like", "quadruped", "arm7", "manipulator", "baxter_simple"):
        m = S.build(name)
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as tf:
            path = tf.name
        dump(m, path)
        res = json.loads(subprocess.check_output([exe, path]).decode())
        os.unlink(path)
        out["models"][name] = {"nq": m.nq, "nv": m.nv, "njoints": m.njoints, **res}
        print(name, {k: v["flops"] for k, v in res.items()})
    with open(os.path.join(ROOT, "profiles", "counted_flops.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
