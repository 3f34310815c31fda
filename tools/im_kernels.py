#!/usr/bin/env python3
"""Kernel times of RNEA / ABA in both boundary layouts (device resident), per launch, with the
library's sampled dispatch timing.  Usage: [BRBD_LIB=...] python tools/im_kernels.py [model] [batch]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pinocchio_amd import capi, parallel, sample_models as S
from util import inputs
name = sys.argv[1] if len(sys.argv) > 1 else "simple_humanoid"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
m = S.build(name); pool = parallel.ModelPool(m, 1); dm = pool.device_model(0)
q, v, a = inputs(m, B)
out = {"model": name, "batch": B, "lib": os.environ.get("BRBD_LIB", "default")}
for lay in ("batch_major", "instance_major"):
    def put(x):
        t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        return t if lay == "batch_major" else torch.from_numpy(np.ascontiguousarray(x.T)).cuda().t()
    qd, vd, ad = put(q), put(v), put(a)
    tau = put(np.zeros((m.nv, B))); ddq = put(np.zeros((m.nv, B)))
    def step():
        parallel.rneaInParallel(1, pool, qd, vd, ad, tau); parallel.abaInParallel(1, pool, qd, vd, tau, ddq)
    for _ in range(300): step()
    torch.cuda.synchronize()
    dm.kernel_time(capi.ALGO_RNEA); dm.kernel_time(capi.ALGO_ABA)
    dm.set_timing(True, dispatch=True, period=8)
    for _ in range(800): step()
    torch.cuda.synchronize(); dm.set_timing(False)
    nr, tr = dm.kernel_time(capi.ALGO_RNEA); na, ta = dm.kernel_time(capi.ALGO_ABA)
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(1000): step()
    torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 1000
    out[lay] = {"rnea_us": tr / nr * 1e3, "aba_us": ta / na * 1e3, "step_us": w * 1e6, "evals_per_s": 2 * B / w}
print(json.dumps(out))
