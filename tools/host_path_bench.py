#!/usr/bin/env python3
"""Host-memory end-to-end rate of the humanoid RNEA + ABA step (pageable Eigen-style matrices, the
reference's calling convention): one JSON line.  Env: BRBD_HOST_THREADS, BRBD_HOST_CHUNK."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pinocchio_amd import parallel, sample_models as S

B = int(os.environ.get("HOST_B", "65536"))
m = S.simple_humanoid(); pool = parallel.ModelPool(m, 1)
rng = np.random.default_rng(0)
q = np.asfortranarray(m.random_configuration(rng, B)); v = np.asfortranarray(rng.uniform(-1, 1, (m.nv, B)))
a = np.asfortranarray(rng.uniform(-1, 1, (m.nv, B)))
tau = np.empty((m.nv, B), order="F"); ddq = np.empty((m.nv, B), order="F")
def step():
    parallel.rneaInParallel(1, pool, q, v, a, tau); parallel.abaInParallel(1, pool, q, v, tau, ddq)
for _ in range(3): step()
t0 = time.perf_counter(); n = 0
while n < 10 or time.perf_counter() - t0 < 1.0:
    step(); n += 1
t = (time.perf_counter() - t0) / n
print(json.dumps({"B": B, "threads": os.environ.get("BRBD_HOST_THREADS", "8"), "chunk": os.environ.get("BRBD_HOST_CHUNK", "default"),
                  "evals_per_s": 2 * B / t, "GBps_host_bytes": 2 * B * (m.nq + 3 * m.nv) * 8 / t / 1e9,
                  "err": float(np.abs(ddq - a).max())}))
