#!/usr/bin/env python3
"""Step time of the headline (RNEA + ABA, humanoid 65536, BATCH_MAJOR) with and without the library's
timing events on the stream: what the events between the launches cost, for every launch timed
(period 1) and for one launch in 8 (brbd_set_timing_period)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pinocchio_amd import capi, parallel, sample_models as S
m = S.simple_humanoid(); B = 65536; pool = parallel.ModelPool(m, 1); dm = pool.device_model(0)
rng = np.random.default_rng(0)
q = torch.from_numpy(np.ascontiguousarray(m.random_configuration(rng, B))).cuda()
v = torch.from_numpy(rng.uniform(-1, 1, (m.nv, B))).cuda(); a = torch.from_numpy(rng.uniform(-1, 1, (m.nv, B))).cuda()
tau = torch.empty_like(a); ddq = torch.empty_like(a)
def step():
    parallel.rneaInParallel(1, pool, q, v, a, tau); parallel.abaInParallel(1, pool, q, v, tau, ddq)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5: step()
torch.cuda.synchronize()
res = {}
for mode, period in ((0, 1), (2, 1), (3, 1), (2, 8), (3, 8)) * 2:
    dm.kernel_time(capi.ALGO_RNEA); dm.kernel_time(capi.ALGO_ABA)
    capi.check(capi.lib().brbd_set_timing(dm.handle, mode))
    capi.check(capi.lib().brbd_set_timing_period(dm.handle, period))
    for _ in range(300): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000): step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 2000
    n_r, ms_r = dm.kernel_time(capi.ALGO_RNEA); n_a, ms_a = dm.kernel_time(capi.ALGO_ABA)
    res.setdefault("timing_%d_period_%d_us" % (mode, period), []).append((round(t * 1e6, 2), n_r, round(ms_r / max(n_r, 1) * 1e3, 2), round(ms_a / max(n_a, 1) * 1e3, 2)))
    capi.check(capi.lib().brbd_set_timing(dm.handle, 0))
    dm.kernel_time(capi.ALGO_RNEA); dm.kernel_time(capi.ALGO_ABA)
print(json.dumps(res))
