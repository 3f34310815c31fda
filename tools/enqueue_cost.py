#!/usr/bin/env python3
"""Host time to ENQUEUE one RNEA + ABA step through the Python wrappers against the device time of
the step: the margin by which one host thread stays ahead of its GPU (MI355X: 20 us against 53)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
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
for n in (50, 200, 200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("steps %d: enqueue %.1f us/step, until done %.1f us/step" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
