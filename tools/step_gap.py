import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from pinocchio_amd import parallel, capi, sample_models as S
m = S.simple_humanoid(); B = 65536
rng = np.random.default_rng(0)
q = torch.from_numpy(np.ascontiguousarray(m.random_configuration(rng, B))).cuda()
v = torch.from_numpy(rng.uniform(-1, 1, (m.nv, B))).cuda()
a = torch.from_numpy(rng.uniform(-1, 1, (m.nv, B))).cuda()
tau = torch.empty_like(a); ddq = torch.empty_like(a)
pool = parallel.ModelPool(m, 1); dm = pool.device_model(0)
def step():
    parallel.rneaInParallel(1, pool, q, v, a, tau)
    parallel.abaInParallel(1, pool, q, v, tau, ddq)
for timing in (0, 1, 2, 0):
    for _ in range(20): step()
    torch.cuda.synchronize()
    capi.check(capi.lib().brbd_set_timing(dm.handle, timing))
    t0 = time.perf_counter()
    for _ in range(300): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    dm.set_timing(False)
    kr, ka = dm.kernel_time(capi.ALGO_RNEA), dm.kernel_time(capi.ALGO_ABA)
    print("timing", timing, "rnea/aba us %.1f %.1f" % (kr[1] / max(kr[0], 1) * 1e3, ka[1] / max(ka[0], 1) * 1e3), "us/step %.1f" % ((t2 - t0) / 300 * 1e6), "host issue us/step %.1f" % ((t1 - t0) / 300 * 1e6))
