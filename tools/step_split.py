"""Experiment: one step (RNEA then ABA over 65536 instances) as S column slices on S streams, each
slice with a handle (workspace) of its own: does overlapping one slice's ramp / tail with another
slice's body pay?"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from pinocchio_amd import parallel, capi, sample_models as S
m = S.simple_humanoid(); B = 65536
rng = np.random.default_rng(0)
q = torch.from_numpy(np.ascontiguousarray(m.random_configuration(rng, B))).cuda()
v = torch.from_numpy(rng.uniform(-1, 1, (m.nv, B))).cuda()
a = torch.from_numpy(rng.uniform(-1, 1, (m.nv, B))).cuda()
tau = torch.empty_like(a); ddq = torch.empty_like(a)
for nsl in (1, 2, 4):
    pools = [parallel.ModelPool(m, 1) for _ in range(nsl)]
    streams = [torch.cuda.Stream() for _ in range(nsl)]
    w = B // nsl
    sl = [slice(k * w, (k + 1) * w) for k in range(nsl)]
    views = [(q[:, s], v[:, s], a[:, s], tau[:, s], ddq[:, s]) for s in sl]
    def step():
        for k in range(nsl):
            with torch.cuda.stream(streams[k]):
                qq, vv, aa, tt, dd = views[k]
                parallel.rneaInParallel(1, pools[k], qq, vv, aa, tt)
                parallel.abaInParallel(1, pools[k], qq, vv, tt, dd)
    for _ in range(20): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    err = float(((ddq - a).norm(dim=0) / a.norm(dim=0)).max())
    print("slices", nsl, "us/step %.1f" % ((t2 - t0) / 300 * 1e6), "host issue us/step %.1f" % ((t1 - t0) / 300 * 1e6), "err %.1e" % err)
    for p in pools: p.release()
