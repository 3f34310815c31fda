"""Experiment: how fast would limb-parallel execution be?  Emulate it with existing kernels on the
sub-chains of simple_humanoid at 4x the batch (4 waves/SIMD worth of tiles)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from pinocchio_amd import capi, parallel, sample_models as S, model as M

def submodel(m, keep):
    out = M.Model("sub")
    remap = {0: 0}
    for i in range(1, m.njoints):
        if i in keep:
            j = out.addJoint(remap[m.parents[i]], m.jtypes[i], m.jointPlacements[i], m.names[i])
            out.inertias[j] = m.inertias[i]
            remap[i] = j
    return out

m = S.simple_humanoid()
def path(i):
    p = []
    while i > 0:
        p.append(i); i = m.parents[i]
    return set(p)
subs = {"full": set(range(1, m.njoints)), "root+waist+larm (11)": path(23), "root+lleg (7)": path(7), "waist+arms+root(18)": path(23) | path(30)}
dev = torch.device("cuda:0")
for name, keep in subs.items():
    sm = submodel(m, keep)
    pool = parallel.ModelPool(sm, 1)
    for B in (65536, 65536 * 4):
        rng = np.random.default_rng(0)
        q = torch.from_numpy(np.ascontiguousarray(sm.random_configuration(rng, B))).to(dev)
        v = torch.from_numpy(rng.uniform(-1, 1, (sm.nv, B))).to(dev)
        a = torch.from_numpy(rng.uniform(-1, 1, (sm.nv, B))).to(dev)
        tau = torch.empty_like(v); ddq = torch.empty_like(v)
        dm = pool.device_model(0)
        for _ in range(5):
            parallel.rneaInParallel(1, pool, q, v, a, tau); parallel.abaInParallel(1, pool, q, v, tau, ddq)
        torch.cuda.synchronize()
        dm.set_timing(True)
        for _ in range(20):
            parallel.rneaInParallel(1, pool, q, v, a, tau); parallel.abaInParallel(1, pool, q, v, tau, ddq)
        torch.cuda.synchronize()
        n1, t1 = dm.kernel_time(capi.ALGO_RNEA); n2, t2 = dm.kernel_time(capi.ALGO_ABA)
        dm.set_timing(False)
        print(f"{name:24s} njoints {sm.njoints-1:2d} B {B:7d} static={dm.static_topology()!r} rnea {t1/n1*1e3:8.1f} us  aba {t2/n2*1e3:8.1f} us", flush=True)
    pool.release()
