"""Fixed cost of a launch: kernel time (library hipEvents, chained and unchained) of short programs
(manipulator, 6 dof) at several batch sizes, next to the humanoid."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from pinocchio_amd import parallel, capi, sample_models as S
for name in ("manipulator", "arm7", "simple_humanoid"):
    m = S.build(name)
    for B in ((64, 4096, 65536) if name != "simple_humanoid" else (64, 4096, 32768, 65536, 131072, 262144, 1048576)):
        rng = np.random.default_rng(0)
        q = torch.from_numpy(np.ascontiguousarray(m.random_configuration(rng, B))).cuda()
        v = torch.from_numpy(rng.uniform(-1, 1, (m.nv, B))).cuda()
        a = torch.from_numpy(rng.uniform(-1, 1, (m.nv, B))).cuda()
        tau = torch.empty_like(a); ddq = torch.empty_like(a)
        pool = parallel.ModelPool(m, 1); dm = pool.device_model(0)
        def step():
            parallel.rneaInParallel(1, pool, q, v, a, tau)
            parallel.abaInParallel(1, pool, q, v, tau, ddq)
        for _ in range(300): step()
        torch.cuda.synchronize()
        out = []
        for mode in (1, 2):
            capi.check(capi.lib().brbd_set_timing(dm.handle, mode))
            for _ in range(300): step()
            torch.cuda.synchronize()
            capi.check(capi.lib().brbd_set_timing(dm.handle, 0))
            kr, ka = dm.kernel_time(capi.ALGO_RNEA), dm.kernel_time(capi.ALGO_ABA)
            out.append("mode %d rnea %.1f aba %.1f us" % (mode, kr[1] / kr[0] * 1e3, ka[1] / ka[0] * 1e3))
        print(name, B, " | ".join(out), flush=True)
        pool.release()
