import sys, numpy as np, torch
sys.path.insert(0, '.')
from pinocchio_amd import parallel, sample_models as S
m = S.build(sys.argv[1] if len(sys.argv) > 1 else 'simple_humanoid')
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
rng = np.random.default_rng(0)
q = torch.from_numpy(np.ascontiguousarray(m.random_configuration(rng, B))).cuda()
v = torch.from_numpy(rng.uniform(-1, 1, (m.nv, B))).cuda()
a = torch.from_numpy(rng.uniform(-1, 1, (m.nv, B))).cuda()
pool = parallel.ModelPool(m, 1)
what = sys.argv[3] if len(sys.argv) > 3 else "rnea_aba"
for _ in range(3):
    if what == "deriv":
        parallel.computeRNEADerivatives(pool, q, v, a)
    else:
        tau = parallel.rneaInParallel(1, pool, q, v, a)
        ddq = parallel.abaInParallel(1, pool, q, v, tau)
torch.cuda.synchronize()
