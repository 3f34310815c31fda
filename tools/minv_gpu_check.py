import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import oracle
from pinocchio_amd import capi, parallel, sample_models as S
from util import inputs
m = S.quadruped()
pool = parallel.ModelPool(m, 1)
for B in (1, 6, 64, 70, 128, 130, 200, 1000, 4096):
    q, v, a = inputs(m, B, 3)
    ref = oracle.Oracle(m).minverse(q)
    qd = torch.from_numpy(np.ascontiguousarray(q)).cuda()
    out = parallel.computeMinverse(pool, qd).cpu().numpy()
    e = np.abs(out - ref)
    badcols = np.nonzero(e.max(axis=0) > 1e-9)[0]
    print(B, "max err %.2e" % e.max(), "bad columns", len(badcols), badcols[:10], badcols[-5:], "oob", capi.lib().brbd_debug_oob_count(pool.device_model(0).handle))
pool.release()
