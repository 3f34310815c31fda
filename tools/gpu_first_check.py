import sys, time
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import numpy as np
from pinocchio_amd import capi, parallel, sample_models as S
import oracle
sys.path.insert(0,os.path.join(ROOT,'tests'))
from util import rel_err_cols, inputs, parity_models as test_models
print("devices", capi.lib().brbd_device_count(), flush=True)
for m in test_models():
    pool = parallel.ModelPool(m, 1)
    O = oracle.Oracle(m)
    B = 200
    q, v, a = inputs(m, B, 1)
    for order in ("F", "C"):
        qq, vv, aa = (np.asarray(x, order=order) for x in (q, v, a))
        tau = parallel.rneaInParallel(1, pool, qq, vv, aa)
        tref = O.rnea(q, v, a)
        e1 = rel_err_cols(tau, tref).max()
        ddq = parallel.abaInParallel(1, pool, qq, vv, np.asarray(tref, order=order))
        dref = O.aba(q, v, tref)
        e2 = rel_err_cols(ddq, dref).max()
        e2b = rel_err_cols(ddq, a).max()
        J, oMi = parallel.computeJointJacobians(pool, qq, want_oMi=True)
        Jr, Mr = O.fk_jacobians(q)
        e3 = np.abs(J - Jr).max(); e4 = np.abs(oMi - Mr).max()
        if m.is_depth_first():
            dq, dv, da = parallel.computeRNEADerivatives(pool, qq, vv, aa)
            rq, rv, ra, _ = O.rnea_derivatives(q, v, a)
            e5 = max(np.abs(dq-rq).max()/max(1,np.abs(rq).max()), np.abs(dv-rv).max()/max(1,np.abs(rv).max()), np.abs(da-ra).max()/max(1,np.abs(ra).max()))
        else:
            e5 = -1
        print(f"{m.name:28s} {order} rnea {e1:.2e} aba {e2:.2e} (vs a {e2b:.2e}) J {e3:.2e} oMi {e4:.2e} deriv {e5:.2e}", flush=True)
    pool.release()
