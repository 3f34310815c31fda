"""The model-compiled kernels address memory through range-checked buffer resources: the hardware
DROPS an out-of-range access, which would hide an indexing or park-offset bug (the GPU fault on
record from round 1, `k_aba<0,1>`, was removed "by construction" that way).  The checking build
(make -C pinocchio_amd/csrc check -> libbrbd_hip_check.so, -DBRBD_SPEC_CHECK) asserts every offset
against its range inside the kernels and counts violations.  This test runs every compiled kernel
of every registered model under it -- both layouts, ragged and full batches, chunked launches,
external forces, both derivative contracts -- and requires zero violations and oracle parity."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK_LIB = os.path.join(ROOT, "pinocchio_amd", "libbrbd_hip_check.so")

SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import torch
import oracle
from pinocchio_amd import capi, parallel, sample_models as S
from util import inputs, rel_err_cols
L = capi.lib()
assert capi.LIB_PATH.endswith("libbrbd_hip_check.so")
total = 0
for name in ("simple_humanoid", "talos_like", "quadruped", "arm7", "manipulator"):
    model = S.build(name)
    pool = parallel.ModelPool(model, 1)
    dm = pool.device_model(0)
    assert dm.compiled_kernels() == name + "_p"
    assert L.brbd_debug_oob_count(dm.handle) == 0
    O = oracle.Oracle(model)
    for B in (1, 130, 4096 + 7):
        q, v, a = inputs(model, B, B)
        f = np.random.default_rng(B).uniform(-1, 1, (6 * model.njoints, B))
        tau_ref = O.rnea(q, v, a, threads=0)
        for order in "FC":
            qq, vv, aa, ff = (np.asarray(x, order=order) for x in (q, v, a, f))
            for dev in (False, True):
                if dev:
                    put = (lambda x: torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")) if order == "C" else \
                          (lambda x: torch.from_numpy(np.ascontiguousarray(x.T)).to("cuda:0").t())
                    qq, vv, aa, ff = put(q), put(v), put(a), put(f)
                get = (lambda t: t.cpu().numpy()) if dev else (lambda t: t)
                tau = parallel.rneaInParallel(1, pool, qq, vv, aa)
                assert rel_err_cols(get(tau), tau_ref).max() < 1e-12
                ddq = parallel.abaInParallel(1, pool, qq, vv, tau)
                assert rel_err_cols(get(ddq), a).max() < 1e-10
                tf = parallel.rnea(pool, qq, vv, aa, ff)
                assert rel_err_cols(get(parallel.aba(pool, qq, vv, tf, ff)), a).max() < 1e-10
                Jr, Mr = O.fk_jacobians(q)
                J2, M2 = parallel.computeJointJacobians(pool, qq, want_oMi=True)
                assert np.abs(get(J2) - Jr).max() < 1e-12 and np.abs(get(M2) - Mr).max() < 1e-12
                assert np.abs(get(parallel.computeJointJacobians(pool, qq)) - Jr).max() < 1e-12   # (layout 2 when staged)
                parallel.forwardKinematics(pool, qq)
                if B <= 130 or name in ("quadruped", "arm7", "manipulator"):
                    parallel.computeRNEADerivatives(pool, qq, vv, aa, want_tau=True)
                    parallel.computeRNEADerivatives(pool, qq, vv, aa)
                if dev and model.is_depth_first():   # compiled computeMinverse (BATCH_MAJOR) + the symmetric variant
                    Mi = get(parallel.computeMinverse(pool, qq))
                    assert np.abs(Mi - O.minverse(q)).max() / max(1.0, np.abs(Mi).max()) < 1e-10
                    if B <= 130:
                        parallel.computeABADerivatives(pool, qq, vv, tau)
    n = L.brbd_debug_oob_count(dm.handle)
    print(name, "out-of-range accesses:", n, flush=True)
    total += n
    pool.release()
# the headline batch and a chunked launch of a kernel with a global park area
for name, B in (("simple_humanoid", 65536), ("talos_like", 300000)):
    model = S.build(name)
    pool = parallel.ModelPool(model, 1)
    q, v, a = inputs(model, B, 5)
    qd, vd, ad = (torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0") for x in (q, v, a))
    tau = parallel.rneaInParallel(1, pool, qd, vd, ad)
    ddq = parallel.abaInParallel(1, pool, qd, vd, tau)
    assert ((ddq - ad).norm(dim=0) / ad.norm(dim=0)).max().item() < 1e-10
    qi, vi, ai = (x.t().contiguous().t() for x in (qd, vd, ad))
    ddqi = parallel.abaInParallel(1, pool, qi, vi, parallel.rneaInParallel(1, pool, qi, vi, ai))
    assert ((ddqi - ad).norm(dim=0) / ad.norm(dim=0)).max().item() < 1e-10
    n = L.brbd_debug_oob_count(pool.device_model(0).handle)
    print(name, B, "out-of-range accesses:", n, flush=True)
    total += n
    pool.release()
assert total == 0, total
# positive control: the checks must SEE a violation.  BRBD_CHECK_SHRINK=k (checking build only)
# declares matrix k 8 bytes short of what the kernel addresses (k = 8: the park area, 16 bytes), so
# its last element is out of range: dropped by the hardware as always, counted by the checks.
import os
model = S.build("talos_like")
q, v, a = inputs(model, 128, 3)
f = np.random.default_rng(1).uniform(-1, 1, (6 * model.njoints, 128))
seen = {}
for k, what in ((0, "q"), (3, "out"), (7, "fext"), (8, "parks")):
    for order in "CF":
        pool = parallel.ModelPool(model, 1)
        h = pool.device_model(0).handle
        qq, vv, aa, ff = (np.asarray(x, order=order) for x in (q, v, a, f))
        os.environ["BRBD_CHECK_SHRINK"] = str(k)
        if k == 8:
            parallel.computeRNEADerivatives(pool, qq, vv, aa)   # (a kernel that parks in the global area)
        else:
            parallel.aba(pool, qq, vv, aa, ff)
        del os.environ["BRBD_CHECK_SHRINK"]
        seen[what, order] = L.brbd_debug_oob_count(h)
        pool.release()
print("positive control:", seen, flush=True)
assert all(n > 0 for n in seen.values()), seen
print("BOUNDS-OK")
"""


def test_compiled_kernels_make_no_out_of_range_access():
    assert os.path.exists(CHECK_LIB), "build it with `make -C pinocchio_amd/csrc check` (__graft_entry__.build() does)"
    env = dict(os.environ, BRBD_LIB=CHECK_LIB)
    r = subprocess.run([sys.executable, "-u", "-c", SCRIPT % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BOUNDS-OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_regular_build_reports_no_counter():
    from pinocchio_amd import capi, sample_models as S
    dm = capi.DeviceModel(S.manipulator(), 0)
    assert capi.lib().brbd_debug_oob_count(dm.handle) == -1
    dm.close()
