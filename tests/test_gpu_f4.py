"""SURVEY 8f rank 4 on the GPU, through the C ABI, against the oracle restatements
(tests/test_oracle_f4.py pins those): batched integrate (joint-configuration.hxx), computeMinverse
(aba.hxx:841-877) and computeABADerivatives (aba-derivatives.hxx:381-454), both boundary layouts,
host and device memory, ragged batches, every supported joint kind."""
import numpy as np
import pytest

import oracle
from pinocchio_amd import capi, model as M, parallel, sample_models as S
from util import inputs

pytestmark = pytest.mark.gpu
MODELS = [S.simple_humanoid(), S.quadruped(), S.arm7(), S.talos_like(),
          S.random_tree(301, 9, root_kind=M.JOINT_FREEFLYER, armature=True),
          S.random_tree(302, 10, kinds=[M.JOINT_RX, M.JOINT_PY, M.JOINT_SPHERICAL, M.JOINT_RUBZ, M.JOINT_REVOLUTE_UNALIGNED,
                                        M.JOINT_PRISMATIC_UNALIGNED, M.JOINT_REVOLUTE_UNBOUNDED_UNALIGNED])]
IDS = [m.name for m in MODELS]


# ddq_dq / ddq_dv = -Minv * dtau_dq|dv: north_star's 1e-10.  Measured on MI355X over 2048 instances of
# each model above (tools/f4_errors.py, profiles/r03_f4_errors.json): worst 9e-13 in this metric
# (1.4e-12 per instance in the Frobenius norm), on the talos-class model whose cond(M) reaches 1e5;
# every other model stays below 1.1e-13.  Round 2 tested these two at 1e-9 without a measurement.
TOL_DDQ = 1e-10


def _rel(x, y):
    return np.abs(x - y).max() / max(1.0, np.abs(y).max())


@pytest.mark.parametrize("order", ["F", "C"])
@pytest.mark.parametrize("model", MODELS, ids=IDS)
def test_f4_host_memspace_vs_oracle(model, order):
    pool = parallel.ModelPool(model, 1)
    O = oracle.Oracle(model)
    B = 130
    q, v, tau = inputs(model, B, 21)
    qq, vv, tt = (np.asarray(x, order=order) for x in (q, v, tau))
    assert np.abs(parallel.integrate(pool, qq, vv) - O.integrate(q, v)).max() < 1e-13
    small = 1e-6 * v  # Taylor branches of exp3 / exp6
    assert np.abs(parallel.integrate(pool, qq, np.asarray(small, order=order)) - O.integrate(q, small)).max() < 1e-13
    assert _rel(parallel.computeMinverse(pool, qq), O.minverse(q)) < 1e-10
    dq, dv, Minv, ddq = parallel.computeABADerivatives(pool, qq, vv, tt, want_ddq=True)
    rq, rv, rM, rdd = O.aba_derivatives(q, v, tau)
    assert _rel(ddq, rdd) < 1e-10 and _rel(Minv, rM) < 1e-10
    assert _rel(dq, rq) < TOL_DDQ and _rel(dv, rv) < TOL_DDQ
    pool.release()


@pytest.mark.parametrize("layout", ["batch_major", "instance_major"])
def test_f4_device_memspace_rollout_step(layout):
    """One step of a sampling-based MPC rollout entirely on the device: ddq = aba, semi-implicit Euler
    with integrate, sensitivities from computeABADerivatives -- against the same steps on the oracle."""
    import torch
    model = S.quadruped()
    pool = parallel.ModelPool(model, 1)
    O = oracle.Oracle(model)
    B, dt = 4096 + 5, 1e-2
    q, v, tau = inputs(model, B, 33)

    def put(x):
        t = torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
        return t if layout == "batch_major" else torch.from_numpy(np.ascontiguousarray(x.T)).to("cuda:0").t()
    qd, vd, td = put(q), put(v), put(tau)
    dq, dv, Minv = parallel.computeABADerivatives(pool, qd, vd, td)   # (ddq not requested: internal temporary)
    ddq = parallel.abaInParallel(1, pool, qd, vd, td)
    v1 = vd + dt * ddq
    q1 = parallel.integrate(pool, qd, dt * v1)
    torch.cuda.synchronize()
    idx = np.random.default_rng(0).choice(B, 64, replace=False)
    rq, rv, rM, rdd = O.aba_derivatives(q[:, idx], v[:, idx], tau[:, idx])
    assert _rel(dq[:, idx].cpu().numpy(), rq) < TOL_DDQ and _rel(dv[:, idx].cpu().numpy(), rv) < TOL_DDQ
    assert _rel(Minv[:, idx].cpu().numpy(), rM) < 1e-10
    v1r = v[:, idx] + dt * rdd
    assert np.abs(q1[:, idx].cpu().numpy() - O.integrate(q[:, idx], dt * v1r)).max() < 1e-12
    # Minv * M == 1 for every instance (M from dtau_da of computeRNEADerivatives, symmetrised)
    nv = model.nv
    da = parallel.computeRNEADerivatives(pool, qd, vd, td)[2]
    Mu = torch.triu(da.reshape(nv, nv, B).permute(2, 1, 0))
    Mf = Mu + torch.triu(Mu, 1).transpose(1, 2)
    Mi = Minv.reshape(nv, nv, B).permute(2, 1, 0)
    eye = torch.eye(nv, dtype=torch.float64, device=Mi.device)
    assert (torch.bmm(Mi, Mf) - eye).abs().max().item() < 1e-9
    pool.release()


def test_f4_argument_checks():
    m = M.Model("not_dfs")
    j1 = m.addJoint(0, M.JOINT_RX, np.eye(4), "a")
    m.addJoint(0, M.JOINT_RY, np.eye(4), "b")
    m.addJoint(j1, M.JOINT_RZ, np.eye(4), "c")
    for j in (1, 2, 3):
        m.appendBodyToJoint(j, M.Inertia(1.0, [0.1, 0, 0], np.eye(3)))
    pool = parallel.ModelPool(m, 1)
    z = np.zeros((3, 4))
    with pytest.raises(capi.BrbdUnsupported):
        parallel.computeMinverse(pool, z)
    with pytest.raises(capi.BrbdUnsupported):
        parallel.computeABADerivatives(pool, z, z, z)
    assert parallel.integrate(pool, z, z).shape == (3, 4)      # integrate needs no ordering
    with pytest.raises(capi.BrbdInvalidArgument):
        parallel.integrate(pool, z, np.zeros((2, 4)))
    pool.release()


def test_f4_more_than_48_dofs():
    """computeABADerivatives beyond the nv <= 48 of round 2 (now <= 96): a 70-dof random tree."""
    model = S.random_tree(777, 52, kinds=[M.JOINT_RX, M.JOINT_RY, M.JOINT_RZ, M.JOINT_SPHERICAL, M.JOINT_PX])
    assert 48 < model.nv <= 96
    pool = parallel.ModelPool(model, 1)
    O = oracle.Oracle(model)
    q, v, tau = inputs(model, 70, 5)
    for order in "FC":
        qq, vv, tt = (np.asarray(x, order=order) for x in (q, v, tau))
        dq, dv, Minv, ddq = parallel.computeABADerivatives(pool, qq, vv, tt, want_ddq=True)
        rq, rv, rM, rdd = O.aba_derivatives(q, v, tau)
        assert _rel(ddq, rdd) < 1e-10 and _rel(Minv, rM) < 1e-10
        assert _rel(dq, rq) < TOL_DDQ and _rel(dv, rv) < TOL_DDQ
    pool.release()
