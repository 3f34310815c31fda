"""Oracle restatements of SURVEY 8f rank 4 -- computeMinverse (aba.hxx:841-877), integrate
(joint-configuration.hxx; liegroup/special-euclidean.hpp:773-812, special-orthogonal.hpp) and
computeABADerivatives (aba-derivatives.hxx:381-454) -- pinned by the reference's own test
identities: Minv == M^-1 (unittest/aba.cpp test_computeMinverse), integrate against the matrix
exponential of the twist and the group properties (unittest/joint-configurations.cpp), ABA
derivatives against finite differences over integrate (unittest/aba-derivatives.cpp) and
ddq_dtau == Minv."""
import numpy as np
import pytest
import scipy.linalg

import oracle
from pinocchio_amd import model as M, sample_models as S
from util import inputs

MODELS = [S.simple_humanoid(), S.quadruped(), S.arm7(), S.manipulator(),
          S.random_tree(301, 9, root_kind=M.JOINT_FREEFLYER, armature=True),
          S.random_tree(302, 10, kinds=[M.JOINT_RX, M.JOINT_PY, M.JOINT_SPHERICAL, M.JOINT_RUBZ, M.JOINT_REVOLUTE_UNALIGNED,
                                        M.JOINT_PRISMATIC_UNALIGNED, M.JOINT_REVOLUTE_UNBOUNDED_UNALIGNED])]
IDS = [m.name for m in MODELS]


def mass_matrix(O, q):
    nv = O.nv
    Q = np.repeat(q.reshape(-1, 1), nv, axis=1)
    Z = np.zeros((nv, nv))
    saved = O.model.gravity.copy()
    return O.rnea(Q, Z, np.eye(nv)) - O.rnea(Q, Z, Z)


@pytest.mark.parametrize("model", MODELS, ids=IDS)
def test_minverse_is_the_inverse_of_the_mass_matrix(model):
    O = oracle.Oracle(model)
    q, _, _ = inputs(model, 3, 5)
    nv = model.nv
    Minv = O.minverse(q)
    for b in range(3):
        Mi = Minv[:, b].reshape(nv, nv).T  # column-major per instance
        Mm = mass_matrix(O, q[:, b])
        ref = np.linalg.inv(Mm)
        assert np.abs(np.triu(Mi) - np.triu(ref)).max() < 1e-10 * max(1.0, np.abs(ref).max())
        # strictly lower part: zero except inside the diagonal blocks of multi-dof joints
        low = np.tril(Mi, -1).copy()
        for i in range(1, model.njoints):
            iv, n = model.idx_vs[i], model.nvs[i]
            low[iv:iv + n, iv:iv + n] = 0
        assert np.abs(low).max() == 0.0


def _se3(R, t):
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return T


def _quat_R(x, y, z, w):
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


@pytest.mark.parametrize("scale", [1.0, 1e-3, 1e-9, 0.0])
def test_integrate_free_flyer_is_the_se3_exponential(scale):
    """q (+) v == q * exp(v^) on SE(3): checked against scipy's matrix exponential of the 4 x 4
    twist matrix, across the Taylor / closed-form switch of exp6 (explog-quaternion.hpp:119-126)."""
    model = S.quadruped()
    O = oracle.Oracle(model)
    rng = np.random.default_rng(3)
    q, _, _ = inputs(model, 6, 9)
    v = rng.uniform(-1, 1, (model.nv, 6)) * scale
    qo = O.integrate(q, v)
    for b in range(6):
        T0 = _se3(_quat_R(*q[3:7, b]), q[:3, b])
        xi = np.zeros((4, 4))
        w, lin = v[3:6, b], v[:3, b]
        xi[:3, :3] = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        xi[:3, 3] = lin
        T1 = T0 @ scipy.linalg.expm(xi)
        assert np.abs(_quat_R(*qo[3:7, b]) - T1[:3, :3]).max() < 1e-12
        assert np.abs(qo[:3, b] - T1[:3, 3]).max() < 1e-12
        assert abs(np.linalg.norm(qo[3:7, b]) - 1) < 1e-12
        assert np.array_equal(qo[7:, b], q[7:, b] + v[6:, b])  # revolute: q + v (vector space)


@pytest.mark.parametrize("model", MODELS, ids=IDS)
def test_integrate_group_properties(model):
    O = oracle.Oracle(model)
    q, v, _ = inputs(model, 5, 2)
    assert np.abs(O.integrate(q, 0 * v) - q).max() < 1e-15      # neutral element
    back = O.integrate(O.integrate(q, v), -v)                    # exp(v) exp(-v) = 1 (one-parameter subgroup)
    assert np.abs(back - q).max() < 1e-12
    half = O.integrate(O.integrate(q, 0.5 * v), 0.5 * v)
    assert np.abs(half - O.integrate(q, v)).max() < 1e-12


@pytest.mark.parametrize("model", MODELS, ids=IDS)
def test_aba_derivatives_against_finite_differences(model):
    """unittest/aba-derivatives.cpp: forward differences with alpha = 1e-8 over integrate(q, alpha e_k),
    tolerance sqrt(alpha); ddq_dtau == Minv == M^-1 (full, symmetric)."""
    O = oracle.Oracle(model)
    nv = model.nv
    q, v, tau = inputs(model, 1, 4)
    dq, dv, Minv, ddq = O.aba_derivatives(q, v, tau)
    dq, dv, Minv = (x[:, 0].reshape(nv, nv).T for x in (dq, dv, Minv))
    assert np.abs(ddq - O.aba(q, v, tau)).max() == 0.0
    alpha = 1e-8
    Qp = O.integrate(np.repeat(q, nv, axis=1), alpha * np.eye(nv))
    V, T = np.repeat(v, nv, axis=1), np.repeat(tau, nv, axis=1)
    fdq = (O.aba(Qp, V, T) - ddq) / alpha
    fdv = (O.aba(np.repeat(q, nv, axis=1), V + alpha * np.eye(nv), T) - ddq) / alpha
    assert np.linalg.norm(dq - fdq) < np.sqrt(alpha) * max(1.0, np.linalg.norm(fdq))
    assert np.linalg.norm(dv - fdv) < np.sqrt(alpha) * max(1.0, np.linalg.norm(fdv))
    Mm = mass_matrix(O, q[:, 0])
    assert np.abs(Minv @ Mm - np.eye(nv)).max() < 1e-9
    assert np.abs(Minv - Minv.T).max() == 0.0
