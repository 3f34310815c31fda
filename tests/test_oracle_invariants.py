"""The reference's own test invariants for this path, re-expressed on the oracle.
Each test cites the reference unittest it restates."""
import numpy as np
import pytest

import oracle
from oracle import naive_np as N
from pinocchio_amd import model as M, sample_models as S
from util import parity_models, inputs, rel_err_cols

MODELS = parity_models()
IDS = [m.name for m in MODELS]


@pytest.mark.parametrize("model", MODELS, ids=IDS)
def test_aba_of_rnea_is_identity(model):
    """unittest/aba.cpp:143-154,253-258: aba(q, v, rnea(q, v, a)) ~ a."""
    O = oracle.Oracle(model)
    q, v, a = inputs(model, 16, 1)
    ddq = O.aba(q, v, O.rnea(q, v, a))
    assert rel_err_cols(ddq, a).max() < 1e-10


@pytest.mark.parametrize("model", MODELS, ids=IDS)
def test_world_equals_local_aba(model):
    """unittest/aba.cpp:157-159: WORLD and LOCAL conventions agree."""
    O = oracle.Oracle(model)
    q, v, a = inputs(model, 8, 2)
    tau = O.rnea(q, v, a)
    assert rel_err_cols(O.aba(q, v, tau), O.aba(q, v, tau, convention="local")).max() < 1e-10


@pytest.mark.parametrize("model", MODELS[:8], ids=IDS[:8])
def test_rnea_is_affine_in_a_with_spd_matrix(model):
    """unittest/rnea.cpp:198-206 (rnea = M a + nle) and :103-133 (nle = rnea(a=0)):
    tau(a) - tau(0) is linear in a with a symmetric positive-definite matrix."""
    O = oracle.Oracle(model)
    q, v, a = inputs(model, 1, 3)
    nv = model.nv
    nle = O.rnea(q, v, 0 * a)[:, 0]
    Mm = np.zeros((nv, nv))
    for k in range(nv):
        e = np.zeros((nv, 1))
        e[k] = 1
        Mm[:, k] = O.rnea(q, v, e)[:, 0] - nle
    assert np.abs(Mm - Mm.T).max() < 1e-11 * max(1, np.abs(Mm).max())
    assert np.linalg.eigvalsh(0.5 * (Mm + Mm.T)).min() > 0
    tau = O.rnea(q, v, a)[:, 0]
    assert np.linalg.norm(tau - (Mm @ a[:, 0] + nle)) < 1e-11 * np.linalg.norm(tau)


@pytest.mark.parametrize("model", MODELS[:10], ids=IDS[:10])
def test_jacobian_times_qdot_is_joint_velocity(model):
    """unittest/joint-jacobian.cpp:39-57: J_world * qdot ~ oMi.act(v_joint) for every joint: the
    world spatial velocity of joint i is the sum of the columns on its support."""
    O, T = oracle.Oracle(model), N.Tables(model)
    q, v, _ = inputs(model, 1, 4)
    J, oMi = O.fk_jacobians(q)
    J = J[:, 0].reshape(model.nv, 6).T
    # independent velocity recursion in the naive body-frame algebra, moved to the world frame
    vs = [np.zeros(6, dtype=N.LD)]
    H = N.fk(T, q[:, 0])
    for i in range(1, model.njoints):
        Rl, pl, Sj = N._kin(T, i, np.asarray(q[:, 0], dtype=N.LD))
        X = N.xform(Rl, pl)
        iv = model.idx_vs[i]
        vs.append(X @ vs[model.parents[i]] + Sj @ np.asarray(v[iv: iv + model.nvs[i], 0], dtype=N.LD))
    for i in range(1, model.njoints):
        # support columns of joint i
        cols = []
        j = i
        while j > 0:
            cols += list(range(model.idx_vs[j], model.idx_vs[j] + model.nvs[j]))
            j = model.parents[j]
        ov = J[:, cols] @ v[cols, 0]
        R, p = np.asarray(H[i][:3, :3], dtype=float), np.asarray(H[i][:3, 3], dtype=float)
        w_loc, v_loc = np.asarray(vs[i][:3], dtype=float), np.asarray(vs[i][3:], dtype=float)
        w = R @ w_loc
        lin = R @ v_loc + np.cross(p, w)
        assert np.linalg.norm(ov - np.concatenate([lin, w])) < 1e-11 * max(1.0, np.linalg.norm(ov))


@pytest.mark.parametrize("model", [m for m in MODELS if m.is_depth_first()][:9],
                         ids=[m.name for m in MODELS if m.is_depth_first()][:9])
def test_rnea_derivatives_vs_finite_differences(model):
    """unittest/rnea-derivatives.cpp:158-303: forward differences with alpha = 1e-8, tolerance
    sqrt(alpha); dtau/da upper triangle == mass matrix (:293-299)."""
    O, T = oracle.Oracle(model), N.Tables(model)
    q, v, a = inputs(model, 1, 5)
    nv = model.nv
    dq, dv, da, tau = O.rnea_derivatives(q, v, a)
    dq, dv, da = (x[:, 0].reshape(nv, nv).T for x in (dq, dv, da))
    tau0 = O.rnea(q, v, a)[:, 0]
    assert np.linalg.norm(tau[:, 0] - tau0) < 1e-12 * np.linalg.norm(tau0)
    alpha = 1e-8
    fdq, fdv, Mm = (np.zeros((nv, nv)) for _ in range(3))
    for k in range(nv):
        e = np.zeros(nv)
        e[k] = alpha
        qp = np.asarray(N.integrate(T, q[:, 0], e), dtype=float)
        fdq[:, k] = (O.rnea(qp, v[:, 0], a[:, 0])[:, 0] - tau0) / alpha
        fdv[:, k] = (O.rnea(q[:, 0], v[:, 0] + e, a[:, 0])[:, 0] - tau0) / alpha
        e1 = np.zeros(nv)
        e1[k] = 1
        Mm[:, k] = O.rnea(q[:, 0], 0 * e1, e1)[:, 0] - O.rnea(q[:, 0], 0 * e1, 0 * e1)[:, 0]
    tol = np.sqrt(alpha)
    assert np.linalg.norm(dq - fdq) < tol * max(1.0, np.linalg.norm(fdq))
    assert np.linalg.norm(dv - fdv) < tol * max(1.0, np.linalg.norm(fdv))
    assert np.abs(np.triu(da) - np.triu(Mm)).max() < 1e-11 * max(1.0, np.abs(Mm).max())


def test_pendulum_closed_form():
    """single revolute pendulum under gravity, tau = m g l sin(q) + (I + m l^2) qdd."""
    m = M.Model("pendulum")
    j = m.addJoint(0, M.JOINT_RY, np.eye(4), "pivot")
    mass, l, Iyy = 2.0, 0.7, 0.05
    m.appendBodyToJoint(j, M.Inertia(mass, [0, 0, -l], np.diag([0.01, Iyy, 0.02])))
    O = oracle.Oracle(m)
    for qq, qd, qdd in [(0.3, 0.0, 0.0), (-1.1, 0.7, 0.4), (2.0, -1.3, -2.5)]:
        tau = O.rnea([qq], [qd], [qdd])[0, 0]
        expect = mass * 9.81 * l * np.sin(qq) + (Iyy + mass * l * l) * qdd
        assert abs(tau - expect) < 1e-12 * max(1.0, abs(expect))
        assert abs(O.aba([qq], [qd], [tau])[0, 0] - qdd) < 1e-12


def test_free_body_closed_form():
    """free-flying single body: with zero joint torque the body-frame acceleration
    is pure gravity + gyroscopic terms; momentum form I w' + w x I w = 0, v' = R^T g - w x v."""
    m = M.Model("free_body")
    j = m.addJoint(0, M.JOINT_FREEFLYER, np.eye(4), "root")
    I = np.diag([0.3, 0.5, 0.9])
    m.appendBodyToJoint(j, M.Inertia(3.0, [0, 0, 0], I))
    O = oracle.Oracle(m)
    rng = np.random.default_rng(0)
    q = m.random_configuration(rng, 1)
    v = rng.uniform(-1, 1, (6, 1))
    ddq = O.aba(q, v, np.zeros((6, 1)))[:, 0]
    x, y, z, w = q[3:, 0]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    vl, om = v[:3, 0], v[3:, 0]
    wdot = np.linalg.solve(I, -np.cross(om, I @ om))
    vdot = R.T @ np.array([0, 0, -9.81]) - np.cross(om, vl)
    assert np.linalg.norm(ddq - np.concatenate([vdot, wdot])) < 1e-12 * np.linalg.norm(ddq)


def test_batched_equals_serial_bitwise():
    """unittest/parallel-rnea.cpp:46-54, parallel-aba.cpp:46-54: the OpenMP batch loop gives
    bitwise the same result as a serial loop (batch 128, max threads)."""
    model = S.simple_humanoid()
    O = oracle.Oracle(model)
    q, v, a = inputs(model, 128, 6)
    th = oracle.max_threads()
    tau_p, tau_s = O.rnea(q, v, a, threads=th), O.rnea(q, v, a, threads=1)
    assert np.array_equal(tau_p, tau_s)
    assert np.array_equal(O.aba(q, v, tau_p, threads=th), O.aba(q, v, tau_s, threads=1))


def test_external_forces_world_and_local_aba_agree_and_invert_rnea():
    """unittest/aba.cpp (test_aba_with_fext): aba(q, v, rnea(q, v, a, fext), fext) == a, in both
    conventions; zero forces give the plain algorithms; rnea is affine in fext."""
    import numpy as np
    import oracle
    from pinocchio_amd import sample_models as S
    from util import inputs
    for m in (S.simple_humanoid(), S.random_tree(11, 10), S.arm7()):
        O = oracle.Oracle(m)
        q, v, a = inputs(m, 7, 5)
        rng = np.random.default_rng(5)
        f1, f2 = (rng.uniform(-3, 3, (6 * m.njoints, 7)) for _ in range(2))
        tau = O.rnea_fext(q, v, a, f1)
        assert np.abs(O.aba_fext(q, v, tau, f1) - a).max() < 1e-9
        assert np.abs(O.aba_fext(q, v, tau, f1, "local") - a).max() < 1e-9
        assert np.abs(O.rnea_fext(q, v, a, 0 * f1) - O.rnea(q, v, a)).max() == 0.0
        lin = O.rnea_fext(q, v, a, f1 + f2) - O.rnea_fext(q, v, a, f1) - O.rnea_fext(q, v, a, f2) + O.rnea(q, v, a)
        assert np.abs(lin).max() < 1e-10
        # the universe's entry is ignored
        f3 = f1.copy(); f3[:6] = 99.0
        assert np.abs(O.rnea_fext(q, v, a, f3) - tau).max() == 0.0


def test_joint_jacobian_frames_are_consistent():
    """unittest/joint-jacobian.cpp:39-57: WORLD columns = the support columns of the full Jacobian;
    LOCAL = oMi.actInv(WORLD); LOCAL_WORLD_ALIGNED = rotation of LOCAL into the world basis; the
    LOCAL Jacobian times v is the joint's spatial velocity (here: RNEA-free check through the
    velocity recursion of the naive implementation is replaced by J_world v == oMi.act(J_local v))."""
    import numpy as np
    import oracle
    from pinocchio_amd import sample_models as S
    from util import inputs
    for m in (S.simple_humanoid(), S.random_tree(21, 10)):
        O = oracle.Oracle(m)
        q, v, _ = inputs(m, 4, 2)
        Jfull, oMi = O.fk_jacobians(q)
        for j in (1, m.njoints // 2, m.njoints - 1):
            Jw, Jl, Ja = (O.joint_jacobian(q, j, rf) for rf in (0, 1, 2))
            support = []
            i = j
            while i > 0:
                support += list(range(m.idx_vs[i], m.idx_vs[i] + m.nvs[i]))
                i = m.parents[i]
            for b in range(4):
                W, L, A_ = (x[:, b].reshape(m.nv, 6).T for x in (Jw, Jl, Ja))
                F = Jfull[:, b].reshape(m.nv, 6).T
                other = [c for c in range(m.nv) if c not in support]
                assert np.abs(W[:, support] - F[:, support]).max() == 0.0 and np.abs(W[:, other]).max() == 0.0
                R = oMi[12 * j:12 * j + 9, b].reshape(3, 3).T
                t = oMi[12 * j + 9:12 * j + 12, b]
                assert np.abs(np.vstack([R @ L[:3], R @ L[3:]]) - A_).max() < 1e-12
                back = np.vstack([R @ L[:3] + np.cross(t, (R @ L[3:]).T).T, R @ L[3:]])   # oMi.act(LOCAL)
                assert np.abs(back - W).max() < 1e-12
