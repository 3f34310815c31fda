"""Parity pinned by VALUE: the numbers the reference's own tests hold for this path
(tests/golden/reference_value_cpp.json, transcribed from /root/reference/unittest/value.cpp:31-179 and
unittest/joint-prismatic.cpp:137-216 by tools/gen_reference_fixture.py), with the tolerance each
BOOST_CHECK uses (Eigen isApprox).

CPU tests: the oracle (C restatement and the independent numpy implementation) on the model that
pinocchio_amd/urdf.py builds from the reference's simple_humanoid.urdf (a data fixture,
tests/golden/simple_humanoid.urdf) -- this also pins the URDF converter's joint order and inertia
merging, and the committed table pinocchio_amd/models/simple_humanoid_ff.json.
GPU tests (-m gpu): the HIP path through the C ABI -- compiled and table-driven kernels, both
boundary layouts, host and device memory.
"""
import json
import os

import numpy as np
import pytest

import oracle
from oracle import naive_np
from pinocchio_amd import model as M, sample_models as S, urdf

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLD, "reference_value_cpp.json")) as _f:
    REF = json.load(_f)
CASES = REF["simple_humanoid"]["cases"]


def is_approx(x, y, prec):
    """Eigen::DenseBase::isApprox: ||x - y|| <= prec * min(||x||, ||y||)."""
    x, y = np.asarray(x, dtype=np.float64).ravel(), np.asarray(y, dtype=np.float64).ravel()
    return np.linalg.norm(x - y) <= prec * min(np.linalg.norm(x), np.linalg.norm(y))


def humanoid_from_urdf():
    with open(os.path.join(GOLD, "simple_humanoid.urdf")) as f:
        m = urdf.build_model_from_urdf(f.read(), free_flyer=True, name="simple_humanoid")
    m.gravity = np.asarray(REF["simple_humanoid"]["gravity"], dtype=np.float64)  # value.cpp:37
    return m


def px_model():
    """unittest/joint-prismatic.cpp:143-147: one PX joint at the universe, Inertia(1, (0.5,0,0), I)."""
    d = REF["prismatic_px"]["model"]
    m = M.Model("px")
    j = m.addJoint(0, M.JOINT_PX, np.eye(4), "root")
    m.appendBodyToJoint(j, M.Inertia(d["mass"], d["lever"], np.eye(3)))
    return m


def stacked(key):
    return np.asfortranarray(np.array([c[key] for c in CASES], dtype=np.float64).T)


def test_urdf_converter_reproduces_the_committed_table():
    """unittest/urdf.cpp:247-252: njoints == 31 with a free-flyer; nq 36 / nv 35; and the table the
    kernels are registered for is exactly what the converter makes of the reference's URDF."""
    m = humanoid_from_urdf()
    assert (m.njoints, m.nq, m.nv) == (31, 36, 35) and m.names[1] == "root_joint"
    t, c = m.tables(), S.simple_humanoid().tables()
    for k in ("parents", "jtype", "idx_q", "idx_v", "placement", "inertia", "armature"):
        assert np.array_equal(t[k], c[k]), k


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_value_cpp(case):
    m = humanoid_from_urdf()
    tau = oracle.Oracle(m).rnea(case["q"], case["v"], case["a"])[:, 0]
    assert is_approx(case["expected"], tau, case["tol"]), np.abs(tau - case["expected"]).max()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_independent_numpy_implementation_matches_value_cpp(case):
    m = humanoid_from_urdf()
    tau = np.asarray(naive_np.rnea(naive_np.Tables(m), case["q"], case["v"], case["a"]), dtype=np.float64)
    assert is_approx(case["expected"], tau, case["tol"])


def test_oracle_matches_joint_prismatic_cpp():
    m = px_model()
    O = oracle.Oracle(m)
    for c in REF["prismatic_px"]["rnea"]:
        tau = O.rnea(c["q"], c["v"], c["a"])[:, 0]
        if c.get("absolute"):  # expected == 0: isApprox(0, x) holds only for x == 0
            assert np.abs(tau - c["expected"]).max() <= c["tol"]
        else:
            assert is_approx(c["expected"], tau, c["tol"])
    for c in REF["prismatic_px"]["mass_matrix"]:  # crba's M == dtau/da of computeRNEADerivatives
        da = O.rnea_derivatives(c["q"], [0.0], [0.0])[2][:, 0].reshape(1, 1)
        assert is_approx(c["expected"], da, c["tol"])


# ---------------------------------------------------------------------------------------------
# GPU: the HIP path through the C ABI
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("compiled", [True, False], ids=["compiled", "table_driven"])
@pytest.mark.parametrize("order", ["F", "C"], ids=["instance_major", "batch_major"])
def test_hip_rnea_matches_value_cpp(compiled, order):
    import torch
    from pinocchio_amd import parallel
    m = humanoid_from_urdf()
    pool = parallel.ModelPool(m, 1)
    dm = pool.device_model(0)
    dm.use_compiled(compiled)
    q, v, a = (np.asarray(stacked(k), order=order) for k in ("q", "v", "a"))
    # host memory (the reference's calling convention)
    tau = parallel.rneaInParallel(1, pool, q, v, a)
    for k, c in enumerate(CASES):
        assert is_approx(c["expected"], tau[:, k], c["tol"]), (c["name"], "host")
    # device memory
    def put(x):
        t = torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
        return t if order == "C" else torch.from_numpy(np.ascontiguousarray(x.T)).to("cuda:0").t()
    taud = parallel.rneaInParallel(1, pool, put(q), put(v), put(a))
    torch.cuda.synchronize()
    taud = taud.cpu().numpy()
    for k, c in enumerate(CASES):
        assert is_approx(c["expected"], taud[:, k], c["tol"]), (c["name"], "device")
    # ... and the value-pinned tau fed back through ABA returns the recipe's accelerations
    # (unittest/aba.cpp:143-154) for the cases with a unit-norm-free quaternion (R = I from q = 0)
    ddq = parallel.abaInParallel(1, pool, q, v, np.asarray(tau, order=order))
    assert np.abs(ddq - stacked("a")).max() < 1e-9
    pool.release()


@pytest.mark.gpu
def test_hip_gravity_signature_falls_back_or_matches():
    """gravity -9.8 keeps the 0/0/generic pattern of the registered humanoid: the compiled kernels
    must still be selected (numeric constants come from the run-time table)."""
    from pinocchio_amd import parallel
    pool = parallel.ModelPool(humanoid_from_urdf(), 1)
    assert pool.device_model(0).compiled_kernels() == "simple_humanoid_p"
    pool.release()


@pytest.mark.gpu
def test_hip_matches_joint_prismatic_cpp():
    from pinocchio_amd import parallel
    m = px_model()
    pool = parallel.ModelPool(m, 1)
    for c in REF["prismatic_px"]["rnea"]:
        col = lambda x: np.asfortranarray(np.array(x, dtype=np.float64).reshape(1, 1))
        tau = parallel.rneaInParallel(1, pool, col(c["q"]), col(c["v"]), col(c["a"]))[:, 0]
        if c.get("absolute"):
            assert np.abs(tau - c["expected"]).max() <= c["tol"]
        else:
            assert is_approx(c["expected"], tau, c["tol"])
    for c in REF["prismatic_px"]["mass_matrix"]:
        col = lambda x: np.asfortranarray(np.array(x, dtype=np.float64).reshape(1, 1))
        da = parallel.computeRNEADerivatives(pool, col(c["q"]), col([0.0]), col([0.0]))[2]
        assert is_approx(c["expected"], da[:, 0], c["tol"])
    pool.release()
