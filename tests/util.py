"""Shared helpers for the parity tests."""
import numpy as np

from pinocchio_amd import model as M
from pinocchio_amd import sample_models as S


def rel_err_cols(x, y):
    """Eigen isApprox semantics per instance (column): ||x-y|| / min(||x||,||y||)
    (e.g. reference unittest/aba.cpp:154)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    num = np.linalg.norm(x - y, axis=0)
    den = np.minimum(np.linalg.norm(x, axis=0), np.linalg.norm(y, axis=0))
    den = np.where(den > 0, den, 1.0)
    return num / den


def inputs(model, B, seed=0):
    rng = np.random.default_rng(seed)
    q = model.random_configuration(rng, B)
    v = np.asfortranarray(rng.uniform(-1, 1, (model.nv, B)))
    a = np.asfortranarray(rng.uniform(-1, 1, (model.nv, B)))
    return q, v, a


def parity_models():
    """Models every parity test runs on: the built-ins + random trees over all kernel-supported
    joint kinds, with and without free-flyer root / armature, including multi-root forests."""
    kinds = [M.JOINT_RX, M.JOINT_RY, M.JOINT_RZ, M.JOINT_PX, M.JOINT_PY, M.JOINT_PZ, M.JOINT_SPHERICAL,
             M.JOINT_RUBX, M.JOINT_RUBY, M.JOINT_RUBZ, M.JOINT_FREEFLYER]
    out = [S.simple_humanoid(), S.manipulator(), S.humanoid(), S.arm7(), S.quadruped(), S.talos_like(), S.baxter_simple()]
    for s in range(8):
        out.append(S.random_tree(100 + s, 8 + s, kinds=kinds if s % 2 else None,
                                 root_kind=(M.JOINT_FREEFLYER if s % 3 else None),
                                 armature=bool(s % 2), chain_bias=0.4 + 0.07 * s))
    return out
