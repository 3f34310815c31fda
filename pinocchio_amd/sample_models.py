"""Built-in constant models used by the tests, the harness and bench.py.

* ``simple_humanoid()``  -- table flattened from the reference's
  ``models/simple_humanoid.urdf`` (+ free-flyer root) by ``tools/gen_model_tables.py``
  and committed as ``pinocchio_amd/models/simple_humanoid_ff.json``: FF + 29 axis-aligned
  revolute joints, nq 36 / nv 35 (the headline config of BASELINE.json).
* ``baxter_simple()``    -- same for ``models/baxter_simple.urdf`` (fixed base): exercises
  fixed-joint merging, rpy placements and prismatic fingers.
* ``manipulator()``      -- restatement of ``buildModels::manipulator``
  (``include/pinocchio/multibody/sample-models.hxx:57-141,222-226``) as data.
* ``humanoid()``         -- restatement of ``buildModels::humanoid`` (``sample-models.hxx:312-405``,
  free-flyer root): the reference's fully deterministic test humanoid, FF + 28 revolute, nq 35 / nv 34.
* ``arm7()``             -- Franka-Panda-class 7-dof RZ chain (stand-in: the real URDF is
  not available offline; kinematics follow the public DH table, inertias synthetic).
* ``quadruped()``        -- ANYmal-class FF + 4 x (RX, RY, RY), nq 19 / nv 18 (stand-in).
* ``talos_like()``       -- Talos-class FF + 32 revolute, nq 39 / nv 38 (stand-in).
* ``random_tree()``      -- seeded random trees over every supported joint kind (tests).
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import numpy as np

from . import model as M

_HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name: str) -> M.Model:
    with open(os.path.join(_HERE, "models", name + ".json")) as f:
        return M.Model.from_json(f.read())


def simple_humanoid() -> M.Model:
    return _load("simple_humanoid_ff")


def baxter_simple() -> M.Model:
    return _load("baxter_simple")


def _T(R: Optional[np.ndarray] = None, t: Sequence[float] = (0, 0, 0)) -> np.ndarray:
    P = np.eye(4)
    if R is not None:
        P[:3, :3] = R
    P[:3, 3] = t
    return P


def _rotx(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _roty(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _rotz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def manipulator() -> M.Model:
    """buildModels::manipulator: RX, RY, RZ shoulder, RY elbow (+1 m z), RX wrist1 (+1 m z),
    RY wrist2; Ijoint = (0.1, 0, 0.01*I), Iarm = (1, (0,0,0.5), I)
    (sample-models.hxx:72-77,84-127)."""
    m = M.Model("manipulator")
    Ijoint = M.Inertia(0.1, np.zeros(3), np.eye(3) * 0.01)
    Iarm = M.Inertia(1.0, np.array([0, 0, 0.5]), np.eye(3))
    Marm = _T(t=(0, 0, 1))
    Id = np.eye(4)
    spec = [
        (M.JOINT_RX, Id, Ijoint, "shoulder1"), (M.JOINT_RY, Id, Ijoint, "shoulder2"),
        (M.JOINT_RZ, Id, Iarm, "shoulder3"), (M.JOINT_RY, Marm, Iarm, "elbow"),
        (M.JOINT_RX, Marm, Ijoint, "wrist1"), (M.JOINT_RY, Id, Iarm, "wrist2"),
    ]
    parent = 0
    for jt, P, Y, name in spec:
        parent = m.addJoint(parent, jt, P, name + "_joint")
        m.inertias[parent] = Y  # sample-models.hxx:86 overwrites the random inertia
    return m


def humanoid() -> M.Model:
    """buildModels::humanoid(model, usingFF = true) (sample-models.hxx:312-405): free-flyer root with
    Ijoint; two legs = addManipulator (sample-models.hxx:57-141) at SE3(rotate(pi, X), (0, -+0.2, -0.1))
    whose last joint placement gets the rotation rotate(pi/2, Y) (:362-365); chest RX (Ijoint), RY
    (Iarm); head RX at (0, 0, 1) (Ijoint), RY (Iarm); two arms = addManipulator on the chest at
    SE3(rotate(pi, X), (0, -+0.3, 1)).  rotate() is Eigen's AngleAxis::toRotationMatrix: with
    cos(pi) = -1, sin(pi) = 1.2e-16 it gives the same entries as the plain axis rotations below."""
    m = M.Model("humanoid")
    Ijoint = M.Inertia(0.1, np.zeros(3), np.eye(3) * 0.01)
    Iarm = M.Inertia(1.0, np.array([0, 0, 0.5]), np.eye(3))
    Marm = _T(t=(0, 0, 1))
    Id = np.eye(4)

    def add(parent, jt, P, Y, name):
        j = m.addJoint(parent, jt, P, name + "_joint")
        m.inertias[j] = Y
        return j

    def add_manipulator(root, Mroot, pre):
        j = root
        for jt, P, Y, name in ((M.JOINT_RX, Mroot, Ijoint, "shoulder1"), (M.JOINT_RY, Id, Ijoint, "shoulder2"),
                               (M.JOINT_RZ, Id, Iarm, "shoulder3"), (M.JOINT_RY, Marm, Iarm, "elbow"),
                               (M.JOINT_RX, Marm, Ijoint, "wrist1"), (M.JOINT_RY, Id, Iarm, "wrist2")):
            j = add(j, jt, P, Y, pre + name)
        return j

    ff = add(0, M.JOINT_FREEFLYER, Id, Ijoint, "root")
    add_manipulator(ff, _T(_rotx(np.pi), (0, -0.2, -0.1)), "rleg_")
    add_manipulator(ff, _T(_rotx(np.pi), (0, 0.2, -0.1)), "lleg_")
    for foot in (7, 13):  # :362-365
        m.jointPlacements[foot][:3, :3] = _roty(np.pi / 2)
    c1 = add(ff, M.JOINT_RX, Id, Ijoint, "chest1")
    chest = add(c1, M.JOINT_RY, Id, Iarm, "chest2")
    h1 = add(chest, M.JOINT_RX, Marm, Ijoint, "head1")
    add(h1, M.JOINT_RY, Id, Iarm, "head2")
    add_manipulator(chest, _T(_rotx(np.pi), (0, -0.3, 1.0)), "rarm_")
    add_manipulator(chest, _T(_rotx(np.pi), (0, 0.3, 1.0)), "larm_")
    assert (m.nq, m.nv, m.njoints) == (35, 34, 30)
    return m


def arm7() -> M.Model:
    """Panda-class 7-dof arm: every joint RZ, placements from the public modified-DH table
    (d1 .333, d3 .316, a4 .0825, a5 -.0825, d5 .384, a7 .088); synthetic inertias."""
    m = M.Model("arm7_panda_class")
    h = np.pi / 2
    P = [
        _T(t=(0, 0, 0.333)), _T(_rotx(-h)), _T(_rotx(h), (0, -0.316, 0)), _T(_rotx(h), (0.0825, 0, 0)),
        _T(_rotx(-h), (-0.0825, 0.384, 0)), _T(_rotx(h)), _T(_rotx(h), (0.088, 0, 0)),
    ]
    mass = [4.97, 0.65, 3.23, 3.59, 1.23, 1.67, 0.74]
    com = [(0.004, 0.002, -0.175), (-0.003, -0.029, 0.004), (0.028, 0.039, -0.067), (-0.053, 0.105, 0.027),
           (-0.012, 0.041, -0.038), (0.060, -0.014, -0.010), (0.010, -0.004, 0.062)]
    diag = [(0.70, 0.71, 0.009), (0.008, 0.028, 0.026), (0.037, 0.036, 0.011), (0.026, 0.020, 0.028),
            (0.036, 0.029, 0.008), (0.002, 0.004, 0.005), (0.013, 0.010, 0.005)]
    off = [(-0.0002, 0.007, 0.019), (-0.004, 0.010, 0.0007), (-0.005, -0.011, -0.013), (0.008, -0.001, 0.004),
           (-0.002, -0.005, 0.0002), (0.0001, -0.0001, 0.0003), (-0.0004, -0.001, -0.0005)]
    parent = 0
    for k in range(7):
        parent = m.addJoint(parent, M.JOINT_RZ, P[k], f"panda_joint{k + 1}")
        I = np.diag(diag[k])
        I[0, 1] = I[1, 0] = off[k][0]
        I[0, 2] = I[2, 0] = off[k][1]
        I[1, 2] = I[2, 1] = off[k][2]
        m.appendBodyToJoint(parent, M.Inertia(mass[k], com[k], I))
    return m


def _box_inertia(mass, sx, sy, sz):
    return np.diag([mass * (sy * sy + sz * sz) / 12, mass * (sx * sx + sz * sz) / 12,
                    mass * (sx * sx + sy * sy) / 12])


def quadruped() -> M.Model:
    """ANYmal-class quadruped: FF + LF/LH/RF/RH x (HAA: RX, HFE: RY, KFE: RY)."""
    m = M.Model("quadruped_anymal_class")
    base = m.addJoint(0, M.JOINT_FREEFLYER, np.eye(4), "root_joint")
    m.appendBodyToJoint(base, M.Inertia(16.8, (-0.001, -0.0007, 0.045), _box_inertia(16.8, 0.53, 0.27, 0.24)))
    for name, sx, sy in (("LF", 1, 1), ("LH", -1, 1), ("RF", 1, -1), ("RH", -1, -1)):
        haa = m.addJoint(base, M.JOINT_RX, _T(t=(sx * 0.277, sy * 0.116, 0.0)), name + "_HAA")
        m.appendBodyToJoint(haa, M.Inertia(1.42, (sx * 0.064, -sy * 0.015, 0.0), _box_inertia(1.42, 0.1, 0.08, 0.08)))
        hfe = m.addJoint(haa, M.JOINT_RY, _T(t=(sx * 0.0635, sy * 0.041, 0.0)), name + "_HFE")
        m.appendBodyToJoint(hfe, M.Inertia(1.63, (0.0, sy * 0.018, -0.21), _box_inertia(1.63, 0.06, 0.06, 0.25)))
        kfe = m.addJoint(hfe, M.JOINT_RY, _T(t=(0.0, sy * 0.109, -0.25)), name + "_KFE")
        m.appendBodyToJoint(kfe, M.Inertia(0.47, (sx * 0.03, -sy * 0.01, -0.09), _box_inertia(0.47, 0.04, 0.04, 0.3)))
    return m


def talos_like() -> M.Model:
    """Talos-class humanoid: FF + 2 x leg(RZ,RX,RY,RY,RY,RX) + torso(RZ,RY) +
    2 x arm(RZ,RX,RZ,RY,RZ,RX,RY) + gripper(RY) each + head(RY,RZ) = 32 revolute,
    nq 39 / nv 38.  Topology/dof of the real robot, synthetic geometry and inertias."""
    m = M.Model("talos_class")
    rng = np.random.default_rng(20250704)

    def body(j, mass, size):
        com = rng.uniform(-0.03, 0.03, 3)
        I = _box_inertia(mass, *size) + np.eye(3) * 1e-4
        m.appendBodyToJoint(j, M.Inertia(mass, com, I))

    base = m.addJoint(0, M.JOINT_FREEFLYER, np.eye(4), "root_joint")
    body(base, 15.4, (0.2, 0.3, 0.2))
    for side, sy in (("left", 1), ("right", -1)):
        p = base
        spec = [(M.JOINT_RZ, (-0.02, sy * 0.085, -0.027), 2.3), (M.JOINT_RX, (0, 0, 0), 1.9),
                (M.JOINT_RY, (0, 0, 0), 6.2), (M.JOINT_RY, (0, 0, -0.38), 3.6),
                (M.JOINT_RY, (0, 0, -0.325), 1.2), (M.JOINT_RX, (0, 0, 0), 1.6)]
        for k, (jt, t, mass) in enumerate(spec):
            p = m.addJoint(p, jt, _T(t=t), f"leg_{side}_{k + 1}_joint")
            body(p, mass, (0.1, 0.1, 0.25))
    t1 = m.addJoint(base, M.JOINT_RZ, _T(t=(0, 0, 0.0722)), "torso_1_joint")
    body(t1, 2.3, (0.15, 0.2, 0.1))
    t2 = m.addJoint(t1, M.JOINT_RY, _T(), "torso_2_joint")
    body(t2, 17.5, (0.25, 0.35, 0.4))
    for side, sy in (("left", 1), ("right", -1)):
        p = t2
        spec = [(M.JOINT_RZ, (0, sy * 0.1575, 0.232), 1.4), (M.JOINT_RX, (0.00493, sy * 0.1365, 0.04673), 1.7),
                (M.JOINT_RZ, (0, 0, 0), 2.3), (M.JOINT_RY, (0.02, 0, -0.273), 1.5),
                (M.JOINT_RZ, (-0.02, 0, -0.2643), 1.8), (M.JOINT_RX, (0, 0, 0), 0.36),
                (M.JOINT_RY, (0, 0, 0), 1.0), (M.JOINT_RY, (0, sy * 0.02, -0.12), 0.3)]
        for k, (jt, t, mass) in enumerate(spec):
            nm = f"arm_{side}_{k + 1}_joint" if k < 7 else f"gripper_{side}_joint"
            p = m.addJoint(p, jt, _T(t=t), nm)
            body(p, mass, (0.08, 0.08, 0.2))
    h1 = m.addJoint(t2, M.JOINT_RY, _T(t=(0, 0, 0.316)), "head_1_joint")
    body(h1, 0.7, (0.1, 0.1, 0.1))
    h2 = m.addJoint(h1, M.JOINT_RZ, _T(), "head_2_joint")
    body(h2, 1.4, (0.15, 0.15, 0.2))
    assert (m.nq, m.nv, m.njoints) == (39, 38, 34)
    return m


def random_tree(seed: int, njoints: int = 12, kinds: Optional[Sequence[int]] = None,
                root_kind: Optional[int] = M.JOINT_FREEFLYER, armature: bool = True,
                chain_bias: float = 0.6) -> M.Model:
    """Seeded random tree in depth-first numbering with random placements and inertias
    (the role buildModels::humanoidRandom plays in the reference's tests,
    sample-models.hxx:237-310; its Eigen/rand() stream cannot be reproduced)."""
    rng = np.random.default_rng(seed)
    kinds = list(kinds) if kinds is not None else [
        M.JOINT_RX, M.JOINT_RY, M.JOINT_RZ, M.JOINT_PX, M.JOINT_PY, M.JOINT_PZ, M.JOINT_SPHERICAL]
    m = M.Model(f"random_tree_{seed}")

    def rand_R():
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

    def rand_inertia():
        mass = rng.uniform(0.5, 5.0)
        A = rng.uniform(-1, 1, (3, 3))
        I = A @ A.T * 0.05 + np.eye(3) * 0.02
        return M.Inertia(mass, rng.uniform(-0.2, 0.2, 3), I)

    # parents drawn so that the numbering stays depth-first: the parent of a new joint is
    # always on the current "rightmost" path
    path = [0]
    for i in range(1, njoints):
        if i == 1 and root_kind is not None:
            jt = root_kind
        else:
            jt = int(rng.choice(kinds))
        if len(path) > 1 and rng.uniform() > chain_bias:
            depth = int(rng.integers(1 if root_kind is not None and i > 1 else 0, len(path)))
            path = path[: depth + 1]
        parent = path[-1]
        P = _T(rand_R(), rng.uniform(-0.4, 0.4, 3)) if i > 1 or root_kind is None else np.eye(4)
        axis = rng.standard_normal(3)
        jid = m.addJoint(parent, jt, P, f"j{i}", axis=axis)
        m.appendBodyToJoint(jid, rand_inertia(), _T(rand_R(), rng.uniform(-0.1, 0.1, 3)))
        path.append(jid)
    if armature:
        m.armature = rng.uniform(0.0, 0.3, m.nv)
    assert m.is_depth_first()
    return m


BUILTIN = {
    "simple_humanoid": simple_humanoid,
    "baxter_simple": baxter_simple,
    "manipulator": manipulator,
    "humanoid": humanoid,
    "arm7": arm7,
    "quadruped": quadruped,
    "talos_like": talos_like,
}


def build(name: str) -> M.Model:
    if name not in BUILTIN:
        raise KeyError(f"unknown model {name!r}; known: {sorted(BUILTIN)}")
    return BUILTIN[name]()
