"""Host-side constant model description: the input of the BatchModel flattener.

Mirrors the fields of ``pinocchio::ModelTpl`` that the hot path reads
(reference ``include/pinocchio/multibody/model.hpp:107-132,154,176,248``) and the
two construction calls that decide them, ``Model::addJoint``
(``multibody/model.hxx:144-179``: joint_id = njoints++, idx_q = nq, idx_v = nv in
insertion order, parent < njoints) and ``Model::appendBodyToJoint``
(``multibody/model.hxx:620-625``: inertias[j] += placement.act(Y)).

This is NOT a re-implementation of pinocchio's Model: no frames, limits, names
lookup, geometry.  It only carries the constant tree the kernels need and knows
how to hand it to the C ABI (``include/brbd.h: brbd_model_desc``).
"""
from __future__ import annotations

import ctypes
import json
from typing import List, Optional, Sequence

import numpy as np

# enum brbd_joint_type (include/brbd.h)
JOINT_RX, JOINT_RY, JOINT_RZ = 0, 1, 2
JOINT_PX, JOINT_PY, JOINT_PZ = 3, 4, 5
JOINT_FREEFLYER = 6
JOINT_SPHERICAL = 7
JOINT_REVOLUTE_UNALIGNED = 8
JOINT_PRISMATIC_UNALIGNED = 9
JOINT_RUBX, JOINT_RUBY, JOINT_RUBZ = 10, 11, 12
JOINT_REVOLUTE_UNBOUNDED_UNALIGNED = 13

JOINT_NQ = (1, 1, 1, 1, 1, 1, 7, 4, 1, 1, 2, 2, 2, 2)
JOINT_NV = (1, 1, 1, 1, 1, 1, 6, 3, 1, 1, 1, 1, 1, 1)

# pinocchio shortname() of each kind (e.g. joint-revolute.hpp:848-851)
JOINT_SHORTNAMES = (
    "JointModelRX", "JointModelRY", "JointModelRZ",
    "JointModelPX", "JointModelPY", "JointModelPZ",
    "JointModelFreeFlyer", "JointModelSpherical",
    "JointModelRevoluteUnaligned", "JointModelPrismaticUnaligned",
    "JointModelRUBX", "JointModelRUBY", "JointModelRUBZ",
    "JointModelRevoluteUnboundedUnaligned",
)
SHORTNAME_TO_TYPE = {n: i for i, n in enumerate(JOINT_SHORTNAMES)}


class brbd_model_desc(ctypes.Structure):
    """ctypes image of ``struct brbd_model_desc`` (include/brbd.h)."""

    _fields_ = [
        ("njoints", ctypes.c_int32),
        ("nq", ctypes.c_int32),
        ("nv", ctypes.c_int32),
        ("parents", ctypes.POINTER(ctypes.c_int32)),
        ("jtype", ctypes.POINTER(ctypes.c_int32)),
        ("idx_q", ctypes.POINTER(ctypes.c_int32)),
        ("idx_v", ctypes.POINTER(ctypes.c_int32)),
        ("placement", ctypes.POINTER(ctypes.c_double)),
        ("inertia", ctypes.POINTER(ctypes.c_double)),
        ("axis", ctypes.POINTER(ctypes.c_double)),
        ("armature", ctypes.POINTER(ctypes.c_double)),
        ("gravity", ctypes.c_double * 3),
    ]


def _sym3_from_matrix(I: np.ndarray) -> np.ndarray:
    """Symmetric3 data order (xx, xy, yy, xz, yz, zz), spatial/symmetric3.hpp:43-52."""
    return np.array([I[0, 0], I[0, 1], I[1, 1], I[0, 2], I[1, 2], I[2, 2]], dtype=np.float64)


def _sym3_to_matrix(d: np.ndarray) -> np.ndarray:
    return np.array([[d[0], d[1], d[3]], [d[1], d[2], d[4]], [d[3], d[4], d[5]]], dtype=np.float64)


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


class Inertia:
    """(mass, lever, rotational inertia about the COM); spatial/inertia.hpp:286-291."""

    def __init__(self, mass: float, lever: Sequence[float], inertia: np.ndarray):
        self.mass = float(mass)
        self.lever = np.asarray(lever, dtype=np.float64).reshape(3).copy()
        self.inertia = np.asarray(inertia, dtype=np.float64).reshape(3, 3).copy()

    @staticmethod
    def Zero() -> "Inertia":
        return Inertia(0.0, np.zeros(3), np.zeros((3, 3)))

    def se3Action(self, R: np.ndarray, t: np.ndarray) -> "Inertia":
        """M.act(Y) = (m, t + R c, R I R^T); spatial/inertia.hpp:885-893."""
        return Inertia(self.mass, t + R @ self.lever, R @ self.inertia @ R.T)

    def __add__(self, o: "Inertia") -> "Inertia":
        """spatial/inertia.hpp:654-670 (__plus__)."""
        eps = np.finfo(np.float64).eps
        mab = self.mass + o.mass
        mab_inv = 1.0 / max(mab, eps)
        AB = self.lever - o.lever
        sx = _skew(AB)
        return Inertia(
            mab,
            (self.mass * self.lever + o.mass * o.lever) * mab_inv,
            self.inertia + o.inertia - (self.mass * o.mass * mab_inv) * (sx @ sx),
        )

    def to_vec10(self) -> np.ndarray:
        return np.concatenate([[self.mass], self.lever, _sym3_from_matrix(self.inertia)])


class Model:
    """Constant kinematic tree (joint 0 = universe)."""

    def __init__(self, name: str = "model"):
        self.name = name
        self.njoints = 1
        self.nq = 0
        self.nv = 0
        self.parents: List[int] = [0]
        self.names: List[str] = ["universe"]
        self.jtypes: List[int] = [-1]
        self.idx_qs: List[int] = [0]
        self.idx_vs: List[int] = [0]
        self.nqs: List[int] = [0]
        self.nvs: List[int] = [0]
        self.jointPlacements: List[np.ndarray] = [np.eye(4)]  # homogeneous 4x4
        self.inertias: List[Inertia] = [Inertia.Zero()]
        self.axes: List[np.ndarray] = [np.zeros(3)]
        self.armature = np.zeros(0)
        self.gravity = np.array([0.0, 0.0, -9.81])  # model.hpp:250-251,282

    # -- construction (multibody/model.hxx:144-179, 620-625) -----------------
    def addJoint(self, parent: int, jtype: int, placement: np.ndarray, name: str,
                 axis: Optional[Sequence[float]] = None) -> int:
        if not (0 <= parent < self.njoints):
            raise ValueError("The index of the parent joint is not valid.")
        if not (0 <= jtype < len(JOINT_NQ)):
            raise ValueError(f"unsupported joint type {jtype}")
        placement = np.asarray(placement, dtype=np.float64)
        assert placement.shape == (4, 4)
        jid = self.njoints
        self.njoints += 1
        self.parents.append(int(parent))
        self.names.append(name)
        self.jtypes.append(int(jtype))
        self.idx_qs.append(self.nq)
        self.idx_vs.append(self.nv)
        self.nqs.append(JOINT_NQ[jtype])
        self.nvs.append(JOINT_NV[jtype])
        self.nq += JOINT_NQ[jtype]
        self.nv += JOINT_NV[jtype]
        self.jointPlacements.append(placement.copy())
        self.inertias.append(Inertia.Zero())
        ax = np.zeros(3) if axis is None else np.asarray(axis, dtype=np.float64).reshape(3)
        if jtype in (JOINT_REVOLUTE_UNALIGNED, JOINT_PRISMATIC_UNALIGNED, JOINT_REVOLUTE_UNBOUNDED_UNALIGNED):
            n = np.linalg.norm(ax)
            if not n > 0:
                raise ValueError("unaligned joint needs a non-zero axis")
            ax = ax / n  # parsers/urdf/model.hxx:619-626 axis.normalized()
        self.axes.append(ax)
        self.armature = np.concatenate([self.armature, np.zeros(JOINT_NV[jtype])])
        return jid

    def appendBodyToJoint(self, joint_id: int, Y: Inertia, placement: Optional[np.ndarray] = None):
        P = np.eye(4) if placement is None else np.asarray(placement, dtype=np.float64)
        self.inertias[joint_id] = self.inertias[joint_id] + Y.se3Action(P[:3, :3], P[:3, 3])

    # -- structure queries -----------------------------------------------------
    def children(self, i: int) -> List[int]:
        return [j for j in range(1, self.njoints) if self.parents[j] == i]

    def is_depth_first(self) -> bool:
        """True when every subtree is a contiguous index range (what
        Data::computeNvSubtree assumes, multibody/data.hxx:205-230)."""
        last = list(range(self.njoints))
        for j in range(self.njoints - 1, 0, -1):
            p = self.parents[j]
            last[p] = max(last[p], last[j])
        for i in range(1, self.njoints):
            for j in range(i + 1, last[i] + 1):
                p = j
                while p > i:
                    p = self.parents[p]
                if p != i:
                    return False
        return True

    # -- flat tables -------------------------------------------------------------
    def tables(self) -> dict:
        n = self.njoints
        placement = np.zeros((n, 12))
        inertia = np.zeros((n, 10))
        axis = np.zeros((n, 3))
        for i in range(n):
            P = self.jointPlacements[i]
            placement[i, :9] = P[:3, :3].T.reshape(9)  # column-major rotation
            placement[i, 9:] = P[:3, 3]
            inertia[i] = self.inertias[i].to_vec10()
            axis[i] = self.axes[i]
        return dict(
            name=self.name, njoints=n, nq=self.nq, nv=self.nv,
            parents=np.asarray(self.parents, dtype=np.int32),
            jtype=np.asarray([max(t, 0) for t in self.jtypes], dtype=np.int32),
            idx_q=np.asarray(self.idx_qs, dtype=np.int32),
            idx_v=np.asarray(self.idx_vs, dtype=np.int32),
            placement=placement, inertia=inertia, axis=axis,
            armature=np.asarray(self.armature, dtype=np.float64).copy(),
            gravity=np.asarray(self.gravity, dtype=np.float64).copy(),
            names=list(self.names),
        )

    def to_desc(self):
        """Return (brbd_model_desc, keepalive) for the C ABI."""
        t = self.tables()
        keep = {}

        def ip(a):
            a = np.ascontiguousarray(a, dtype=np.int32)
            keep[id(a)] = a
            return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))

        def dp(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep[id(a)] = a
            return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))

        d = brbd_model_desc()
        d.njoints, d.nq, d.nv = t["njoints"], t["nq"], t["nv"]
        d.parents, d.jtype = ip(t["parents"]), ip(t["jtype"])
        d.idx_q, d.idx_v = ip(t["idx_q"]), ip(t["idx_v"])
        d.placement, d.inertia, d.axis = dp(t["placement"]), dp(t["inertia"]), dp(t["axis"])
        arm = t["armature"] if t["nv"] > 0 else np.zeros(1)
        d.armature = dp(arm)
        for k in range(3):
            d.gravity[k] = float(t["gravity"][k])
        return d, keep

    # -- (de)serialisation: how fixtures are committed -----------------------------
    def to_json(self) -> str:
        t = self.tables()
        out = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in t.items()}
        return json.dumps(out, indent=1)

    @staticmethod
    def from_json(s: str) -> "Model":
        t = json.loads(s)
        m = Model(t.get("name", "model"))
        n = t["njoints"]
        names = t.get("names") or ["universe"] + [f"joint{i}" for i in range(1, n)]
        for i in range(1, n):
            p12 = np.asarray(t["placement"][i], dtype=np.float64)
            P = np.eye(4)
            P[:3, :3] = p12[:9].reshape(3, 3).T
            P[:3, 3] = p12[9:]
            jid = m.addJoint(t["parents"][i], t["jtype"][i], P, names[i], axis=t["axis"][i])
            y = np.asarray(t["inertia"][i], dtype=np.float64)
            m.inertias[jid] = Inertia(y[0], y[1:4], _sym3_to_matrix(y[4:]))
            assert m.idx_qs[jid] == t["idx_q"][i] and m.idx_vs[jid] == t["idx_v"][i]
        m.armature = np.asarray(t["armature"], dtype=np.float64).reshape(-1)[: m.nv].copy()
        m.gravity = np.asarray(t["gravity"], dtype=np.float64)
        assert m.nq == t["nq"] and m.nv == t["nv"]
        return m

    def copy(self) -> "Model":
        return Model.from_json(self.to_json())

    # -- inputs ---------------------------------------------------------------------
    def neutral(self) -> np.ndarray:
        q = np.zeros(self.nq)
        for i in range(1, self.njoints):
            jt, iq = self.jtypes[i], self.idx_qs[i]
            if jt == JOINT_FREEFLYER:
                q[iq + 6] = 1.0
            elif jt == JOINT_SPHERICAL:
                q[iq + 3] = 1.0
            elif jt in (JOINT_RUBX, JOINT_RUBY, JOINT_RUBZ, JOINT_REVOLUTE_UNBOUNDED_UNALIGNED):
                q[iq] = 1.0
        return q

    def random_configuration(self, rng: np.random.Generator, B: int) -> np.ndarray:
        """(nq, B) Fortran-ordered like Eigen.  Mirrors
        benchmark/timings-parallel.cpp:48-61 / model-fixture.hpp:42-46: revolute and
        prismatic q ~ U[-1,1], free-flyer position ~ U[-1,1]^3, quaternions uniform
        on S^3 and normalised (math/quaternion.hpp:114-136)."""
        q = np.empty((self.nq, B), order="F")
        for i in range(1, self.njoints):
            jt, iq = self.jtypes[i], self.idx_qs[i]
            if jt == JOINT_FREEFLYER:
                q[iq:iq + 3] = rng.uniform(-1, 1, (3, B))
                quat = rng.standard_normal((4, B))
                q[iq + 3:iq + 7] = quat / np.linalg.norm(quat, axis=0)
            elif jt == JOINT_SPHERICAL:
                quat = rng.standard_normal((4, B))
                q[iq:iq + 4] = quat / np.linalg.norm(quat, axis=0)
            elif jt in (JOINT_RUBX, JOINT_RUBY, JOINT_RUBZ, JOINT_REVOLUTE_UNBOUNDED_UNALIGNED):
                th = rng.uniform(-np.pi, np.pi, B)
                q[iq] = np.cos(th)
                q[iq + 1] = np.sin(th)
            else:
                q[iq] = rng.uniform(-1, 1, B)
        return q
