"""Per-joint-kind kinematics on symbolic values: liMi = placement * M(q), the motion subspace S,
and the depth-first visit program.

Reference: joint-revolute.hpp:132-166,801-830 (aligned revolute), joint-revolute-unbounded.hpp:153-162,
joint-revolute-unaligned.hpp:667-708, joint-prismatic.hpp:262-272, joint-prismatic-unaligned.hpp,
joint-free-flyer.hpp:332-374, joint-spherical.hpp:525-589.
"""
from __future__ import annotations

from typing import List, Tuple

from .. import model as M
from . import algebra as A
from .emit import ONE, ZERO, Emitter, V

REV = (M.JOINT_RX, M.JOINT_RY, M.JOINT_RZ)
PRI = (M.JOINT_PX, M.JOINT_PY, M.JOINT_PZ)
RUB = (M.JOINT_RUBX, M.JOINT_RUBY, M.JOINT_RUBZ)


class ModelView:
    """Numeric tables of one model + constant access with folding."""

    def __init__(self, model: M.Model, dense: bool = False, arm_generic: bool = False):
        """``dense``: every geometric / inertial constant generic (any model with this tree);
        ``arm_generic``: the armature of 1-dof joints generic (rotor inertias, aba.hxx:178-179,
        rnea.hxx:158: run-time constants instead of part of the pattern).  Gravity is always generic."""
        self.model = model
        self.t = model.tables()
        self.dense = dense
        self.arm_generic = arm_generic or dense
        self.n = model.njoints
        self.parents = [int(p) for p in self.t["parents"]]
        self.jtype = [int(x) for x in self.t["jtype"]]
        self.idx_q = [int(x) for x in self.t["idx_q"]]
        self.idx_v = [int(x) for x in self.t["idx_v"]]
        self.children: List[List[int]] = [[] for _ in range(self.n)]
        for i in range(1, self.n):
            self.children[self.parents[i]].append(i)
        self.dof_nv = {}  # dof index -> nv of the joint that owns it
        for i in range(1, self.n):
            for k in range(self.nv(i)):
                self.dof_nv[self.idx_v[i] + k] = self.nv(i)

    def generic(self, ident) -> bool:
        """Is this constant a table entry whatever its value?"""
        _, field, k = ident
        if field == "g":
            return True
        if field == "arm":  # (multi-dof joints keep their pattern: a free-flyer without armature is solved directly)
            return self.arm_generic and self.dof_nv.get(k, 0) == 1
        return self.dense

    def nv(self, i: int) -> int:
        return M.JOINT_NV[self.jtype[i]]

    def value(self, ident) -> float:
        """Numeric value of a model constant (joint, field, index)."""
        i, field, k = ident
        if field == "R":  # row-major index 3*r + c; table holds the rotation column-major
            r, c = divmod(k, 3)
            return float(self.t["placement"][i][3 * c + r])
        if field == "t":
            return float(self.t["placement"][i][9 + k])
        if field == "m":
            return float(self.t["inertia"][i][0])
        if field == "c":
            return float(self.t["inertia"][i][1 + k])
        if field == "I":
            return float(self.t["inertia"][i][4 + k])
        if field == "arm":
            return float(self.t["armature"][k])
        if field == "axis":
            return float(self.t["axis"][i][k])
        if field == "g":
            return float(self.t["gravity"][k])
        raise KeyError(field)

    def c(self, e: Emitter, i: int, field: str, k: int = 0) -> V:
        ident = (i, field, k)
        return e.const(self.value(ident), ident, self.generic(ident))

    def placement(self, e: Emitter, i: int):
        R = [[self.c(e, i, "R", 3 * r + c) for c in range(3)] for r in range(3)]
        t = [self.c(e, i, "t", k) for k in range(3)]
        return R, t

    def inertia(self, e: Emitter, i: int):
        return self.c(e, i, "m"), [self.c(e, i, "c", k) for k in range(3)], [self.c(e, i, "I", k) for k in range(6)]

    def gravity(self, e: Emitter):
        return [self.c(e, 0, "g", k) for k in range(3)]

    def signature(self) -> str:
        """What a model must share with the one this kernel was generated for: the tree, the joint
        kinds, the rows of q / v each joint owns (the bodies hard-code them) and, for every constant
        that is not generic in this variant, whether it is exactly 0 / +1 / -1."""
        head = "n%d;p%s;j%s;q%s;v%s;" % (self.n, ",".join(map(str, self.parents[1:])), ",".join(map(str, self.jtype[1:])),
                                         ",".join(map(str, self.idx_q[1:])), ",".join(map(str, self.idx_v[1:])))
        pat = []
        for ident in all_const_idents(self):
            x = self.value(ident)
            ch = "0" if x == 0.0 else ("1" if x == 1.0 else ("m" if x == -1.0 else "g"))
            pat.append("g" if self.generic(ident) else ch)
        return head + "".join(pat)


def all_const_idents(mv: "ModelView"):
    """Canonical order of every model constant (the host recomputes the pattern in this order)."""
    out = [(0, "g", k) for k in range(3)]
    for i in range(1, mv.n):
        out += [(i, "R", k) for k in range(9)] + [(i, "t", k) for k in range(3)] + [(i, "m", 0)]
        out += [(i, "c", k) for k in range(3)] + [(i, "I", k) for k in range(6)]
        out += [(i, "axis", k) for k in range(3)]
        out += [(i, "arm", mv.idx_v[i] + k) for k in range(mv.nv(i))]
    return out


def visit_program(mv: ModelView) -> List[Tuple[str, int]]:
    """Depth-first visits: ('F', i) forward, ('B', i) backward (leaves: backward follows at once).
    Children in DECREASING index order so child->parent sums accumulate in the order of the
    reference's `for (i = njoints-1; i > 0; --i)` loops (rnea.hxx:152-155, aba.hxx:276-279)."""
    out: List[Tuple[str, int]] = []

    def rec(i: int):
        out.append(("F", i))
        for c in sorted(mv.children[i], reverse=True):
            rec(c)
        out.append(("B", i))

    for r in sorted(mv.children[0], reverse=True):
        rec(r)
    return out


def forward_program(mv: ModelView) -> List[int]:
    return [i for k, i in visit_program(mv) if k == "F"]


def joint_q(e: Emitter, mv: ModelView, i: int) -> List[V]:
    return [e.inp("Q", mv.idx_q[i] + k) for k in range(M.JOINT_NQ[mv.jtype[i]])]


def axis_rotation(e: Emitter, ax: List[V], s: V, c: V) -> A.Mat3:
    """Rodrigues rotation about a unit axis (joint-revolute-unaligned.hpp:667-708 uses
    Eigen::AngleAxis(theta, axis).toRotationMatrix(): c 1 + s [a]x + (1-c) a a^T)."""
    omc = e.sub(ONE, c)
    R = [[ZERO] * 3 for _ in range(3)]
    for r in range(3):
        for k in range(3):
            base = e.mul(e.mul(omc, ax[r]), ax[k])
            if r == k:
                R[r][k] = e.add(base, c)
            else:
                # [a]x = [[0,-az,ay],[az,0,-ax],[-ay,ax,0]]
                j = 3 - r - k
                sign_pos = (r, k) in ((0, 2), (1, 0), (2, 1))
                sa = e.mul(s, ax[j])
                R[r][k] = e.add(base, sa if sign_pos else -sa)
    return R


def joint_transform(e: Emitter, mv: ModelView, i: int, sc=None):
    """liMi = placement * M(q) as (R, t), plus the (s, c) pair of revolute kinds (or None).
    ``sc``: reuse a known (sin, cos) instead of evaluating sincos again."""
    jt = mv.jtype[i]
    P, pt = mv.placement(e, i)
    if jt in REV or jt in RUB:
        k = jt - M.JOINT_RX if jt in REV else jt - M.JOINT_RUBX
        if sc is None:
            if jt in REV:
                sc = e.sincos(e.inp("Q", mv.idx_q[i]))
            else:  # joint-revolute-unbounded.hpp:153-162: q = (cos, sin)
                sc = (e.inp("Q", mv.idx_q[i] + 1), e.inp("Q", mv.idx_q[i]))
        s, c = sc
        ka, kb = (k + 1) % 3, (k + 2) % 3
        # joint-revolute.hpp:132-166: column k unchanged, column ka = c*Pa + s*Pb, column kb by cross
        ck = A.col(P, k)
        na = [e.fma(c, P[r][ka], e.mul(s, P[r][kb])) for r in range(3)]
        nb = A.cross(e, ck, na)
        R = [[ZERO] * 3 for _ in range(3)]
        for r in range(3):
            R[r][k], R[r][ka], R[r][kb] = ck[r], na[r], nb[r]
        return R, pt, sc
    if jt in PRI:
        k = jt - M.JOINT_PX
        q = e.inp("Q", mv.idx_q[i])
        return P, [e.fma(q, P[r][k], pt[r]) for r in range(3)], None
    if jt in (M.JOINT_REVOLUTE_UNALIGNED, M.JOINT_REVOLUTE_UNBOUNDED_UNALIGNED):
        if sc is None:
            if jt == M.JOINT_REVOLUTE_UNALIGNED:
                sc = e.sincos(e.inp("Q", mv.idx_q[i]))
            else:  # joint-revolute-unbounded-unaligned.hpp:190-200: q = (cos, sin)
                sc = (e.inp("Q", mv.idx_q[i] + 1), e.inp("Q", mv.idx_q[i]))
        ax = [mv.c(e, i, "axis", k) for k in range(3)]
        return A.rot_mul(e, P, axis_rotation(e, ax, sc[0], sc[1])), pt, sc
    if jt == M.JOINT_PRISMATIC_UNALIGNED:
        q = e.inp("Q", mv.idx_q[i])
        ax = [mv.c(e, i, "axis", k) for k in range(3)]
        Pa = A.rot_vec(e, P, ax)
        return P, [e.fma(q, Pa[r], pt[r]) for r in range(3)], None
    if jt == M.JOINT_FREEFLYER:
        q = joint_q(e, mv, i)
        Rq = A.quat_to_R(e, q[3], q[4], q[5], q[6])
        return A.rot_mul(e, P, Rq), A.rot_vec(e, P, q[0:3], pt), None
    if jt == M.JOINT_SPHERICAL:
        q = joint_q(e, mv, i)
        Rq = A.quat_to_R(e, q[0], q[1], q[2], q[3])
        return A.rot_mul(e, P, Rq), pt, None
    raise ValueError("joint kind %d is not supported by the model compiler" % jt)


def subspace(e: Emitter, mv: ModelView, i: int) -> List[List[V]]:
    """Columns of S in the joint frame, each a 6-vector [lin; ang]."""
    jt = mv.jtype[i]

    def unit(k):
        return [ONE if j == k else ZERO for j in range(6)]

    if jt in REV:
        return [unit(3 + jt - M.JOINT_RX)]
    if jt in RUB:
        return [unit(3 + jt - M.JOINT_RUBX)]
    if jt in PRI:
        return [unit(jt - M.JOINT_PX)]
    if jt in (M.JOINT_REVOLUTE_UNALIGNED, M.JOINT_REVOLUTE_UNBOUNDED_UNALIGNED):
        return [[ZERO] * 3 + [mv.c(e, i, "axis", k) for k in range(3)]]
    if jt == M.JOINT_PRISMATIC_UNALIGNED:
        return [[mv.c(e, i, "axis", k) for k in range(3)] + [ZERO] * 3]
    if jt == M.JOINT_FREEFLYER:
        return [unit(k) for k in range(6)]
    if jt == M.JOINT_SPHERICAL:
        return [unit(3 + k) for k in range(3)]
    raise ValueError(jt)


def s_times(e: Emitter, S: List[List[V]], x: List[V]) -> List[V]:
    """S x (6-vector)"""
    out = [ZERO] * 6
    for col, xk in zip(S, x):
        out = [e.fma(col[j], xk, out[j]) for j in range(6)]
    return out


def st_times(e: Emitter, S: List[List[V]], f: List[V]) -> List[V]:
    """S^T f (nv-vector)"""
    return [e.dot(col, f) for col in S]
