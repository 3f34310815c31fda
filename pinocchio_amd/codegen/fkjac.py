"""Model compiler: forwardKinematics(q) + computeJointJacobians(q)
(algorithm/kinematics.hxx:52-97, algorithm/jacobian.hxx:38-95) as straight-line code.

One forward sweep: liMi, oMi = oMparent * liMi, J[:, cols_i] = oMi.act(S_i) in the WORLD frame.
Outputs: OUT = J (per instance 6 x nv column-major), OUT2 = oMi (per joint rotation column-major
then translation; joint 0 = identity), the layout of include/brbd.h.

Joints are visited in index order (parents come first) and every output is emitted in increasing
element order: the INSTANCE_MAJOR kernels (layout 2) collect 64 rows at a time and write them out
through LDS as soon as a chunk is complete (csrc/brbd_spec.hpp: SpecOutStream).  ``monotone(mv)``
tells whether a model's dof numbering allows that (J rows follow idx_v).
"""
from __future__ import annotations

from . import algebra as A
from . import joints as J
from .emit import ONE, ZERO, Emitter


def monotone(mv: J.ModelView) -> bool:
    return all(mv.parents[i] < i for i in range(1, mv.n)) and all(
        mv.idx_v[i + 1] == mv.idx_v[i] + mv.nv(i) for i in range(1, mv.n - 1))


def generate(mv: J.ModelView, e: Emitter) -> Emitter:
    kin = {}
    first = True
    order = list(range(1, mv.n)) if monotone(mv) else J.forward_program(mv)
    for i in order:
        p = mv.parents[i]
        e.begin_visit("FK %d" % i)
        if first:
            for k in range(12):  # joint 0 = identity
                e.out(k, ONE if k in (0, 4, 8) else ZERO, which=2)
            first = False
        Rl, tl, _ = J.joint_transform(e, mv, i)
        if p == 0:  # kinematics.hxx:66-69
            R, t = Rl, tl
        else:
            vals = e.unpark(kin[p])
            Rp, tp = [vals[0:3], vals[3:6], vals[6:9]], vals[9:12]
            R, t = A.rot_mul(e, Rp, Rl), A.rot_vec(e, Rp, tl, tp)
        if mv.children[i]:
            kin[i] = e.park([x for row in R for x in row] + list(t), "stack")
        for c in range(3):
            for r in range(3):
                e.out(12 * i + 3 * c + r, R[r][c], which=2)
        for k in range(3):
            e.out(12 * i + 9 + k, t[k], which=2)
        S = J.subspace(e, mv, i)
        for k, col in enumerate(S):  # jacobian.hxx:59
            Jc = A.act_motion(e, R, t, col)
            for r in range(6):
                e.out(6 * (mv.idx_v[i] + k) + r, Jc[r])
    return e
