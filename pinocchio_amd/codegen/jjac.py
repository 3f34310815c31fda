"""Model compiler: computeJointJacobians(q) + getJointJacobian(joint, rf, J)
(algorithm/jacobian.hxx:38-95, 175-283, 315-326) as straight-line code, for ANY joint of the model.

The joint is a run-time argument: the host passes the support chain of the joint (itself and its
ancestors) as a bit mask, and every visit of the body sits in a wave-uniform `if (ON_CHAIN(i))`;
joints off the chain cost one scalar branch.  Joints are visited in index order, so the parent of a
chain joint is the chain joint visited before it and ONE world placement is updated in place.
The reference frame is a compile-time variant (WANT: 1 WORLD, 3 LOCAL, 5 LOCAL_WORLD_ALIGNED):

* WORLD: one walk; the column of a chain joint is oMi.act(S) (jacobian.hxx:59, :233-247 copies it).
* LOCAL / LOCAL_WORLD_ALIGNED need the placement oMj of the joint itself before the first column
  can be written: a first walk over the chain computes it (forward kinematics only, ~50
  instructions per chain joint), the second walk writes oMj.actInv(column) (:262-276) or
  column.linear -= oMj.translation x column.angular (:248-261).
* Every column off the chain is written as zero (the reference leaves them to the caller's setZero()).

All 6 nv rows are emitted in increasing order in both branches, so the INSTANCE_MAJOR kernels
stream them out through LDS 64 rows at a time like codegen/fkjac.py.
"""
from __future__ import annotations

from . import algebra as A
from . import joints as J
from .emit import ZERO, Emitter, V


def _state(prefix: str):
    R = [[V("%sR%d" % (prefix, 3 * r + c)) for c in range(3)] for r in range(3)]
    t = [V("%st%d" % (prefix, k)) for k in range(3)]
    return R, t


def _assign(e: Emitter, R, t, Rn, tn):
    for r in range(3):
        for c in range(3):
            e.stmt("%s = %s;" % (R[r][c].name, Rn[r][c].code()))
    for k in range(3):
        e.stmt("%s = %s;" % (t[k].name, tn[k].code()))


def generate(mv: J.ModelView, e: Emitter) -> Emitter:
    jR, jt = _state("jj")   # world placement of the joint itself (after the first walk)
    cR, ct = _state("jc")   # world placement of the chain joint the walk stands on
    order = list(range(1, mv.n))
    assert all(mv.parents[i] < i for i in order)

    e.begin_visit("JJ init")
    names = [x.name for row in jR for x in row] + [x.name for x in jt] + [x.name for row in cR for x in row] + [x.name for x in ct]
    e.stmt("double " + ", ".join("%s = 0.0" % n for n in names) + ";")

    # first walk: oMj (variants that express the Jacobian relative to the joint's own placement)
    for i in order:
        e.begin_visit("JJ walk1 %d" % i)
        with e.block("(WANT & 6) && ON_CHAIN(%d)" % i):
            Rl, tl, _ = J.joint_transform(e, mv, i)
            if mv.parents[i] == 0:
                R, t = Rl, tl
            else:
                R, t = A.rot_mul(e, jR, Rl), A.rot_vec(e, jR, tl, jt)
            _assign(e, jR, jt, R, t)

    # second walk: the columns
    for i in order:
        e.begin_visit("JJ walk2 %d" % i)
        iv, nv = mv.idx_v[i], mv.nv(i)
        with e.block("ON_CHAIN(%d)" % i):
            Rl, tl, _ = J.joint_transform(e, mv, i)
            if mv.parents[i] == 0:
                R, t = Rl, tl
            else:
                R, t = A.rot_mul(e, cR, Rl), A.rot_vec(e, cR, tl, ct)
            _assign(e, cR, ct, R, t)
            cols = [A.act_motion(e, R, t, col) for col in J.subspace(e, mv, i)]   # jacobian.hxx:59
            with e.block("(WANT & 6) == 0"):   # WORLD (:233-247)
                for k, Jc in enumerate(cols):
                    for r in range(6):
                        e.out(6 * (iv + k) + r, Jc[r])
            with e.block("WANT & 2"):          # LOCAL (:262-276): oMj.actInv(column)
                for k, Jc in enumerate(cols):
                    Jl = A.actinv_motion(e, jR, jt, Jc)
                    for r in range(6):
                        e.out(6 * (iv + k) + r, Jl[r])
            with e.block("WANT & 4"):          # LOCAL_WORLD_ALIGNED (:248-261): lin -= oMj.t x ang
                for k, Jc in enumerate(cols):
                    lin = A.cross_add(e, Jc[3:6], jt, Jc[0:3])
                    for r in range(6):
                        e.out(6 * (iv + k) + r, lin[r] if r < 3 else Jc[r])
        e.stmt("else {")
        for k in range(nv):
            for r in range(6):
                e.out(6 * (iv + k) + r, ZERO)
        e.stmt("}")
    return e
