"""Model compiler: RNEA (algorithm/rnea.hxx:45-161) as straight-line code for one model.

Depth-first walk.  Forward visit of joint i (rnea.hxx:45-79): liMi, v_i, a_gf_i,
f_i = Y a_gf + v x* (Y v).  Backward visit (rnea.hxx:92-107): tau_i = S^T f_i (+ armature * a,
:158), f_parent += liMi.act(f_i).  What has to survive the walk of a subtree is parked (emit.py):
f_i with the (sin, cos) of the joint (its rotation is rebuilt from them, cheaper than parking 9
doubles), (v, a_gf) of joints with children, and the contributions of finished children.  A park
that is consumed by the very next visit never leaves the registers, so chains hand v / a_gf down
and f up for free.  A leaf's backward visit is fused into its forward visit.
"""
from __future__ import annotations

from . import algebra as A
from . import joints as J
from .emit import ZERO, Emitter


def generate(mv: J.ModelView, e: Emitter) -> Emitter:
    va = {}       # joint -> Park of v(6) + a_gf(6)
    fpark = {}    # joint -> Park of f(6) + (s, c) | f(6) + R(9) + t(3)
    contrib = {}  # joint -> Park of the sum of the finished children's liMi.act(f)

    def backward(i, f, R, t):
        S = J.subspace(e, mv, i)
        tau = J.st_times(e, S, f)  # rnea.hxx:103
        iv = mv.idx_v[i]
        for k in range(mv.nv(i)):
            arm = mv.c(e, i, "arm", iv + k)
            e.out(iv + k, e.fma(arm, e.inp("X", iv + k), tau[k]) if not arm.is_zero() else tau[k])  # :158
        p = mv.parents[i]
        if p != 0:  # :105-106
            up = A.act_force(e, R, t, f)
            if p in contrib:  # running sum of the finished children
                up = A.vadd(e, e.unpark(contrib[p]), up)
            contrib[p] = e.park(up, "stack")

    for kind, i in J.visit_program(mv):
        p = mv.parents[i]
        leaf = not mv.children[i]
        if kind == "F":
            e.begin_visit("FWD %d%s" % (i, " + BWD (leaf)" if leaf else ""))
            if p == 0:  # rnea.hxx:136-138: v[0] = 0, a_gf[0] = -gravity
                g = mv.gravity(e)
                vp, ap = A.zeros(6), [-g[0], -g[1], -g[2], ZERO, ZERO, ZERO]
            else:
                st = e.unpark(va[p])
                vp, ap = st[:6], st[6:]
            R, t, sc = J.joint_transform(e, mv, i)
            S = J.subspace(e, mv, i)
            iv = mv.idx_v[i]
            vJ = J.s_times(e, S, [e.inp("V", iv + k) for k in range(mv.nv(i))])
            Sa = J.s_times(e, S, [e.inp("X", iv + k) for k in range(mv.nv(i))])
            # :63-65 velocity propagation is guarded by parent > 0
            v = vJ if p == 0 else A.vadd(e, A.actinv_motion(e, R, t, vp), vJ)
            # :67-69 a_gf = c (= 0) + v x vJ + S a + liMi.actInv(a_gf[parent])   (not guarded)
            a = A.vadd(e, A.vadd(e, A.motion_cross(e, v, vJ), Sa), A.actinv_motion(e, R, t, ap))
            m, c, I = mv.inertia(e, i)
            h = A.inertia_mul(e, m, c, I, v)                                      # :73
            f = A.vadd(e, A.inertia_mul(e, m, c, I, a), A.force_cross(e, v, h))  # :74-75
            # rnea.hxx:205 (kernel variants with external forces): f[i] -= fext[i], joint frame
            f = e.cond_sub(f, lambda: [e.inp("F", 6 * i + k) for k in range(6)])
            if leaf:
                backward(i, f, R, t)
            else:
                va[i] = e.park(v + a, "stack")
                keep = list(sc) if sc is not None else [x for row in R for x in row] + list(t)
                fpark[i] = e.park(f + keep, "stack")
        elif not leaf:
            e.begin_visit("BWD %d" % i)
            keep = e.unpark(fpark.pop(i))
            f = keep[:6]
            if len(keep) == 8:
                R, t, _ = J.joint_transform(e, mv, i, sc=(keep[6], keep[7]))
            else:
                R, t = [keep[6:9], keep[9:12], keep[12:15]], keep[15:18]
            if i in contrib:
                f = A.vadd(e, f, e.unpark(contrib.pop(i)))
            backward(i, f, R, t)
    return e
