"""Model compiler: computeRNEADerivatives (algorithm/rnea-derivatives.hxx:263-541) as straight-line
code: dtau/dq (OUT), dtau/dv (OUT2), dtau/da = M upper triangle (OUT3), tau (OUT4); each matrix is
nv x nv column-major per instance (element (row, col) at col * nv + row), every element written.

World-frame recursion like the reference.  Forward visit of joint i (:263-338): oMi, J = oMi.act(S),
ov, oa_gf (world form of :285-305: oa_gf_i = oa_gf_p + J a + ov_i x (J v)), and per dof the columns
dVdq = ov_p x J, dAdq = oa_gf_p x J + ov_p x dVdq, dAdv = ov_i x J + dVdq (:319-331).  Backward visit
(:378-459): composite oYcrb (packed symmetric 6x6, additive in the world frame), doYcrb =
oYcrb.variation(ov) + forceCrossMatrix(oh) (:335-337, written as -(P + P^T) with P = Y crm(ov)),
of; then tau = J^T of, dFda = Y J, dFdq = doY dVdq + Y dAdq (+ J x* of), dFdv = doY J + Y dAdv,
dYtJ = J^T doY.

The reference fills, at an ANCESTOR's backward visit, the blocks that couple it with every dof of
its subtree, which keeps 24 doubles per dof of the whole subtree alive until the root.  Here the
same dot products are evaluated at the DESCENDANT's backward visit, streaming over the joints on
its root path (one short visit per ancestor): what stays alive is the path, not the subtree.
"""
from __future__ import annotations

from . import algebra as A
from . import joints as J
from .emit import ZERO, Emitter, V


def crm_col(e: Emitter, v, j: int):
    """column j of the motion cross matrix of v: v x e_j"""
    unit = [V(num=1.0) if k == j else ZERO for k in range(6)]
    return A.motion_cross(e, v, unit)


def generate(mv: J.ModelView, e: Emitter) -> Emitter:
    nvt = mv.model.nv
    written = {1: set(), 2: set(), 3: set()}

    def put(which, row, col, val):
        e.out(col * nvt + row, val, which)
        written[which].add((row, col))

    kin = {}      # joint -> Park R(9) t(3) ov(6) oa_gf(6)
    vec = {}      # joint -> Park per dof: J, dVdq, dAdq, dAdv (24 per dof)
    acc = {}      # joint -> Park of the finished children's oYcrb (21) + doYcrb (36) + of (6)

    def mat6_mul(Mx, x):
        return [e.dot(Mx[r], x) for r in range(6)]

    def backward(i, R, t, ov, oa_gf, Jc, cols):
        """Jc: the joint's world-frame columns; cols(k) -> (J, dVdq, dAdq, dAdv) of dof k"""
        p = mv.parents[i]
        nv, iv = mv.nv(i), mv.idx_v[i]
        m, c, I = mv.inertia(e, i)
        cw, Iw = A.inertia_to_world(e, R, t, c, I)            # :302
        Y = A.inertia_to_sym6(e, m, cw, Iw)
        oh = A.sym6_mulvec(e, Y, ov)                           # :307
        of = A.vadd(e, A.sym6_mulvec(e, Y, oa_gf), A.force_cross(e, ov, oh))  # :308
        # :335 oYcrb.variation(ov) = crf(ov) Y - Y crm(ov) = -(P^T + P), P = Y crm(ov)
        P = [A.sym6_mulvec(e, Y, crm_col(e, ov, j)) for j in range(6)]   # P[j] = column j
        dY = [[-e.add(P[cc][r], P[r][cc]) for cc in range(6)] for r in range(6)]
        # :337 addForceCrossMatrix(oh): skew(-f.lin) on the LA and AL blocks, skew(-f.ang) on AA
        def skew(v3):
            return [[ZERO, -v3[2], v3[1]], [v3[2], ZERO, -v3[0]], [-v3[1], v3[0], ZERO]]
        sl, sa = skew([-x for x in oh[0:3]]), skew([-x for x in oh[3:6]])
        for r in range(3):
            for cc in range(3):
                dY[r][3 + cc] = e.add(dY[r][3 + cc], sl[r][cc])
                dY[3 + r][cc] = e.add(dY[3 + r][cc], sl[r][cc])
                dY[3 + r][3 + cc] = e.add(dY[3 + r][3 + cc], sa[r][cc])
        if i in acc:  # children's composite quantities (:453-458)
            vals = e.unpark(acc.pop(i))
            Y = A.vadd(e, Y, vals[0:21])
            dY = [[e.add(dY[r][cc], vals[21 + 6 * r + cc]) for cc in range(6)] for r in range(6)]
            of = A.vadd(e, of, vals[57:63])
        # hand the composite quantities to the parent (:453-458)
        if p > 0:
            up = list(Y) + [dY[r][cc] for r in range(6) for cc in range(6)] + list(of)
            if p in acc:
                up = A.vadd(e, e.unpark(acc[p]), up)
            acc[p] = e.park(up, "stack")
        # own columns, one dof at a time (a free-flyer's six columns of dFda / dFdq / dFdv at once
        # would not fit the register file); the values above stay in registers across the sub-visits
        own_vals = []
        for k in range(nv):
            if k > 0:
                e.begin_visit("BWD %d, dof %d" % (i, k))
            Jk, dVk, dAqk, dAvk = cols(k)
            dFda = A.sym6_mulvec(e, Y, Jk)                                             # :419
            if p > 0:                                                                  # :424-430
                dFdq = A.vadd(e, mat6_mul(dY, dVk), A.sym6_mulvec(e, Y, dAqk))
            else:
                dFdq = A.sym6_mulvec(e, Y, dAqk)
            tau = A.dot6(e, Jk, of)                                                    # :409-410
            arm = mv.c(e, i, "arm", iv + k)
            if not arm.is_zero():
                tau = e.fma(arm, e.inp("X", iv + k), tau)                              # :538-540
            e.out(iv + k, tau, 4)
            for r in range(nv):
                mrk = A.dot6(e, Jc[r], dFda)                                           # :420-421
                if r == k:
                    mrk = e.add(mrk, arm)
                put(3, iv + r, iv + k, mrk)
                put(1, iv + r, iv + k, A.dot6(e, Jc[r], dFdq))                         # :437-438
            dFdq = A.vadd(e, dFdq, A.force_cross(e, Jk, of))                           # :440
            dFdv = A.vadd(e, mat6_mul(dY, Jk), A.sym6_mulvec(e, Y, dAvk))              # :443-444
            for r in range(nv):
                put(2, iv + r, iv + k, A.dot6(e, Jc[r], dFdv))                         # :450-451
            if p > 0:
                dYtJ = [e.dot(Jk, [dY[r][cc] for r in range(6)]) for cc in range(6)]   # :432
                own_vals += dFda + dYtJ + dFdq + dFdv
        # the blocks that couple this joint with every joint on its root path
        a = p
        own = None
        while a > 0:
            if own is None:
                # keep the own columns across the ancestor visits
                own = e.park(own_vals, "tail")
            nva, iva = mv.nv(a), mv.idx_v[a]
            for r in range(nva):  # (one visit per dof of the ancestor: a free-flyer's 6 x 24 values)
                e.begin_visit("PAIR %d x %d%s" % (i, a, "" if nva == 1 else ", dof %d" % r))
                o = e.unpark(own)
                va = e.unpark(vec[a][r])
                Ja, dVa, dAqa, dAva = (va[6 * j:6 * j + 6] for j in range(4))
                for k in range(nv):
                    fa, yt, fq, fv = (o[24 * k + 6 * j:24 * k + 6 * j + 6] for j in range(4))
                    put(3, iva + r, iv + k, A.dot6(e, Ja, fa))                                   # :420-421
                    put(1, iv + k, iva + r, A.dot6(e, yt, dVa, A.dot6(e, fa, dAqa)))             # :433-435
                    put(1, iva + r, iv + k, A.dot6(e, Ja, fq))                                   # :437-438
                    put(2, iv + k, iva + r, A.dot6(e, yt, Ja, A.dot6(e, fa, dAva)))              # :446-448
                    put(2, iva + r, iv + k, A.dot6(e, Ja, fv))                                   # :450-451
            a = mv.parents[a]

    for kind, i in J.visit_program(mv):
        p = mv.parents[i]
        leaf = not mv.children[i]
        nv, iv = mv.nv(i), mv.idx_v[i]
        if kind == "F":
            e.begin_visit("FWD %d%s" % (i, " + BWD (leaf)" if leaf else ""))
            if p == 0:  # :504  oa_gf[0] = -gravity, ov[0] = 0
                g = mv.gravity(e)
                Rp = tp = None
                ovp, oap = A.zeros(6), [-g[0], -g[1], -g[2], ZERO, ZERO, ZERO]
            else:
                st = e.unpark(kin[p])
                Rp, tp, ovp, oap = [st[0:3], st[3:6], st[6:9]], st[9:12], st[12:18], st[18:24]
            Rl, tl, _ = J.joint_transform(e, mv, i)
            if p == 0:
                R, t = Rl, tl
            else:
                R, t = A.rot_mul(e, Rp, Rl), A.rot_vec(e, Rp, tl, tp)
            Jc = [A.act_motion(e, R, t, col) for col in J.subspace(e, mv, i)]  # :319
            vq = [e.inp("V", iv + k) for k in range(nv)]
            aq = [e.inp("X", iv + k) for k in range(nv)]
            ovJ = A.zeros(6)
            for k in range(nv):
                ovJ = A.vaxpy(e, vq[k], Jc[k], ovJ)
            ov = A.vadd(e, ovp, ovJ)
            oa = A.vadd(e, oap, A.motion_cross(e, ov, ovJ))
            for k in range(nv):
                oa = A.vaxpy(e, aq[k], Jc[k], oa)
            cols = []
            for k in range(nv):
                dJ = A.motion_cross(e, ov, Jc[k])                 # :320
                dAdq = A.motion_cross(e, oap, Jc[k])              # :321
                if p > 0:                                         # :323-328
                    dVdq = A.motion_cross(e, ovp, Jc[k])
                    dAdq = A.vadd(e, dAdq, A.motion_cross(e, ovp, dVdq))
                    dAdv = A.vadd(e, dJ, dVdq)
                else:
                    dVdq, dAdv = A.zeros(6), dJ                   # :331
                cols.append((Jc[k], dVdq, dAdq, dAdv))
            if leaf:
                backward(i, R, t, ov, oa, Jc, lambda k: cols[k])
            else:
                vec[i] = [e.park([x for vv in cset for x in vv], "stack") for cset in cols]
                kin[i] = e.park([x for row in R for x in row] + list(t) + ov + oa, "stack")
        elif not leaf:
            e.begin_visit("BWD %d" % i)
            st = e.unpark(kin[i])
            R, t, ov, oa = [st[0:3], st[3:6], st[6:9]], st[9:12], st[12:18], st[18:24]
            if p > 0:  # (ancestors of later joints read these parks too)
                vv = [e.unpark(pk) for pk in vec[i]]
                backward(i, R, t, ov, oa, [x[0:6] for x in vv],
                         lambda k: tuple(vv[k][6 * j:6 * j + 6] for j in range(4)))
            else:      # root: each dof's dVdq / dAdq / dAdv only in its own sub-visit
                def col(k):
                    x = e.unpark(vec[i][k])
                    return tuple(x[6 * j:6 * j + 6] for j in range(4))
                Jroot = [e.unpark(pk)[0:6] for pk in vec[i]]
                backward(i, R, t, ov, oa, Jroot, col)
    # everything else is structurally zero (unrelated branches; lower triangle of dtau/da)
    e.begin_visit("ZEROS")
    for which in (1, 2, 3):
        for col in range(nvt):
            for row in range(nvt):
                if (row, col) not in written[which]:
                    e.out(col * nvt + row, ZERO, which)
    return e
