"""Model compiler: computeRNEADerivatives (algorithm/rnea-derivatives.hxx:263-541) as straight-line
code: dtau/dq (OUT), dtau/dv (OUT2), dtau/da = M upper triangle (OUT3), tau (OUT4); each matrix is
nv x nv column-major per instance (element (row, col) at col * nv + row), every element written.

World-frame recursion like the reference.  Forward visit of joint i (:263-338): oMi, J = oMi.act(S),
ov, oa_gf (world form of :285-305: oa_gf_i = oa_gf_p + J a + ov_i x (J v)), and per dof the columns
dVdq = ov_p x J, dAdq = oa_gf_p x J + ov_p x dVdq, dAdv = ov_i x J + dVdq (:319-331).  Backward visit
(:378-459): composite oYcrb as the 10 numbers of a rigid-body inertia about the world origin (m,
h = m c, I_o: additive over the subtree), doYcrb = oYcrb.variation(ov) + forceCrossMatrix(oh)
(:335-337) as the 9 numbers of the variation (dh, dI: it keeps the block structure) plus the
composite oh, of; then tau = J^T of, dFda = Y J, dFdq = doY dVdq + Y dAdq (+ J x* of), dFdv = doY J + Y dAdv,
dYtJ = J^T doY.

The reference fills, at an ANCESTOR's backward visit, the blocks that couple it with every dof of
its subtree, which keeps 24 doubles per dof of the whole subtree alive until the root.  Here the
same dot products are evaluated from the DESCENDANT's side, streaming over the joints on its root
path: what stays alive is the path, not the subtree.  The pair visits of a chain of joints whose
backward visits follow each other (a limb on the way up) are deferred and done together, ancestor
by ancestor (at most CHAIN dofs at a time): the ancestor's columns are loaded once per group, and
the composite quantities pass from child to parent in registers.  The columns of a joint attached
to the universe are recomputed from its placement and velocity (18 doubles) instead of parked
(24 per dof): with the output matrices, the parked state is what the kernel's HBM time is made of.
"""
from __future__ import annotations

import os

from . import algebra as A
from . import joints as J
from .emit import ZERO, Emitter, V


CHAIN = int(os.environ.get("BRBD_CG_CHAIN", "3"))  # dofs per deferred group of pair visits


def generate(mv: J.ModelView, e: Emitter) -> Emitter:
    nvt = mv.model.nv
    written = {1: set(), 2: set(), 3: set()}

    def put(which, row, col, val):
        e.out(col * nvt + row, val, which)
        written[which].add((row, col))

    kin = {}      # joint -> Park R(9) t(3) ov(6) [oa_gf(6) in kin_oa for joints attached to the universe]
    kin_oa = {}
    vec = {}      # joint -> Park per dof: J, dVdq, dAdq, dAdv (24 per dof); none for universe children
    deferred = [] # (joint, Park of its own dFda / dYtJ / dFdq / dFdv columns), deepest first

    def root_col(i, R, t, ov, k):
        """columns of dof k of a joint attached to the universe (:331: dVdq = 0, oa_gf[0] = -gravity)"""
        g = mv.gravity(e)
        oap = [-g[0], -g[1], -g[2], ZERO, ZERO, ZERO]
        Jk = A.act_motion(e, R, t, J.subspace(e, mv, i)[k])
        return (Jk, A.zeros(6), A.motion_cross(e, oap, Jk), A.motion_cross(e, ov, Jk))

    def pair(iv_k, o, iva_r, ca):
        fa, yt, fq, fv = (o[6 * j:6 * j + 6] for j in range(4))
        Ja, dVa, dAqa, dAva = ca
        put(3, iva_r, iv_k, A.dot6(e, Ja, fa))                                   # :420-421
        put(1, iv_k, iva_r, A.dot6(e, yt, dVa, A.dot6(e, fa, dAqa)))             # :433-435
        put(1, iva_r, iv_k, A.dot6(e, Ja, fq))                                   # :437-438
        put(2, iv_k, iva_r, A.dot6(e, yt, Ja, A.dot6(e, fa, dAva)))              # :446-448
        put(2, iva_r, iv_k, A.dot6(e, Ja, fv))                                   # :450-451

    def flush():
        """the blocks that couple the deferred joints with every joint on their root path"""
        group = list(deferred)
        del deferred[:]
        if not group:
            return
        members = [d for d, _ in group]
        names = "+".join(str(d) for d in members)
        a = mv.parents[members[0]]
        while a > 0:
            below = group[:members.index(a)] if a in members else group
            nva, iva = mv.nv(a), mv.idx_v[a]
            if mv.parents[a] == 0:
                e.begin_visit("PAIR %s x %d" % (names, a))
                st = e.unpark(kin[a])
                owns = [(d, e.unpark(own)) for d, own in below]
                for r in range(nva):
                    ca = root_col(a, [st[0:3], st[3:6], st[6:9]], st[9:12], st[12:18], r)
                    for d, o in owns:
                        for k in range(mv.nv(d)):
                            pair(mv.idx_v[d] + k, o[24 * k:24 * k + 24], iva + r, ca)
            else:
                for r in range(nva):
                    e.begin_visit("PAIR %s x %d%s" % (names, a, "" if nva == 1 else ", dof %d" % r))
                    va = e.unpark(vec[a][r])
                    ca = tuple(va[6 * j:6 * j + 6] for j in range(4))
                    for d, own in below:
                        o = e.unpark(own)
                        for k in range(mv.nv(d)):
                            pair(mv.idx_v[d] + k, o[24 * k:24 * k + 24], iva + r, ca)
            a = mv.parents[a]
    acc = {}      # joint -> Park of the finished children's composite m, h, I_o (oYcrb), dh, dI (doYcrb
                  # without its force-cross part), oh, of: 31 doubles

    def backward(i, R, t, ov, oa_gf, Jc, cols):
        """Jc: the joint's world-frame columns; cols(k) -> (J, dVdq, dAdq, dAdv) of dof k"""
        p = mv.parents[i]
        nv, iv = mv.nv(i), mv.idx_v[i]
        m, c, I = mv.inertia(e, i)
        cw, Iw = A.inertia_to_world(e, R, t, c, I)            # :302
        h, Io = A.inertia_origin(e, m, cw, Iw)
        oh = A.rbi_mul(e, m, h, Io, ov)                        # :307
        of = A.vadd(e, A.rbi_mul(e, m, h, Io, oa_gf), A.force_cross(e, ov, oh))  # :308
        dh, dI = A.rbi_variation(e, m, h, Io, ov)              # :335 (of the body's own inertia)
        if i in acc:  # children's composite quantities (:453-458)
            vals = e.unpark(acc.pop(i))
            m, h, Io = e.add(m, vals[0]), A.vadd(e, h, vals[1:4]), A.vadd(e, Io, vals[4:10])
            dh, dI = A.vadd(e, dh, vals[10:13]), A.vadd(e, dI, vals[13:19])
            oh, of = A.vadd(e, oh, vals[19:25]), A.vadd(e, of, vals[25:31])
        # :337 doYcrb += forceCrossMatrix(oh): not symmetric, carried as oh itself
        hp, hm, hn = A.vadd(e, dh, oh[0:3]), A.vsub(e, dh, oh[0:3]), oh[3:6]

        def Ymul(x):
            return A.rbi_mul(e, m, h, Io, x)

        def dYmul(x, transpose=False):
            return A.rbi_dot_mul(e, hp, hm, dI, hn, x, transpose)

        # own columns, one dof at a time (a free-flyer's six columns of dFda / dFdq / dFdv at once
        # would not fit the register file); the values above stay in registers across the sub-visits
        own_vals = []
        for k in range(nv):
            if k > 0:
                e.begin_visit("BWD %d, dof %d" % (i, k))
            Jk, dVk, dAqk, dAvk = cols(k)
            dFda = Ymul(Jk)                                                            # :419
            if p > 0:                                                                  # :424-430
                dFdq = A.vadd(e, dYmul(dVk), Ymul(dAqk))
            else:
                dFdq = Ymul(dAqk)
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
            dFdv = A.vadd(e, dYmul(Jk), Ymul(dAvk))                                    # :443-444
            for r in range(nv):
                put(2, iv + r, iv + k, A.dot6(e, Jc[r], dFdv))                         # :450-451
            if p > 0:
                dYtJ = dYmul(Jk, transpose=True)                                       # :432
                own_vals += dFda + dYtJ + dFdq + dFdv
        # hand the composite quantities to the parent (:453-458); in registers when the
        # parent's backward visit is the next one
        if p > 0:
            up = [m] + list(h) + list(Io) + list(dh) + list(dI) + list(oh) + list(of)
            if p in acc:
                up = A.vadd(e, e.unpark(acc[p]), up)
            # (the running sum of a joint attached to the universe lives across whole limbs: "hand")
            acc[p] = e.park(up, "stack" if mv.parents[p] > 0 else "hand")
        if p > 0:
            deferred.append((i, e.park(own_vals, "tail")))

    program = list(J.visit_program(mv))
    for step, (kind, i) in enumerate(program):
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
                Rp, tp, ovp = [st[0:3], st[3:6], st[6:9]], st[9:12], st[12:18]
                oap = st[18:24] if mv.parents[p] > 0 else e.unpark(kin_oa[p])
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
            elif p > 0:
                vec[i] = [e.park([x for vv in cset for x in vv], "stack") for cset in cols]
                kin[i] = e.park([x for row in R for x in row] + list(t) + ov + oa, "stack")
            else:
                kin[i] = e.park([x for row in R for x in row] + list(t) + ov, "stack")
                kin_oa[i] = e.park(oa, "stack")
        elif not leaf:
            if deferred and sum(mv.nv(d) for d, _ in deferred) + nv > CHAIN:
                flush()
            e.begin_visit("BWD %d" % i)
            st = e.unpark(kin[i])
            R, t, ov = [st[0:3], st[3:6], st[6:9]], st[9:12], st[12:18]
            if p > 0:  # (ancestors of later joints read these parks too)
                vv = [e.unpark(pk) for pk in vec[i]]
                backward(i, R, t, ov, st[18:24], [x[0:6] for x in vv],
                         lambda k: tuple(vv[k][6 * j:6 * j + 6] for j in range(4)))
            else:
                Jroot = [A.act_motion(e, R, t, col) for col in J.subspace(e, mv, i)]
                backward(i, R, t, ov, e.unpark(kin_oa[i]), Jroot, lambda k: root_col(i, R, t, ov, k))
        # pair visits of the joints whose backward visits have run, once the walk turns downwards
        # again (or the group is full)
        nxt = program[step + 1][0] if step + 1 < len(program) else "F"
        if deferred and (nxt == "F" or sum(mv.nv(d) for d, _ in deferred) >= CHAIN):
            flush()
    # everything else is structurally zero (unrelated branches; lower triangle of dtau/da): spread
    # over the whole body (Emitter.out_filler)
    for which in (1, 2, 3):
        for col in range(nvt):
            for row in range(nvt):
                if (row, col) not in written[which]:
                    e.out_filler(col * nvt + row, ZERO, which)
    return e
