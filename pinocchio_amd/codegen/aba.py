"""Model compiler: ABA, WORLD convention (algorithm/aba.hxx:102-294) as straight-line code.

Per tree hanging off the universe: passes 1 + 2 fused in one depth-first walk, then pass 3.

* Forward visit (aba.hxx:102-139): liMi, oMi = oMparent * liMi, J = oMi.act(S), ov.  The walk keeps
  the world placement / velocity of the joint it stands on in registers only: after the backward
  visit of a child it steps back onto the parent by undoing the composition
  (oMp = oMi * liMi^-1, ov_p = ov_i - J vq; rounding-level difference to the forward values), so
  no kinematic state is stacked.  The own world inertia and bias force (:134-138) are evaluated
  when they are needed, at the backward visit.
* Backward visit (aba.hxx:153-193): u, U, D, Dinv, UDinv; Ia -= UDinv U^T;
  of += Ia * a_bias + UDinv u; the (Ia, of) contribution is parked for the parent (registers when
  the parent's backward visit is next).  Pass 3 needs UDinv and Dinv u of every joint: that
  hand-off record (7 doubles for a 1-dof joint) and the joint's (sin, cos) are parked until then.
* A joint attached to the universe solves for its acceleration at once (a_gf[0] = -gravity is
  known): the free-flyer without armature uses  a = Ia^-1 (oMi.act(tau) - of),
  ddq = oMi.actInv(a - a_gf[0]), which equals Dinv (u - U^T a_gf[0]) with D = J^T Ia J because J is
  the invertible action matrix of oMi (se3-tpl.hpp:178-191).
* Pass 3 (aba.hxx:207-232): a_i = a_parent + ov_parent x ov_i; ddq = Dinv u - UDinv^T a_i;
  a_i += J ddq.  Kinematics are recomputed from the parked (sin, cos) instead of storing J and the
  bias of every joint (12 more doubles per joint than the on-chip storage can hold).
"""
from __future__ import annotations

import os
from typing import List

from .. import model as M
from . import algebra as A
from . import joints as J
from .emit import ZERO, Emitter, V


# Pass 3 needs sin / cos of every joint again.  Evaluating them a second time (~40 instructions)
# is cheaper than carrying 2 more doubles per joint from pass 1 to pass 3 once the on-chip park
# storage is full and they would travel through HBM.
RECOMPUTE_SINCOS = os.environ.get("BRBD_CG_RESINCOS", "1") == "1"


def chol_factor(e: Emitter, D: List[List[V]]):
    """D = L L^T (lower).  Returns L with the RECIPROCAL diagonal stored on the diagonal."""
    n = len(D)
    L = [[ZERO] * n for _ in range(n)]
    for j in range(n):
        d = D[j][j]
        for k in range(j):
            d = e.fma(-L[j][k], L[j][k], d)
        rd = e.rsqrt(d)
        L[j][j] = rd
        for i in range(j + 1, n):
            s = D[i][j]
            for k in range(j):
                s = e.fma(-L[i][k], L[j][k], s)
            L[i][j] = e.mul(s, rd)
    return L


def chol_solve(e: Emitter, L, x: List[V]) -> List[V]:
    n = len(x)
    y = list(x)
    for i in range(n):  # L y = x
        s = y[i]
        for k in range(i):
            s = e.fma(-L[i][k], y[k], s)
        y[i] = e.mul(s, L[i][i])
    for i in range(n - 1, -1, -1):  # L^T z = y
        s = y[i]
        for k in range(i + 1, n):
            s = e.fma(-L[k][i], y[k], s)
        y[i] = e.mul(s, L[i][i])
    return y


def spd_inverse(e: Emitter, D: List[List[V]]) -> List[List[V]]:
    """Inverse of a small SPD matrix through its Cholesky factor (replaces matrix_inversion,
    math/matrix-inverse.hpp:142-147)."""
    n = len(D)
    L = chol_factor(e, D)  # diagonal holds 1 / L_jj
    Li = [[ZERO] * n for _ in range(n)]
    for j in range(n):
        Li[j][j] = L[j][j]
        for i in range(j + 1, n):
            s = ZERO
            for k in range(j, i):
                s = e.fma(-L[i][k], Li[k][j], s)
            Li[i][j] = e.mul(s, L[i][i])
    inv = [[ZERO] * n for _ in range(n)]
    for i in range(n):
        for j in range(i, n):
            s = ZERO
            for k in range(j, n):
                s = e.fma(Li[k][i], Li[k][j], s)
            inv[i][j] = inv[j][i] = s
    return inv


def sym6_full(Ia: List[V]) -> List[List[V]]:
    return [[Ia[A.sidx(i, j)] for j in range(6)] for i in range(6)]


# Tails.  A tail is a chain of up to TAIL joints that ends in a leaf.  Pass 2 does NOT keep the
# hand-off records of tail joints; pass 3, arriving at the head of a tail with the parent's state,
# walks down the tail, re-runs its backward visits (they depend on nothing outside the tail) and
# only then sweeps down again for the accelerations.  Re-running a backward visit costs ~300
# instructions (1 200 cycles); carrying its 7 doubles through HBM costs as much bandwidth as two
# input vectors, and the kernel is bound by that traffic.
TAIL = int(os.environ.get("BRBD_CG_TAIL", "2"))


def find_tails(mv: J.ModelView, limit: int):
    """joint -> the tail (list, head first, leaf last) it heads."""
    tails = {}
    if limit <= 0:
        return tails
    for leaf in range(1, mv.n):
        if mv.children[leaf] or mv.parents[leaf] == 0:
            continue
        chain = [leaf]
        while len(chain) < limit:
            p = mv.parents[chain[-1]]
            if p == 0 or mv.parents[p] == 0 or len(mv.children[p]) != 1:
                break
            chain.append(p)
        chain.reverse()
        tails[chain[0]] = chain
    return tails


def generate(mv: J.ModelView, e: Emitter) -> Emitter:
    g = None
    scpark = {}   # joint -> Park (s, c)
    hand = {}     # joint -> Park UDinv (6 nv) + Dinv u (nv)
    contrib = {}  # joint -> Park of the sum of the finished children's (Ia 21, of 6)
    tails = find_tails(mv, TAIL)
    in_tail = {j for ch in tails.values() for j in ch}

    def compose(p, Rp, tp, Rl, tl):
        if p == 0:  # aba.hxx:119-122
            return Rl, tl
        return A.rot_mul(e, Rp, Rl), A.rot_vec(e, Rp, tl, tp)  # se3-tpl.hpp:311-315

    def own_inertia(i, R, t, ov):
        """oYcrb = oMi.act(Y), its 6x6 matrix, of = ov x* (oYcrb ov)  (aba.hxx:134-138)"""
        m, c, I = mv.inertia(e, i)
        cw, Iw = A.inertia_to_world(e, R, t, c, I)
        Ia = A.inertia_to_sym6(e, m, cw, Iw)
        of = A.force_cross(e, ov, A.inertia_mul(e, m, cw, Iw, ov))
        # aba.hxx:338 (kernel variants with external forces): of[i] -= oMi.act(fext[i])
        of = e.cond_sub(of, lambda: A.act_force(e, R, t, [e.inp("F", 6 * i + k) for k in range(6)]))
        return Ia, of

    def backward_root(i, R, t, Jc, Ia, of):
        """Joint attached to the universe: a_gf[0] = -gravity is known, solve at once."""
        nv, iv = mv.nv(i), mv.idx_v[i]
        tau = [e.inp("X", iv + k) for k in range(nv)]
        arm = [mv.c(e, i, "arm", iv + k) for k in range(nv)]
        a0 = [-g[0], -g[1], -g[2], ZERO, ZERO, ZERO]  # aba.hxx:261
        if mv.jtype[i] == M.JOINT_FREEFLYER and all(x.is_zero() for x in arm):
            fw = A.act_force(e, R, t, tau)
            rhs = A.vsub(e, fw, of)
            a = chol_solve(e, chol_factor(e, sym6_full(Ia)), rhs)
            dd = A.actinv_motion(e, R, t, A.vsub(e, a, a0))
            for k in range(6):
                e.out(iv + k, dd[k])
            return a
        U = [A.sym6_mulvec(e, Ia, Jc[k]) for k in range(nv)]
        x = [e.sub(e.sub(tau[k], A.dot6(e, Jc[k], of)), A.dot6(e, U[k], a0)) for k in range(nv)]
        D = [[ZERO] * nv for _ in range(nv)]
        for r in range(nv):
            for c in range(r, nv):
                d = A.dot6(e, Jc[r], U[c])
                D[r][c] = D[c][r] = e.add(d, arm[r]) if r == c else d
        if nv == 1:
            dd = [e.mul(x[0], e.rcp(D[0][0]))]
        else:
            dd = chol_solve(e, chol_factor(e, D), x)
        a = a0
        for k in range(nv):
            e.out(iv + k, dd[k])
            a = A.vaxpy(e, dd[k], Jc[k], a)
        return a

    def backward_core(i, Jc, bias, Ia, of, want_parent=True):
        """aba.hxx:153-193 for a joint with a moving parent.  Returns (UDinv columns, Dinv u,
        contribution (Ia 21 + of 6) for the parent or None)."""
        nv, iv = mv.nv(i), mv.idx_v[i]
        tau = [e.inp("X", iv + k) for k in range(nv)]
        arm = [mv.c(e, i, "arm", iv + k) for k in range(nv)]
        u = [e.sub(tau[k], A.dot6(e, Jc[k], of)) for k in range(nv)]  # :173
        U = [A.sym6_mulvec(e, Ia, Jc[k]) for k in range(nv)]          # :175
        D = [[ZERO] * nv for _ in range(nv)]
        for r in range(nv):
            for c in range(r, nv):
                d = A.dot6(e, Jc[r], U[c])                            # :176
                D[r][c] = D[c][r] = e.add(d, arm[r]) if r == c else d  # :179
        if nv == 1:
            Dinv = [[e.rcp(D[0][0])]]
        else:
            Dinv = spd_inverse(e, D)                                  # :181
        UD = [[e.dot([U[r][k] for r in range(nv)], [Dinv[r][c] for r in range(nv)]) for k in range(6)]
              for c in range(nv)]                                     # :182
        Dinvu = [e.dot(Dinv[r], u) for r in range(nv)]
        if not want_parent:
            return UD, Dinvu, None
        Ia2 = list(Ia)
        for a_ in range(6):
            for b_ in range(a_, 6):
                s = Ia[A.sidx(a_, b_)]
                for c in range(nv):
                    s = e.fma(-UD[c][a_], U[c][b_], s)                # :186
                Ia2[A.sidx(a_, b_)] = s
        pa = A.sym6_mulvec(e, Ia2, bias, of)                          # :188
        for c in range(nv):
            pa = A.vaxpy(e, u[c], UD[c], pa)                          # :189
        return UD, Dinvu, Ia2 + pa

    def step_back(R, t, ov, Rl, tl, ovJ):
        """kinematics of the parent from the child's: oMp = oMi * liMi^-1, ov_p = ov_i - J vq"""
        Rp = [[e.dot(R[r], Rl[c]) for c in range(3)] for r in range(3)]  # R * Rl^T
        tp = A.vsub(e, t, A.rot_vec(e, Rp, tl))
        return Rp, tp, A.vsub(e, ov, ovJ)

    def kinematics(i, Rp, tp, ovp, sc=None):
        """world placement, Jacobian columns, velocity of joint i from its parent's"""
        p = mv.parents[i]
        Rl, tl, sc = J.joint_transform(e, mv, i, sc=sc)
        R, t = compose(p, Rp, tp, Rl, tl)
        nv, iv = mv.nv(i), mv.idx_v[i]
        Jc = [A.act_motion(e, R, t, col) for col in J.subspace(e, mv, i)]  # aba.hxx:124
        vq = [e.inp("V", iv + k) for k in range(nv)]
        ovJ = A.zeros(6)
        for k in range(nv):
            ovJ = A.vaxpy(e, vq[k], Jc[k], ovJ)                            # :126
        ov = ovJ if p == 0 else A.vadd(e, ovp, ovJ)                        # :127-128
        return R, t, ov, Jc, ovJ, Rl, tl, sc

    def flat(R, t, *vecs):
        out = [x for row in R for x in row] + list(t)
        for v in vecs:
            out += list(v)
        return out

    def pass3_joint(i, ap, bias, Jc, UD, Dinvu, leaf):
        """:223-227  a_i = a_parent + bias; ddq = Dinv u - UDinv^T a_i; a_i += J ddq"""
        nv, iv = mv.nv(i), mv.idx_v[i]
        a = A.vadd(e, ap, bias)
        dd = []
        for k in range(nv):
            d = e.sub(Dinvu[k], A.dot6(e, UD[k], a))
            dd.append(d)
            e.out(iv + k, d)
        if not leaf:
            for k in range(nv):
                a = A.vaxpy(e, dd[k], Jc[k], a)
        return a

    def pass3_tail(chain, kin_parent):
        """Pass 3 over a tail whose hand-off records were not kept: down (kinematics), up (the
        backward visits again), down (accelerations)."""
        n = len(chain)
        keep = {}   # joint -> Park of R, t, ov, J, bias (needed by its backward visit and the sweep)
        rec = {}    # joint -> Park of UDinv, Dinv u
        state = None
        for k, i in enumerate(chain):
            leaf = k == n - 1
            e.begin_visit("TAIL down %d%s" % (i, " + up (leaf)" if leaf else ""))
            if k == 0:
                vals = e.unpark(kin_parent)
                Rp, tp, ovp = [vals[0:3], vals[3:6], vals[6:9]], vals[9:12], vals[12:18]
                a_head = vals[18:24]
            else:
                Rp, tp, ovp = state
            R, t, ov, Jc, ovJ, _, _, _ = kinematics(i, Rp, tp, ovp)
            bias = A.motion_cross(e, ovp, ov)
            state = (R, t, ov)
            if leaf:
                Ia, of = own_inertia(i, R, t, ov)
                UD, Dinvu, up = backward_core(i, Jc, bias, Ia, of, want_parent=n > 1)
                rec[i] = e.park([x for col in UD for x in col] + Dinvu + [x for col in Jc for x in col] + bias,
                                "tail")
                carry = up
            else:
                keep[i] = e.park(flat(R, t, ov) + [x for col in Jc for x in col] + bias, "tail")
        for k in range(n - 2, -1, -1):
            i = chain[k]
            nv = mv.nv(i)
            e.begin_visit("TAIL up %d" % i)
            vals = e.unpark(keep[i])
            R, t, ov = [vals[0:3], vals[3:6], vals[6:9]], vals[9:12], vals[12:18]
            Jc = [vals[18 + 6 * c:24 + 6 * c] for c in range(nv)]
            bias = vals[18 + 6 * nv:24 + 6 * nv]
            Ia, of = own_inertia(i, R, t, ov)
            Ia = A.vadd(e, Ia, carry[:21])
            of = A.vadd(e, of, carry[21:])
            UD, Dinvu, carry = backward_core(i, Jc, bias, Ia, of, want_parent=k > 0)
            rec[i] = e.park([x for col in UD for x in col] + Dinvu + [x for col in Jc for x in col] + bias, "tail")
        a = a_head
        for k, i in enumerate(chain):
            nv = mv.nv(i)
            e.begin_visit("TAIL sweep %d" % i)
            vals = e.unpark(rec[i])
            UD = [vals[6 * c:6 * c + 6] for c in range(nv)]
            Dinvu = vals[6 * nv:7 * nv]
            Jc = [vals[7 * nv + 6 * c:7 * nv + 6 * c + 6] for c in range(nv)]
            bias = vals[13 * nv:13 * nv + 6]
            a = pass3_joint(i, a, bias, Jc, UD, Dinvu, k == n - 1)

    def pass3(root, Rr, tr, ovr, ar):
        """aba.hxx:207-232 over the subtree of `root` (whose own ddq is already written)."""
        # state of a joint for its children: Park of R (9), t (3), ov (6), a_gf (6)
        kin = {root: e.park(flat(Rr, tr, ovr, ar), "stack")}
        order = []

        def rec(i):
            for c in sorted(mv.children[i], reverse=True):
                order.append(c)
                rec(c)

        rec(root)
        for i in order:
            p = mv.parents[i]
            if i in tails:
                pass3_tail(tails[i], kin[p])
                continue
            if i in in_tail:
                continue
            e.begin_visit("PASS3 %d" % i)
            vals = e.unpark(kin[p])
            Rp, tp, ovp, ap = [vals[0:3], vals[3:6], vals[6:9]], vals[9:12], vals[12:18], vals[18:24]
            sc = tuple(e.unpark(scpark[i])) if (i in scpark and not RECOMPUTE_SINCOS) else None
            R, t, ov, Jc, _, _, _, _ = kinematics(i, Rp, tp, ovp, sc=sc)
            bias = A.motion_cross(e, ovp, ov)  # the bias of :129-132
            nv = mv.nv(i)
            h = e.unpark(hand[i])
            a = pass3_joint(i, ap, bias, Jc, [h[6 * k:6 * k + 6] for k in range(nv)], h[6 * nv:7 * nv],
                            not mv.children[i])
            if mv.children[i]:
                kin[i] = e.park(flat(R, t, ov, a), "stack")

    for root in sorted(mv.children[0], reverse=True):
        cur = None  # (R, t, ov) of the joint the walk stands on

        def walk(i):
            nonlocal cur, g
            p = mv.parents[i]
            leaf = not mv.children[i]
            e.begin_visit("FWD %d%s" % (i, " + BWD (leaf)" if leaf else ""))
            if g is None or p == 0:
                g = mv.gravity(e)
            R, t, ov, Jc, ovJ, Rl, tl, sc = kinematics(i, cur[0] if cur else None, cur[1] if cur else None,
                                                       cur[2] if cur else None)
            if sc is not None and p != 0 and not RECOMPUTE_SINCOS and i not in in_tail:
                scpark[i] = e.park(list(sc), "hand")
            elif sc is not None and not leaf:
                scpark[i] = e.park(list(sc), "stack")
            cur = (R, t, ov)
            if not leaf:
                for c in sorted(mv.children[i], reverse=True):
                    walk(c)
                e.begin_visit("BWD %d" % i)
                if p == 0:
                    g = mv.gravity(e)
                R, t, ov = cur
                sc = tuple(e.unpark(scpark[i])) if i in scpark else None
                Rl, tl, _ = J.joint_transform(e, mv, i, sc=sc)
                nv, iv = mv.nv(i), mv.idx_v[i]
                Jc = [A.act_motion(e, R, t, col) for col in J.subspace(e, mv, i)]
                vq = [e.inp("V", iv + k) for k in range(nv)]
                ovJ = A.zeros(6)
                for k in range(nv):
                    ovJ = A.vaxpy(e, vq[k], Jc[k], ovJ)
            Ia, of = own_inertia(i, R, t, ov)
            if i in contrib:
                vals = e.unpark(contrib.pop(i))
                Ia = A.vadd(e, Ia, vals[:21])
                of = A.vadd(e, of, vals[21:])
            if p == 0:
                return backward_root(i, R, t, Jc, Ia, of)
            # bias acceleration ov_parent x ov_i (:129-132); ov_parent = ov - J vq
            bias = A.motion_cross(e, A.vsub(e, ov, ovJ), ov)
            UD, Dinvu, acc = backward_core(i, Jc, bias, Ia, of)
            if i not in in_tail:
                hand[i] = e.park([x for col in UD for x in col] + Dinvu, "hand")
            # :190-191  oYaba[parent] += Ia, of[parent] += of: one running accumulator per parent (the
            # first finished child starts it, later ones add to it), consumed by the parent's backward
            if p in contrib:
                acc = A.vadd(e, e.unpark(contrib[p]), acc)
            contrib[p] = e.park(acc, "stack")
            cur = step_back(R, t, ov, Rl, tl, ovJ)
            return None

        a_root = walk(root)
        pass3(root, cur[0], cur[1], cur[2], a_root)
    return e
