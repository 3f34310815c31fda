"""Model compiler: ABA in the LOCAL (body-frame) convention as straight-line code.

The reference has the articulated-body algorithm twice: `abaWorldConvention` (algorithm/aba.hxx:
102-294, what abaInParallel calls, restated by codegen/aba.py) and `abaLocalConvention`
(aba.hxx:378-542, the default of `aba()`), which its own tests hold equal to 1e-12
(unittest/aba.cpp:157-159).  Both produce the same ddq; they differ in the frame the recursion is
written in.  For a model compiler that difference is large: in the joint frames the motion
subspace of an axis-aligned joint is a unit vector, a joint placement whose rotation is the
identity (or any signed permutation) keeps its zeros, the body inertia is a CONSTANT of the model,
U = Ia S is a column of Ia and D one of its diagonal entries -- so most of the 6 x 6 work of the
world-frame recursion folds away; what is left is the transformation of the articulated inertia
to the parent frame (`internal::SE3actOn`, aba.hxx:25-76).  codegen/compile.py generates both
bodies for a model and keeps the one with fewer instructions.

Per tree hanging off the universe, passes 1 + 2 fused in one depth-first walk, then pass 3:

* Forward visit (aba.hxx:378-403): liMi = placement * M(q), v_i = liMi.actInv(v_parent) + S vq.
  The walk keeps the velocity of the joint it stands on in registers only; after the backward
  visit of a child it steps back onto the parent (v_parent = liMi.act(v_i - S vq)).
* Backward visit (aba.hxx:416-443 and the joints' calc_aba, e.g. joint-revolute.hpp:832-846):
  bias c = v x S vq, pA = v x* (Y v), Ia = Y (+ the children's contributions),
  u = tau - S^T pA, U = Ia S, D = S^T U + armature, UDinv = U D^-1, Ia -= UDinv U^T,
  pA += Ia c + UDinv u; parent += liMi.act(Ia), liMi.act(pA).  Without armature the row and
  column of Ia along a 1-dof joint's axis are exactly zero after the update and are dropped.
* A joint attached to the universe solves at once: a_gf = c + liMi.actInv(-gravity) is known.
* Pass 3 (aba.hxx:456-477): a_i = liMi.actInv(a_parent) + c; ddq = Dinv u - UDinv^T a_i;
  a_i += S ddq.  The hand-off record per joint is UDinv, k = Dinv u - UDinv^T c and the joint's
  (sin, cos); the bias c is rebuilt from the velocity recursion (or parked, PARK_BIAS).
"""
from __future__ import annotations

import os
from typing import List

from .. import model as M
from . import algebra as A
from . import joints as J
from .aba import chol_factor, chol_solve, find_tails, spd_inverse, sym6_full
from .emit import ONE, ZERO, Emitter, V, num

TAIL = int(os.environ.get("BRBD_CG_LTAIL", "2"))  # default tail length (compile.py walks a ladder of them)
# tail joints: keep (sin, cos) from pass 1 for pass 3's re-run (2 doubles per tail joint) or evaluate again
TAIL_PARK_SC = os.environ.get("BRBD_CG_LTAILSC", "0") == "1"
# pass 3 needs the bias c_i = v_i x S vq again: parked with the hand-off record (4 doubles for an
# axis-aligned revolute joint) or rebuilt from the velocity recursion (~20 instructions)
PARK_BIAS = os.environ.get("BRBD_CG_LBIAS", "0") == "1"
# ... and sin / cos of the joint angle: parked (2 doubles) or evaluated again (~35 instructions)
RECOMPUTE_SINCOS = os.environ.get("BRBD_CG_LRESINCOS", "0") == "1"
# hand-off record: UDinv along the axis of a 1-dof joint without armature is U[ax] / U[ax] = 1, not stored;
# the joint velocity is parked with it or read again from the boundary matrix in pass 3
UD_AXIS_ONE = os.environ.get("BRBD_CG_LUDONE", "1") == "1"
REREAD_VQ = os.environ.get("BRBD_CG_LREREADVQ", "1") == "1"
# inertia transform behind a plain axis rotation through the double-angle form: 4 % fewer operations,
# measured (humanoid, 65 536, same box): BATCH_MAJOR ABA 22.87 -> 22.66 us, but the INSTANCE_MAJOR
# kernel, which stages its tile in registers, then spills 90 instead of 40 VGPRs: 28.8 -> 34.4 us.
# Off by default (experiment hook).
AXIS_CONGRUENCE = os.environ.get("BRBD_CG_AXISCONG", "0") == "1"


def sym6_blocks(Ia: List[V]):
    """packed symmetric 6x6 -> A (lin-lin), B (lin-ang), C (ang-ang) as 3x3 nested lists"""
    Ab = [[Ia[A.sidx(r, c)] for c in range(3)] for r in range(3)]
    Bb = [[Ia[A.sidx(r, 3 + c)] for c in range(3)] for r in range(3)]
    Cb = [[Ia[A.sidx(3 + r, 3 + c)] for c in range(3)] for r in range(3)]
    return Ab, Bb, Cb


def rot_congruence(e: Emitter, R, X, symmetric: bool):
    """R X R^T"""
    T = A.rot_mul(e, R, X)
    out = [[ZERO] * 3 for _ in range(3)]
    for r in range(3):
        for c in range(r if symmetric else 0, 3):
            out[r][c] = e.dot(T[r], R[c])
            if symmetric:
                out[c][r] = out[r][c]
    return out


def axis_rotation_of(R):
    """(k, c, s) when R is a plain rotation about coordinate axis k -- what an axis-aligned revolute
    joint behind an identity placement has (R[k][k] = 1, the (ka, kb) block [[c, -s], [s, c]]) -- else None."""
    for k in range(3):
        ka, kb = (k + 1) % 3, (k + 2) % 3
        if not (R[k][k].is_num and R[k][k].num == 1.0):
            continue
        if not all(R[k][j].is_zero() and R[j][k].is_zero() for j in (ka, kb)):
            continue
        c, s = R[ka][ka], R[kb][ka]
        if c.is_num or s.is_num:
            continue
        if R[kb][kb].key() == c.key() and R[ka][kb].key() == (-s).key():
            return k, c, s
    return None


def axis_congruence(e: Emitter, k: int, c: V, s: V, X, symmetric: bool, sq):
    """R X R^T for a rotation about coordinate axis k, through the double-angle form of the 2 x 2
    block: 12 instead of 22 operations for a symmetric X (8 when row / column k are zero), 20 instead
    of 24 for a general one.  sq = (c^2, s^2, c s, 2 c s, c^2 - s^2), shared by the three blocks."""
    c2, s2, cs, cs2, cd = sq
    p, r = (k + 1) % 3, (k + 2) % 3
    Y = [[ZERO] * 3 for _ in range(3)]
    Y[k][k] = X[k][k]
    Y[k][p] = e.fma(c, X[k][p], -e.mul(s, X[k][r]))
    Y[k][r] = e.fma(s, X[k][p], e.mul(c, X[k][r]))
    a, b, g, d = X[p][p], X[p][r], X[r][p], X[r][r]
    if symmetric:
        Y[p][k], Y[r][k] = Y[k][p], Y[k][r]
        a1 = e.fma(c2, a, e.fma(s2, d, -e.mul(cs2, b)))
        d1 = e.sub(e.add(a, d), a1)
        b1 = e.fma(cs, e.sub(a, d), e.mul(cd, b))
        Y[p][p], Y[r][r], Y[p][r], Y[r][p] = a1, d1, b1, b1
        return Y
    Y[p][k] = e.fma(c, X[p][k], -e.mul(s, X[r][k]))
    Y[r][k] = e.fma(s, X[p][k], e.mul(c, X[r][k]))
    u = e.mul(cs, e.sub(a, d))
    a1 = e.fma(c2, a, e.fma(s2, d, -e.mul(cs, e.add(b, g))))
    Y[p][p] = a1
    Y[r][r] = e.sub(e.add(a, d), a1)
    Y[p][r] = e.fma(c2, b, e.fma(-s2, g, u))
    Y[r][p] = e.fma(c2, g, e.fma(-s2, b, u))
    return Y


def sym6_act(e: Emitter, R, t, Ia: List[V]) -> List[V]:
    """Xf Ia Xf^T with Xf = [[R, 0], [[t]x R, R]] the force action matrix of liMi = (R, t): an
    articulated inertia expressed in the parent frame (internal::SE3actOn, aba.hxx:25-76).
    With T = [t]x:  A' = R A R^T;  B' = Br - A' T;  C' = Cr + T N + (T N)^T,  N = Br - A' T / 2
    (Br = R B R^T, Cr = R C R^T)."""
    Ab, Bb, Cb = sym6_blocks(Ia)
    ax = axis_rotation_of(R) if AXIS_CONGRUENCE else None
    if ax is not None:
        k, c, s = ax
        cs = e.mul(c, s)
        c2, s2 = e.mul(c, c), e.mul(s, s)
        sq = (c2, s2, cs, e.add(cs, cs), e.sub(c2, s2))
        A1 = axis_congruence(e, k, c, s, Ab, True, sq)
        Br = axis_congruence(e, k, c, s, Bb, False, sq)
        Cr = axis_congruence(e, k, c, s, Cb, True, sq)
    else:
        A1 = rot_congruence(e, R, Ab, True)
        Br = rot_congruence(e, R, Bb, False)
        Cr = rot_congruence(e, R, Cb, True)
    th = [e.mul(x, num(0.5)) for x in t]
    # row r of A' T is A'[r] x t
    N = [A.cross_add(e, [-x for x in A1[r]], th, Br[r]) for r in range(3)]
    two = num(2.0)
    B1 = [[e.fma(two, N[r][c], -Br[r][c]) for c in range(3)] for r in range(3)]
    TN = [A.cross(e, t, [N[0][c], N[1][c], N[2][c]]) for c in range(3)]  # TN[c][r] = (T N)[r][c]
    C1 = [[ZERO] * 3 for _ in range(3)]
    for r in range(3):
        for c in range(r, 3):
            C1[r][c] = e.add(e.add(Cr[r][c], TN[c][r]), TN[r][c])
    out = [ZERO] * 21
    for r in range(3):
        for c in range(3):
            if c >= r:
                out[A.sidx(r, c)] = A1[r][c]
                out[A.sidx(3 + r, 3 + c)] = C1[r][c]
            out[A.sidx(r, 3 + c)] = B1[r][c]
    return out


def unit_axis(col: List[V]):
    """index of the 1 when the subspace column is a unit vector, else None"""
    ones = [j for j, x in enumerate(col) if x.is_num and x.num == 1.0]
    if len(ones) == 1 and all(x.is_zero() for j, x in enumerate(col) if j != ones[0]):
        return ones[0]
    return None


def generate(mv: J.ModelView, e: Emitter, tail: int = None) -> Emitter:
    keep = {}     # joint -> Park of what its backward visit needs from its forward visit
    hand = {}     # joint -> Park of UDinv (6 nv) + k (nv) [+ bias] [+ (s, c)]
    contrib = {}  # joint -> Park of the sum of the finished children's (Ia 21, pA 6) in its frame
    tailsc = {}   # tail joint -> Park of (s, c)
    tails = find_tails(mv, TAIL if tail is None else tail)
    in_tail = {j for ch in tails.values() for j in ch}

    def transform(i, sc=None):
        R, t, sc = J.joint_transform(e, mv, i, sc=sc)
        return R, t, sc

    def joint_velocity(i, vq=None):
        """S, S vq and vq itself (read from the boundary matrix unless a parked copy is at hand:
        every re-read is HBM traffic again, the inputs stream past the L2)"""
        S = J.subspace(e, mv, i)
        iv = mv.idx_v[i]
        if vq is None:
            vq = [e.inp("V", iv + k) for k in range(mv.nv(i))]
        return S, J.s_times(e, S, vq), list(vq)

    def body_terms(i, v, vJ):
        """bias c = v x vJ (aba.hxx:398), Ia = Y.matrix() (:400), pA = v x* (Y v) (:401-402)"""
        m, c, I = mv.inertia(e, i)
        bias = A.motion_cross(e, v, vJ)
        pA = A.force_cross(e, v, A.inertia_mul(e, m, c, I, v))
        # aba.hxx:572 (kernel variants with external forces): f[i] -= fext[i], joint frame
        pA = e.cond_sub(pA, lambda: [e.inp("F", 6 * i + k) for k in range(6)])
        return bias, A.inertia_to_sym6(e, m, c, I), pA

    def project(i, S, bias, Ia, pA, want_parent=True):
        """aba.hxx:431-439 + calc_aba.  Returns UDinv columns, k = Dinv u - UDinv^T c, and the
        (Ia 21, pA 6) left for the parent (in this joint's frame) or None."""
        nv, iv = mv.nv(i), mv.idx_v[i]
        tau = [e.inp("X", iv + k) for k in range(nv)]
        arm = [mv.c(e, i, "arm", iv + k) for k in range(nv)]
        u = [e.sub(tau[k], A.dot6(e, S[k], pA)) for k in range(nv)]      # :431
        U = [A.sym6_mulvec(e, Ia, S[k]) for k in range(nv)]
        D = [[ZERO] * nv for _ in range(nv)]
        for r in range(nv):
            for c in range(r, nv):
                d = A.dot6(e, S[r], U[c])
                D[r][c] = D[c][r] = e.add(d, arm[r]) if r == c else d
        Dinv = [[e.rcp(D[0][0])]] if nv == 1 else spd_inverse(e, D)
        UD = [[e.dot([U[r][k] for r in range(nv)], [Dinv[r][c] for r in range(nv)]) for k in range(6)]
              for c in range(nv)]
        ax = unit_axis(S[0]) if (nv == 1 and arm[0].is_zero()) else None
        if ax is not None and UD_AXIS_ONE:
            UD[0][ax] = ONE  # U[ax] / D with D = U[ax]: 1 (the quotient's rounding aside), and one double less to hand over
        kk = [e.sub(e.dot(Dinv[r], u), A.dot6(e, UD[r], bias)) for r in range(nv)]
        if not want_parent:
            return UD, kk, None
        Ia2 = list(Ia)
        for a_ in range(6):
            for b_ in range(a_, 6):
                if ax is not None and ax in (a_, b_):
                    # U - (U[ax] / D) U with D = U[ax]: exactly zero without armature
                    Ia2[A.sidx(a_, b_)] = ZERO
                    continue
                s = Ia[A.sidx(a_, b_)]
                for c in range(nv):
                    s = e.fma(-UD[c][a_], U[c][b_], s)
                Ia2[A.sidx(a_, b_)] = s
        pa = A.sym6_mulvec(e, Ia2, bias, pA)                              # :438-439
        for c in range(nv):
            pa = A.vaxpy(e, u[c], UD[c], pa)
        return UD, kk, Ia2 + pa

    def solve_root(i, R, t, S, bias, Ia, pA):
        """Joint attached to the universe: a_gf = c + liMi.actInv(-gravity) is known (:468)."""
        nv, iv = mv.nv(i), mv.idx_v[i]
        g = mv.gravity(e)
        a0 = A.vadd(e, A.actinv_motion(e, R, t, [-g[0], -g[1], -g[2], ZERO, ZERO, ZERO]), bias)
        tau = [e.inp("X", iv + k) for k in range(nv)]
        arm = [mv.c(e, i, "arm", iv + k) for k in range(nv)]
        if mv.jtype[i] == M.JOINT_FREEFLYER and all(x.is_zero() for x in arm):
            # S = 1: ddq = Ia^-1 (tau - pA) - a_gf, a = Ia^-1 (tau - pA)
            a = chol_solve(e, chol_factor(e, sym6_full(Ia)), A.vsub(e, tau, pA))
            dd = A.vsub(e, a, a0)
            for k in range(6):
                e.out(iv + k, dd[k])
            return a
        U = [A.sym6_mulvec(e, Ia, S[k]) for k in range(nv)]
        x = [e.sub(e.sub(tau[k], A.dot6(e, S[k], pA)), A.dot6(e, U[k], a0)) for k in range(nv)]
        D = [[ZERO] * nv for _ in range(nv)]
        for r in range(nv):
            for c in range(r, nv):
                d = A.dot6(e, S[r], U[c])
                D[r][c] = D[c][r] = e.add(d, arm[r]) if r == c else d
        dd = [e.mul(x[0], e.rcp(D[0][0]))] if nv == 1 else chol_solve(e, chol_factor(e, D), x)
        a = a0
        for k in range(nv):
            e.out(iv + k, dd[k])
            a = A.vaxpy(e, dd[k], S[k], a)
        return a

    def to_parent(i, R, t, acc):
        """:440-441  Yaba[parent] += SE3actOn(liMi, Ia), f[parent] += liMi.act(pa): one running sum per parent"""
        p = mv.parents[i]
        up = sym6_act(e, R, t, acc[:21]) + A.act_force(e, R, t, acc[21:])
        if p in contrib:
            up = A.vadd(e, e.unpark(contrib[p]), up)
        contrib[p] = e.park(up, "stack")

    def pass3_joint(i, a_in, bias, S, UD, kk, leaf):
        """:468-472 with a_in = liMi.actInv(a_parent): ddq = k - UDinv^T a_in; a = a_in + c + S ddq"""
        nv, iv = mv.nv(i), mv.idx_v[i]
        dd = []
        for k in range(nv):
            d = e.sub(kk[k], A.dot6(e, UD[k], a_in))
            dd.append(d)
            e.out(iv + k, d)
        if leaf:
            return None
        a = A.vadd(e, a_in, bias)
        for k in range(nv):
            a = A.vaxpy(e, dd[k], S[k], a)
        return a

    def pass3_tail(chain, kin_parent):
        """Pass 3 over a tail whose hand-off records were not kept: down (velocities), up (the
        backward visits again), down (accelerations)."""
        n = len(chain)
        st = {}   # joint -> Park of (R, t non-constant entries as (s, c)), v, vJ
        rec = {}
        vals = None
        carry = None
        for k, i in enumerate(chain):
            leaf = k == n - 1
            e.begin_visit("TAIL down %d%s" % (i, " + up (leaf)" if leaf else ""))
            if k == 0:
                vals = e.unpark(kin_parent)
                vp, a_head = vals[0:6], vals[6:12]
            R, t, sc = transform(i, tuple(e.unpark(tailsc[i])) if i in tailsc else None)
            S, vJ, _ = joint_velocity(i)
            v = A.vadd(e, A.actinv_motion(e, R, t, vp), vJ)
            vp = v
            if leaf:
                bias, Ia, pA = body_terms(i, v, vJ)
                UD, kk, carry = project(i, S, bias, Ia, pA, want_parent=n > 1)
                if carry is not None:
                    carry = sym6_act(e, R, t, carry[:21]) + A.act_force(e, R, t, carry[21:])
                rec[i] = e.park([x for col in UD for x in col] + kk + bias + (list(sc) if sc else []), "tail")
            else:
                st[i] = e.park(v + vJ + (list(sc) if sc else []), "tail")
        for k in range(n - 2, -1, -1):
            i = chain[k]
            e.begin_visit("TAIL up %d" % i)
            vals_i = e.unpark(st[i])
            v, vJ = vals_i[0:6], vals_i[6:12]
            sc = tuple(vals_i[12:14]) if len(vals_i) > 12 else None
            R, t, sc = transform(i, sc)
            S = J.subspace(e, mv, i)
            bias, Ia, pA = body_terms(i, v, vJ)
            Ia = A.vadd(e, Ia, carry[:21])
            pA = A.vadd(e, pA, carry[21:])
            UD, kk, carry = project(i, S, bias, Ia, pA, want_parent=k > 0)
            if carry is not None:
                carry = sym6_act(e, R, t, carry[:21]) + A.act_force(e, R, t, carry[21:])
            rec[i] = e.park([x for col in UD for x in col] + kk + bias + (list(sc) if sc else []), "tail")
        a = a_head
        for k, i in enumerate(chain):
            nv = mv.nv(i)
            e.begin_visit("TAIL sweep %d" % i)
            r = e.unpark(rec[i])
            UD = [r[6 * c:6 * c + 6] for c in range(nv)]
            kk = r[6 * nv:7 * nv]
            bias = r[7 * nv:7 * nv + 6]
            sc = tuple(r[7 * nv + 6:7 * nv + 8]) if len(r) > 7 * nv + 6 else None
            R, t, _ = transform(i, sc)
            a = pass3_joint(i, A.actinv_motion(e, R, t, a), bias, J.subspace(e, mv, i), UD, kk, k == n - 1)

    def pass3(root, v_root, a_root):
        """aba.hxx:456-477 over the subtree of `root` (whose own ddq is already written)."""
        kin = {root: e.park(list(v_root) + list(a_root), "stack")}  # v (6), a_gf (6) of a joint, for its children
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
            vp, ap = vals[0:6], vals[6:12]
            nv = mv.nv(i)
            h = e.unpark(hand[i])
            UD = [h[6 * k:6 * k + 6] for k in range(nv)]
            kk = h[6 * nv:7 * nv]
            pos = 7 * nv
            bias = vq = None
            if PARK_BIAS:
                bias = h[pos:pos + 6]
                pos += 6
            elif not REREAD_VQ:
                vq = h[pos:pos + nv]
                pos += nv
            sc = tuple(h[pos:pos + 2]) if len(h) >= pos + 2 else None
            R, t, _ = transform(i, sc)
            S = J.subspace(e, mv, i)
            leaf = not mv.children[i]
            v = None
            if bias is None or not leaf:  # (the velocity also goes down to the children: tails start from it)
                _, vJ, _ = joint_velocity(i, vq)
                v = A.vadd(e, A.actinv_motion(e, R, t, vp), vJ)
                if bias is None:
                    bias = A.motion_cross(e, v, vJ)
            a = pass3_joint(i, A.actinv_motion(e, R, t, ap), bias, S, UD, kk, leaf)
            if not leaf:
                kin[i] = e.park(list(v) + list(a), "stack")

    for root in sorted(mv.children[0], reverse=True):
        cur = None  # velocity of the joint the walk stands on

        def walk(i):
            nonlocal cur
            p = mv.parents[i]
            leaf = not mv.children[i]
            e.begin_visit("FWD %d%s" % (i, " + BWD (leaf)" if leaf else ""))
            R, t, sc = transform(i)
            S, vJ, vq = joint_velocity(i)
            v = vJ if p == 0 else A.vadd(e, A.actinv_motion(e, R, t, cur), vJ)  # aba.hxx:394-396
            cur = v
            if not leaf:
                keep[i] = e.park(vq + (list(sc) if sc is not None else []), "stack")
                for c in sorted(mv.children[i], reverse=True):
                    walk(c)
                e.begin_visit("BWD %d" % i)
                v = cur
                kept = e.unpark(keep[i])
                vq = kept[:mv.nv(i)]
                R, t, sc = transform(i, tuple(kept[mv.nv(i):]) if sc is not None else None)
                S, vJ, _ = joint_velocity(i, vq)
            bias, Ia, pA = body_terms(i, v, vJ)
            if i in contrib:
                vals = e.unpark(contrib.pop(i))
                Ia = A.vadd(e, Ia, vals[:21])
                pA = A.vadd(e, pA, vals[21:])
            if p == 0:
                return solve_root(i, R, t, S, bias, Ia, pA)
            UD, kk, acc = project(i, S, bias, Ia, pA)
            if i not in in_tail:
                rec = [x for col in UD for x in col] + kk
                rec += bias if PARK_BIAS else ([] if REREAD_VQ else vq)
                if sc is not None and not RECOMPUTE_SINCOS:
                    rec += list(sc)
                hand[i] = e.park(rec, "hand")
            elif TAIL_PARK_SC and sc is not None:
                tailsc[i] = e.park(list(sc), "hand")
            to_parent(i, R, t, acc)
            cur = A.act_motion(e, R, t, A.vsub(e, v, vJ))  # step back onto the parent
            return None

        a_root = walk(root)
        pass3(root, cur, a_root)
    return e
