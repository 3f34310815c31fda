"""Deliberately naive, structurally independent implementation of RNEA / ABA / FK /
Jacobians written from the textbook equations (Featherstone, "Rigid Body Dynamics
Algorithms", Tables 5.1 and 7.1), NOT from the reference's code:

* spatial vectors are Featherstone-ordered [angular; linear] (the reference and the
  engine use [linear; angular]), everything is dense 6x6 Pluecker algebra;
* all recursions are in BODY coordinates (the engine's ABA is in WORLD coordinates);
* arithmetic is numpy longdouble (x87 80-bit on this host).

TEST INFRASTRUCTURE ONLY.  It is the second implementation that pins the C restatement
(oracle/brbd_oracle.c) at <= 1e-12; see tests/test_oracle_vs_naive.py.
"""
from __future__ import annotations

import numpy as np

LD = np.longdouble

RX, RY, RZ, PX, PY, PZ, FF, SPH, RU, PU, RUBX, RUBY, RUBZ, RUBU = range(14)


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=LD)


def quat_R(x, y, z, w):
    """R = I + 2 w [v]x + 2 [v]x^2: equals Eigen's toRotationMatrix polynomial, also for
    non-unit quaternions (the reference does not normalise, joint-free-flyer.hpp:338-339)."""
    V = skew(np.array([x, y, z], dtype=LD))
    return np.eye(3, dtype=LD) + 2 * w * V + 2 * (V @ V)


def axis_R(axis, c, s):
    K = skew(axis)
    return np.eye(3, dtype=LD) + s * K + (1 - c) * (K @ K)


def joint(jt, axis, q, vj):
    """Return (R, p, S[6 x nv] in [ang;lin], vJ) of the joint transform M(q)."""
    e = np.eye(3, dtype=LD)
    Z = np.zeros(3, dtype=LD)
    if jt in (RX, RY, RZ, RUBX, RUBY, RUBZ):
        if jt <= RZ:
            k = jt - RX
            c, s = np.cos(LD(q[0])), np.sin(LD(q[0]))
        else:
            k = jt - RUBX
            c, s = LD(q[0]), LD(q[1])
        S = np.concatenate([e[k], Z]).reshape(6, 1)
        return axis_R(e[k], c, s), Z, S
    if jt in (PX, PY, PZ):
        k = jt - PX
        S = np.concatenate([Z, e[k]]).reshape(6, 1)
        return e, e[k] * LD(q[0]), S
    if jt == FF:
        S = np.zeros((6, 6), dtype=LD)
        S[3:, :3] = e  # v_joint = [v_lin, w] (pinocchio order) -> [w; v_lin]
        S[:3, 3:] = e
        return quat_R(*[LD(x) for x in q[3:7]]), np.array(q[:3], dtype=LD), S
    if jt == SPH:
        S = np.zeros((6, 3), dtype=LD)
        S[:3, :] = e
        return quat_R(*[LD(x) for x in q[:4]]), Z, S
    if jt == RU:
        a = np.array(axis, dtype=LD)
        S = np.concatenate([a, Z]).reshape(6, 1)
        return axis_R(a, np.cos(LD(q[0])), np.sin(LD(q[0]))), Z, S
    if jt == RUBU:
        a = np.array(axis, dtype=LD)
        S = np.concatenate([a, Z]).reshape(6, 1)
        return axis_R(a, LD(q[0]), LD(q[1])), Z, S
    if jt == PU:
        a = np.array(axis, dtype=LD)
        S = np.concatenate([Z, a]).reshape(6, 1)
        return e, a * LD(q[0]), S
    raise ValueError(jt)


def xform(R, p):
    """Pluecker motion transform parent -> child for a child frame placed at (R, p) in the
    parent: X = [[E, 0], [-E [p]x, E]], E = R^T."""
    E = R.T
    X = np.zeros((6, 6), dtype=LD)
    X[:3, :3] = E
    X[3:, 3:] = E
    X[3:, :3] = -E @ skew(p)
    return X


def crm(v):
    M = np.zeros((6, 6), dtype=LD)
    M[:3, :3] = skew(v[:3])
    M[3:, 3:] = skew(v[:3])
    M[3:, :3] = skew(v[3:])
    return M


def crf(v):
    return -crm(v).T


def spatial_inertia(y10):
    m = LD(y10[0])
    c = np.array(y10[1:4], dtype=LD)
    d = [LD(x) for x in y10[4:10]]
    Ic = np.array([[d[0], d[1], d[3]], [d[1], d[2], d[4]], [d[3], d[4], d[5]]], dtype=LD)
    C = skew(c)
    I = np.zeros((6, 6), dtype=LD)
    I[:3, :3] = Ic + m * (C @ C.T)
    I[:3, 3:] = m * C
    I[3:, :3] = m * C.T
    I[3:, 3:] = m * np.eye(3, dtype=LD)
    return I


class Tables:
    def __init__(self, model):
        t = model.tables()
        self.n = t["njoints"]
        self.nq, self.nv = t["nq"], t["nv"]
        self.parents = t["parents"]
        self.jtype = t["jtype"]
        self.idx_q, self.idx_v = t["idx_q"], t["idx_v"]
        self.axis = t["axis"]
        self.armature = np.array(t["armature"], dtype=LD)
        self.gravity = np.array(t["gravity"], dtype=LD)
        self.Rp, self.pp, self.I = [None], [None], [None]
        for i in range(1, self.n):
            p12 = np.array(t["placement"][i], dtype=LD)
            self.Rp.append(p12[:9].reshape(3, 3).T.copy())
            self.pp.append(p12[9:].copy())
            self.I.append(spatial_inertia(t["inertia"][i]))
        self.nvj = [0] + [joint(self.jtype[i], self.axis[i], model.neutral()[self.idx_q[i]:], None)[2].shape[1]
                          for i in range(1, self.n)]
        self.nqj = [0] + [model.nqs[i] for i in range(1, self.n)]


def _kin(T, i, q):
    R, p, S = joint(T.jtype[i], T.axis[i], q[T.idx_q[i]: T.idx_q[i] + T.nqj[i]], None)
    # liMi = placement * M(q)
    Rl = T.Rp[i] @ R
    pl = T.pp[i] + T.Rp[i] @ p
    return Rl, pl, S


def rnea(T: Tables, q, v, a):
    q, v, a = (np.asarray(x, dtype=LD) for x in (q, v, a))
    n = T.n
    X, S, vs, as_, fs = [None] * n, [None] * n, [None] * n, [None] * n, [None] * n
    vs[0] = np.zeros(6, dtype=LD)
    as_[0] = np.concatenate([np.zeros(3, dtype=LD), -T.gravity])
    for i in range(1, n):
        Rl, pl, S[i] = _kin(T, i, q)
        X[i] = xform(Rl, pl)
        iv = T.idx_v[i]
        vJ = S[i] @ v[iv: iv + T.nvj[i]]
        vs[i] = X[i] @ vs[T.parents[i]] + vJ
        as_[i] = X[i] @ as_[T.parents[i]] + S[i] @ a[iv: iv + T.nvj[i]] + crm(vs[i]) @ vJ
        fs[i] = T.I[i] @ as_[i] + crf(vs[i]) @ (T.I[i] @ vs[i])
    tau = np.zeros(T.nv, dtype=LD)
    for i in range(n - 1, 0, -1):
        iv = T.idx_v[i]
        tau[iv: iv + T.nvj[i]] = S[i].T @ fs[i]
        p = T.parents[i]
        if p > 0:
            fs[p] = fs[p] + X[i].T @ fs[i]
    return tau + T.armature * a


def aba(T: Tables, q, v, tau):
    q, v, tau = (np.asarray(x, dtype=LD) for x in (q, v, tau))
    n = T.n
    X, S, vs, cs, IA, pA = ([None] * n for _ in range(6))
    U, Dinv, u = [None] * n, [None] * n, [None] * n
    vs[0] = np.zeros(6, dtype=LD)
    for i in range(1, n):
        Rl, pl, S[i] = _kin(T, i, q)
        X[i] = xform(Rl, pl)
        iv = T.idx_v[i]
        vJ = S[i] @ v[iv: iv + T.nvj[i]]
        vs[i] = X[i] @ vs[T.parents[i]] + vJ
        cs[i] = crm(vs[i]) @ vJ
        IA[i] = T.I[i].copy()
        pA[i] = crf(vs[i]) @ (T.I[i] @ vs[i])
    for i in range(n - 1, 0, -1):
        iv, nvj = T.idx_v[i], T.nvj[i]
        U[i] = IA[i] @ S[i]
        D = S[i].T @ U[i] + np.diag(T.armature[iv: iv + nvj])
        Dinv[i] = np.linalg.inv(D.astype(np.float64)).astype(LD)
        # two Newton steps lift the float64 inverse to longdouble accuracy
        for _ in range(2):
            Dinv[i] = Dinv[i] @ (2 * np.eye(nvj, dtype=LD) - D @ Dinv[i])
        u[i] = tau[iv: iv + nvj] - S[i].T @ pA[i]
        p = T.parents[i]
        if p > 0:
            Ia = IA[i] - U[i] @ Dinv[i] @ U[i].T
            pa = pA[i] + Ia @ cs[i] + U[i] @ Dinv[i] @ u[i]
            IA[p] = IA[p] + X[i].T @ Ia @ X[i]
            pA[p] = pA[p] + X[i].T @ pa
    ddq = np.zeros(T.nv, dtype=LD)
    acc = [None] * n
    acc[0] = np.concatenate([np.zeros(3, dtype=LD), -T.gravity])
    for i in range(1, n):
        iv, nvj = T.idx_v[i], T.nvj[i]
        ap = X[i] @ acc[T.parents[i]] + cs[i]
        ddq[iv: iv + nvj] = Dinv[i] @ (u[i] - U[i].T @ ap)
        acc[i] = ap + S[i] @ ddq[iv: iv + nvj]
    return ddq


def fk(T: Tables, q):
    """Homogeneous 4x4 world placements, index 0 = identity."""
    q = np.asarray(q, dtype=LD)
    H = [np.eye(4, dtype=LD)]
    for i in range(1, T.n):
        Rl, pl, _ = _kin(T, i, q)
        L = np.eye(4, dtype=LD)
        L[:3, :3] = Rl
        L[:3, 3] = pl
        H.append(H[T.parents[i]] @ L)
    return H


def jacobian_world(T: Tables, q):
    """6 x nv, rows [lin; ang] like the reference's data.J, columns = Ad(oMi) S_i with
    the spatial-velocity ("velocity of the point at the world origin") convention."""
    q = np.asarray(q, dtype=LD)
    H = fk(T, q)
    J = np.zeros((6, T.nv), dtype=LD)
    for i in range(1, T.n):
        _, _, S = _kin(T, i, q)
        R, p = H[i][:3, :3], H[i][:3, 3]
        for c in range(T.nvj[i]):
            w_loc, v_loc = S[:3, c], S[3:, c]
            w = R @ w_loc
            vv = R @ v_loc + np.cross(p, w)
            J[:3, T.idx_v[i] + c] = vv
            J[3:, T.idx_v[i] + c] = w
    return J


def integrate(T: Tables, q, dv):
    """q (+) dv: configuration retraction used by the finite-difference checks
    (stand-in for pinocchio::integrate; exact exp on SO(3), first order in the FF
    translation, which is enough for alpha = 1e-8 forward differences)."""
    q = np.array(q, dtype=LD)
    dv = np.asarray(dv, dtype=LD)
    out = q.copy()

    def quat_mul_exp(qx, w):
        th = np.sqrt(w @ w)
        if th < 1e-30:
            dq = np.array([w[0] / 2, w[1] / 2, w[2] / 2, 1], dtype=LD)
        else:
            dq = np.concatenate([np.sin(th / 2) * w / th, [np.cos(th / 2)]])
        x1, y1, z1, w1 = qx
        x2, y2, z2, w2 = dq
        return np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                         w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                         w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
                         w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], dtype=LD)

    for i in range(1, T.n):
        jt, iq, iv = T.jtype[i], T.idx_q[i], T.idx_v[i]
        if jt == FF:
            R = quat_R(*q[iq + 3: iq + 7])
            out[iq: iq + 3] = q[iq: iq + 3] + R @ dv[iv: iv + 3]
            out[iq + 3: iq + 7] = quat_mul_exp(q[iq + 3: iq + 7], dv[iv + 3: iv + 6])
        elif jt == SPH:
            out[iq: iq + 4] = quat_mul_exp(q[iq: iq + 4], dv[iv: iv + 3])
        elif jt in (RUBX, RUBY, RUBZ, RUBU):
            c, s = q[iq], q[iq + 1]
            dc, ds = np.cos(dv[iv]), np.sin(dv[iv])
            out[iq], out[iq + 1] = c * dc - s * ds, s * dc + c * ds
        else:
            out[iq] = q[iq] + dv[iv]
    return out
