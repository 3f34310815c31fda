"""6-D spatial algebra on symbolic values (emit.V).

Same conventions as the reference: spatial vectors are [linear(3); angular(3)]
(multibody/joint/joint-revolute.hpp:56-60); rotations are 3x3 nested lists R[r][c].
Every helper cites the reference formula it restates; zeros fold in the emitter, so e.g. the
action of an identity placement or the cross product with a unit axis costs what it should.
"""
from __future__ import annotations

from typing import List, Sequence

from .emit import ONE, ZERO, Emitter, V

Vec = List[V]
Mat3 = List[List[V]]


def zeros(n: int) -> Vec:
    return [ZERO] * n


def eye3() -> Mat3:
    return [[ONE if r == c else ZERO for c in range(3)] for r in range(3)]


def vadd(e: Emitter, a: Sequence[V], b: Sequence[V]) -> Vec:
    return [e.add(x, y) for x, y in zip(a, b)]


def vsub(e: Emitter, a: Sequence[V], b: Sequence[V]) -> Vec:
    return [e.sub(x, y) for x, y in zip(a, b)]


def vscale(e: Emitter, s: V, a: Sequence[V]) -> Vec:
    return [e.mul(s, x) for x in a]


def vaxpy(e: Emitter, s: V, a: Sequence[V], y: Sequence[V]) -> Vec:
    """s * a + y"""
    return [e.fma(s, x, z) for x, z in zip(a, y)]


def cross(e: Emitter, a: Sequence[V], b: Sequence[V]) -> Vec:
    return [
        e.fma(a[1], b[2], -e.mul(a[2], b[1])),
        e.fma(a[2], b[0], -e.mul(a[0], b[2])),
        e.fma(a[0], b[1], -e.mul(a[1], b[0])),
    ]


def cross_add(e: Emitter, a: Sequence[V], b: Sequence[V], y: Sequence[V]) -> Vec:
    """a x b + y"""
    return [
        e.fma(a[1], b[2], e.fma(-a[2], b[1], y[0])),
        e.fma(a[2], b[0], e.fma(-a[0], b[2], y[1])),
        e.fma(a[0], b[1], e.fma(-a[1], b[0], y[2])),
    ]


def rot_vec(e: Emitter, R: Mat3, x: Sequence[V], init: Sequence[V] = (ZERO, ZERO, ZERO)) -> Vec:
    """R x (+ init)"""
    return [e.dot(R[r], x, init[r]) for r in range(3)]


def rotT_vec(e: Emitter, R: Mat3, x: Sequence[V]) -> Vec:
    """R^T x"""
    return [e.dot([R[0][c], R[1][c], R[2][c]], x) for c in range(3)]


def rot_mul(e: Emitter, A: Mat3, B: Mat3) -> Mat3:
    return [[e.dot(A[r], [B[0][c], B[1][c], B[2][c]]) for c in range(3)] for r in range(3)]


def col(R: Mat3, k: int) -> Vec:
    return [R[0][k], R[1][k], R[2][k]]


# ---- SE3 actions -------------------------------------------------------------------------------
def act_motion(e: Emitter, R: Mat3, t: Sequence[V], m: Sequence[V]) -> Vec:
    """M.act(m): ang' = R ang; lin' = R lin + t x ang'  (spatial/motion-dense.hpp:261-266)"""
    ang = rot_vec(e, R, m[3:6])
    lin = rot_vec(e, R, m[0:3])
    return cross_add(e, t, ang, lin) + ang


def actinv_motion(e: Emitter, R: Mat3, t: Sequence[V], m: Sequence[V]) -> Vec:
    """M.actInv(m): lin' = R^T (lin - t x ang); ang' = R^T ang  (motion-dense.hpp:276-282)"""
    tmp = cross_add(e, [-x for x in t], m[3:6], m[0:3])
    return rotT_vec(e, R, tmp) + rotT_vec(e, R, m[3:6])


def act_force(e: Emitter, R: Mat3, t: Sequence[V], f: Sequence[V]) -> Vec:
    """M.act(f): lin' = R lin; ang' = R ang + t x lin'  (spatial/force-dense.hpp:217-223)"""
    lin = rot_vec(e, R, f[0:3])
    ang = rot_vec(e, R, f[3:6])
    return lin + cross_add(e, t, lin, ang)


def motion_cross(e: Emitter, v: Sequence[V], m: Sequence[V]) -> Vec:
    """v x m: lin = v.lin x m.ang + v.ang x m.lin; ang = v.ang x m.ang  (motion-dense.hpp:224-229)"""
    lin = cross_add(e, v[0:3], m[3:6], cross(e, v[3:6], m[0:3]))
    return lin + cross(e, v[3:6], m[3:6])


def force_cross(e: Emitter, v: Sequence[V], f: Sequence[V]) -> Vec:
    """v x* f: lin = v.ang x f.lin; ang = v.ang x f.ang + v.lin x f.lin  (force-dense.hpp:180-185)"""
    lin = cross(e, v[3:6], f[0:3])
    ang = cross_add(e, v[3:6], f[3:6], cross(e, v[0:3], f[0:3]))
    return lin + ang


def sym3_mul(e: Emitter, I: Sequence[V], w: Sequence[V], init: Sequence[V] = (ZERO, ZERO, ZERO)) -> Vec:
    """Symmetric3 (xx, xy, yy, xz, yz, zz) times vector (spatial/symmetric3.hpp:43-52)"""
    return [
        e.dot([I[0], I[1], I[3]], w, init[0]),
        e.dot([I[1], I[2], I[4]], w, init[1]),
        e.dot([I[3], I[4], I[5]], w, init[2]),
    ]


def inertia_mul(e: Emitter, m: V, c: Sequence[V], I: Sequence[V], v: Sequence[V]) -> Vec:
    """Y v: f.lin = m (v.lin - c x v.ang); f.ang = I v.ang + c x f.lin  (spatial/inertia.hpp:741-748)"""
    lin = vscale(e, m, cross_add(e, [-x for x in c], v[3:6], v[0:3]))
    ang = cross_add(e, c, lin, sym3_mul(e, I, v[3:6]))
    return lin + ang


def inertia_to_world(e: Emitter, R: Mat3, t: Sequence[V], c: Sequence[V], I: Sequence[V]):
    """oMi.act(Y) = (m, t + R c, R I R^T)  (spatial/inertia.hpp:885-893, symmetric3.hpp:560-601)"""
    cw = rot_vec(e, R, c, t)
    Isym = [[I[0], I[1], I[3]], [I[1], I[2], I[4]], [I[3], I[4], I[5]]]
    T = rot_mul(e, R, Isym)  # R I
    Iw = [
        e.dot(T[0], R[0]), e.dot(T[0], R[1]), e.dot(T[1], R[1]),
        e.dot(T[0], R[2]), e.dot(T[1], R[2]), e.dot(T[2], R[2]),
    ]
    return cw, Iw


# ---- rigid-body inertia about the world origin: (m, h = m c, I_o); additive over bodies --------------
def inertia_origin(e: Emitter, m: V, c: Sequence[V], I: Sequence[V]):
    """(m, c, I_c) -> h = m c and I_o = I_c - m [c]x [c]x, the blocks of Inertia::matrix()
    (spatial/inertia.hpp:480-491): [[m 1, -[h]x], [[h]x, I_o]]."""
    h = vscale(e, m, c)
    Io = [
        e.add(I[0], e.fma(h[1], c[1], e.mul(h[2], c[2]))), e.fma(-h[0], c[1], I[1]),
        e.add(I[2], e.fma(h[0], c[0], e.mul(h[2], c[2]))), e.fma(-h[0], c[2], I[3]),
        e.fma(-h[1], c[2], I[4]), e.add(I[5], e.fma(h[0], c[0], e.mul(h[1], c[1]))),
    ]
    return h, Io


def rbi_mul(e: Emitter, m: V, h: Sequence[V], Io: Sequence[V], x: Sequence[V]) -> Vec:
    """[[m 1, -[h]x], [[h]x, I_o]] (a, b) = (m a + b x h, h x a + I_o b)"""
    a, b = x[0:3], x[3:6]
    return cross_add(e, b, h, vscale(e, m, a)) + sym3_mul(e, Io, b, cross(e, h, a))


def rbi_variation(e: Emitter, m: V, h: Sequence[V], Io: Sequence[V], v: Sequence[V]):
    """Inertia::variation(v) = crf(v) Y - Y crm(v) (spatial/inertia.hpp:584-603) keeps the block
    structure [[0, -[dh]x], [[dh]x, dI]]: dh = m v.lin + w x h,
    dI = [w]x I_o - I_o [w]x - ([v.lin]x [h]x + [h]x [v.lin]x), symmetric."""
    vl, w = v[0:3], v[3:6]
    dh = cross_add(e, w, h, vscale(e, m, vl))
    rows = [[Io[0], Io[1], Io[3]], [Io[1], Io[2], Io[4]], [Io[3], Io[4], Io[5]]]
    W = [cross(e, w, rows[c]) for c in range(3)]   # W[c][r] = ([w]x I_o)[r][c]
    vh = e.dot(vl, h)
    vh2 = e.add(vh, vh)

    def entry(r, c):
        x = e.add(W[c][r], W[r][c])
        x = e.fma(-h[r], vl[c], e.fma(-vl[r], h[c], x))
        return e.add(x, vh2) if r == c else x

    dI = [entry(0, 0), entry(0, 1), entry(1, 1), entry(0, 2), entry(1, 2), entry(2, 2)]
    return dh, dI


def rbi_dot_mul(e: Emitter, hp: Sequence[V], hm: Sequence[V], dI: Sequence[V], n: Sequence[V],
                x: Sequence[V], transpose: bool = False) -> Vec:
    """(dY + forceCrossMatrix(oh)) x with dY = [[0, -[dh]x], [[dh]x, dI]], oh = (f, n), hp = dh + f,
    hm = dh - f:  (b x hp, hm x a + dI b + b x n); transposed: (b x hm, hp x a + dI b + n x b)."""
    a, b = x[0:3], x[3:6]
    if transpose:
        return cross(e, b, hm) + cross_add(e, hp, a, sym3_mul(e, dI, b, cross(e, n, b)))
    return cross(e, b, hp) + cross_add(e, hm, a, sym3_mul(e, dI, b, cross(e, b, n)))


# ---- packed symmetric 6x6 (21 entries, upper triangle, row-major) -----------------------------------
def sidx(i: int, j: int) -> int:
    if i > j:
        i, j = j, i
    return (i * (13 - i)) // 2 + (j - i)


def inertia_to_sym6(e: Emitter, m: V, c: Sequence[V], I: Sequence[V]) -> Vec:
    """Inertia::matrix() = [[m 1, -m[c]x], [m[c]x, I - m[c]x[c]x]]  (spatial/inertia.hpp:480-491)"""
    A = zeros(21)
    mc = vscale(e, m, c)
    A[sidx(0, 0)] = m
    A[sidx(1, 1)] = m
    A[sidx(2, 2)] = m
    A[sidx(0, 4)] = mc[2]
    A[sidx(0, 5)] = -mc[1]
    A[sidx(1, 3)] = -mc[2]
    A[sidx(1, 5)] = mc[0]
    A[sidx(2, 3)] = mc[1]
    A[sidx(2, 4)] = -mc[0]
    A[sidx(3, 3)] = e.add(I[0], e.fma(mc[1], c[1], e.mul(mc[2], c[2])))
    A[sidx(3, 4)] = e.fma(-mc[0], c[1], I[1])
    A[sidx(3, 5)] = e.fma(-mc[0], c[2], I[3])
    A[sidx(4, 4)] = e.add(I[2], e.fma(mc[0], c[0], e.mul(mc[2], c[2])))
    A[sidx(4, 5)] = e.fma(-mc[1], c[2], I[4])
    A[sidx(5, 5)] = e.add(I[5], e.fma(mc[0], c[0], e.mul(mc[1], c[1])))
    return A


def sym6_mulvec(e: Emitter, A: Sequence[V], x: Sequence[V], init: Sequence[V] = None) -> Vec:
    init = init if init is not None else zeros(6)
    return [e.dot([A[sidx(i, j)] for j in range(6)], x, init[i]) for i in range(6)]


def dot6(e: Emitter, a: Sequence[V], b: Sequence[V], init: V = ZERO) -> V:
    return e.dot(a, b, init)


def quat_to_R(e: Emitter, x: V, y: V, z: V, w: V) -> Mat3:
    """Eigen Quaternion::toRotationMatrix, no normalisation (used at joint-free-flyer.hpp:338-339,
    joint-spherical.hpp:528-529)"""
    two = V(num=2.0)
    tx, ty, tz = e.mul(two, x), e.mul(two, y), e.mul(two, z)
    twx, twy, twz = e.mul(tx, w), e.mul(ty, w), e.mul(tz, w)
    txx, txy, txz = e.mul(tx, x), e.mul(ty, x), e.mul(tz, x)
    tyy, tyz, tzz = e.mul(ty, y), e.mul(tz, y), e.mul(tz, z)
    one = ONE
    return [
        [e.sub(one, e.add(tyy, tzz)), e.sub(txy, twz), e.add(txz, twy)],
        [e.add(txy, twz), e.sub(one, e.add(txx, tzz)), e.sub(tyz, twx)],
        [e.sub(txz, twy), e.add(tyz, twx), e.sub(one, e.add(txx, tyy))],
    ]
