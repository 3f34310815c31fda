"""CPU oracle: ctypes front-end of oracle/libbrbd_oracle.so (C restatement of the
reference's rnea / aba / FK+Jacobians / RNEA-derivatives) and the independent
numpy implementation in oracle/naive_np.py.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under pinocchio_amd/ imports this package.
RNEA is pinned by value against the reference's own test vectors (tests/test_reference_values.py);
ABA / Jacobians / derivatives by the reference's invariants (see oracle/brbd_oracle.h).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbrbd_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """gcc-compile oracle/brbd_oracle.c -> oracle/libbrbd_oracle.so (host only)."""
    src = os.path.join(_HERE, "brbd_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libbrbd_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        vp = ctypes.c_void_p
        L.brbd_oracle_rnea.argtypes = [vp, dp, dp, dp, dp]
        L.brbd_oracle_aba_world.argtypes = [vp, dp, dp, dp, dp]
        L.brbd_oracle_aba_local.argtypes = [vp, dp, dp, dp, dp]
        L.brbd_oracle_rnea_fext.argtypes = [vp, dp, dp, dp, dp, dp]
        L.brbd_oracle_aba_fext.argtypes = [vp, dp, dp, dp, dp, ctypes.c_int, dp]
        L.brbd_oracle_joint_jacobian.argtypes = [vp, dp, ctypes.c_int, ctypes.c_int, dp]
        L.brbd_oracle_fk_jacobians.argtypes = [vp, dp, dp, dp]
        L.brbd_oracle_rnea_derivatives.argtypes = [vp, dp, dp, dp, dp, dp, dp, dp]
        i64 = ctypes.c_int64
        L.brbd_oracle_rnea_batch.argtypes = [vp, i64, dp, dp, dp, dp, ctypes.c_int]
        L.brbd_oracle_aba_batch.argtypes = [vp, i64, dp, dp, dp, dp, ctypes.c_int]
        L.brbd_oracle_fk_jacobians_batch.argtypes = [vp, i64, dp, dp, dp, ctypes.c_int]
        L.brbd_oracle_rnea_derivatives_batch.argtypes = [vp, i64, dp, dp, dp, dp, dp, dp, dp, ctypes.c_int]
        L.brbd_oracle_minverse.argtypes = [vp, dp, dp]
        L.brbd_oracle_integrate.argtypes = [vp, dp, dp, dp]
        L.brbd_oracle_aba_derivatives.argtypes = [vp, dp, dp, dp, dp, dp, dp, dp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _cols(x, rows):
    """(rows, B) array -> Fortran-contiguous float64 (instance-major columns)."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x.reshape(rows, 1)
    assert x.shape[0] == rows, (x.shape, rows)
    return np.asfortranarray(x)


class Oracle:
    """Bound to one pinocchio_amd.model.Model."""

    def __init__(self, model):
        self.model = model
        self.desc, self._keep = model.to_desc()
        self.nq, self.nv, self.njoints = model.nq, model.nv, model.njoints
        self._ref = ctypes.byref(self.desc)

    # ---- batched (instance-major (dof,B) in, (dof,B) out) ------------------------
    def rnea(self, q, v, a, threads=1):
        q, v, a = _cols(q, self.nq), _cols(v, self.nv), _cols(a, self.nv)
        B = q.shape[1]
        tau = np.zeros((self.nv, B), order="F")
        rc = lib().brbd_oracle_rnea_batch(self._ref, B, _p(q), _p(v), _p(a), _p(tau), threads)
        assert rc > 0, rc
        return tau

    def aba(self, q, v, tau, threads=1, convention="world"):
        q, v, tau = _cols(q, self.nq), _cols(v, self.nv), _cols(tau, self.nv)
        B = q.shape[1]
        ddq = np.zeros((self.nv, B), order="F")
        if convention == "world":
            rc = lib().brbd_oracle_aba_batch(self._ref, B, _p(q), _p(v), _p(tau), _p(ddq), threads)
            assert rc > 0, rc
        else:
            for b in range(B):
                qb, vb, tb = np.ascontiguousarray(q[:, b]), np.ascontiguousarray(v[:, b]), np.ascontiguousarray(tau[:, b])
                out = np.zeros(self.nv)
                rc = lib().brbd_oracle_aba_local(self._ref, _p(qb), _p(vb), _p(tb), _p(out))
                assert rc == 0, rc
                ddq[:, b] = out
        return ddq

    def rnea_fext(self, q, v, a, fext):
        """rnea(model, data, q, v, a, fext) per column; fext (6*njoints, B), joint frame."""
        q, v, a, f = _cols(q, self.nq), _cols(v, self.nv), _cols(a, self.nv), _cols(fext, 6 * self.njoints)
        tau = np.zeros((self.nv, q.shape[1]), order="F")
        for b in range(q.shape[1]):
            cols = [np.ascontiguousarray(x[:, b]) for x in (q, v, a, f)]
            out = np.zeros(self.nv)
            assert lib().brbd_oracle_rnea_fext(self._ref, *[_p(c) for c in cols], _p(out)) == 0
            tau[:, b] = out
        return tau

    def aba_fext(self, q, v, tau, fext, convention="world"):
        q, v, tau, f = _cols(q, self.nq), _cols(v, self.nv), _cols(tau, self.nv), _cols(fext, 6 * self.njoints)
        ddq = np.zeros((self.nv, q.shape[1]), order="F")
        for b in range(q.shape[1]):
            cols = [np.ascontiguousarray(x[:, b]) for x in (q, v, tau, f)]
            out = np.zeros(self.nv)
            rc = lib().brbd_oracle_aba_fext(self._ref, *[_p(c) for c in cols], 0 if convention == "world" else 1, _p(out))
            assert rc == 0, rc
            ddq[:, b] = out
        return ddq

    def joint_jacobian(self, q, joint, rf):
        """getJointJacobian(joint, rf) per column: (6*nv, B); rf 0 WORLD, 1 LOCAL, 2 LOCAL_WORLD_ALIGNED."""
        q = _cols(q, self.nq)
        J = np.zeros((6 * self.nv, q.shape[1]), order="F")
        for b in range(q.shape[1]):
            qb, out = np.ascontiguousarray(q[:, b]), np.zeros(6 * self.nv)
            assert lib().brbd_oracle_joint_jacobian(self._ref, _p(qb), joint, rf, _p(out)) == 0
            J[:, b] = out
        return J

    def fk_jacobians(self, q, threads=1):
        """Returns J (6*nv, B) [per instance 6 x nv column-major] and oMi (12*njoints, B)."""
        q = _cols(q, self.nq)
        B = q.shape[1]
        J = np.zeros((6 * self.nv, B), order="F")
        oMi = np.zeros((12 * self.njoints, B), order="F")
        rc = lib().brbd_oracle_fk_jacobians_batch(self._ref, B, _p(q), _p(J), _p(oMi), threads)
        assert rc > 0, rc
        return J, oMi

    def rnea_derivatives(self, q, v, a, threads=1):
        """Returns dtau_dq, dtau_dv, dtau_da as (nv*nv, B) (column-major nv x nv each), tau (nv,B)."""
        q, v, a = _cols(q, self.nq), _cols(v, self.nv), _cols(a, self.nv)
        B = q.shape[1]
        nn = self.nv * self.nv
        dq, dv, da = (np.zeros((nn, B), order="F") for _ in range(3))
        tau = np.zeros((self.nv, B), order="F")
        rc = lib().brbd_oracle_rnea_derivatives_batch(self._ref, B, _p(q), _p(v), _p(a), _p(dq), _p(dv),
                                                      _p(da), _p(tau), threads)
        assert rc > 0, rc
        return dq, dv, da, tau


def _per_column(fn, ins, out_rows):
    """Run a single-instance oracle function over the columns of (rows, B) inputs."""
    B = ins[0].shape[1]
    outs = [np.zeros((r, B), order="F") for r in out_rows]
    for b in range(B):
        cols = [np.ascontiguousarray(x[:, b]) for x in ins]
        res = [np.zeros(r) for r in out_rows]
        rc = fn(*[_p(c) for c in cols], *[_p(r) for r in res])
        assert rc == 0, rc
        for o, r in zip(outs, res):
            o[:, b] = r
    return outs


def _minverse(self, q):
    """computeMinverse per column: (nv*nv, B), per instance column-major, upper triangle."""
    q = _cols(q, self.nq)
    return _per_column(lambda *a: lib().brbd_oracle_minverse(self._ref, *a), [q], [self.nv * self.nv])[0]


def _integrate(self, q, v):
    """integrate(model, q, v) per column: (nq, B)."""
    q, v = _cols(q, self.nq), _cols(v, self.nv)
    return _per_column(lambda *a: lib().brbd_oracle_integrate(self._ref, *a), [q, v], [self.nq])[0]


def _aba_derivatives(self, q, v, tau):
    """computeABADerivatives per column: ddq_dq, ddq_dv, ddq_dtau (= Minv, full) as (nv*nv, B), ddq (nv, B)."""
    q, v, tau = _cols(q, self.nq), _cols(v, self.nv), _cols(tau, self.nv)
    nn = self.nv * self.nv
    return _per_column(lambda *a: lib().brbd_oracle_aba_derivatives(self._ref, *a), [q, v, tau], [nn, nn, nn, self.nv])


Oracle.minverse = _minverse
Oracle.integrate = _integrate
Oracle.aba_derivatives = _aba_derivatives


def max_threads() -> int:
    return int(lib().brbd_oracle_max_threads())
