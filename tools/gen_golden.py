#!/usr/bin/env python3
"""Generate tests/golden/*.npz: small input/output vectors for the built-in models.

These are REGRESSION vectors, not the value pin: the reference cannot be run offline, so they are
captured from the build's own oracle pair: the C restatement (oracle/brbd_oracle.c) is accepted
only where it agrees with the independent numpy-longdouble implementation (oracle/naive_np.py)
to <= 1e-12 (RNEA, FK, J) / 1e-10 (ABA).  The numbers the reference itself holds for this path
(unittest/value.cpp, unittest/joint-prismatic.cpp) are a separate fixture,
tests/golden/reference_value_cpp.json (tools/gen_reference_fixture.py, tests/test_reference_values.py).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import naive_np as N  # noqa: E402
from pinocchio_amd import sample_models as S  # noqa: E402

B = 16


def rel(x, y):
    return np.linalg.norm(np.asarray(x, float) - np.asarray(y, float)) / max(np.linalg.norm(np.asarray(y, float)), 1e-300)


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    names = sys.argv[1:] or ("simple_humanoid", "manipulator", "humanoid", "arm7", "quadruped", "talos_like", "baxter_simple")
    for name in names:  # (name some models on the command line to regenerate only those)
        m = S.build(name)
        O, T = oracle.Oracle(m), N.Tables(m)
        rng = np.random.default_rng(2025)
        q = m.random_configuration(rng, B)
        v = rng.uniform(-1, 1, (m.nv, B))
        a = rng.uniform(-1, 1, (m.nv, B))
        tau = O.rnea(q, v, a)
        ddq = O.aba(q, v, tau)
        J, oMi = O.fk_jacobians(q)
        for b in range(B):
            assert rel(tau[:, b], N.rnea(T, q[:, b], v[:, b], a[:, b])) < 1e-12
            assert rel(ddq[:, b], N.aba(T, q[:, b], v[:, b], tau[:, b])) < 1e-10
            assert np.abs(J[:, b].reshape(m.nv, 6).T - np.asarray(N.jacobian_world(T, q[:, b]), float)).max() < 1e-12
        d = dict(q=q, v=v, a=a, tau=tau, ddq=ddq, J=J, oMi=oMi)
        if m.nv <= 20:
            dq, dv, da, _ = O.rnea_derivatives(q, v, a)
            d.update(dtau_dq=dq, dtau_dv=dv, dtau_da=da)
        np.savez_compressed(os.path.join(out, name + ".npz"), **d)
        print(name, {k: x.shape for k, x in d.items()})
        # external forces and joint Jacobians (<name>_ext.npz).  Accepted only where the two agree with
        # each other through tau(fext) = tau(0) - sum_i J_i(LOCAL)^T fext_i, and aba inverts rnea.
        fext = rng.uniform(-2, 2, (6 * m.njoints, B))
        tau_f = O.rnea_fext(q, v, a, fext)
        ddq_f = O.aba_fext(q, v, tau_f, fext)
        acc = tau.copy()
        for i in range(1, m.njoints):
            Jl = O.joint_jacobian(q, i, 1)
            for b in range(B):
                acc[:, b] -= Jl[:, b].reshape(m.nv, 6) @ fext[6 * i:6 * i + 6, b]
        assert np.abs(acc - tau_f).max() < 1e-11 * max(1.0, np.abs(tau_f).max()), name
        assert np.abs(ddq_f - a).max() < 1e-9, name
        assert np.abs(O.aba_fext(q, v, tau_f, fext, "local") - ddq_f).max() < 1e-9, name
        jl = m.njoints - 1
        e = dict(fext=fext, tau_fext=tau_f, ddq_fext=ddq_f, joint=np.array(jl),
                 J_world=O.joint_jacobian(q, jl, 0), J_local=O.joint_jacobian(q, jl, 1),
                 J_lwa=O.joint_jacobian(q, jl, 2))
        np.savez_compressed(os.path.join(out, name + "_ext.npz"), **e)


if __name__ == "__main__":
    main()
