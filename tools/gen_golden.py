#!/usr/bin/env python3
"""Generate tests/golden/*.npz: small input/output vectors for the built-in models.

The reference holds no golden numbers for this path and cannot be run offline, so the vectors are
captured from the build's own oracle pair: the C restatement (oracle/brbd_oracle.c) is accepted
only where it agrees with the independent numpy-longdouble implementation (oracle/naive_np.py)
to <= 1e-12 (RNEA, FK, J) / 1e-10 (ABA).  "Parity unpinned by value" w.r.t. the reference.
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
    for name in ("simple_humanoid", "manipulator", "arm7", "quadruped", "talos_like", "baxter_simple"):
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


if __name__ == "__main__":
    main()
