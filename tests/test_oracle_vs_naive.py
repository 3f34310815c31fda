"""Pins the C restatement (oracle/brbd_oracle.c) against the structurally independent numpy
longdouble implementation (oracle/naive_np.py, Featherstone body-frame [ang;lin] algebra) at
<= 1e-12 -- the two-implementation pin of SURVEY 8c(2).  (The value pin against the numbers the
reference's own tests hold is tests/test_reference_values.py.)"""
import numpy as np
import pytest

import oracle
from oracle import naive_np as N
from util import parity_models, inputs, rel_err_cols

MODELS = parity_models()


@pytest.mark.parametrize("model", MODELS, ids=[m.name for m in MODELS])
def test_rnea_aba_fk_jac(model):
    O, T = oracle.Oracle(model), N.Tables(model)
    B = 3
    q, v, a = inputs(model, B, 7)
    tau = O.rnea(q, v, a)
    ddq = O.aba(q, v, tau)
    J, oMi = O.fk_jacobians(q)
    for b in range(B):
        tn = np.asarray(N.rnea(T, q[:, b], v[:, b], a[:, b]), dtype=float)
        an = np.asarray(N.aba(T, q[:, b], v[:, b], tau[:, b]), dtype=float)
        assert rel_err_cols(tau[:, b:b + 1], tn[:, None]).max() < 1e-12
        # ABA error is amplified by cond(M): tolerance of the reference's aba tests (1e-12 isApprox,
        # unittest/aba.cpp:154) relaxed by the conditioning of the random trees
        assert rel_err_cols(ddq[:, b:b + 1], an[:, None]).max() < 1e-10
        Jn = np.asarray(N.jacobian_world(T, q[:, b]), dtype=float)
        assert np.abs(J[:, b].reshape(model.nv, 6).T - Jn).max() < 1e-12
        H = N.fk(T, q[:, b])
        oM = oMi[:, b].reshape(model.njoints, 12)
        for i in range(model.njoints):
            Hi = np.asarray(H[i], dtype=float)
            assert np.abs(Hi[:3, :3].T.reshape(9) - oM[i, :9]).max() < 1e-12
            assert np.abs(Hi[:3, 3] - oM[i, 9:]).max() < 1e-12
