"""Oracle vs the committed golden vectors (tests/golden/*.npz, made by tools/gen_golden.py)."""
import os

import numpy as np
import pytest

import oracle
from pinocchio_amd import sample_models as S
from util import rel_err_cols

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["simple_humanoid", "manipulator", "humanoid", "arm7", "quadruped", "talos_like", "baxter_simple"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    m = S.build(name)
    O = oracle.Oracle(m)
    assert rel_err_cols(O.rnea(g["q"], g["v"], g["a"]), g["tau"]).max() < 1e-13
    assert rel_err_cols(O.aba(g["q"], g["v"], g["tau"]), g["ddq"]).max() < 1e-11
    J, oMi = O.fk_jacobians(g["q"])
    assert np.abs(J - g["J"]).max() < 1e-13
    assert np.abs(oMi - g["oMi"]).max() < 1e-13
    if "dtau_dq" in g:
        dq, dv, da, _ = O.rnea_derivatives(g["q"], g["v"], g["a"])
        for x, y in ((dq, g["dtau_dq"]), (dv, g["dtau_dv"]), (da, g["dtau_da"])):
            assert np.abs(x - y).max() < 1e-12 * max(1.0, np.abs(y).max())


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_golden_external_forces_and_joint_jacobians(name):
    g, e = np.load(os.path.join(GOLD, name + ".npz")), np.load(os.path.join(GOLD, name + "_ext.npz"))
    O = oracle.Oracle(S.build(name))
    assert rel_err_cols(O.rnea_fext(g["q"], g["v"], g["a"], e["fext"]), e["tau_fext"]).max() < 1e-13
    assert rel_err_cols(O.aba_fext(g["q"], g["v"], e["tau_fext"], e["fext"]), e["ddq_fext"]).max() < 1e-11
    j = int(e["joint"])
    for rf, key in ((0, "J_world"), (1, "J_local"), (2, "J_lwa")):
        assert np.abs(O.joint_jacobian(g["q"], j, rf) - e[key]).max() < 1e-13


def test_model_tables_are_stable():
    """The committed simple_humanoid table has the dimensions SURVEY section 8 states."""
    m = S.simple_humanoid()
    assert (m.njoints, m.nq, m.nv) == (31, 36, 35)
    assert m.jtypes[1] == 6 and all(t in (0, 1, 2) for t in m.jtypes[2:])
    assert m.is_depth_first()
    assert abs(sum(y.mass for y in m.inertias) - 130.8) < 1e-9  # sum of the URDF link masses
