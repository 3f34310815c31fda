"""C-ABI library: loads, exports every symbol include/brbd.h declares, and its host-side logic
(argument validation = the reference's std::invalid_argument checks, flattener) behaves -- no
compute call is made here (no GPU in the CPU tier)."""
import ctypes
import os
import re

import numpy as np
import pytest

from pinocchio_amd import capi, model as M, parallel, sample_models as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "brbd.h")).read()
    declared = set(re.findall(r"\b(brbd_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"brbd_model_desc"}
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    L = capi.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.brbd_version()


def _create(model, device=0):
    desc, keep = model.to_desc()
    h = ctypes.c_void_p()
    rc = capi.lib().brbd_model_create(ctypes.byref(desc), device, ctypes.byref(h))
    return rc, h, capi.last_error()


def test_flattener_rejects_bad_tables():
    m = S.manipulator()
    t = m.tables()
    # parents[i] >= i  (multibody/model.hxx:144-151 requires parent < njoints at insertion)
    bad = m.copy()
    bad.parents[2] = 5
    rc, _, err = _create(bad)
    assert rc == capi.BRBD_ERR_INVALID_ARGUMENT and "parent" in err
    bad = m.copy()
    bad.jtypes[3] = 99
    rc, _, err = _create(bad)
    assert rc == capi.BRBD_ERR_INVALID_ARGUMENT
    bad = m.copy()
    bad.idx_vs[3] = bad.idx_vs[2]
    rc, _, err = _create(bad)
    assert rc == capi.BRBD_ERR_INVALID_ARGUMENT and "idx_v" in err
    rc = capi.lib().brbd_model_create(None, 0, ctypes.byref(ctypes.c_void_p()))
    assert rc == capi.BRBD_ERR_INVALID_ARGUMENT


def test_unaligned_joints_pass_the_flattener(have_gpu):
    m = M.Model("unaligned")
    m.addJoint(0, M.JOINT_REVOLUTE_UNALIGNED, np.eye(4), "j1", axis=[1, 1, 0])
    m.appendBodyToJoint(1, M.Inertia(1.0, [0, 0, 0.1], np.eye(3)))
    rc, h, err = _create(m)
    assert rc == (capi.BRBD_OK if have_gpu else capi.BRBD_ERR_NO_DEVICE), err
    # a zero axis cannot be canonicalised
    d, keep = m.to_desc()
    z = np.zeros((m.njoints, 3))
    d.axis = z.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    out = ctypes.c_void_p()
    assert capi.lib().brbd_model_create(ctypes.byref(d), 0, ctypes.byref(out)) == capi.BRBD_ERR_INVALID_ARGUMENT
    assert "zero axis" in capi.lib().brbd_last_error().decode()


def test_canonical_form_of_unaligned_joints_is_the_same_robot():
    """The flattener turns unaligned joints into RZ / PZ joints of a rotated joint frame
    (brbd_model_canonical_form).  The oracle, which knows unaligned joints natively
    (joint-revolute-unaligned.hpp:667-708), must give the same RNEA / ABA / world Jacobians /
    derivatives for both descriptions, and oMi = oMi_canonical * post."""
    import oracle
    from util import inputs
    kinds = [M.JOINT_RX, M.JOINT_PY, M.JOINT_REVOLUTE_UNALIGNED, M.JOINT_PRISMATIC_UNALIGNED, M.JOINT_SPHERICAL,
             M.JOINT_REVOLUTE_UNALIGNED, M.JOINT_RUBZ, M.JOINT_REVOLUTE_UNBOUNDED_UNALIGNED]
    for seed in range(4):
        m = S.random_tree(500 + seed, 10, kinds=kinds, root_kind=M.JOINT_FREEFLYER if seed % 2 else None)
        assert any(t in (M.JOINT_REVOLUTE_UNALIGNED, M.JOINT_PRISMATIC_UNALIGNED, M.JOINT_REVOLUTE_UNBOUNDED_UNALIGNED) for t in m.jtypes[1:])
        mc, post = capi.canonical_form(m)
        assert not any(t in (M.JOINT_REVOLUTE_UNALIGNED, M.JOINT_PRISMATIC_UNALIGNED, M.JOINT_REVOLUTE_UNBOUNDED_UNALIGNED) for t in mc.jtypes[1:])
        O, Oc = oracle.Oracle(m), oracle.Oracle(mc)
        q, v, a = inputs(m, 6, seed)
        tau = O.rnea(q, v, a)
        assert np.abs(Oc.rnea(q, v, a) - tau).max() < 1e-11 * max(1.0, np.abs(tau).max())
        assert np.abs(Oc.aba(q, v, tau) - a).max() < 1e-9
        J, oMi = O.fk_jacobians(q)
        Jc, oMic = Oc.fk_jacobians(q)
        assert np.abs(J - Jc).max() < 1e-12
        for i in range(1, m.njoints):
            Rc = oMic[12 * i:12 * i + 9].T.reshape(-1, 3, 3).transpose(0, 2, 1)   # column-major -> (B, 3, 3)
            R = oMi[12 * i:12 * i + 9].T.reshape(-1, 3, 3).transpose(0, 2, 1)
            assert np.abs(Rc @ post[i] - R).max() < 1e-12
            assert np.abs(oMic[12 * i + 9:12 * i + 12] - oMi[12 * i + 9:12 * i + 12]).max() < 1e-12
        if m.is_depth_first():
            for x, y in zip(O.rnea_derivatives(q, v, a), Oc.rnea_derivatives(q, v, a)):
                assert np.abs(x - y).max() < 1e-10 * max(1.0, np.abs(x).max())


def test_no_cpu_fallback_without_device(have_gpu):
    """On a box without a GPU model creation must fail loudly, never compute on the CPU."""
    if have_gpu:
        pytest.skip("GPU present")
    rc, _, err = _create(S.manipulator())
    assert rc == capi.BRBD_ERR_NO_DEVICE and "no CPU fallback" in err
    pool = parallel.ModelPool(S.manipulator(), 2)
    with pytest.raises(capi.BrbdLibraryError):
        parallel.rneaInParallel(1, pool, np.zeros((6, 4)), np.zeros((6, 4)), np.zeros((6, 4)))


def test_pool_argument_checks_match_reference():
    """algorithm/parallel/rnea.hpp:53-54,61-68 -> std::invalid_argument before any work."""
    m = S.manipulator()
    z = np.zeros((6, 4))
    with pytest.raises(capi.BrbdInvalidArgument, match="at least one element"):
        parallel.rneaInParallel(1, parallel.ModelPool(m, 0), z, z, z)
    with pytest.raises(capi.BrbdInvalidArgument, match="too small"):
        parallel.abaInParallel(3, parallel.ModelPool(m, 2), z, z, z)
    pool = parallel.ModelPool(m, 4)
    with pytest.raises(capi.BrbdInvalidArgument):
        parallel.rneaInParallel(1, pool, np.zeros((5, 4)), z, z)       # q.rows() != nq
    with pytest.raises(capi.BrbdInvalidArgument):
        parallel.rneaInParallel(1, pool, z, np.zeros((6, 3)), z)       # column mismatch
    with pytest.raises(capi.BrbdInvalidArgument):
        parallel.abaInParallel(1, pool, z, z, np.zeros((7, 4)))        # tau rows (check the reference forgets)
    assert pool.size() == 4 and len(pool.getModels()) == 4
    pool.resize(2)
    assert pool.size() == 2
    with pytest.raises(capi.BrbdInvalidArgument):
        pool.getModel(5)


def test_layout_detection():
    q = np.zeros((6, 4), order="F")
    a = parallel._describe(q, 6, "q")
    assert (a.layout, a.ld) == (capi.INSTANCE_MAJOR, 6)
    a = parallel._describe(np.zeros((6, 4)), 6, "q")
    assert (a.layout, a.ld) == (capi.BATCH_MAJOR, 4)
    # a C-ordered v next to a Fortran-ordered q is copied into q's ordering (Eigen copies too)
    a = parallel._describe(np.zeros((6, 4)), 6, "v", capi.INSTANCE_MAJOR)
    assert (a.layout, a.ld) == (capi.INSTANCE_MAJOR, 6) and a.keep.flags.f_contiguous
    import torch
    t = torch.zeros((4, 6), dtype=torch.float64).t()
    a = parallel._describe(t, 6, "q")
    assert (a.layout, a.ld) == (capi.INSTANCE_MAJOR, 6)
    a = parallel._describe(torch.zeros((6, 4), dtype=torch.float64), 6, "q")
    assert (a.layout, a.ld) == (capi.BATCH_MAJOR, 4)
    with pytest.raises(capi.BrbdInvalidArgument):
        parallel._describe(torch.zeros((6, 4), dtype=torch.float32), 6, "q")


def test_urdf_flattening_rules():
    """parsers/urdf/model.hxx:113-146,331-346,493-527: FF root carries the root link inertia,
    fixed joints merge inertias, aligned axes map to RX/RY/RZ, prismatic to PX/PY/PZ."""
    from pinocchio_amd.urdf import build_model_from_urdf
    urdf = """<robot name="r">
      <link name="base"><inertial><origin xyz="0 0 0.1"/><mass value="2"/>
        <inertia ixx="1" ixy="0" ixz="0" iyy="1" iyz="0" izz="1"/></inertial></link>
      <link name="l1"><inertial><origin xyz="0.1 0 0"/><mass value="1"/>
        <inertia ixx="0.1" ixy="0" ixz="0" iyy="0.2" iyz="0" izz="0.3"/></inertial></link>
      <link name="tool"><inertial><origin xyz="0 0 0.05"/><mass value="0.5"/>
        <inertia ixx="0.01" ixy="0" ixz="0" iyy="0.01" iyz="0" izz="0.01"/></inertial></link>
      <link name="l2"><inertial><origin xyz="0 0 0"/><mass value="1"/>
        <inertia ixx="0.1" ixy="0" ixz="0" iyy="0.1" iyz="0" izz="0.1"/></inertial></link>
      <joint name="b_rev" type="revolute"><parent link="base"/><child link="l1"/>
        <origin xyz="0 0 1" rpy="0 0 1.5707963267948966"/><axis xyz="0 1 0"/></joint>
      <joint name="fix" type="fixed"><parent link="l1"/><child link="tool"/><origin xyz="0.2 0 0"/></joint>
      <joint name="a_pri" type="prismatic"><parent link="base"/><child link="l2"/>
        <origin xyz="1 0 0"/><axis xyz="0 0 1"/></joint>
      <joint name="c_neg" type="revolute"><parent link="tool"/><child link="l3"/>
        <origin xyz="0 0 0.1"/><axis xyz="-1 0 0"/></joint>
      <link name="l3"/>
    </robot>"""
    m = build_model_from_urdf(urdf, free_flyer=True)
    assert m.names[1:] == ["root_joint", "a_pri", "b_rev", "c_neg"]  # siblings in joint-name order
    assert m.jtypes[1:] == [M.JOINT_FREEFLYER, M.JOINT_PZ, M.JOINT_RY, M.JOINT_REVOLUTE_UNALIGNED]
    assert (m.nq, m.nv) == (7 + 3, 6 + 3)
    assert abs(m.inertias[1].mass - 2.0) < 1e-15
    # l1 + tool merged: mass 1.5, com = (1*0.1 + 0.5*0.2)/1.5 in x
    assert abs(m.inertias[3].mass - 1.5) < 1e-15
    assert np.allclose(m.inertias[3].lever, [(0.1 + 0.5 * 0.2) / 1.5, 0, 0.5 * 0.05 / 1.5])
    # c_neg hangs off the fixed `tool` frame: placement = fix origin * joint origin
    assert np.allclose(m.jointPlacements[4][:3, 3], [0.2, 0, 0.1])
    assert np.allclose(m.axes[4], [-1, 0, 0])
    R = m.jointPlacements[3][:3, :3]
    assert np.allclose(R, [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-15)
    assert m.is_depth_first()


def test_json_round_trip():
    m = S.random_tree(5, 9)
    m2 = M.Model.from_json(m.to_json())
    t1, t2 = m.tables(), m2.tables()
    for k in ("parents", "jtype", "idx_q", "idx_v", "placement", "inertia", "axis", "armature", "gravity"):
        assert np.array_equal(t1[k], t2[k]), k


def test_compiled_kernel_signature_carries_the_dof_layout():
    """The generated bodies hard-code the rows of q / v / tau of every joint, so idx_q / idx_v are
    part of the signature that selects them (pinocchio_amd/codegen/joints.py; recomputed on the host
    by model_signature in csrc/brbd_capi.hip)."""
    from pinocchio_amd.codegen import joints
    m = S.arm7()
    p = m.copy()
    for k, j in enumerate(range(m.njoints - 1, 0, -1)):
        p.idx_qs[j] = k
        p.idx_vs[j] = k
    a, b = joints.ModelView(m).signature(), joints.ModelView(p).signature()
    assert a != b and ";q0,1,2,3,4,5,6;v0,1,2,3,4,5,6;" in a and ";q6,5,4,3,2,1,0;" in b
    # the library's registered signatures are the generator's (same text on both sides)
    reg = open(os.path.join(ROOT, "pinocchio_amd", "csrc", "gen", "spec_arm7_p.hip")).read()
    assert a in reg
