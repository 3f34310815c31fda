"""Parity tests proper: the HIP path, called through the C ABI (pinocchio_amd.capi /
pinocchio_amd.parallel are thin ctypes plumbing over include/brbd.h), against the CPU oracle on
the same seeded inputs, against the committed golden vectors, and -- at BASELINE.json's full
sizes -- through size-independent properties.

Tolerances (north_star: <= 1e-10 relative in fp64), Eigen isApprox semantics per instance
(||x - y|| <= tol * min(||x||, ||y||), as in the reference's tests):
  RNEA 1e-12, FK / Jacobians 1e-12 (absolute, entries are O(1)), ABA 1e-10, RNEA derivatives 1e-10.
"""
import ctypes
import os

import numpy as np
import pytest

import oracle
from pinocchio_amd import capi, model as M, parallel, sample_models as S
from util import inputs, parity_models, rel_err_cols

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODELS = parity_models()
IDS = [m.name for m in MODELS]
TOL_RNEA, TOL_KIN, TOL_ABA, TOL_DERIV = 1e-12, 1e-12, 1e-10, 1e-10


def _mat_rel(x, y):
    return np.abs(x - y).max() / max(1.0, np.abs(y).max())


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    return torch


@pytest.mark.parametrize("order", ["F", "C"])
@pytest.mark.parametrize("model", MODELS, ids=IDS)
def test_all_algorithms_vs_oracle_host_memspace(model, order):
    """Host pointers (the reference's calling convention) in both boundary layouts; B = 200 is
    ragged (3 full waves + 8 lanes)."""
    pool = parallel.ModelPool(model, 1)
    O = oracle.Oracle(model)
    q, v, a = inputs(model, 200, 11)
    qq, vv, aa = (np.asarray(x, order=order) for x in (q, v, a))
    tau_ref = O.rnea(q, v, a)
    tau = parallel.rneaInParallel(1, pool, qq, vv, aa)
    assert tau.flags["F_CONTIGUOUS" if order == "F" else "C_CONTIGUOUS"]
    assert rel_err_cols(tau, tau_ref).max() < TOL_RNEA
    ddq = parallel.abaInParallel(1, pool, qq, vv, np.asarray(tau_ref, order=order))
    assert rel_err_cols(ddq, O.aba(q, v, tau_ref)).max() < TOL_ABA
    assert rel_err_cols(ddq, a).max() < TOL_ABA  # unittest/aba.cpp:143-154
    J, oMi = parallel.computeJointJacobians(pool, qq, want_oMi=True)
    Jr, Mr = O.fk_jacobians(q)
    assert np.abs(J - Jr).max() < TOL_KIN and np.abs(oMi - Mr).max() < TOL_KIN
    assert np.abs(parallel.forwardKinematics(pool, qq) - Mr).max() < TOL_KIN
    if model.is_depth_first():
        dq, dv, da, tau2 = parallel.computeRNEADerivatives(pool, qq, vv, aa, want_tau=True)
        rq, rv, ra, _ = O.rnea_derivatives(q, v, a)
        assert _mat_rel(dq, rq) < TOL_DERIV and _mat_rel(dv, rv) < TOL_DERIV and _mat_rel(da, ra) < TOL_DERIV
        assert rel_err_cols(tau2, tau_ref).max() < TOL_RNEA
    pool.release()


@pytest.mark.parametrize("name", ["simple_humanoid", "manipulator", "humanoid", "arm7", "quadruped", "talos_like", "baxter_simple"])
def test_golden_vectors(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    pool = parallel.ModelPool(S.build(name), 1)
    assert rel_err_cols(parallel.rneaInParallel(1, pool, g["q"], g["v"], g["a"]), g["tau"]).max() < TOL_RNEA
    assert rel_err_cols(parallel.abaInParallel(1, pool, g["q"], g["v"], g["tau"]), g["ddq"]).max() < TOL_ABA
    J, oMi = parallel.computeJointJacobians(pool, g["q"], want_oMi=True)
    assert np.abs(J - g["J"]).max() < TOL_KIN and np.abs(oMi - g["oMi"]).max() < TOL_KIN
    e = np.load(os.path.join(GOLD, name + "_ext.npz"))
    assert rel_err_cols(parallel.rnea(pool, g["q"], g["v"], g["a"], e["fext"]), e["tau_fext"]).max() < TOL_RNEA
    assert rel_err_cols(parallel.aba(pool, g["q"], g["v"], e["tau_fext"], e["fext"]), e["ddq_fext"]).max() < TOL_ABA
    for rf, key in ((capi.WORLD, "J_world"), (capi.LOCAL, "J_local"), (capi.LOCAL_WORLD_ALIGNED, "J_lwa")):
        assert np.abs(parallel.getJointJacobian(pool, g["q"], int(e["joint"]), rf) - e[key]).max() < TOL_KIN
    if "dtau_dq" in g:
        dq, dv, da = parallel.computeRNEADerivatives(pool, g["q"], g["v"], g["a"])
        assert _mat_rel(dq, g["dtau_dq"]) < TOL_DERIV
        assert _mat_rel(dv, g["dtau_dv"]) < TOL_DERIV
        assert _mat_rel(da, g["dtau_da"]) < TOL_DERIV
    pool.release()


@pytest.mark.parametrize("B", [1, 63, 64, 65, 129])
def test_ragged_batches(B):
    model = S.quadruped()
    pool = parallel.ModelPool(model, 1)
    O = oracle.Oracle(model)
    q, v, a = inputs(model, B, B)
    tau = parallel.rneaInParallel(1, pool, q, v, a)
    assert tau.shape == (model.nv, B)
    assert rel_err_cols(tau, O.rnea(q, v, a)).max() < TOL_RNEA
    assert rel_err_cols(parallel.abaInParallel(1, pool, q, v, tau), a).max() < TOL_ABA
    pool.release()


def test_empty_batch():
    model = S.manipulator()
    pool = parallel.ModelPool(model, 1)
    z = np.zeros((6, 0))
    assert parallel.rneaInParallel(1, pool, z, z, z).shape == (6, 0)
    assert parallel.abaInParallel(1, pool, z, z, z).shape == (6, 0)
    pool.release()


def test_empty_batch_new_entry_points():
    model = S.manipulator()
    pool = parallel.ModelPool(model, 1)
    z = np.zeros((6, 0))
    f = np.zeros((6 * model.njoints, 0))
    assert parallel.rnea(pool, z, z, z, f).shape == (6, 0)
    assert parallel.aba(pool, z, z, z, f).shape == (6, 0)
    assert parallel.getJointJacobian(pool, z, 3, capi.WORLD).shape == (36, 0)
    with pytest.raises(capi.BrbdInvalidArgument):   # fext must have 6 rows per joint, universe included
        parallel.rnea(pool, np.zeros((6, 4)), np.zeros((6, 4)), np.zeros((6, 4)), np.zeros((6 * (model.njoints - 1), 4)))
    pool.release()


def test_in_place_overloads_and_mixed_orders():
    """pin.rneaInParallel(n, pool, q, v, a, tau) / abaInParallel(..., a) write the caller's array
    (bindings/python/algorithm/parallel/rnea.cpp:44-63)."""
    model = S.arm7()
    pool = parallel.ModelPool(model, 2)
    O = oracle.Oracle(model)
    q, v, a = inputs(model, 40, 3)
    tau = np.full((model.nv, 40), np.nan, order="F")
    ret = parallel.rneaInParallel(2, pool, q, np.ascontiguousarray(v), a, tau)
    assert ret is tau and rel_err_cols(tau, O.rnea(q, v, a)).max() < TOL_RNEA
    out_c = np.full((model.nv, 40), np.nan)  # C-ordered output next to Fortran-ordered inputs
    parallel.abaInParallel(1, pool, q, v, tau, out_c)
    assert rel_err_cols(out_c, a).max() < TOL_ABA
    pool.release()


@pytest.mark.parametrize("layout", ["batch_major", "instance_major"])
def test_device_memspace_with_padding(torch_mod, layout):
    """Device-resident tensors, used in place, with leading dimensions larger than needed."""
    torch = torch_mod
    model = S.simple_humanoid()
    pool = parallel.ModelPool(model, 1)
    O = oracle.Oracle(model)
    B, pad = 300, 17
    q, v, a = inputs(model, B, 5)
    dev = torch.device("cuda", 0)

    def put(x):
        rows = x.shape[0]
        if layout == "batch_major":
            big = torch.full((rows, B + pad), float("nan"), dtype=torch.float64, device=dev)
            big[:, :B] = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
            return big, big[:, :B]
        big = torch.full((B, rows + pad), float("nan"), dtype=torch.float64, device=dev)
        big[:, :rows] = torch.from_numpy(np.ascontiguousarray(x.T)).to(dev)
        return big, big[:, :rows].t()

    (_, qd), (_, vd), (_, ad) = put(q), put(v), put(a)
    big_t, tau = put(np.zeros((model.nv, B)))
    parallel.rneaInParallel(1, pool, qd, vd, ad, tau)
    torch.cuda.synchronize()
    assert rel_err_cols(tau.cpu().numpy(), O.rnea(q, v, a)).max() < TOL_RNEA
    # padding untouched
    if layout == "batch_major":
        assert torch.isnan(big_t[:, B:]).all()
    else:
        assert torch.isnan(big_t[:, model.nv:]).all()
    ddq = parallel.abaInParallel(1, pool, qd, vd, tau)
    torch.cuda.synchronize()
    assert rel_err_cols(ddq.cpu().numpy(), a).max() < TOL_ABA
    pool.release()


def test_full_size_headline_properties(torch_mod):
    """BASELINE.json configs[2] size (simple_humanoid, batch 65536): size-independent properties.
    (1) aba(q, v, rnea(q, v, a)) == a per instance (unittest/aba.cpp:143-154);
    (2) rnea is affine in a: rnea(a1 + a2) - rnea(a1) - rnea(a2) + rnea(0) == 0 (rnea.cpp:198-206);
    (3) 256 randomly chosen instances match the oracle; (4) relaunch is bitwise reproducible."""
    torch = torch_mod
    model = S.simple_humanoid()
    pool = parallel.ModelPool(model, 1)
    B = 65536
    q, v, a = inputs(model, B, 65536)
    dev = torch.device("cuda", 0)
    qd, vd, ad = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (q, v, a))
    tau = parallel.rneaInParallel(1, pool, qd, vd, ad)
    ddq = parallel.abaInParallel(1, pool, qd, vd, tau)
    err = ((ddq - ad).norm(dim=0) / ad.norm(dim=0)).max().item()
    assert err < TOL_ABA, err
    a2 = torch.flip(ad, dims=[1])
    lin = (parallel.rneaInParallel(1, pool, qd, vd, ad + a2) - tau
           - parallel.rneaInParallel(1, pool, qd, vd, a2) + parallel.rneaInParallel(1, pool, qd, vd, 0 * ad))
    assert (lin.norm(dim=0) / tau.norm(dim=0)).max().item() < 1e-12
    idx = np.random.default_rng(0).choice(B, 256, replace=False)
    O = oracle.Oracle(model)
    tr = O.rnea(q[:, idx], v[:, idx], a[:, idx], threads=0)
    assert rel_err_cols(tau[:, idx].cpu().numpy(), tr).max() < TOL_RNEA
    assert rel_err_cols(ddq[:, idx].cpu().numpy(), O.aba(q[:, idx], v[:, idx], tr, threads=0)).max() < TOL_ABA
    assert torch.equal(parallel.abaInParallel(1, pool, qd, vd, tau), ddq)
    pool.release()


def test_full_size_fk_jacobians_arm(torch_mod):
    """BASELINE.json configs[1]: 7-dof arm FK + Jacobians, batch 1M: more tiles than resident
    waves (grid-stride path).  Properties: J columns are unit twists (|angular| == 1), the
    rotation of every oMi is orthonormal; 128 sampled instances match the oracle."""
    torch = torch_mod
    model = S.arm7()
    pool = parallel.ModelPool(model, 1)
    B = 1 << 20
    rng = np.random.default_rng(1)
    q = rng.uniform(-1, 1, (model.nq, B))
    qd = torch.from_numpy(q).to("cuda:0")
    J, oMi = parallel.computeJointJacobians(pool, qd, want_oMi=True)
    Jv = J.view(model.nv, 6, B)
    assert (Jv[:, 3:, :].norm(dim=1) - 1).abs().max().item() < 1e-13
    Rm = oMi.view(model.njoints, 12, B)[1:, :9, :].reshape(-1, 3, 3, B)  # column-major: [col][row]
    gram = torch.einsum("jcrb,jdrb->jcdb", Rm, Rm)
    eye = torch.eye(3, dtype=torch.float64, device=gram.device)[None, :, :, None]
    assert (gram - eye).abs().max().item() < 1e-13
    idx = rng.choice(B, 128, replace=False)
    Jr, Mr = oracle.Oracle(model).fk_jacobians(q[:, idx])
    assert np.abs(J[:, idx].cpu().numpy() - Jr).max() < TOL_KIN
    assert np.abs(oMi[:, idx].cpu().numpy() - Mr).max() < TOL_KIN
    pool.release()


def test_derivatives_vs_finite_differences_on_device(torch_mod):
    """unittest/rnea-derivatives.cpp:158-303 through the HIP path only: dtau/dv by forward
    differences of the HIP RNEA (alpha = 1e-8, tolerance sqrt(alpha)); dtau/da symmetric upper
    triangle == M built from HIP RNEA columns (:293-299)."""
    model = S.quadruped()
    pool = parallel.ModelPool(model, 1)
    nv = model.nv
    q, v, a = inputs(model, 1, 9)
    dq, dv, da = parallel.computeRNEADerivatives(pool, q, v, a)
    dv, da = dv[:, 0].reshape(nv, nv).T, da[:, 0].reshape(nv, nv).T
    alpha = 1e-8
    Q = np.repeat(q, nv, axis=1)
    V = np.repeat(v, nv, axis=1) + alpha * np.eye(nv)
    Aa = np.repeat(a, nv, axis=1)
    tau0 = parallel.rneaInParallel(1, pool, q, v, a)
    fdv = (parallel.rneaInParallel(1, pool, np.asfortranarray(Q), np.asfortranarray(V), np.asfortranarray(Aa)) - tau0) / alpha
    assert np.linalg.norm(dv - fdv) < np.sqrt(alpha) * max(1.0, np.linalg.norm(fdv))
    Z = np.zeros((nv, nv), order="F")
    Mm = (parallel.rneaInParallel(1, pool, np.asfortranarray(Q), Z, np.asfortranarray(np.eye(nv)))
          - parallel.rneaInParallel(1, pool, np.asfortranarray(Q), Z, Z))
    assert np.abs(np.triu(da) - np.triu(Mm)).max() < 1e-11 * max(1.0, np.abs(Mm).max())
    pool.release()


def test_c_abi_error_codes(torch_mod):
    torch = torch_mod
    L = capi.lib()
    model = S.manipulator()
    dm = capi.DeviceModel(model, 0)
    x = torch.zeros((6, 8), dtype=torch.float64, device="cuda:0")
    p = x.data_ptr()
    # NULL pointer / too-small leading dimension / bad layout -> INVALID_ARGUMENT, nothing launched
    assert L.brbd_rnea(dm.handle, 8, None, 8, p, 8, p, 8, p, 8, capi.BATCH_MAJOR, capi.DEVICE, None) == capi.BRBD_ERR_INVALID_ARGUMENT
    assert "q is NULL" in capi.last_error()
    assert L.brbd_rnea(dm.handle, 8, p, 4, p, 8, p, 8, p, 8, capi.BATCH_MAJOR, capi.DEVICE, None) == capi.BRBD_ERR_INVALID_ARGUMENT
    assert L.brbd_aba(dm.handle, 8, p, 8, p, 8, p, 3, p, 8, capi.BATCH_MAJOR, capi.DEVICE, None) == capi.BRBD_ERR_INVALID_ARGUMENT
    assert "tau" in capi.last_error()
    assert L.brbd_rnea(dm.handle, 8, p, 8, p, 8, p, 8, p, 8, 7, capi.DEVICE, None) == capi.BRBD_ERR_INVALID_ARGUMENT
    assert L.brbd_rnea(dm.handle, -1, p, 8, p, 8, p, 8, p, 8, capi.BATCH_MAJOR, capi.DEVICE, None) == capi.BRBD_ERR_INVALID_ARGUMENT
    assert L.brbd_model_create(ctypes.byref(model.to_desc()[0]), 99, ctypes.byref(ctypes.c_void_p())) == capi.BRBD_ERR_INVALID_ARGUMENT
    assert dm.workspace_bytes(capi.ALGO_ABA, 65536) > 0
    dm.close()
    # RNEA derivatives need depth-first numbering (Data::computeNvSubtree, data.hxx:205-230)
    m = M.Model("not_dfs")
    j1 = m.addJoint(0, M.JOINT_RX, np.eye(4), "a")
    j2 = m.addJoint(0, M.JOINT_RY, np.eye(4), "b")
    j3 = m.addJoint(j1, M.JOINT_RZ, np.eye(4), "c")
    for j in (j1, j2, j3):
        m.appendBodyToJoint(j, M.Inertia(1.0, [0.1, 0, 0], np.eye(3)))
    assert not m.is_depth_first()
    pool = parallel.ModelPool(m, 1)
    z = np.zeros((3, 4))
    with pytest.raises(capi.BrbdUnsupported):
        parallel.computeRNEADerivatives(pool, z, z, z)
    # ...while RNEA / ABA handle any parents[i] < i ordering
    O = oracle.Oracle(m)
    q, v, a = inputs(m, 32, 2)
    tau = parallel.rneaInParallel(1, pool, q, v, a)
    assert rel_err_cols(tau, O.rnea(q, v, a)).max() < TOL_RNEA
    assert rel_err_cols(parallel.abaInParallel(1, pool, q, v, tau), a).max() < TOL_ABA
    pool.release()


def test_transposes_and_kernel_timing(torch_mod):
    torch = torch_mod
    L = capi.lib()
    dof, B = 37, 1000
    x = torch.randn((B, dof), dtype=torch.float64, device="cuda:0")  # instance-major
    y = torch.empty((dof, B), dtype=torch.float64, device="cuda:0")
    z = torch.empty_like(x)
    capi.check(L.brbd_transpose_to_batch_major(0, dof, B, x.data_ptr(), dof, y.data_ptr(), B, None))
    capi.check(L.brbd_transpose_to_instance_major(0, dof, B, y.data_ptr(), B, z.data_ptr(), dof, None))
    torch.cuda.synchronize()
    assert torch.equal(y, x.t()) and torch.equal(z, x)
    model = S.quadruped()
    pool = parallel.ModelPool(model, 1)
    dm = pool.device_model(0)
    q, v, a = inputs(model, 4096, 1)
    qd, vd, ad = (torch.from_numpy(np.ascontiguousarray(t)).to("cuda:0") for t in (q, v, a))
    dm.set_timing(True)
    for _ in range(3):
        parallel.rneaInParallel(1, pool, qd, vd, ad)
    n, ms = dm.kernel_time(capi.ALGO_RNEA)
    dm.set_timing(False)
    assert n == 3 and 0 < ms < 1000
    assert dm.kernel_time(capi.ALGO_RNEA) == (0, 0.0)
    pool.release()


# ---- model-compiled kernels (pinocchio_amd/codegen, csrc/brbd_spec.hpp) --------------------------
BUILTIN_NAMES = ["simple_humanoid", "manipulator", "humanoid", "arm7", "quadruped", "talos_like", "baxter_simple"]


def _perturbed(model):
    """Same tree and joint kinds, every placement / inertia generic: must select the dense kernels."""
    rng = np.random.default_rng(5)
    m = model.copy()
    for i in range(1, m.njoints):
        a = rng.uniform(-0.3, 0.3, 3)
        ca, sa = np.cos(a), np.sin(a)
        Rx = np.array([[1, 0, 0], [0, ca[0], -sa[0]], [0, sa[0], ca[0]]])
        Ry = np.array([[ca[1], 0, sa[1]], [0, 1, 0], [-sa[1], 0, ca[1]]])
        Rz = np.array([[ca[2], -sa[2], 0], [sa[2], ca[2], 0], [0, 0, 1]])
        P = m.jointPlacements[i].copy()
        P[:3, :3] = P[:3, :3] @ Rx @ Ry @ Rz
        P[:3, 3] += rng.uniform(0.01, 0.05, 3)
        m.jointPlacements[i] = P
        Y = m.inertias[i]
        A_ = rng.uniform(-1, 1, (3, 3))
        m.inertias[i] = M.Inertia(Y.mass * 1.1 + 0.1, Y.lever + rng.uniform(0.01, 0.02, 3),
                                  Y.inertia + A_ @ A_.T * 0.01)
    return m


def _check_all(pool, model, O, B=130, seed=17):
    """Every algorithm in both boundary layouts (each is a kernel of its own) against the oracle."""
    q, v, a = inputs(model, B, seed)
    tau_ref = O.rnea(q, v, a)
    ddq_ref = O.aba(q, v, tau_ref)
    Jr, Mr = O.fk_jacobians(q)
    der = O.rnea_derivatives(q, v, a) if model.is_depth_first() else None
    for order in "FC":
        qq, vv, aa, tt = (np.asarray(x, order=order) for x in (q, v, a, tau_ref))
        if der is not None:
            dq, dv, da, tau2 = parallel.computeRNEADerivatives(pool, qq, vv, aa, want_tau=True)
            assert _mat_rel(dq, der[0]) < TOL_DERIV and _mat_rel(dv, der[1]) < TOL_DERIV
            assert _mat_rel(da, der[2]) < TOL_DERIV and rel_err_cols(tau2, tau_ref).max() < TOL_RNEA
            dq, dv, da = parallel.computeRNEADerivatives(pool, qq, vv, aa)  # (no tau: stores dropped)
            assert _mat_rel(dq, der[0]) < TOL_DERIV and _mat_rel(dv, der[1]) < TOL_DERIV
        assert rel_err_cols(parallel.rneaInParallel(1, pool, qq, vv, aa), tau_ref).max() < TOL_RNEA
        assert rel_err_cols(parallel.abaInParallel(1, pool, qq, vv, tt), ddq_ref).max() < TOL_ABA
        J, oMi = parallel.computeJointJacobians(pool, qq, want_oMi=True)
        assert np.abs(J - Jr).max() < TOL_KIN and np.abs(oMi - Mr).max() < TOL_KIN
        assert np.abs(parallel.computeJointJacobians(pool, qq) - Jr).max() < TOL_KIN  # J only
        assert np.abs(parallel.forwardKinematics(pool, qq) - Mr).max() < TOL_KIN      # oMi only


@pytest.mark.parametrize("name", BUILTIN_NAMES)
def test_kernel_selection_pattern_dense_table(name):
    """Registered models run on the kernels compiled for them ("_p": zero pattern folded); a model
    with the same tree but generic constants selects the dense kernels ("_d"); forcing the
    table-driven kernels gives the same results.  All three against the oracle."""
    model = S.build(name)
    pool = parallel.ModelPool(model, 1)
    dm = pool.device_model(0)
    # baxter_simple and the reference's test humanoid are not registered by default (codegen/build.py: REGISTERED)
    assert dm.compiled_kernels() == (name + "_p" if name not in ("baxter_simple", "humanoid") else "")
    O = oracle.Oracle(model)
    _check_all(pool, model, O)
    dm.use_compiled(False)
    _check_all(pool, model, O)
    dm.use_compiled(True)
    pool.release()
    pm = _perturbed(model)
    pool = parallel.ModelPool(pm, 1)
    # dense kernels are only built on request (BRBD_SPEC_VARIANTS=p,d); otherwise table-driven
    assert pool.device_model(0).compiled_kernels() in (name + "_d", "")
    _check_all(pool, pm, oracle.Oracle(pm))
    pool.release()


@pytest.mark.parametrize("name", ["simple_humanoid", "quadruped", "arm7", "baxter_simple", "unaligned_tree"])
def test_external_forces(torch_mod, name):
    """brbd_rnea_fext / brbd_aba_fext (rnea.hxx:171-219, aba.hxx:296-354): compiled and table-driven
    kernels, both layouts, host and device memory, against the oracle; aba(rnea(a, fext), fext) == a."""
    torch = torch_mod
    if name == "unaligned_tree":
        model = S.random_tree(502, 10, kinds=[M.JOINT_RX, M.JOINT_REVOLUTE_UNALIGNED, M.JOINT_PRISMATIC_UNALIGNED,
                                              M.JOINT_SPHERICAL, M.JOINT_REVOLUTE_UNBOUNDED_UNALIGNED])
    else:
        model = S.build(name)
    pool = parallel.ModelPool(model, 1)
    dm = pool.device_model(0)
    O = oracle.Oracle(model)
    B = 150
    q, v, a = inputs(model, B, 31)
    f = np.random.default_rng(31).uniform(-2, 2, (6 * model.njoints, B))
    tau_ref = O.rnea_fext(q, v, a, f)
    assert np.abs(tau_ref - O.rnea(q, v, a)).max() > 1e-2
    for compiled in (True, False):
        dm.use_compiled(compiled)
        for order in "FC":
            qq, vv, aa, ff, tt = (np.asarray(x, order=order) for x in (q, v, a, f, tau_ref))
            tau = parallel.rnea(pool, qq, vv, aa, ff)
            assert rel_err_cols(tau, tau_ref).max() < TOL_RNEA
            ddq = parallel.aba(pool, qq, vv, tt, ff)
            assert rel_err_cols(ddq, a).max() < TOL_ABA
            # plain calls are unaffected by the variants
            assert rel_err_cols(parallel.rneaInParallel(1, pool, qq, vv, aa), O.rnea(q, v, a)).max() < TOL_RNEA
    dm.use_compiled(True)
    dev = torch.device("cuda", 0)
    qd, vd, ad, fd = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (q, v, a, f))
    taud = parallel.rnea(pool, qd, vd, ad, fd)
    ddqd = parallel.aba(pool, qd, vd, taud, fd)
    torch.cuda.synchronize()
    assert rel_err_cols(taud.cpu().numpy(), tau_ref).max() < TOL_RNEA
    assert rel_err_cols(ddqd.cpu().numpy(), a).max() < TOL_ABA
    pool.release()


@pytest.mark.parametrize("name", ["simple_humanoid", "arm7", "unaligned_tree"])
def test_joint_jacobian(name):
    """brbd_joint_jacobian = computeJointJacobians + getJointJacobian(joint, rf)
    (jacobian.hxx:38-95, 175-283) in the three reference frames, both layouts, against the oracle."""
    if name == "unaligned_tree":
        model = S.random_tree(503, 10, kinds=[M.JOINT_RY, M.JOINT_REVOLUTE_UNALIGNED, M.JOINT_PRISMATIC_UNALIGNED,
                                              M.JOINT_SPHERICAL, M.JOINT_REVOLUTE_UNBOUNDED_UNALIGNED])
    else:
        model = S.build(name)
    pool = parallel.ModelPool(model, 1)
    O = oracle.Oracle(model)
    q, _, _ = inputs(model, 100, 41)
    for joint in (1, model.njoints // 2, model.njoints - 1):
        for rf in (capi.WORLD, capi.LOCAL, capi.LOCAL_WORLD_ALIGNED):
            ref = O.joint_jacobian(q, joint, rf)
            for order in "FC":
                J = parallel.getJointJacobian(pool, np.asarray(q, order=order), joint, rf)
                assert np.abs(J - ref).max() < TOL_KIN, (joint, rf, order)
    with pytest.raises(capi.BrbdInvalidArgument):
        parallel.getJointJacobian(pool, q, model.njoints, capi.LOCAL)
    with pytest.raises(capi.BrbdInvalidArgument):
        parallel.getJointJacobian(pool, q, 1, 3)
    pool.release()


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_unaligned_joints(seed):
    """RevoluteUnaligned / PrismaticUnaligned (joint-revolute-unaligned.hpp:667-708): the flattener
    canonicalises them onto the RZ / PZ code of the table-driven kernels; every algorithm, both
    boundary layouts, against the oracle, which evaluates them natively."""
    kinds = [M.JOINT_RX, M.JOINT_PY, M.JOINT_REVOLUTE_UNALIGNED, M.JOINT_PRISMATIC_UNALIGNED, M.JOINT_SPHERICAL,
             M.JOINT_REVOLUTE_UNALIGNED, M.JOINT_RUBZ, M.JOINT_REVOLUTE_UNBOUNDED_UNALIGNED]
    model = S.random_tree(500 + seed, 10, kinds=kinds, root_kind=M.JOINT_FREEFLYER if seed % 2 else None)
    assert any(t in (M.JOINT_REVOLUTE_UNALIGNED, M.JOINT_PRISMATIC_UNALIGNED, M.JOINT_REVOLUTE_UNBOUNDED_UNALIGNED) for t in model.jtypes[1:])
    pool = parallel.ModelPool(model, 1)
    assert pool.device_model(0).compiled_kernels() == ""
    _check_all(pool, model, oracle.Oracle(model))
    pool.release()


@pytest.mark.parametrize("order", ["F", "C"])
def test_derivatives_prezeroed_contract(torch_mod, order):
    """brbd_model_outputs_prezeroed(1): the reference's contract (rnea-derivatives.hpp:104-106).
    On zeroed outputs the result equals the default mode's; entries of unrelated branches are not
    written at all (a NaN planted there survives); host and device memory."""
    torch = torch_mod
    model = S.quadruped()
    pool = parallel.ModelPool(model, 1)
    assert pool.device_model(0).compiled_kernels() == "quadruped_p"
    B, nv = 200, model.nv
    q, v, a = inputs(model, B, 21)
    ref = oracle.Oracle(model).rnea_derivatives(q, v, a)
    qq, vv, aa = (np.asarray(x, order=order) for x in (q, v, a))
    outs = [np.zeros((nv * nv, B), order=order) for _ in range(3)]
    r = parallel.computeRNEADerivatives(pool, qq, vv, aa, out=outs)
    assert r[0] is outs[0]
    for x, y in zip(outs, ref[:3]):
        assert _mat_rel(x, y) < TOL_DERIV
    # device memory, a NaN in a structurally zero entry: joint 2 (first leg) x joint 5 (second leg)
    dev = torch.device("cuda", 0)
    def put(x):
        t = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        return t if order == "C" else torch.from_numpy(np.ascontiguousarray(x.T)).to(dev).t()
    qd, vd, ad = put(q), put(v), put(a)
    outd = [put(np.zeros((nv * nv, B))) for _ in range(3)]
    row, col = model.idx_vs[2], model.idx_vs[5]
    for o in outd:
        o[col * nv + row, :] = float("nan")
    parallel.computeRNEADerivatives(pool, qd, vd, ad, out=outd)
    torch.cuda.synchronize()
    for o, y in zip(outd, ref[:3]):
        got = o.cpu().numpy()
        assert np.isnan(got[col * nv + row]).all()
        got[col * nv + row] = 0.0
        assert _mat_rel(got, y) < TOL_DERIV
    # back to the default contract: everything is written again
    d2 = parallel.computeRNEADerivatives(pool, qd, vd, ad)
    torch.cuda.synchronize()
    assert _mat_rel(d2[0].cpu().numpy(), ref[0]) < TOL_DERIV
    pool.release()


def test_unregistered_model_uses_table_driven_kernels():
    pool = parallel.ModelPool(S.random_tree(100, 8), 1)
    assert pool.device_model(0).compiled_kernels() == ""
    pool.release()


def test_compiled_kernels_armature_and_gravity_change_selection():
    """Armature (rotor inertias: aba.hxx:178-179, rnea.hxx:158; the reference's own tests set it,
    unittest/rnea.cpp:198-206) and gravity are RUN-TIME constants of the compiled kernels: a
    registered tree with armature on its 1-dof joints keeps compiled kernels (the "_a" variant), one
    with another gravity vector keeps the pattern kernels; both match the oracle on every algorithm."""
    for name in ("quadruped", "simple_humanoid"):
        model = S.build(name)
        arm = np.linspace(0.05, 0.2, model.nv)
        arm[:6] = 0.0  # (free-flyer: no rotor)
        model.armature = arm
        pool = parallel.ModelPool(model, 1)
        assert pool.device_model(0).compiled_kernels() == name + "_a"
        _check_all(pool, model, oracle.Oracle(model))
        pool.release()
    model = S.quadruped()
    model.gravity = np.array([0.3, -0.2, -9.0])
    pool = parallel.ModelPool(model, 1)
    assert pool.device_model(0).compiled_kernels() == "quadruped_p"
    _check_all(pool, model, oracle.Oracle(model))
    pool.release()
    # armature on the free-flyer itself is not a pattern any built-in kernel was generated for
    model = S.quadruped()
    model.armature = np.full(model.nv, 0.1)
    pool = parallel.ModelPool(model, 1)
    assert pool.device_model(0).compiled_kernels() == ""
    _check_all(pool, model, oracle.Oracle(model))
    pool.release()


def test_compiled_kernels_large_angles():
    """spec_sincos is branch-free: max abs error against glibc 2.2e-16 for |q| <= 1e12 and 4.5e-16 at
    1e15 (tools/ubench/sincos_check.c), so the results keep the standard tolerances at any angle a
    caller can mean; for |q| >= 2^50 rad (1.1e15) it returns NaN by design, while the
    table-driven kernels follow the library path like the reference."""
    model = S.arm7()
    pool = parallel.ModelPool(model, 1)
    O = oracle.Oracle(model)
    q, v, a = inputs(model, 64, 23)
    q[0, :16] *= 1.0e4
    q[1, 16:32] *= 3.0e9
    q[2, 32:48] = -7.0e12
    assert rel_err_cols(parallel.rneaInParallel(1, pool, q, v, a), O.rnea(q, v, a)).max() < TOL_RNEA
    assert np.abs(parallel.computeJointJacobians(pool, q) - O.fk_jacobians(q)[0]).max() < TOL_KIN
    tau_ref = O.rnea(q, v, a)
    assert rel_err_cols(parallel.abaInParallel(1, pool, q, v, tau_ref), a).max() < TOL_ABA
    q[3, 48:] = 3.0e17
    tau = parallel.rneaInParallel(1, pool, q, v, a)
    assert np.isnan(tau[:, 48:]).any() and not np.isnan(tau[:, :48]).any()
    pool.device_model(0).use_compiled(False)
    tau = parallel.rneaInParallel(1, pool, q, v, a)
    assert rel_err_cols(tau, O.rnea(q, v, a)).max() < TOL_RNEA
    pool.release()


def test_dispatch_timestamps_timing_mode(torch_mod):
    """brbd_set_timing(m, 3): single-dispatch calls of the compiled kernels are timed by their own
    begin / end timestamps: as many records as launches, each shorter than the chained-event time."""
    torch = torch_mod
    model = S.simple_humanoid()
    pool = parallel.ModelPool(model, 1)
    dm = pool.device_model(0)
    q, v, a = (torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0") for x in inputs(model, 4096, 3))
    times = {}
    for mode in (2, 3):
        for _ in range(5):
            parallel.rneaInParallel(1, pool, q, v, a)
        torch.cuda.synchronize()
        dm.kernel_time(capi.ALGO_RNEA)
        capi.check(capi.lib().brbd_set_timing(dm.handle, mode))
        for _ in range(20):
            tau = parallel.rneaInParallel(1, pool, q, v, a)
        torch.cuda.synchronize()
        capi.check(capi.lib().brbd_set_timing(dm.handle, 0))
        n, ms = dm.kernel_time(capi.ALGO_RNEA)
        assert n == 20 and ms > 0
        times[mode] = ms / n
    assert times[3] <= times[2]
    assert rel_err_cols(tau.cpu().numpy(), oracle.Oracle(model).rnea(*inputs(model, 4096, 3))).max() < TOL_RNEA
    pool.release()


def test_timing_period_samples_launches(torch_mod):
    """brbd_set_timing_period(m, p): every p-th launch of each algorithm is timed, in every timing
    mode; a timed launch never spans an untimed one (its time stays that of ONE launch); period 1
    restores timing of every launch."""
    torch = torch_mod
    model = S.simple_humanoid()
    pool = parallel.ModelPool(model, 1)
    dm = pool.device_model(0)
    q, v, a = (torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0") for x in inputs(model, 4096, 3))
    tau = parallel.rneaInParallel(1, pool, q, v, a)
    for _ in range(5):
        parallel.abaInParallel(1, pool, q, v, tau)
    torch.cuda.synchronize()
    per_launch = {}
    for mode, period, expect in ((2, 1, 24), (2, 8, 3), (3, 8, 3), (1, 5, 5), (3, 1, 24)):
        dm.kernel_time(capi.ALGO_RNEA), dm.kernel_time(capi.ALGO_ABA)
        dm.set_timing(True, chained=mode == 2, dispatch=mode == 3, period=period)
        for _ in range(24):
            parallel.rneaInParallel(1, pool, q, v, a, tau)
            ddq = parallel.abaInParallel(1, pool, q, v, tau)
        torch.cuda.synchronize()
        dm.set_timing(False)
        (n_r, ms_r), (n_a, ms_a) = dm.kernel_time(capi.ALGO_RNEA), dm.kernel_time(capi.ALGO_ABA)
        assert n_r == expect and n_a == expect, (mode, period, n_r, n_a)
        per_launch[(mode, period)] = (ms_r / n_r, ms_a / n_a)
    # a sampled launch's time is one launch's, not the 8 steps since the previous sample
    for key in ((2, 8), (3, 8)):
        assert per_launch[key][0] < 3 * per_launch[(2, 1)][0] and per_launch[key][1] < 3 * per_launch[(2, 1)][1]
    with pytest.raises(capi.BrbdInvalidArgument):
        capi.check(capi.lib().brbd_set_timing_period(dm.handle, 0))
    assert rel_err_cols(ddq.cpu().numpy(), a.cpu().numpy()).max() < TOL_ABA
    pool.release()
