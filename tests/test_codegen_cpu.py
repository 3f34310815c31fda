"""The model compiler (pinocchio_amd/codegen) checked on the CPU: the straight-line bodies it emits
are compiled by gcc into a scalar function (codegen/compile.py: CpuKernel) and compared with the
oracle on the same inputs.  The very same text is what hipcc compiles for the GPU.  Also: the
storage planner never lets two live parks share a slot, and the model signature the generator
writes is the one the C++ host recomputes (tests/test_capi_host.py covers the host side)."""
import numpy as np
import pytest

import oracle
from pinocchio_amd import model as M, sample_models as S
from pinocchio_amd.codegen import compile as C, emit, joints, rnea, aba
from util import inputs, rel_err_cols

KINDS_ALL = [M.JOINT_RX, M.JOINT_RY, M.JOINT_RZ, M.JOINT_PX, M.JOINT_PY, M.JOINT_PZ, M.JOINT_SPHERICAL,
             M.JOINT_RUBX, M.JOINT_RUBY, M.JOINT_RUBZ, M.JOINT_FREEFLYER, M.JOINT_REVOLUTE_UNALIGNED,
             M.JOINT_PRISMATIC_UNALIGNED]


def _models():
    out = [(S.simple_humanoid(), False), (S.quadruped(), True), (S.arm7(), False), (S.baxter_simple(), True)]
    for s in range(4):
        out.append((S.random_tree(300 + s, 9, kinds=KINDS_ALL, root_kind=None if s % 2 else M.JOINT_FREEFLYER,
                                  armature=bool(s % 2)), bool(s % 2)))
    return out


@pytest.mark.parametrize("model,dense", _models(), ids=lambda x: getattr(x, "name", str(x)))
def test_generated_bodies_match_oracle(model, dense):
    mv = joints.ModelView(model, dense=dense)
    O = oracle.Oracle(model)
    q, v, a = inputs(model, 12, 3)
    tau_ref = O.rnea(q, v, a)
    # tiny budgets force every storage class (registers, LDS, global) to be exercised
    tau = C.CpuKernel(mv, "rnea", reg_budget=6, lds_budget=16).run(q, v, a, (model.nv, 0, 0))[0]
    assert rel_err_cols(tau, tau_ref).max() < 1e-12
    ddq = C.CpuKernel(mv, "aba", reg_budget=8, lds_budget=20).run(q, v, tau_ref, (model.nv, 0, 0))[0]
    assert rel_err_cols(ddq, O.aba(q, v, tau_ref)).max() < 1e-10
    J, oMi, _ = C.CpuKernel(mv, "fkjac", reg_budget=0, lds_budget=12).run(
        q, None, None, (6 * model.nv, 12 * model.njoints, 0))
    Jr, Mr = O.fk_jacobians(q)
    assert np.abs(J - Jr).max() < 1e-12 and np.abs(oMi - Mr).max() < 1e-12


def test_pattern_folding_shrinks_the_humanoid():
    """simple_humanoid has identity placement rotations and diagonal inertias: the pattern kernels
    must need far fewer fp64 operations than the dense ones."""
    m = S.simple_humanoid()
    kp = C.build_kernel(joints.ModelView(m, dense=False), "rnea")
    kd = C.build_kernel(joints.ModelView(m, dense=True), "rnea")
    assert kp.stats["fp"] < 0.6 * kd.stats["fp"]
    assert len(kp.const_list) < 0.2 * len(kd.const_list)


@pytest.mark.parametrize("gen", [rnea.generate, aba.generate])
def test_storage_plan_has_no_overlapping_live_slots(gen):
    mv = joints.ModelView(S.talos_like())
    dry = gen(mv, emit.Emitter())
    plan, lds, glb = emit.plan_storage(dry.parks, 10, 24, 10)
    live = {"lds": [], "glb": []}
    for pk in dry.parks:
        cls, slot = plan[pk.pid]
        if cls == "reg":
            continue
        n = sum(1 for x in pk.vals if not x.is_num)
        n += n & 1
        assert slot % 2 == 0 and slot + n <= (lds if cls == "lds" else glb)
        for (s2, n2, a2, b2) in live[cls]:
            overlap_time = a2 < pk.end and pk.start < b2
            overlap_space = s2 < slot + n and slot < s2 + n2
            assert not (overlap_time and overlap_space)
        live[cls].append((slot, n, pk.start, pk.end))
    assert lds <= 24


def test_signature_distinguishes_pattern_and_dense():
    m = S.quadruped()
    sp = joints.ModelView(m, dense=False).signature()
    sd = joints.ModelView(m, dense=True).signature()
    assert sp != sd and sp.split(";")[:3] == sd.split(";")[:3]
    m2 = S.quadruped()
    m2.armature = np.full(m2.nv, 0.1)
    assert joints.ModelView(m2, dense=True).signature() != sd  # armature pattern is always exact
