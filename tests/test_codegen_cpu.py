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
             M.JOINT_REVOLUTE_UNBOUNDED_UNALIGNED,
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
    J, oMi = C.CpuKernel(mv, "fkjac", reg_budget=0, lds_budget=12).run(
        q, None, None, (6 * model.nv, 12 * model.njoints, 0))[:2]
    Jr, Mr = O.fk_jacobians(q)
    assert np.abs(J - Jr).max() < 1e-12 and np.abs(oMi - Mr).max() < 1e-12


@pytest.mark.parametrize("model,dense", _models()[1:3] + _models()[4:6], ids=lambda x: getattr(x, "name", str(x)))
def test_generated_derivative_bodies_match_oracle(model, dense):
    """codegen/deriv.py (computeRNEADerivatives): all four outputs, every element of the three
    nv x nv matrices written (structural zeros included: the outputs start as NaN here)."""
    mv = joints.ModelView(model, dense=dense)
    q, v, a = inputs(model, 6, 4)
    nn = model.nv * model.nv
    k = C.CpuKernel(mv, "deriv", reg_budget=8, lds_budget=24)
    t = k.kernel.text
    assert t.count("OUT(") + t.count("OUTZ(") == nn and t.count("OUT2(") + t.count("OUT2Z(") == nn
    assert t.count("OUT3(") + t.count("OUT3Z(") == nn and t.count("OUT4(") == model.nv
    dq, dv, da, tau = k.run(q, v, a, (nn, nn, nn, model.nv))
    rq, rv, ra, rt = oracle.Oracle(model).rnea_derivatives(q, v, a)
    for x, r in ((dq, rq), (dv, rv), (da, ra), (tau, rt)):
        assert np.abs(x - r).max() / max(1.0, np.abs(r).max()) < 1e-12


@pytest.mark.parametrize("model,dense", _models()[:2] + _models()[4:6], ids=lambda x: getattr(x, "name", str(x)))
def test_generated_bodies_with_external_forces(model, dense):
    """Kernel variants with WANT bit 5: rnea(q, v, a, fext) (rnea.hxx:205) and aba(q, v, tau, fext)
    (aba.hxx:338).  Without the bit the very same text computes the plain algorithms."""
    mv = joints.ModelView(model, dense=dense)
    O = oracle.Oracle(model)
    q, v, a = inputs(model, 6, 8)
    f = np.random.default_rng(8).uniform(-2, 2, (6 * model.njoints, 6))
    ref = O.rnea_fext(q, v, a, f)
    assert np.abs(ref - O.rnea(q, v, a)).max() > 1e-2
    tau = C.CpuKernel(mv, "rnea", want=33).run(q, v, a, (model.nv, 0, 0), fext=f)[0]
    assert rel_err_cols(tau, ref).max() < 1e-12
    ddq = C.CpuKernel(mv, "aba", want=33, reg_budget=8, lds_budget=20).run(q, v, ref, (model.nv, 0, 0), fext=f)[0]
    assert rel_err_cols(ddq, O.aba_fext(q, v, ref, f)).max() < 1e-10
    assert rel_err_cols(ddq, a).max() < 1e-9          # unittest/aba.cpp: aba(rnea(a, fext), fext) == a
    tau0 = C.CpuKernel(mv, "rnea", want=1).run(q, v, a, (model.nv, 0, 0), fext=f)[0]
    assert rel_err_cols(tau0, O.rnea(q, v, a)).max() < 1e-12


def test_derivative_body_under_the_prezeroed_contract():
    """Without the structural zeros (rnea-derivatives.hpp:104-106: the caller zeroes the outputs)
    the body writes only coupled entries; on zeroed outputs the result is the same."""
    model = S.quadruped()
    mv = joints.ModelView(model)
    q, v, a = inputs(model, 5, 9)
    nn = model.nv * model.nv
    k = C.CpuKernel(mv, "deriv", skip_zeros=True)
    nz = sum(k.kernel.text.count(m) for m in ("OUTZ(", "OUT2Z(", "OUT3Z("))
    assert nz == 408 and k.kernel.text.count("OUT(") == 324 - 108   # 4 legs x 3 other legs x 3 x 3
    got = k.run(q, v, a, (nn, nn, nn, model.nv))   # (CpuKernel.run starts from zeroed outputs)
    ref = oracle.Oracle(model).rnea_derivatives(q, v, a)
    for x, r in zip(got, ref):
        assert np.abs(x - r).max() / max(1.0, np.abs(r).max()) < 1e-12


@pytest.mark.parametrize("model,dense", _models()[:3] + _models()[4:7], ids=lambda x: getattr(x, "name", str(x)))
def test_generated_minverse_bodies_match_oracle(model, dense):
    """codegen/minv.py (computeMinverse): the upper triangle like data.Minv (strictly lower part written as
    zero), and the symmetric variant computeABADerivatives asks for; tiny budgets and the default ones."""
    if not model.is_depth_first():
        with pytest.raises(ValueError):
            C.build_kernel(joints.ModelView(model, dense=dense), "minv")
        return
    mv = joints.ModelView(model, dense=dense)
    q, _, _ = inputs(model, 6, 9)
    nv = model.nv
    ref = oracle.Oracle(model).minverse(q)
    sym = ref.copy()
    for b in range(q.shape[1]):
        Mx = ref[:, b].reshape(nv, nv).T      # [row][col]
        low = np.tril(Mx, -1)
        full = Mx + np.where(low == 0.0, np.triu(Mx, 1).T, 0.0)   # (diagonal blocks of multi-dof joints are already full)
        sym[:, b] = full.T.reshape(-1)
    for want, kw, exp in ((1, dict(reg_budget=6, lds_budget=12), ref), (3, {}, sym)):
        k = C.CpuKernel(mv, "minv", want=want, **kw)
        t = k.kernel.text
        assert t.count("OUT(") + t.count("OUTSYM(") + t.count("OUTZ(") == nv * nv   # every element exactly once
        out = k.run(q, None, None, (nv * nv, 0, 0))[0]
        assert np.abs(out - exp).max() / max(1.0, np.abs(exp).max()) < 1e-12


@pytest.mark.parametrize("model", [S.quadruped(), S.arm7(), S.baxter_simple(),
                                   S.random_tree(301, 9, root_kind=M.JOINT_FREEFLYER, armature=True)],
                         ids=lambda x: getattr(x, "name", str(x)))
def test_generated_product_bodies_match_numpy(model):
    """codegen/prod.py (the closing products of computeABADerivatives, in place): -Minv dtau/dq and
    -Minv dtau/dv from the oracle's matrices; both strategies (Minv in registers / column groups)."""
    mv = joints.ModelView(model)
    nv = model.nv
    nn = nv * nv
    q, v, a = inputs(model, 5, 11)
    O = oracle.Oracle(model)
    dq, dv = O.rnea_derivatives(q, v, a)[:2]
    Mf = O.minverse(q)
    exp1, exp2 = np.zeros_like(dq), np.zeros_like(dv)
    for b in range(q.shape[1]):
        X = Mf[:, b].reshape(nv, nv).T
        X = np.triu(X) + np.triu(X, 1).T
        Mf[:, b] = X.T.reshape(-1)
        exp1[:, b] = (-X @ dq[:, b].reshape(nv, nv).T).T.reshape(-1)
        exp2[:, b] = (-X @ dv[:, b].reshape(nv, nv).T).T.reshape(-1)
    k = C.CpuKernel(mv, "prod", want=3)
    assert k.kernel.lds == 0 and k.kernel.glb == 0
    o1, o2, o3 = k.run(q, None, None, (nn, nn, nn), init_outs=(dq, dv, Mf))[:3]
    assert np.abs(o1 - exp1).max() <= 1e-12 * np.abs(exp1).max()
    assert np.abs(o2 - exp2).max() <= 1e-12 * np.abs(exp2).max()
    assert np.array_equal(o3, Mf)   # Minv is only read


@pytest.mark.parametrize("name,algo", [("quadruped", "minv"), ("simple_humanoid", "minv"), ("quadruped", "deriv"),
                                       ("simple_humanoid", "aba"), ("talos_like", "aba"), ("simple_humanoid", "rnea")])
def test_no_park_slot_is_read_after_another_park_took_it(name, algo):
    """Every load of a parked value must see the slot still owned by that park: walks the emitted text
    with the park behind every access named (Emitter.tag_parks) -- placement of early loads at
    statement granularity and slot reuse within a visit are where this could go wrong."""
    import re
    emit.Emitter.tag_parks = True
    try:
        # (RNEA's default budget keeps the whole path state in registers: give it the round-2 budget,
        # which parks 76 doubles per lane in LDS, so that there are slots to check)
        kw = {"reg_budget": 40} if algo == "rnea" else {}
        text = C.build_kernel(joints.ModelView(S.build(name)), algo, **kw).text
    finally:
        emit.Emitter.tag_parks = False
    owner, n = {}, 0
    for ln in text.splitlines():
        for m_ in re.finditer(r"PK_(LDS|GLB)_(ST|LD)2\((\d+),[^;]*;\s*/\*P(\d+)\*/", ln):
            slot, pid = (m_.group(1), int(m_.group(3))), int(m_.group(4))
            if m_.group(2) == "ST":
                owner[slot] = pid
            else:
                n += 1
                assert owner.get(slot) == pid, ln
    assert n > 0


def test_fine_prefetch_placement_keeps_order():
    """Emitter.text(fine=True): every early load of a parked value comes after the park's stores
    and before its first use; the statements themselves keep their order."""
    mv = joints.ModelView(S.arm7())
    k0 = C.build_kernel(mv, "deriv")
    lines = k0.text.splitlines()
    stored, loaded = {}, {}
    for n, ln in enumerate(lines):
        for mac, d in (("PK_GLB_ST2(", stored), ("PK_LDS_ST2(", stored)):
            if mac in ln:
                d.setdefault((mac[3:6], int(ln.split(mac)[1].split(",")[0])), []).append(n)
        for mac in ("PK_GLB_LD2(", "PK_LDS_LD2("):
            if mac in ln:
                slot = (mac[3:6], int(ln.split(mac)[1].split(",")[0]))
                assert slot in stored and max(x for x in stored[slot] if x < n) < n
    assert len(stored) > 0


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
    # armature: part of the pattern ("p"); generic for 1-dof joints in the "a" and "d" variants, so
    # rotor inertias do not change the kernels; on a multi-dof joint it always is part of the pattern
    m2 = S.quadruped()
    m2.armature = np.concatenate([np.zeros(6), np.full(m2.nv - 6, 0.1)])
    sa = joints.ModelView(m, arm_generic=True).signature()
    assert joints.ModelView(m2).signature() != sp
    assert joints.ModelView(m2, arm_generic=True).signature() == sa
    assert joints.ModelView(m2, dense=True).signature() == sd
    m2.armature = np.full(m2.nv, 0.1)
    assert joints.ModelView(m2, arm_generic=True).signature() != sa
    # gravity is a run-time constant of every variant
    m3 = S.quadruped()
    m3.gravity = np.array([0.3, -0.2, -9.0])
    assert joints.ModelView(m3).signature() == sp


def test_armature_variant_bodies_match_oracle():
    """the "a" variant (armature of 1-dof joints generic) on a model WITH rotor inertias and a tilted gravity"""
    model = S.quadruped()
    model.armature = np.concatenate([np.zeros(6), np.linspace(0.05, 0.2, model.nv - 6)])
    model.gravity = np.array([0.3, -0.2, -9.0])
    mv = joints.ModelView(S.quadruped(), arm_generic=True)   # generated WITHOUT knowing the values
    mv_vals = joints.ModelView(model, arm_generic=True)       # packed from the model that has them
    assert mv.signature() == mv_vals.signature()
    O = oracle.Oracle(model)
    q, v, a = inputs(model, 5)
    for algo in ("rnea", "aba"):
        ck = C.CpuKernel(mv, algo)
        ck.ct = C.pack_constants(mv_vals, ck.kernel.const_list)
        if algo == "rnea":
            tau = ck.run(q, v, a, (model.nv, 0, 0))[0]
            assert rel_err_cols(tau, O.rnea(q, v, a)).max() < 1e-12
        else:
            ddq = ck.run(q, v, tau, (model.nv, 0, 0))[0]
            assert rel_err_cols(ddq, O.aba(q, v, tau)).max() < 1e-10


@pytest.mark.parametrize("name", ["simple_humanoid", "talos_like", "quadruped"])
def test_chunked_staging_bodies(name):
    """The bodies of the INSTANCE_MAJOR kernels with chunked staging (compile.build_kernel(chunked=True),
    csrc/brbd_spec.hpp: SpecIn<2, ., ., 3>): same values, the same constant table as the other
    layouts' body, LDS parks within what the window leaves, and every element read of a group comes
    after the group's commit, which comes after its fetch."""
    import re
    model = S.build(name)
    mv = joints.ModelView(model)
    O = oracle.Oracle(model)
    q, v, a = inputs(model, 9, 5)
    tau_ref = O.rnea(q, v, a)
    for algo, x, ref, tol in (("rnea", a, tau_ref, 1e-12), ("aba", tau_ref, None, 1e-10)):
        base = C.build_kernel(mv, algo)
        ck = C.CpuKernel(mv, algo, chunked=True, like=base)
        k = ck.kernel
        assert list(k.const_list) == list(base.const_list)
        assert k.lds <= C.CHUNK_LDS
        out = ck.run(q, v, x, (model.nv, 0, 0))[0]
        assert rel_err_cols(out, ref if ref is not None else O.aba(q, v, tau_ref)).max() < tol
        state = {}
        for line in k.text.splitlines():
            m = re.match(r"GRP(C?)_([QVX])\((\d+), (\d+)\);", line)
            if m:
                key = (m.group(2), int(m.group(3)), int(m.group(4)))
                if m.group(1):
                    assert state.get(key) == "fetched", (line, state.get(key))
                    state[key] = "committed"
                else:
                    assert key not in state, line
                    state[key] = "fetched"
                continue
            for m in re.finditer(r"IN_([QVX])2\((\d+), (\d+), (\d+), (\d+)\)", line):
                kk, g, j, n = (int(m.group(i)) for i in (2, 3, 4, 5))
                assert (g, j) == divmod(kk, C.CHUNK_WIDTH)
                assert state.get((m.group(1), g, n)) == "committed", line
            assert not re.search(r"IN_[QVX]\(", line), line   # (no element load left outside the groups)
        assert state and all(s == "committed" for s in state.values())


def test_built_kernels_have_no_scratch_with_spilled_sgprs():
    """Register reports of the last build (pinocchio_amd/csrc/gen/*.usage, tools/spec_usage.py): no
    compiled kernel may combine scratch-memory accesses with spilled SGPRs (the combination that
    miscompiled under hipcc 7.2 during development), and the headline kernels
    use no scratch.  A reserved private segment that no instruction touches does not count: the
    kernels that report one are disassembled (codegen/usage.py)."""
    import os
    from pinocchio_amd.codegen import usage
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pinocchio_amd", "csrc", "gen")
    rows = usage.kernel_reports(root)
    if not rows:
        pytest.skip("no build reports in this tree")
    for cur in rows:
        assert not usage.refused(cur), cur
        # the headline kernels (BATCH_MAJOR) touch no scratch at all; the INSTANCE_MAJOR ABA kernel
        # stages its tile in registers and may spill a few VGPRs (never together with SGPRs)
        if "simple_humanoid" in cur["name"] and ("k_rneaILi1" in cur["name"] or "k_abaILi1" in cur["name"]):
            assert int(cur["ScratchInsts"]) == 0, cur


def test_instrumented_kernels_have_no_scratch_with_spilled_sgprs():
    """The same guard for the CHECKING build's objects (csrc/gen_check): round 3 found the instrumented
    talos_like k_aba<1,33> miscomputing (scratch + 1 353 spilled SGPRs, caused by the instrumentation's
    out-of-line reporter) while the product kernel was right; a checking build that computes wrong
    values would fail tests/test_gpu_bounds_check.py for the wrong reason."""
    import os
    from pinocchio_amd.codegen import usage
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pinocchio_amd", "csrc", "gen_check")
    rows = usage.kernel_reports(root)
    if not rows:
        pytest.skip("no checking-build reports in this tree")
    for cur in rows:
        assert not usage.refused(cur), cur


def test_scratch_detector_sees_real_spills(tmp_path):
    """positive control of codegen/usage.py: a kernel that indexes a local array dynamically"""
    import shutil, subprocess
    from pinocchio_amd.codegen import usage
    hipcc = "/opt/rocm/bin/hipcc"
    if not shutil.which(hipcc):
        pytest.skip("no hipcc")
    src = tmp_path / "sp.hip"
    src.write_text("#include <hip/hip_runtime.h>\n__global__ void k_spill(double* x, int n) { double a[64]; "
                   "for (int i = 0; i < 64; ++i) a[i] = x[i] * n; x[threadIdx.x] = a[(n * threadIdx.x) & 63]; }\n")
    obj = tmp_path / "sp.o"
    subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-c", "-o", str(obj), str(src)])
    counts = usage.scratch_instructions(str(obj))
    assert counts and max(counts.values()) > 0


@pytest.mark.parametrize("name", ["arm7", "quadruped", "simple_humanoid"])
def test_generated_joint_jacobian_bodies_match_oracle(name):
    """codegen/jjac.py: one body per model, the joint a run-time chain mask, the frame a variant
    (WANT 1 WORLD, 3 LOCAL, 5 LOCAL_WORLD_ALIGNED), against getJointJacobian restated by the oracle
    (jacobian.hxx:175-283)."""
    import ctypes
    model = S.build(name)
    mv = joints.ModelView(model)
    O = oracle.Oracle(model)
    q = inputs(model, 6)[0]
    for want, rf in ((1, 0), (3, 1), (5, 2)):
        ck = C.CpuKernel(mv, "jjac", want=want)
        for j in sorted({1, model.njoints // 2, model.njoints - 1}):
            mask, i = 0, j
            while i > 0:
                mask |= 1 << i
                i = model.parents[i]
            ctypes.c_ulonglong.in_dll(ck.lib, "spec_chain").value = mask
            Jk = ck.run(q, None, None, (6 * model.nv, 0, 0))[0]
            assert np.abs(Jk - O.joint_jacobian(q, j, rf)).max() < 1e-12
