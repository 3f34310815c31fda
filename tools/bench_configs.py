#!/usr/bin/env python3
"""The other BASELINE.json configs on one GPU (device-resident BATCH_MAJOR), kernel time from the
library's hipEvents: [1] arm7 FK + Jacobians B=1M, [3] talos-class ABA B=1M (one GPU's slice of the
8M), [4] quadruped RNEA derivatives B=262144.  Prints one JSON line per config."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch   # (oracle: only to count the structural non-zeros of one instance, below)
from pinocchio_amd import capi, parallel, sample_models as S

def inputs(m, B):
    rng = np.random.default_rng(0)
    q = torch.from_numpy(np.ascontiguousarray(m.random_configuration(rng, B))).cuda()
    v = torch.from_numpy(rng.uniform(-1, 1, (m.nv, B))).cuda()
    a = torch.from_numpy(rng.uniform(-1, 1, (m.nv, B))).cuda()
    return q, v, a

def timed(dm, algo, fn, reps):
    for _ in range(2): fn()
    torch.cuda.synchronize(); dm.kernel_time(algo); dm.set_timing(True)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); dm.set_timing(False)
    n, ms = dm.kernel_time(algo)
    return ms / max(n, 1) * 1e-3 * (n / reps)   # seconds per call (chunked launches summed)

def wall(fn, reps):
    """seconds per call between two events on the current stream (everything the call enqueues:
    kernels, transposes, copies)"""
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps

which = sys.argv[1:] or ["fkjac", "aba", "rnea", "deriv", "fext", "jjac", "f4", "layouts"]
if "fkjac" in which:
    m = S.arm7(); B = 1 << 20; pool = parallel.ModelPool(m, 1); dm = pool.device_model(0)
    q, _, _ = inputs(m, B)
    t = timed(dm, capi.ALGO_FK_JAC, lambda: parallel.computeJointJacobians(pool, q), 10)
    byts = (m.nq + 6 * m.nv) * 8 * B
    print(json.dumps({"config": "arm7 FK+J (J only) B=1048576", "kernels": dm.compiled_kernels(), "s_per_call": t,
                      "evals_per_s": B / t, "GBps_algorithmic": byts / t / 1e9}))
    t = timed(dm, capi.ALGO_FK_JAC, lambda: parallel.computeJointJacobians(pool, q, want_oMi=True), 10)
    byts = (m.nq + 6 * m.nv + 12 * m.njoints) * 8 * B
    print(json.dumps({"config": "arm7 FK+J (J + oMi) B=1048576", "s_per_call": t, "evals_per_s": B / t,
                      "GBps_algorithmic": byts / t / 1e9}))
    pool.release()
for name in ("aba", "rnea"):
    if name in which:
        m = S.talos_like(); B = 1 << 20; pool = parallel.ModelPool(m, 1); dm = pool.device_model(0)
        q, v, a = inputs(m, B)
        out = torch.empty_like(a)
        f = (lambda: parallel.abaInParallel(1, pool, q, v, a, out)) if name == "aba" else (lambda: parallel.rneaInParallel(1, pool, q, v, a, out))
        t = timed(dm, capi.ALGO_ABA if name == "aba" else capi.ALGO_RNEA, f, 5)
        print(json.dumps({"config": f"talos_like {name.upper()} B=1048576", "kernels": dm.compiled_kernels(), "s_per_call": t,
                          "evals_per_s": B / t, "GBps_algorithmic": (m.nq + 3 * m.nv) * 8 * B / t / 1e9}))
        pool.release()
if "deriv" in which:
    m = S.quadruped(); B = 1 << 18; pool = parallel.ModelPool(m, 1); dm = pool.device_model(0)
    q, v, a = inputs(m, B)
    t = timed(dm, capi.ALGO_RNEA_DERIV, lambda: parallel.computeRNEADerivatives(pool, q, v, a), 5)
    byts = ((m.nq + 2 * m.nv) + 3 * m.nv * m.nv) * 8 * B
    print(json.dumps({"config": "quadruped RNEA derivatives B=262144", "kernels": dm.compiled_kernels(),
                      "s_per_call": t, "evals_per_s": B / t, "GBps_algorithmic": byts / t / 1e9}))
    # the reference's contract: outputs zeroed by the caller, structural zeros not rewritten
    outs = [torch.zeros((m.nv * m.nv, B), dtype=torch.float64, device="cuda") for _ in range(3)]
    t = timed(dm, capi.ALGO_RNEA_DERIV, lambda: parallel.computeRNEADerivatives(pool, q, v, a, out=outs), 5)
    # priced with the bytes this variant moves: the structural zeros are neither read nor written
    import oracle as _oracle
    rng1 = np.random.default_rng(5)
    nz = sum(int(np.count_nonzero(x)) for x in _oracle.Oracle(m).rnea_derivatives(
        m.random_configuration(rng1, 1), rng1.uniform(-1, 1, (m.nv, 1)), rng1.uniform(-1, 1, (m.nv, 1)))[:3])
    byts_pz = ((m.nq + 2 * m.nv) + nz) * 8 * B
    print(json.dumps({"config": "quadruped RNEA derivatives B=262144, outputs pre-zeroed by the caller",
                      "s_per_call": t, "evals_per_s": B / t, "stores_per_eval": nz, "GBps_moved": byts_pz / t / 1e9}))
    pool.release()

if "fext" in which:   # simple_humanoid with external forces (rnea.hxx:171-219, aba.hxx:296-354)
    m = S.simple_humanoid(); B = 1 << 16; pool = parallel.ModelPool(m, 1); dm = pool.device_model(0)
    q, v, a = inputs(m, B)
    f = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (6 * m.njoints, B))).cuda()
    tau = torch.empty_like(a); ddq = torch.empty_like(a)
    tr = timed(dm, capi.ALGO_RNEA, lambda: parallel.rnea(pool, q, v, a, f, tau), 20)
    ta = timed(dm, capi.ALGO_ABA, lambda: parallel.aba(pool, q, v, tau, f, ddq), 20)
    byts = (m.nq + 3 * m.nv + 6 * m.njoints) * 8 * B
    print(json.dumps({"config": "simple_humanoid RNEA / ABA with external forces B=65536", "rnea_s": tr, "aba_s": ta,
                      "rnea_evals_per_s": B / tr, "aba_evals_per_s": B / ta, "rnea_GBps_algorithmic": byts / tr / 1e9}))
    pool.release()
if "jjac" in which:   # getJointJacobian of the last joint in the three reference frames
    import time
    for m, B in ((S.arm7(), 1 << 20), (S.simple_humanoid(), 1 << 18)):
        pool = parallel.ModelPool(m, 1); dm = pool.device_model(0)
        q, _, _ = inputs(m, B)
        for mode in ("compiled", "table-driven"):
            dm.use_compiled(mode == "compiled")
            for rf, rfname in ((capi.WORLD, "WORLD"), (capi.LOCAL, "LOCAL"), (capi.LOCAL_WORLD_ALIGNED, "LOCAL_WORLD_ALIGNED")):
                for _ in range(3): parallel.getJointJacobian(pool, q, m.njoints - 1, rf)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10): parallel.getJointJacobian(pool, q, m.njoints - 1, rf)
                torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
                byts = (m.nq + 6 * m.nv) * 8 * B
                print(json.dumps({"config": f"{m.name} getJointJacobian(last joint, {rfname}) B={B}, {mode} (wall time per call)",
                                  "s_per_call": t, "evals_per_s": B / t, "GBps_algorithmic": byts / t / 1e9}))
        dm.use_compiled(True)
        pool.release()

def wall(fn, reps):
    import time
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

if "f4" in which:   # rollout support, quadruped B = 262144 and humanoid B = 65536
    for m, B in ((S.quadruped(), 1 << 18), (S.simple_humanoid(), 1 << 16)):
        pool = parallel.ModelPool(m, 1)
        q, v, a = inputs(m, B)
        t = wall(lambda: parallel.integrate(pool, q, v), 20)
        print(json.dumps({"config": f"{m.name} integrate B={B} (wall time per call)", "s_per_call": t, "evals_per_s": B / t,
                          "GBps_algorithmic": (2 * m.nq + m.nv) * 8 * B / t / 1e9}))
        t = wall(lambda: parallel.computeMinverse(pool, q), 5)
        print(json.dumps({"config": f"{m.name} computeMinverse B={B} (wall time per call)", "s_per_call": t, "evals_per_s": B / t,
                          "GBps_algorithmic": (m.nq + m.nv * m.nv) * 8 * B / t / 1e9}))
        t = wall(lambda: parallel.computeABADerivatives(pool, q, v, a), 5)
        print(json.dumps({"config": f"{m.name} computeABADerivatives B={B} (wall time per call)", "s_per_call": t, "evals_per_s": B / t,
                          "GBps_algorithmic": (m.nq + 2 * m.nv + 3 * m.nv * m.nv) * 8 * B / t / 1e9}))
        pool.release()
if "layouts" in which:   # the reference's own layout, device resident, against BATCH_MAJOR (humanoid, kernel times)
    m = S.simple_humanoid(); pool = parallel.ModelPool(m, 1); dm = pool.device_model(0)
    for B in (1 << 14, 1 << 16, 1 << 20):
        q, v, a = inputs(m, B)
        qi, vi, ai = (x.t().contiguous().t() for x in (q, v, a))
        tau, taui = torch.empty_like(a), torch.empty_like(ai.t()).t()
        ddq, ddqi = torch.empty_like(a), torch.empty_like(ai.t()).t()
        reps = 40 if B < (1 << 20) else 5
        r = {"config": f"simple_humanoid kernel time by layout B={B}"}
        r["rnea_batch_major_us"] = timed(dm, capi.ALGO_RNEA, lambda: parallel.rneaInParallel(1, pool, q, v, a, tau), reps) * 1e6
        r["aba_batch_major_us"] = timed(dm, capi.ALGO_ABA, lambda: parallel.abaInParallel(1, pool, q, v, tau, ddq), reps) * 1e6
        r["rnea_instance_major_us"] = timed(dm, capi.ALGO_RNEA, lambda: parallel.rneaInParallel(1, pool, qi, vi, ai, taui), reps) * 1e6
        r["aba_instance_major_us"] = timed(dm, capi.ALGO_ABA, lambda: parallel.abaInParallel(1, pool, qi, vi, taui, ddqi), reps) * 1e6
        print(json.dumps(r))
    pool.release()
    # the output-bound kernels in the reference's layout (per-lane addressing, layout 0)
    m = S.quadruped(); B = 1 << 18; pool = parallel.ModelPool(m, 1); dm = pool.device_model(0)
    q, v, a = inputs(m, B)
    qi, vi, ai = (x.t().contiguous().t() for x in (q, v, a))
    nn = m.nv * m.nv
    ob = [torch.empty((nn, B), dtype=torch.float64, device="cuda") for _ in range(3)]
    oi = [torch.empty((B, nn), dtype=torch.float64, device="cuda").t() for _ in range(3)]
    dm.outputs_prezeroed(False)
    def call(qq, vv, aa, oo):
        capi.check(capi.lib().brbd_rnea_derivatives(
            dm.handle, B, qq.data_ptr(), qq.stride(1) if qq.stride(0) == 1 else qq.stride(0), vv.data_ptr(),
            vv.stride(1) if vv.stride(0) == 1 else vv.stride(0), aa.data_ptr(), aa.stride(1) if aa.stride(0) == 1 else aa.stride(0),
            oo[0].data_ptr(), max(oo[0].stride()), oo[1].data_ptr(), max(oo[1].stride()), oo[2].data_ptr(), max(oo[2].stride()),
            None, 0, capi.INSTANCE_MAJOR if (qq.stride(0) == 1 and B > 1) else capi.BATCH_MAJOR, capi.DEVICE,
            torch.cuda.current_stream().cuda_stream))
    tb = wall(lambda: call(q, v, a, ob), 5)
    ti = wall(lambda: call(qi, vi, ai, oi), 5)
    print(json.dumps({"config": "quadruped RNEA derivatives B=262144 by layout (time per call, events on the stream)",
                      "batch_major_us": tb * 1e6, "instance_major_us": ti * 1e6,
                      "via_bm": os.environ.get("BRBD_DISABLE_VIA_BM", "0") != "1", "chunk": os.environ.get("BRBD_VIA_BM_CHUNK", "default")}))
    tb = wall(lambda: parallel.computeMinverse(pool, q), 3)
    ti = wall(lambda: parallel.computeMinverse(pool, qi), 3)
    print(json.dumps({"config": "quadruped computeMinverse B=262144 by layout (time per call incl. output allocation)",
                      "batch_major_us": tb * 1e6, "instance_major_us": ti * 1e6}))
    tb = wall(lambda: parallel.computeABADerivatives(pool, q, v, a), 3)
    ti = wall(lambda: parallel.computeABADerivatives(pool, qi, vi, ai), 3)
    print(json.dumps({"config": "quadruped computeABADerivatives B=262144 by layout (time per call incl. output allocation)",
                      "batch_major_us": tb * 1e6, "instance_major_us": ti * 1e6}))
    pool.release()
    m = S.arm7(); B = 1 << 20; pool = parallel.ModelPool(m, 1); dm = pool.device_model(0)
    q, _, _ = inputs(m, B)
    qi = q.t().contiguous().t()
    tb = timed(dm, capi.ALGO_FK_JAC, lambda: parallel.computeJointJacobians(pool, q), 10)
    ti = timed(dm, capi.ALGO_FK_JAC, lambda: parallel.computeJointJacobians(pool, qi), 10)
    print(json.dumps({"config": "arm7 FK+J (J only) B=1048576 by layout", "batch_major_us": tb * 1e6, "instance_major_us": ti * 1e6}))
    tb2 = timed(dm, capi.ALGO_FK_JAC, lambda: parallel.computeJointJacobians(pool, q, want_oMi=True), 10)
    ti2 = timed(dm, capi.ALGO_FK_JAC, lambda: parallel.computeJointJacobians(pool, qi, want_oMi=True), 10)
    print(json.dumps({"config": "arm7 FK+J (J + oMi) B=1048576 by layout", "batch_major_us": tb2 * 1e6, "instance_major_us": ti2 * 1e6}))
    pool.release()
    m = S.simple_humanoid(); B = 1 << 18; pool = parallel.ModelPool(m, 1); dm = pool.device_model(0)
    q, _, _ = inputs(m, B)
    qi = q.t().contiguous().t()
    tb2 = timed(dm, capi.ALGO_FK_JAC, lambda: parallel.computeJointJacobians(pool, q, want_oMi=True), 5)
    ti2 = timed(dm, capi.ALGO_FK_JAC, lambda: parallel.computeJointJacobians(pool, qi, want_oMi=True), 5)
    print(json.dumps({"config": "simple_humanoid FK+J (J + oMi) B=262144 by layout", "batch_major_us": tb2 * 1e6, "instance_major_us": ti2 * 1e6}))
    pool.release()
