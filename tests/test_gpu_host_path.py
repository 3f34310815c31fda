"""The BRBD_HOST calling convention (host matrices in, host matrices out: what rneaInParallel
receives, algorithm/parallel/rnea.hpp:39-45) through the C ABI: padded leading dimensions on
exactly sized buffers with canaries (nothing outside the logical matrix is read or written),
batches that span several pipeline chunks, registered (pinned) memory, and two streams of one
handle in flight at once."""
import ctypes
import threading

import numpy as np
import pytest

import oracle
from pinocchio_amd import capi, parallel, sample_models as S
from util import inputs, rel_err_cols

pytestmark = pytest.mark.gpu
TOL_RNEA, TOL_ABA, TOL_KIN, TOL_DERIV = 1e-12, 1e-10, 1e-12, 1e-10
CANARY = -7.25e300


def _padded(x, ld, layout):
    """Exactly sized flat buffer holding the (rows, B) matrix x with leading dimension ld: span
    (outer - 1) * ld + inner doubles, canary in the padding, plus guard zones on both sides."""
    rows, B = x.shape
    outer, inner = (rows, B) if layout == capi.BATCH_MAJOR else (B, rows)
    span = (outer - 1) * ld + inner
    guard = 64
    buf = np.full(span + 2 * guard, CANARY)
    view = np.lib.stride_tricks.as_strided(buf[guard:], shape=(outer, inner), strides=(ld * 8, 8))
    view[...] = x if layout == capi.BATCH_MAJOR else x.T
    return buf, view, guard


def _canaries_intact(buf, view, guard, ld, inner):
    assert (buf[:guard] == CANARY).all() and (buf[-guard:] == CANARY).all()
    body = buf[guard:len(buf) - guard]
    mask = np.ones(body.shape, dtype=bool)
    outer = view.shape[0]
    for o in range(outer):
        mask[o * ld:o * ld + inner] = False
    assert (body[mask] == CANARY).all()


@pytest.mark.parametrize("layout", [capi.INSTANCE_MAJOR, capi.BATCH_MAJOR], ids=["instance_major", "batch_major"])
@pytest.mark.parametrize("name", ["simple_humanoid", "random_tree"])
@pytest.mark.parametrize("B", [300, 9000], ids=["direct_copies", "bounce_buffers"])
def test_host_memspace_padded_exact_buffers(layout, name, B):
    """B = 300: the runtime's own pageable copies (hipMemcpy2DAsync); B = 9000: above the 4 MiB
    threshold, the library's worker threads pack / unpack pinned bounce buffers."""
    L = capi.lib()
    model = S.build(name) if name != "random_tree" else S.random_tree(104, 10)
    dm = capi.DeviceModel(model, 0)
    O = oracle.Oracle(model)
    pad = 5
    q, v, a = inputs(model, B, 8)
    tau_ref = O.rnea(q, v, a)
    nan = np.full((model.nv, B), np.nan)

    def mk(x):
        inner = x.shape[1] if layout == capi.BATCH_MAJOR else x.shape[0]
        ld = inner + pad
        buf, view, guard = _padded(x, ld, layout)
        return buf, view, guard, ld, inner

    bq, bv, ba, bt, bd = mk(q), mk(v), mk(a), mk(nan), mk(nan)
    ptr = lambda b: b[1].ctypes.data
    capi.check(L.brbd_rnea(dm.handle, B, ptr(bq), bq[3], ptr(bv), bv[3], ptr(ba), ba[3], ptr(bt), bt[3], layout, capi.HOST, None))
    tau = bt[1] if layout == capi.BATCH_MAJOR else bt[1].T
    assert rel_err_cols(tau, tau_ref).max() < TOL_RNEA
    capi.check(L.brbd_aba(dm.handle, B, ptr(bq), bq[3], ptr(bv), bv[3], ptr(bt), bt[3], ptr(bd), bd[3], layout, capi.HOST, None))
    ddq = bd[1] if layout == capi.BATCH_MAJOR else bd[1].T
    assert rel_err_cols(ddq, a).max() < TOL_ABA
    for b in (bq, bv, ba, bt, bd):
        _canaries_intact(b[0], b[1], b[2], b[3], b[4])
    # derivative outputs and Jacobians take the same path
    if model.is_depth_first():
        nn = model.nv * model.nv
        outs = [mk(np.full((nn, B), np.nan)) for _ in range(3)]
        capi.check(L.brbd_rnea_derivatives(dm.handle, B, ptr(bq), bq[3], ptr(bv), bv[3], ptr(ba), ba[3],
                                           ptr(outs[0]), outs[0][3], ptr(outs[1]), outs[1][3], ptr(outs[2]), outs[2][3],
                                           None, 0, layout, capi.HOST, None))
        ref = O.rnea_derivatives(q, v, a)
        for o, r in zip(outs, ref[:3]):
            got = o[1] if layout == capi.BATCH_MAJOR else o[1].T
            assert np.abs(got - r).max() / max(1.0, np.abs(r).max()) < TOL_DERIV
            _canaries_intact(o[0], o[1], o[2], o[3], o[4])
    bj = mk(np.full((6 * model.nv, B), np.nan))
    capi.check(L.brbd_fk_jacobians(dm.handle, B, ptr(bq), bq[3], ptr(bj), bj[3], None, 0, layout, capi.HOST, None))
    J = bj[1] if layout == capi.BATCH_MAJOR else bj[1].T
    assert np.abs(J - O.fk_jacobians(q)[0]).max() < TOL_KIN
    _canaries_intact(bj[0], bj[1], bj[2], bj[3], bj[4])
    dm.close()


@pytest.mark.parametrize("order", ["F", "C"])
def test_host_pipeline_many_chunks_and_pinned_memory(order, monkeypatch):
    """B = 20000 -> chunks of 4096 + a ragged tail through two staging slots; the same call again
    on page-locked matrices (brbd_host_register)."""
    model = S.quadruped()
    pool = parallel.ModelPool(model, 1)
    O = oracle.Oracle(model)
    B = 20000 + 37
    q, v, a = inputs(model, B, 12)
    tau_ref = O.rnea(q, v, a, threads=0)
    qq, vv, aa = (np.asarray(x, order=order) for x in (q, v, a))
    tau = parallel.rneaInParallel(1, pool, qq, vv, aa)
    assert rel_err_cols(tau, tau_ref).max() < TOL_RNEA
    ddq = parallel.abaInParallel(1, pool, qq, vv, tau)
    assert rel_err_cols(ddq, a).max() < TOL_ABA
    # page-locked matrices: true DMA copies, upload | kernels | download overlap
    import torch

    def pinned(x):
        if order == "C":
            t = torch.empty(x.shape, dtype=torch.float64).pin_memory()
            t.copy_(torch.from_numpy(np.ascontiguousarray(x)))
            return t.numpy()
        t = torch.empty(x.shape[::-1], dtype=torch.float64).pin_memory()
        t.copy_(torch.from_numpy(np.ascontiguousarray(x.T)))
        return t.numpy().T
    pq, pv, pa, pt, pd = pinned(q), pinned(v), pinned(a), pinned(np.zeros_like(a)), pinned(np.zeros_like(a))
    assert parallel.rneaInParallel(1, pool, pq, pv, pa, pt) is pt
    assert np.array_equal(pt, tau)  # same kernels, same chunking: bitwise the pageable result
    parallel.abaInParallel(1, pool, pq, pv, pt, pd)
    assert np.array_equal(pd, ddq)
    # brbd_host_register pins a caller's own range (whole pages).  Containers may forbid it
    # (RLIMIT_MEMLOCK): then it must fail cleanly and leave the library usable.
    L = capi.lib()
    rc = L.brbd_host_register(qq.ctypes.data, qq.nbytes)
    if rc == capi.BRBD_OK:
        tau3 = parallel.rneaInParallel(1, pool, qq, vv, aa)
        capi.check(L.brbd_host_unregister(qq.ctypes.data))
    else:
        assert rc == capi.BRBD_ERR_RUNTIME and "hipHostRegister" in capi.last_error()
        tau3 = parallel.rneaInParallel(1, pool, qq, vv, aa)
    assert np.array_equal(tau3, tau)
    pool.release()


def test_two_streams_of_one_handle_in_flight():
    """SURVEY 8b "Threading": re-entrant per handle + stream.  Two host threads drive two streams
    of ONE handle with different batch sizes (different workspace sizes, table-driven kernels, so
    the workspace is really used); each must get its own results."""
    import torch
    model = S.random_tree(105, 14)
    pool = parallel.ModelPool(model, 1)
    dm = pool.device_model(0)
    O = oracle.Oracle(model)
    results, errors = {}, []

    def work(tag, B, seed):
        try:
            q, v, a = inputs(model, B, seed)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                qd, vd, ad = (torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0", non_blocking=False) for x in (q, v, a))
                for _ in range(20):
                    tau = parallel.rneaInParallel(1, pool, qd, vd, ad)
                    ddq = parallel.abaInParallel(1, pool, qd, vd, tau)
                s.synchronize()
            results[tag] = (rel_err_cols(tau.cpu().numpy(), O.rnea(q, v, a, threads=1)).max(),
                            rel_err_cols(ddq.cpu().numpy(), a).max())
        except Exception as e:  # pragma: no cover
            errors.append(e)

    ts = [threading.Thread(target=work, args=("a", 40000, 1)), threading.Thread(target=work, args=("b", 3000, 2))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for tag in ("a", "b"):
        assert results[tag][0] < TOL_RNEA and results[tag][1] < TOL_ABA, results
    pool.release()


def test_permuted_dof_order_on_a_registered_tree():
    """A description with a registered tree but another idx_q / idx_v layout must not run the
    kernels compiled for the registered layout (they hard-code the rows): the signature carries
    idx_q / idx_v, so it falls back to the table-driven kernels and matches the oracle."""
    model = S.arm7()
    n = model.njoints
    perm = model.copy()
    order = list(range(n - 1, 0, -1))  # joint n-1 owns dof 0, ...
    for k, j in enumerate(order):
        perm.idx_qs[j] = k
        perm.idx_vs[j] = k
    pool = parallel.ModelPool(perm, 1)
    assert pool.device_model(0).compiled_kernels() == ""
    O = oracle.Oracle(perm)
    q, v, a = inputs(perm, 96, 4)
    tau = parallel.rneaInParallel(1, pool, q, v, a)
    assert rel_err_cols(tau, O.rnea(q, v, a)).max() < TOL_RNEA
    assert rel_err_cols(parallel.abaInParallel(1, pool, q, v, tau), a).max() < TOL_ABA
    J = parallel.computeJointJacobians(pool, q)
    assert np.abs(J - O.fk_jacobians(q)[0]).max() < TOL_KIN
    pool.release()
    pool = parallel.ModelPool(model, 1)
    assert pool.device_model(0).compiled_kernels() == "arm7_p"
    pool.release()


@pytest.mark.parametrize("order", ["F", "C"])
def test_one_process_several_devices_column_split(order):
    """rneaInParallelOnDevices / abaInParallelOnDevices: slice k of the columns on devices[k], one host
    thread per device (here three handles on GPU 0; on an 8-GPU node devices=range(8)).  Same numbers
    as the single-device call, ragged split included."""
    model = S.quadruped()
    pool = parallel.ModelPool(model, 4)
    B = 10007
    q, v, a = inputs(model, B, 77)
    qq, vv, aa = (np.asarray(x, order=order) for x in (q, v, a))
    tau1 = parallel.rneaInParallel(1, pool, qq, vv, aa)
    tau3 = parallel.rneaInParallelOnDevices(3, pool, qq, vv, aa, devices=[0, 0, 0])
    assert np.array_equal(tau1, tau3)
    ddq3 = parallel.abaInParallelOnDevices(3, pool, qq, vv, tau3, devices=[0, 0, 0])
    assert np.array_equal(ddq3, parallel.abaInParallel(1, pool, qq, vv, tau1))
    assert rel_err_cols(ddq3, a).max() < TOL_ABA
    with pytest.raises(capi.BrbdInvalidArgument):
        parallel.rneaInParallelOnDevices(1, pool, qq, vv, aa, devices=[])
    pool.release()


def test_release_stream_frees_and_recreates_the_per_stream_buffers():
    """brbd_model_release_stream (include/brbd.h): a stream's workspace / staging share goes away and
    comes back on the next call; results unchanged.  And the table-driven computeMinverse is timed
    like every other enum brbd_algo value (brbd_kernel_time)."""
    import torch
    model = S.quadruped()
    pool = parallel.ModelPool(model, 1)
    dm = pool.device_model(0)
    O = oracle.Oracle(model)
    q, v, a = inputs(model, 5000, 9)
    tau1 = parallel.rneaInParallel(1, pool, q, v, a)   # host memory: staging buffers of the NULL stream
    dm.release_stream(None)
    dm.release_stream(None)                            # (nothing left: a no-op)
    dm.release_stream(ctypes.c_void_p(12345))          # (a stream the handle never saw)
    tau2 = parallel.rneaInParallel(1, pool, q, v, a)
    assert np.array_equal(tau1, tau2)
    assert rel_err_cols(tau1, O.rnea(q, v, a)).max() < 1e-12
    dm.use_compiled(False)
    dm.set_timing(True)
    parallel.computeMinverse(pool, torch.from_numpy(np.ascontiguousarray(q)).cuda())
    n, ms = dm.kernel_time(capi.ALGO_MINVERSE)
    dm.set_timing(False)
    assert n == 1 and ms > 0
    assert dm.workspace_bytes(capi.ALGO_MINVERSE, 64) > 0
    pool.release()
