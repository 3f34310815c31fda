"""brbd_aba_noalloc (include/brbd.h): the entry point that allocates nothing -- device pointers, the
table-driven ABA kernel, its workspace in a fixed file-scope device array.  Checked through the
C ABI against brbd_aba on two calls in a row, the second with a smaller input (the shared array
keeps the first call's contents), plus the stated limits."""
import ctypes

import numpy as np
import pytest

import oracle
from pinocchio_amd import capi, sample_models as S
from util import inputs, rel_err_cols

pytestmark = pytest.mark.gpu

TOL_ABA = 1e-10


def _call(fn, dm, B, q, v, tau, out, layout, *extra):
    ld = (lambda t: t.stride(0) if layout == capi.BATCH_MAJOR else t.stride(1))
    return fn(dm.handle, B, q.data_ptr(), ld(q), v.data_ptr(), ld(v), tau.data_ptr(), ld(tau), out.data_ptr(), ld(out),
              layout, *extra)


@pytest.mark.parametrize("name", ["random_tree", "simple_humanoid", "talos_like"])
@pytest.mark.parametrize("layout", [capi.BATCH_MAJOR, capi.INSTANCE_MAJOR], ids=["batch_major", "instance_major"])
def test_noalloc_matches_brbd_aba_twice(name, layout):
    import torch
    L = capi.lib()
    model = S.random_tree(100, 12) if name == "random_tree" else S.build(name)
    dm = capi.DeviceModel(model, 0)
    dev = torch.device("cuda", 0)
    for B in (5000, 777):  # second call smaller: stale contents of the shared array must not matter
        q, v, tau = inputs(model, B, B)

        def put(x):
            t = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
            return t if layout == capi.BATCH_MAJOR else torch.from_numpy(np.ascontiguousarray(x.T)).to(dev).t()

        qd, vd, td = put(q), put(v), put(tau)
        ref, out = put(np.zeros((model.nv, B))), put(np.full((model.nv, B), np.nan))
        capi.check(_call(L.brbd_aba, dm, B, qd, vd, td, ref, layout, capi.DEVICE, None))
        capi.check(_call(L.brbd_aba_noalloc, dm, B, qd, vd, td, out, layout, None))
        torch.cuda.synchronize()
        # directly against the oracle (aba.hxx:243-294 restated), to the bar of the GPU parity suite
        assert rel_err_cols(out.cpu().numpy(), oracle.Oracle(model).aba(q, v, tau)).max() < TOL_ABA
        # brbd_aba may run the model-compiled kernel: same result, rounding-level differences
        assert rel_err_cols(out.cpu().numpy(), ref.cpu().numpy()).max() < TOL_ABA
        # against the table-driven kernel of brbd_aba the result is bitwise the same
        dm.use_compiled(False)
        capi.check(_call(L.brbd_aba, dm, B, qd, vd, td, ref, layout, capi.DEVICE, None))
        torch.cuda.synchronize()
        dm.use_compiled(True)
        assert torch.equal(out, ref)
    dm.close()


def test_noalloc_limits_and_argument_checks():
    import torch
    L = capi.lib()
    model = S.manipulator()
    dm = capi.DeviceModel(model, 0)
    x = torch.zeros((6, 8), dtype=torch.float64, device="cuda:0")
    p = x.data_ptr()
    big = (1 << 20) + 1
    assert L.brbd_aba_noalloc(dm.handle, big, p, big, p, big, p, big, p, big, capi.BATCH_MAJOR, None) == capi.BRBD_ERR_INVALID_ARGUMENT
    assert "exceeds the maximum" in capi.last_error()
    assert L.brbd_aba_noalloc(dm.handle, 8, p, 8, p, 8, p, 3, p, 8, capi.BATCH_MAJOR, None) == capi.BRBD_ERR_INVALID_ARGUMENT
    assert "tau" in capi.last_error()
    assert L.brbd_aba_noalloc(dm.handle, 0, p, 8, p, 8, p, 8, p, 8, capi.BATCH_MAJOR, None) == capi.BRBD_OK
    dm.close()


def test_noalloc_under_stream_capture_allocates_nothing():
    """The reason the entry point exists: it can be captured into a HIP graph (an allocation or a
    synchronisation inside the capture would fail it), and the replayed graph gives the result."""
    import torch
    L = capi.lib()
    model = S.quadruped()
    dm = capi.DeviceModel(model, 0)
    dm.use_compiled(False)
    B = 1024
    q, v, tau = inputs(model, B, 3)
    dev = torch.device("cuda", 0)
    qd, vd, td = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (q, v, tau))
    ref, out = torch.zeros((model.nv, B), dtype=torch.float64, device=dev), torch.zeros((model.nv, B), dtype=torch.float64, device=dev)
    capi.check(_call(L.brbd_aba, dm, B, qd, vd, td, ref, capi.BATCH_MAJOR, capi.DEVICE, None))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=s):
        capi.check(_call(L.brbd_aba_noalloc, dm, B, qd, vd, td, out, capi.BATCH_MAJOR,
                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    dm.close()
