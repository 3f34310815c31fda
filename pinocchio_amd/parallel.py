"""Python mirror of the reference's data-parallel entry points, running on the HIP engine.

Same names, argument order, return values and error behaviour as the Boost.Python proxies
(reference ``bindings/python/algorithm/parallel/rnea.cpp:15-63``,
``bindings/python/algorithm/parallel/aba.cpp:15-63``,
``include/pinocchio/bindings/python/multibody/pool/model.hpp:43-67``):

    pool = ModelPool(model[, size])
    tau  = rneaInParallel(num_threads, pool, q, v, a)        # (nv, B)
           rneaInParallel(num_threads, pool, q, v, a, tau)   # in place
    ddq  = abaInParallel(num_threads, pool, q, v, tau)       # (nv, B)

plus the batched forms of the single-instance calls of the hot path
(``forwardKinematics`` / ``computeJointJacobians`` / ``computeRNEADerivatives``).

Matrices are ``(dof, B)``.  numpy arrays live on the host and are staged by the C ABI; torch
CUDA(ROCm) tensors are used in place on the current torch stream.  Fortran-ordered arrays
(what Eigen hands the reference: one instance per contiguous column) map to
BRBD_INSTANCE_MAJOR, C-ordered ones to the engine-native BRBD_BATCH_MAJOR; no copy is made for
either.  Everything is computed by libbrbd_hip.so -- there is no CPU path here.
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np

from . import capi
from .capi import BrbdInvalidArgument
from .model import Model

try:  # torch is only the carrier of device memory and streams
    import torch
except Exception:  # pragma: no cover
    torch = None


def omp_get_max_threads() -> int:
    """``pin.omp_get_max_threads()`` (bindings/python/algorithm/parallel/expose-parallel.cpp:24-26).
    The HIP engine does not use host threads; the value only feeds the reference's
    ``num_threads <= pool.size()`` check."""
    return os.cpu_count() or 1


class ModelPool:
    """``pin.ModelPool`` (multibody/pool/model.hpp:19-165).  The reference keeps ``size`` copies
    of Model and Data, one per OpenMP thread; here the replicas are one flattened BatchModel
    per GPU, created lazily, and ``size`` only drives the reference's argument checks."""

    def __init__(self, model: Model, pool_size: Optional[int] = None, compile_kernels: bool = False,
                 plugin_cache: Optional[str] = None):
        """``compile_kernels``: a model the library has no compiled kernels for gets them from the
        plugin cache (``plugin_cache``, $BRBD_PLUGIN_CACHE, ~/.cache/brbd_plugins), built on first use
        with hipcc (codegen/plugin.py: minutes for a humanoid); off by default -- such a model then
        runs on the table-driven kernels."""
        self._compile = bool(compile_kernels)
        self._plugin_cache = plugin_cache
        if pool_size is None:
            pool_size = omp_get_max_threads()  # pool/model.hpp:41
        if pool_size < 0:
            raise BrbdInvalidArgument("pool_size must be >= 0")
        self._model = model
        self._size = int(pool_size)
        self._dev = {}

    def size(self) -> int:
        return self._size

    def resize(self, new_size: int):
        if new_size < 0:
            raise BrbdInvalidArgument("new_size must be >= 0")
        self._size = int(new_size)

    def getModel(self, index: int = 0) -> Model:
        if not (0 <= index < max(self._size, 1)):
            raise BrbdInvalidArgument("Index greater than the size of the model vector.")  # :87-90
        return self._model

    def getModels(self) -> List[Model]:
        return [self._model] * self._size

    def update(self, model: Model):
        """pool/model.hpp:68-73: replace the model of every replica; device tables are rebuilt."""
        self._model = model
        self.release()

    def release(self):
        for d in self._dev.values():
            d.close()
        self._dev = {}

    def device_model(self, device: int = 0, replica: int = 0) -> capi.DeviceModel:
        """The flattened model on GPU `device` (`replica` > 0: a further handle on the same GPU, for
        callers that drive one GPU from several host threads)."""
        key = device if replica == 0 else (device, replica)
        d = self._dev.get(key)
        if d is None:
            d = capi.DeviceModel(self._model, device)
            if self._compile and d.compiled_kernels() == "":
                from .codegen import plugin
                path = plugin.cached_plugin(self._model, self._plugin_cache)
                if path is not None:
                    capi.load_compiled_kernels(path)
                    d.close()
                    d = capi.DeviceModel(self._model, device)
            self._dev[key] = d
        return d


# --------------------------------------------------------------------------------------------
class _Arg:
    __slots__ = ("ptr", "ld", "layout", "space", "B", "keep", "device", "is_torch")


def _describe(x, rows: int, name: str, prefer: Optional[int] = None) -> _Arg:
    """Describe one (rows, B) matrix for the C ABI.  Host numpy arrays whose ordering differs
    from `prefer` (the ordering of q) are copied into it, like the reference's Eigen copies."""
    a = _Arg()
    if torch is not None and isinstance(x, torch.Tensor):
        if x.dtype != torch.float64:
            raise BrbdInvalidArgument(f"{name} must be float64")
        if x.dim() != 2 or x.shape[0] != rows:
            raise BrbdInvalidArgument(f"{name} has shape {tuple(x.shape)}, expected ({rows}, B)")
        a.B = int(x.shape[1])
        s0, s1 = x.stride()
        if a.B == 1 and s0 == 1:
            a.layout, a.ld = capi.INSTANCE_MAJOR, rows
        elif s1 == 1:
            a.layout, a.ld = capi.BATCH_MAJOR, (int(s0) if rows > 1 else max(int(s0), a.B))
        elif s0 == 1:
            a.layout, a.ld = capi.INSTANCE_MAJOR, int(s1)
        else:
            raise BrbdInvalidArgument(f"{name} must have unit stride along one dimension")
        a.space = capi.DEVICE if x.is_cuda else capi.HOST
        a.device = x.device.index if x.is_cuda else None
        a.ptr, a.keep, a.is_torch = x.data_ptr(), x, True
        return a
    arr = np.asarray(x)
    if arr.dtype != np.float64:
        arr = arr.astype(np.float64)
    if arr.ndim == 1:
        arr = arr.reshape(-1, 1)
    if arr.ndim != 2 or arr.shape[0] != rows:
        raise BrbdInvalidArgument(f"{name} has shape {arr.shape}, expected ({rows}, B)")
    if not (arr.flags.c_contiguous or arr.flags.f_contiguous):
        arr = np.asfortranarray(arr)
    a.B = int(arr.shape[1])
    if prefer == capi.INSTANCE_MAJOR and not arr.flags.f_contiguous:
        arr = np.asfortranarray(arr)
    elif prefer == capi.BATCH_MAJOR and not arr.flags.c_contiguous:
        arr = np.ascontiguousarray(arr)
    if prefer is not None:
        a.layout, a.ld = prefer, (rows if prefer == capi.INSTANCE_MAJOR else a.B)
        a.space, a.device, a.ptr, a.keep, a.is_torch = capi.HOST, None, arr.ctypes.data, arr, False
        return a
    if arr.flags.f_contiguous and (a.B == 1 or not arr.flags.c_contiguous):
        a.layout, a.ld = capi.INSTANCE_MAJOR, rows
    else:
        a.layout, a.ld = capi.BATCH_MAJOR, a.B
    a.space, a.device, a.ptr, a.keep, a.is_torch = capi.HOST, None, arr.ctypes.data, arr, False
    return a


def _same(args: List[_Arg]):
    a0 = args[0]
    for a in args[1:]:
        if a.B != a0.B:
            raise BrbdInvalidArgument("The number of columns of the inputs do not match")  # rnea.hpp:64-68
        if a.layout != a0.layout or a.space != a0.space:
            raise BrbdInvalidArgument("all matrices of one call must share memory space and ordering")
    return a0


def _alloc_like(ref_obj, a0: _Arg, rows: int):
    """Output with the ordering/memory of the inputs."""
    if a0.is_torch:
        t = ref_obj
        if a0.layout == capi.BATCH_MAJOR:
            return torch.empty((rows, a0.B), dtype=torch.float64, device=t.device)
        return torch.empty((a0.B, rows), dtype=torch.float64, device=t.device).t()
    return np.empty((rows, a0.B), dtype=np.float64, order="C" if a0.layout == capi.BATCH_MAJOR else "F")


def _stream(a0: _Arg):
    if a0.is_torch and a0.space == capi.DEVICE:
        return torch.cuda.current_stream(a0.device).cuda_stream
    return None


def _pool_checks(num_threads: int, pool: ModelPool):
    # algorithm/parallel/rnea.hpp:53-54 (PINOCCHIO_CHECK_INPUT_ARGUMENT -> std::invalid_argument)
    if not pool.size() > 0:
        raise BrbdInvalidArgument("The pool should have at least one element")
    if not num_threads <= pool.size():
        raise BrbdInvalidArgument("The pool is too small")


def _device_of(a0: _Arg, device: Optional[int]) -> int:
    if a0.space == capi.DEVICE:
        return a0.device or 0
    return 0 if device is None else device


def rneaInParallel(num_threads: int, pool: ModelPool, q, v, a, tau=None, device: Optional[int] = None):
    """``pin.rneaInParallel(num_thread, pool, q, v, a[, tau])`` -> tau (nv, B)
    (algorithm/parallel/rnea.hpp:39-83)."""
    _pool_checks(num_threads, pool)
    m = pool.getModel(0)
    aq = _describe(q, m.nq, "q")
    av, aa = _describe(v, m.nv, "v", aq.layout), _describe(a, m.nv, "a", aq.layout)
    a0 = _same([aq, av, aa])
    ret = tau if tau is not None else _alloc_like(q, a0, m.nv)
    at = _describe(ret, m.nv, "tau", aq.layout)
    _same([aq, at])
    dm = pool.device_model(_device_of(a0, device))
    capi.check(capi.lib().brbd_rnea(dm.handle, a0.B, aq.ptr, aq.ld, av.ptr, av.ld, aa.ptr, aa.ld,
                                    at.ptr, at.ld, a0.layout, a0.space, _stream(a0)))
    return ret if tau is None or at.keep is ret else _copy_back(tau, at)


def rnea(pool: ModelPool, q, v, a, fext, tau=None, device: Optional[int] = None):
    """Batched ``pin.rnea(model, data, q, v, a, fext)`` (algorithm/rnea.hxx:171-219): ``fext`` is
    (6 * njoints, B), rows 6i..6i+5 = the force [linear; angular] on joint i in its own frame
    (entry 0, the universe's, is ignored like in the reference)."""
    m = pool.getModel(0)
    aq = _describe(q, m.nq, "q")
    av, aa = _describe(v, m.nv, "v", aq.layout), _describe(a, m.nv, "a", aq.layout)
    af = _describe(fext, 6 * m.njoints, "fext", aq.layout)
    a0 = _same([aq, av, aa, af])
    ret = tau if tau is not None else _alloc_like(q, a0, m.nv)
    at = _describe(ret, m.nv, "tau", aq.layout)
    _same([aq, at])
    dm = pool.device_model(_device_of(a0, device))
    capi.check(capi.lib().brbd_rnea_fext(dm.handle, a0.B, aq.ptr, aq.ld, av.ptr, av.ld, aa.ptr, aa.ld, af.ptr,
                                         af.ld, at.ptr, at.ld, a0.layout, a0.space, _stream(a0)))
    return ret if tau is None or at.keep is ret else _copy_back(tau, at)


def aba(pool: ModelPool, q, v, tau, fext, a=None, device: Optional[int] = None):
    """Batched ``pin.aba(model, data, q, v, tau, fext)``, WORLD convention (algorithm/aba.hxx:296-354);
    ``fext`` as for :func:`rnea`."""
    m = pool.getModel(0)
    aq = _describe(q, m.nq, "q")
    av, at = _describe(v, m.nv, "v", aq.layout), _describe(tau, m.nv, "tau", aq.layout)
    af = _describe(fext, 6 * m.njoints, "fext", aq.layout)
    a0 = _same([aq, av, at, af])
    ret = a if a is not None else _alloc_like(q, a0, m.nv)
    ad = _describe(ret, m.nv, "a", aq.layout)
    _same([aq, ad])
    dm = pool.device_model(_device_of(a0, device))
    capi.check(capi.lib().brbd_aba_fext(dm.handle, a0.B, aq.ptr, aq.ld, av.ptr, av.ld, at.ptr, at.ld, af.ptr,
                                        af.ld, ad.ptr, ad.ld, a0.layout, a0.space, _stream(a0)))
    return ret if a is None or ad.keep is ret else _copy_back(a, ad)


def abaInParallel(num_threads: int, pool: ModelPool, q, v, tau, a=None, device: Optional[int] = None):
    """``pin.abaInParallel(num_thread, pool, q, v, tau[, a])`` -> ddq (nv, B), WORLD convention
    (algorithm/parallel/aba.hpp:40-84)."""
    _pool_checks(num_threads, pool)
    m = pool.getModel(0)
    aq = _describe(q, m.nq, "q")
    av, at = _describe(v, m.nv, "v", aq.layout), _describe(tau, m.nv, "tau", aq.layout)
    a0 = _same([aq, av, at])
    ret = a if a is not None else _alloc_like(q, a0, m.nv)
    ad = _describe(ret, m.nv, "a", aq.layout)
    _same([aq, ad])
    dm = pool.device_model(_device_of(a0, device))
    capi.check(capi.lib().brbd_aba(dm.handle, a0.B, aq.ptr, aq.ld, av.ptr, av.ld, at.ptr, at.ld,
                                   ad.ptr, ad.ld, a0.layout, a0.space, _stream(a0)))
    return ret if a is None or ad.keep is ret else _copy_back(a, ad)


def _copy_back(dst, arg: _Arg):
    """In-place overloads: the caller's array was not directly usable (wrong dtype / strides)."""
    np.copyto(np.asarray(dst), arg.keep)
    return dst


def computeJointJacobians(pool: ModelPool, q, want_oMi: bool = False, device: Optional[int] = None):
    """Batched ``forwardKinematics(q)`` + ``computeJointJacobians(q)``
    (algorithm/jacobian.hxx:38-95): returns J of shape (6*nv, B) -- per instance the 6 x nv
    WORLD-frame ``data.J`` in column-major order -- and, if asked, oMi (12*njoints, B)."""
    m = pool.getModel(0)
    aq = _describe(q, m.nq, "q")
    J = _alloc_like(q, aq, 6 * m.nv)
    aJ = _describe(J, 6 * m.nv, "J")
    oMi = _alloc_like(q, aq, 12 * m.njoints) if want_oMi else None
    aM = _describe(oMi, 12 * m.njoints, "oMi") if want_oMi else None
    dm = pool.device_model(_device_of(aq, device))
    capi.check(capi.lib().brbd_fk_jacobians(dm.handle, aq.B, aq.ptr, aq.ld, aJ.ptr, aJ.ld,
                                            aM.ptr if aM else None, aM.ld if aM else 0,
                                            aq.layout, aq.space, _stream(aq)))
    return (J, oMi) if want_oMi else J


def forwardKinematics(pool: ModelPool, q, device: Optional[int] = None):
    """Batched ``forwardKinematics(q)`` (algorithm/kinematics.hxx:52-97): oMi (12*njoints, B),
    per joint the rotation in column-major order then the translation."""
    m = pool.getModel(0)
    aq = _describe(q, m.nq, "q")
    oMi = _alloc_like(q, aq, 12 * m.njoints)
    aM = _describe(oMi, 12 * m.njoints, "oMi")
    dm = pool.device_model(_device_of(aq, device))
    capi.check(capi.lib().brbd_fk_jacobians(dm.handle, aq.B, aq.ptr, aq.ld, None, 0, aM.ptr, aM.ld,
                                            aq.layout, aq.space, _stream(aq)))
    return oMi


def getJointJacobian(pool: ModelPool, q, joint_id: int, reference_frame: int = capi.LOCAL,
                     device: Optional[int] = None):
    """Batched ``computeJointJacobians(q)`` + ``getJointJacobian(joint_id, reference_frame)``
    (algorithm/jacobian.hxx:38-95, 175-283): (6*nv, B), per instance a column-major 6 x nv matrix.
    ``reference_frame``: capi.WORLD / LOCAL / LOCAL_WORLD_ALIGNED."""
    m = pool.getModel(0)
    aq = _describe(q, m.nq, "q")
    J = _alloc_like(q, aq, 6 * m.nv)
    aj = _describe(J, 6 * m.nv, "J", aq.layout)
    dm = pool.device_model(_device_of(aq, device))
    capi.check(capi.lib().brbd_joint_jacobian(dm.handle, aq.B, aq.ptr, aq.ld, int(joint_id), int(reference_frame),
                                              aj.ptr, aj.ld, aq.layout, aq.space, _stream(aq)))
    return J


def computeRNEADerivatives(pool: ModelPool, q, v, a, want_tau: bool = False, device: Optional[int] = None,
                           out=None):
    """Batched ``computeRNEADerivatives`` (algorithm/rnea-derivatives.hxx:472-541): returns
    (dtau_dq, dtau_dv, dtau_da), each (nv*nv, B) with per-instance column-major nv x nv blocks;
    dtau_da holds the upper triangle of M like the reference.  ``out`` = (dtau_dq, dtau_dv,
    dtau_da) already ZEROED by the caller selects the reference's contract
    (rnea-derivatives.hpp:104-106): they are filled in place and structural zeros are not
    rewritten."""
    m = pool.getModel(0)
    aq = _describe(q, m.nq, "q")
    av, aa = _describe(v, m.nv, "v", aq.layout), _describe(a, m.nv, "a", aq.layout)
    a0 = _same([aq, av, aa])
    nn = m.nv * m.nv
    outs = list(out) if out is not None else [_alloc_like(q, a0, nn) for _ in range(3)]
    d = [_describe(o, nn, "d", aq.layout) for o in outs]
    _same([aq] + d)
    if out is not None and any(x.keep is not o for x, o in zip(d, outs)):
        raise BrbdInvalidArgument("out matrices must be usable in place (float64, ordered like q)")
    tau = _alloc_like(q, a0, m.nv) if want_tau else None
    at = _describe(tau, m.nv, "tau") if want_tau else None
    dm = pool.device_model(_device_of(a0, device))
    dm.outputs_prezeroed(out is not None)
    capi.check(capi.lib().brbd_rnea_derivatives(
        dm.handle, a0.B, aq.ptr, aq.ld, av.ptr, av.ld, aa.ptr, aa.ld, d[0].ptr, d[0].ld, d[1].ptr, d[1].ld,
        d[2].ptr, d[2].ld, at.ptr if at else None, at.ld if at else 0, a0.layout, a0.space, _stream(a0)))
    return (outs[0], outs[1], outs[2], tau) if want_tau else tuple(outs)


def integrate(pool: ModelPool, q, v, device: Optional[int] = None):
    """Batched ``pin.integrate(model, q, v)`` (algorithm/joint-configuration.hxx): (nq, B)."""
    m = pool.getModel(0)
    aq = _describe(q, m.nq, "q")
    av = _describe(v, m.nv, "v", aq.layout)
    a0 = _same([aq, av])
    out = _alloc_like(q, a0, m.nq)
    ao = _describe(out, m.nq, "qout", aq.layout)
    dm = pool.device_model(_device_of(a0, device))
    capi.check(capi.lib().brbd_integrate(dm.handle, a0.B, aq.ptr, aq.ld, av.ptr, av.ld, ao.ptr, ao.ld, a0.layout,
                                         a0.space, _stream(a0)))
    return out


def computeMinverse(pool: ModelPool, q, device: Optional[int] = None):
    """Batched ``pin.computeMinverse(model, data, q)`` (algorithm/aba.hxx:841-877): (nv*nv, B), per
    instance column-major; upper triangle like ``data.Minv``, strictly lower part zero."""
    m = pool.getModel(0)
    aq = _describe(q, m.nq, "q")
    out = _alloc_like(q, aq, m.nv * m.nv)
    ao = _describe(out, m.nv * m.nv, "Minv", aq.layout)
    dm = pool.device_model(_device_of(aq, device))
    capi.check(capi.lib().brbd_minverse(dm.handle, aq.B, aq.ptr, aq.ld, ao.ptr, ao.ld, aq.layout, aq.space, _stream(aq)))
    return out


def computeABADerivatives(pool: ModelPool, q, v, tau, want_ddq: bool = False, device: Optional[int] = None):
    """Batched ``pin.computeABADerivatives(model, data, q, v, tau)`` (algorithm/aba-derivatives.hxx:381-454):
    (ddq_dq, ddq_dv, ddq_dtau), each (nv*nv, B) column-major per instance, ddq_dtau = Minv full;
    with ``want_ddq`` also ddq (nv, B)."""
    m = pool.getModel(0)
    aq = _describe(q, m.nq, "q")
    av, at = _describe(v, m.nv, "v", aq.layout), _describe(tau, m.nv, "tau", aq.layout)
    a0 = _same([aq, av, at])
    nn = m.nv * m.nv
    outs = [_alloc_like(q, a0, nn) for _ in range(3)]
    d = [_describe(o, nn, "d", aq.layout) for o in outs]
    acc = _alloc_like(q, a0, m.nv) if want_ddq else None
    aa = _describe(acc, m.nv, "ddq", aq.layout) if want_ddq else None
    dm = pool.device_model(_device_of(a0, device))
    capi.check(capi.lib().brbd_aba_derivatives(
        dm.handle, a0.B, aq.ptr, aq.ld, av.ptr, av.ld, at.ptr, at.ld, d[0].ptr, d[0].ld, d[1].ptr, d[1].ld,
        d[2].ptr, d[2].ld, aa.ptr if aa else None, aa.ld if aa else 0, a0.layout, a0.space, _stream(a0)))
    return (outs[0], outs[1], outs[2], acc) if want_ddq else tuple(outs)


# --------------------------------------------------------------------------------------------
# One process, several GPUs: the static column split of the reference's OpenMP loop
# (algorithm/parallel/rnea.hpp:74-82: thread t gets a contiguous block of columns) with one host
# thread + one device model + one stream context per GPU instead of one thread per core.  Host
# matrices only (each slice is staged by its own device's pipeline); no collective, no peer copies.
def _sharded(fn_name, pool: ModelPool, devices, mats, rows, out_index):
    import threading
    from .launch import column_slice
    m = pool.getModel(0)
    descs = [_describe(x, r, n) for x, r, n in zip(mats, rows, ("q", "v", "x", "out"))]
    a0 = _same(descs)
    if a0.space != capi.HOST:
        raise BrbdInvalidArgument("the multi-device helpers take host matrices (device tensors already live on one GPU)")
    G = len(devices)
    if G < 1:
        raise BrbdInvalidArgument("devices must name at least one GPU")
    handles = [pool.device_model(d, replica=k) for k, d in enumerate(devices)]
    fn = getattr(capi.lib(), fn_name)
    errors = []

    def work(k):
        lo, hi = column_slice(a0.B, k, G)
        if hi <= lo:
            return
        # slice k of every matrix: same leading dimension, shifted base pointer
        ptrs = [d.ptr + 8 * (lo * d.ld if d.layout == capi.INSTANCE_MAJOR else lo) for d in descs]
        rc = fn(handles[k].handle, hi - lo, ptrs[0], descs[0].ld, ptrs[1], descs[1].ld, ptrs[2], descs[2].ld,
                ptrs[3], descs[3].ld, a0.layout, capi.HOST, None)
        if rc != capi.BRBD_OK:
            errors.append((rc, capi.last_error()))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise capi.BrbdRuntimeError("a device slice failed: %r" % (errors[0],))
    return descs[out_index].keep


def rneaInParallelOnDevices(num_threads: int, pool: ModelPool, q, v, a, devices, tau=None):
    """rneaInParallel with the batch split into contiguous column slices, slice k on GPU devices[k]
    (one host thread per GPU, all slices in flight at once)."""
    _pool_checks(num_threads, pool)
    m = pool.getModel(0)
    aq = _describe(q, m.nq, "q")
    out = tau if tau is not None else _alloc_like(q, aq, m.nv)
    return _sharded("brbd_rnea", pool, devices, (q, v, a, out), (m.nq, m.nv, m.nv, m.nv), 3)


def abaInParallelOnDevices(num_threads: int, pool: ModelPool, q, v, tau, devices, a=None):
    """abaInParallel with slice k of the columns on GPU devices[k]."""
    _pool_checks(num_threads, pool)
    m = pool.getModel(0)
    aq = _describe(q, m.nq, "q")
    out = a if a is not None else _alloc_like(q, aq, m.nv)
    return _sharded("brbd_aba", pool, devices, (q, v, tau, out), (m.nq, m.nv, m.nv, m.nv), 3)
