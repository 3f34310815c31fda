"""ctypes binding of the C ABI in include/brbd.h (pinocchio_amd/libbrbd_hip.so).

This is plumbing only: every symbol declared in include/brbd.h is bound here with its exact
C signature.  There is no fallback: if the shared library is missing or cannot be loaded the
import of the compute path fails loudly with BrbdLibraryError.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

from .model import brbd_model_desc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BRBD_LIB") or os.path.join(_HERE, "libbrbd_hip.so")  # BRBD_LIB: A/B builds

BRBD_OK = 0
BRBD_ERR_INVALID_ARGUMENT = -1
BRBD_ERR_UNSUPPORTED = -2
BRBD_ERR_RUNTIME = -3
BRBD_ERR_NO_DEVICE = -4

INSTANCE_MAJOR = 0
BATCH_MAJOR = 1
HOST = 0
DEVICE = 1

ALGO_RNEA, ALGO_ABA, ALGO_FK_JAC, ALGO_RNEA_DERIV, ALGO_MINVERSE, ALGO_ABA_DERIV_PRODUCTS, ALGO_JOINT_JAC = 0, 1, 2, 3, 4, 5, 6
# pinocchio::ReferenceFrame (multibody/fwd.hpp:46-54)
WORLD, LOCAL, LOCAL_WORLD_ALIGNED = 0, 1, 2

# every extern "C" symbol of include/brbd.h (checked by tests/test_capi_host.py)
SYMBOLS = (
    "brbd_last_error", "brbd_version", "brbd_device_count",
    "brbd_model_create", "brbd_model_destroy", "brbd_model_release_stream", "brbd_model_dims", "brbd_workspace_bytes",
    "brbd_rnea", "brbd_aba", "brbd_rnea_fext", "brbd_aba_fext", "brbd_fk_jacobians", "brbd_joint_jacobian", "brbd_rnea_derivatives",
    "brbd_transpose_to_batch_major", "brbd_transpose_to_instance_major",
    "brbd_set_timing", "brbd_set_timing_period", "brbd_kernel_time", "brbd_model_compiled_kernels", "brbd_model_use_compiled", "brbd_model_outputs_prezeroed", "brbd_model_canonical_form",
    "brbd_load_compiled_kernels", "brbd_integrate", "brbd_minverse", "brbd_aba_derivatives",
    "brbd_aba_noalloc", "brbd_host_register", "brbd_host_unregister", "brbd_debug_oob_count",
)


class BrbdLibraryError(RuntimeError):
    """libbrbd_hip.so is missing / not loadable: there is no CPU fallback."""


class BrbdInvalidArgument(ValueError):
    """Counterpart of std::invalid_argument thrown by the reference's argument checks
    (include/pinocchio/macros.hpp:201-239)."""


class BrbdUnsupported(BrbdInvalidArgument):
    """Joint kind outside the supported set (rejected at flatten time)."""


class BrbdRuntimeError(RuntimeError):
    """HIP failure (reference counterpart: none; mapped to std::runtime_error in the C++ shim)."""


_lib = None


def canonical_form(model):
    """(canonical Model, post[njoints, 3, 3]) as the table-driven kernels see ``model``: unaligned
    joints turned into RZ / PZ joints of a rotated joint frame (include/brbd.h:
    brbd_model_canonical_form).  Host only."""
    import numpy as np
    from . import model as M
    desc, keep = model.to_desc()
    n = model.njoints
    jt = np.zeros(n, dtype=np.int32)
    P, Y, post = np.zeros((n, 12)), np.zeros((n, 10)), np.zeros((n, 9))
    check(lib().brbd_model_canonical_form(ctypes.byref(desc), jt.ctypes.data, P.ctypes.data, Y.ctypes.data,
                                          post.ctypes.data))
    t = model.tables()
    t["jtype"], t["placement"], t["inertia"] = jt.tolist(), P.tolist(), Y.tolist()
    out = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in t.items()}
    import json
    return M.Model.from_json(json.dumps(out)), post.reshape(n, 3, 3)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BrbdLibraryError(
            f"{LIB_PATH} not found: build it with `make -C pinocchio_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`); the engine has no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    # libamdhip64.so.7, same as /opt/rocm's).  Load torch FIRST so that our library's
    # DT_NEEDED libamdhip64.so.7 resolves to the runtime torch uses; otherwise two runtimes
    # coexist, device pointers / streams cannot be shared and the second one sees no device.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch-less deployment uses /opt/rocm's runtime
        pass
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise BrbdLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    dp = ctypes.POINTER(ctypes.c_double)
    vp = ctypes.c_void_p
    i64, i32 = ctypes.c_int64, ctypes.c_int32
    ci = ctypes.c_int
    L.brbd_last_error.restype = ctypes.c_char_p
    L.brbd_last_error.argtypes = []
    L.brbd_version.restype = ctypes.c_char_p
    L.brbd_version.argtypes = []
    L.brbd_device_count.restype = ci
    L.brbd_device_count.argtypes = []
    L.brbd_model_create.restype = ci
    L.brbd_model_create.argtypes = [ctypes.POINTER(brbd_model_desc), ci, ctypes.POINTER(vp)]
    L.brbd_model_destroy.restype = ci
    L.brbd_model_destroy.argtypes = [vp]
    L.brbd_model_release_stream.restype = ci
    L.brbd_model_release_stream.argtypes = [vp, vp]
    L.brbd_model_dims.restype = ci
    L.brbd_model_dims.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.brbd_workspace_bytes.restype = i64
    L.brbd_workspace_bytes.argtypes = [vp, ci, i64]
    L.brbd_rnea.restype = ci
    L.brbd_rnea.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, ci, ci, vp]
    L.brbd_aba.restype = ci
    L.brbd_aba.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, ci, ci, vp]
    L.brbd_load_compiled_kernels.restype = ci
    L.brbd_load_compiled_kernels.argtypes = [ctypes.c_char_p]
    L.brbd_integrate.restype = ci
    L.brbd_integrate.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, ci, ci, vp]
    L.brbd_minverse.restype = ci
    L.brbd_minverse.argtypes = [vp, i64, vp, i64, vp, i64, ci, ci, vp]
    L.brbd_aba_derivatives.restype = ci
    L.brbd_aba_derivatives.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, ci, ci, vp]
    L.brbd_aba_noalloc.restype = ci
    L.brbd_aba_noalloc.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, ci, vp]
    L.brbd_debug_oob_count.restype = i64
    L.brbd_debug_oob_count.argtypes = [vp]
    L.brbd_host_register.restype = ci
    L.brbd_host_register.argtypes = [vp, ctypes.c_size_t]
    L.brbd_host_unregister.restype = ci
    L.brbd_host_unregister.argtypes = [vp]
    L.brbd_rnea_fext.restype = ci
    L.brbd_rnea_fext.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, ci, ci, vp]
    L.brbd_aba_fext.restype = ci
    L.brbd_aba_fext.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, ci, ci, vp]
    L.brbd_fk_jacobians.restype = ci
    L.brbd_fk_jacobians.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, ci, ci, vp]
    L.brbd_joint_jacobian.restype = ci
    L.brbd_joint_jacobian.argtypes = [vp, i64, vp, i64, ci, ci, vp, i64, ci, ci, vp]
    L.brbd_rnea_derivatives.restype = ci
    L.brbd_rnea_derivatives.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64,
                                        vp, i64, ci, ci, vp]
    L.brbd_transpose_to_batch_major.restype = ci
    L.brbd_transpose_to_batch_major.argtypes = [ci, i64, i64, vp, i64, vp, i64, vp]
    L.brbd_transpose_to_instance_major.restype = ci
    L.brbd_transpose_to_instance_major.argtypes = [ci, i64, i64, vp, i64, vp, i64, vp]
    L.brbd_model_compiled_kernels.restype = ctypes.c_char_p
    L.brbd_model_compiled_kernels.argtypes = [vp]
    L.brbd_model_use_compiled.restype = ci
    L.brbd_model_use_compiled.argtypes = [vp, ci]
    L.brbd_model_canonical_form.restype = ci
    L.brbd_model_canonical_form.argtypes = [vp, vp, vp, vp, vp]
    L.brbd_model_outputs_prezeroed.restype = ci
    L.brbd_model_outputs_prezeroed.argtypes = [vp, ci]
    L.brbd_set_timing.restype = ci
    L.brbd_set_timing.argtypes = [vp, ci]
    L.brbd_set_timing_period.restype = ci
    L.brbd_set_timing_period.argtypes = [vp, ci]
    L.brbd_kernel_time.restype = ci
    L.brbd_kernel_time.argtypes = [vp, ci, ctypes.POINTER(i32), dp]
    _lib = L
    return L


def load_compiled_kernels(path: str) -> int:
    """Load a kernel plugin built by pinocchio_amd.codegen.plugin (include/brbd.h:
    brbd_load_compiled_kernels); returns the number of kernel sets added."""
    rc = lib().brbd_load_compiled_kernels(os.fsencode(path))
    if rc < 0:
        check(rc)
    return rc


def last_error() -> str:
    return lib().brbd_last_error().decode()


def check(rc: int):
    if rc == BRBD_OK:
        return
    msg = last_error()
    if rc == BRBD_ERR_INVALID_ARGUMENT:
        raise BrbdInvalidArgument(msg)
    if rc == BRBD_ERR_UNSUPPORTED:
        raise BrbdUnsupported(msg)
    if rc == BRBD_ERR_NO_DEVICE:
        raise BrbdLibraryError(msg)
    raise BrbdRuntimeError(f"brbd error {rc}: {msg}")


class DeviceModel:
    """Owns one `brbd_model*` (flattened BatchModel + device table + workspace)."""

    def __init__(self, model, device: int = 0):
        self.model = model
        self.device = int(device)
        desc, keep = model.to_desc()
        self._keep = keep
        h = ctypes.c_void_p()
        check(lib().brbd_model_create(ctypes.byref(desc), self.device, ctypes.byref(h)))
        self.handle = h
        self.nq, self.nv, self.njoints = model.nq, model.nv, model.njoints

    def close(self):
        if getattr(self, "handle", None):
            lib().brbd_model_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def release_stream(self, stream=None):
        """free the per-stream buffers the handle keeps for ``stream`` (include/brbd.h)"""
        check(lib().brbd_model_release_stream(self.handle, stream))

    def workspace_bytes(self, algo: int, B: int) -> int:
        return int(lib().brbd_workspace_bytes(self.handle, algo, B))

    def compiled_kernels(self) -> str:
        return lib().brbd_model_compiled_kernels(self.handle).decode()

    def use_compiled(self, on: bool):
        check(lib().brbd_model_use_compiled(self.handle, 1 if on else 0))

    def outputs_prezeroed(self, on: bool):
        """Derivative outputs under the reference's contract (caller zeroed them): see include/brbd.h"""
        check(lib().brbd_model_outputs_prezeroed(self.handle, 1 if on else 0))

    def set_timing(self, on: bool, chained: bool = False, dispatch: bool = False, period: int = 1):
        """hipEvents around every kernel launch; ``chained``: one shared event between consecutive
        launches instead of two; ``dispatch``: the compiled kernels' dispatches carry their own begin /
        end timestamps; ``period``: only every period-th launch of an algorithm is timed
        (include/brbd.h: brbd_set_timing, brbd_set_timing_period)."""
        check(lib().brbd_set_timing(self.handle, (3 if dispatch else 2 if chained else 1) if on else 0))
        check(lib().brbd_set_timing_period(self.handle, max(int(period), 1) if on else 1))

    def kernel_time(self, algo: int):
        n = ctypes.c_int32(0)
        ms = ctypes.c_double(0.0)
        check(lib().brbd_kernel_time(self.handle, algo, ctypes.byref(n), ctypes.byref(ms)))
        return int(n.value), float(ms.value)


def _ptr(x) -> Optional[int]:
    """Raw address of a numpy array (host) or torch tensor (device)."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return x.ctypes.data
