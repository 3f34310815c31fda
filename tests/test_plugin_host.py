"""Kernel plugins (pinocchio_amd/codegen/plugin.py + brbd_load_compiled_kernels): the model compiler
runs on a model the library was not built with, hipcc cross-compiles its kernels for gfx950 and the
library accepts the result -- host side only (no GPU needed to build or to load)."""
import os

import pytest

from pinocchio_amd import capi, sample_models as S
from pinocchio_amd.codegen import joints, plugin


def test_build_and_load_a_plugin_for_an_unregistered_model(tmp_path):
    model = S.random_tree(778, 4)
    out = str(tmp_path / "libbrbd_spec_tiny.so")
    rep = plugin.build_plugin(model, out, name="tiny tree!")   # the name is sanitised into an identifier
    assert rep["name"] == "tiny_tree__p" and os.path.exists(out)
    assert len(rep["kernels"]) == 35 and not rep["refused"]   # rnea 5, aba 5, fkjac 9, deriv 4, minv 2, prod 1, jjac 9
    assert not any(int(k["ScratchInsts"]) != 0 and int(k["SGPRs Spill"]) > 8 for k in rep["kernels"])
    assert capi.load_compiled_kernels(out) == 1
    assert capi.load_compiled_kernels(out) == 0                # same signature: not added twice


def test_loading_rejects_what_is_not_a_plugin(tmp_path):
    with pytest.raises(capi.BrbdInvalidArgument, match="cannot load"):
        capi.load_compiled_kernels(str(tmp_path / "missing.so"))
    with pytest.raises(capi.BrbdInvalidArgument, match="not a brbd kernel plugin"):
        capi.load_compiled_kernels(capi.LIB_PATH.replace("libbrbd_hip.so", "../oracle/libbrbd_oracle.so"))


def test_a_urdf_with_a_fixed_base_is_a_different_kernel_set():
    """The reference's simple_humanoid.urdf WITHOUT free-flyer (unittest/urdf.cpp:52-60: nq == 29) is a
    model the library has no kernels for: its signature differs from the registered one, so
    `python -m pinocchio_amd.codegen.plugin --urdf ... --fixed-base` is how it gets compiled kernels
    (the 29-joint ABA / derivative kernels take minutes to compile: only the generator side here)."""
    urdf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "simple_humanoid.urdf")
    from pinocchio_amd import urdf as U
    with open(urdf) as f:
        fixed = U.build_model_from_urdf(f.read(), free_flyer=False)
    assert fixed.nq == 29 and fixed.nv == 29
    assert joints.ModelView(fixed).signature() != joints.ModelView(S.simple_humanoid()).signature()


def test_plugin_cache_builds_once_and_keys_on_the_signature(tmp_path):
    """codegen/plugin.cached_plugin: the first request builds (a 4-joint tree: seconds), the second is a
    cache hit; another robot of the same kind (same signature) shares the entry, another tree does not."""
    from pinocchio_amd import model as M, sample_models as S
    from pinocchio_amd.codegen import plugin
    m1 = S.random_tree(7, 4)
    cache = str(tmp_path / "cache")
    assert plugin.cached_plugin(m1, cache, build_missing=False) is None
    p1 = plugin.cached_plugin(m1, cache)
    assert p1 is not None and os.path.exists(p1)
    t = os.path.getmtime(p1)
    assert plugin.cached_plugin(m1, cache) == p1 and os.path.getmtime(p1) == t
    assert plugin.cache_key(S.random_tree(8, 4)) != plugin.cache_key(m1)
