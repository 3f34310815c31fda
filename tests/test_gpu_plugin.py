"""A model the library was NOT built with gets compiled kernels from a run-time plugin
(pinocchio_amd/plugins/libbrbd_spec_baxter_simple.so, built by __graft_entry__.build() with
pinocchio_amd/codegen/plugin.py): selected by signature, parity with the oracle on every algorithm
and layout, and faster than the table-driven kernels.  Runs in a child process: a loaded plugin
stays registered for the life of a process and would change which kernels later tests get."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGIN = os.path.join(ROOT, "pinocchio_amd", "plugins", "libbrbd_spec_baxter_simple.so")

SCRIPT = r"""
import sys, time
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np, torch
import oracle
from pinocchio_amd import capi, parallel, sample_models as S
from util import inputs
import test_gpu_parity as T
m = S.baxter_simple()
pool = parallel.ModelPool(m, 1)
assert pool.device_model(0).compiled_kernels() == ""          # not built in
pool.release()
assert capi.load_compiled_kernels(%(plugin)r) == 1
pool = parallel.ModelPool(m, 1)
dm = pool.device_model(0)
assert dm.compiled_kernels() == "baxter_simple_x_p", dm.compiled_kernels()
T._check_all(pool, m, oracle.Oracle(m))                        # every algorithm, both layouts, vs the oracle
# same tree with other numbers (another robot of this kind): same plugin, constants from its own table
m2 = S.baxter_simple()
for i in range(1, m2.njoints):
    m2.inertias[i].mass *= 1.3
    m2.jointPlacements[i][:3, 3] *= 1.1
pool2 = parallel.ModelPool(m2, 1)
assert pool2.device_model(0).compiled_kernels() == "baxter_simple_x_p"
T._check_all(pool2, m2, oracle.Oracle(m2))
pool2.release()
B = 65536
q, v, a = inputs(m, B, 1)
qd, vd, ad = (torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0") for x in (q, v, a))
rate = {}
for comp in (True, False):
    dm.use_compiled(comp)
    for _ in range(50):
        tau = parallel.rneaInParallel(1, pool, qd, vd, ad); ddq = parallel.abaInParallel(1, pool, qd, vd, tau)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(100):
        tau = parallel.rneaInParallel(1, pool, qd, vd, ad); ddq = parallel.abaInParallel(1, pool, qd, vd, tau)
    torch.cuda.synchronize()
    rate[comp] = 2 * B * 100 / (time.perf_counter() - t)
print("plugin %%.3e evals/s, table-driven %%.3e evals/s" %% (rate[True], rate[False]))
assert rate[True] > 2.0 * rate[False]
print("PLUGIN-OK")
"""


def test_plugin_kernels_for_a_model_unknown_at_build_time():
    assert os.path.exists(PLUGIN), "build it with __graft_entry__.build() (pinocchio_amd.codegen.plugin)"
    r = subprocess.run([sys.executable, "-u", "-c", SCRIPT % {"root": ROOT, "tests": os.path.join(ROOT, "tests"),
                                                                "plugin": PLUGIN}],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PLUGIN-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


CACHE = os.path.join(ROOT, "pinocchio_amd", "plugins", "cache")
SCRIPT_CACHE = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np
import oracle
from pinocchio_amd import parallel, sample_models as S
from pinocchio_amd.codegen import plugin
from util import inputs, rel_err_cols
m = S.random_tree(7, 4)
assert plugin.cached_plugin(m, %(cache)r, build_missing=False) is not None, "run __graft_entry__.build()"
pool = parallel.ModelPool(m, 1)
assert pool.device_model(0).compiled_kernels() == ""
pool.release()
pool = parallel.ModelPool(m, 1, compile_kernels=True, plugin_cache=%(cache)r)     # cache hit: nothing is compiled here
assert pool.device_model(0).compiled_kernels().startswith("m_")
q, v, a = inputs(m, 300, 4)
O = oracle.Oracle(m)
tau = parallel.rneaInParallel(1, pool, q, v, a)
assert rel_err_cols(tau, O.rnea(q, v, a)).max() < 1e-12
assert rel_err_cols(parallel.abaInParallel(1, pool, q, v, tau), a).max() < 1e-10
print("CACHE-OK")
"""


def test_model_pool_takes_compiled_kernels_from_the_plugin_cache():
    r = subprocess.run([sys.executable, "-u", "-c", SCRIPT_CACHE % {"root": ROOT, "tests": os.path.join(ROOT, "tests"),
                                                                      "cache": CACHE}],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CACHE-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
