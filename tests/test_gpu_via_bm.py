"""INSTANCE_MAJOR device calls of the output-bound algorithms (computeRNEADerivatives,
computeMinverse, computeABADerivatives) run through batch-major copies of column chunks
(brbd_capi.hip, via_batch_major): same kernels, so the results must equal the BATCH_MAJOR call's
bit for bit, over several chunks, ragged tails and padded leading dimensions, and match the oracle."""
import numpy as np
import pytest

import oracle
from pinocchio_amd import parallel, sample_models as S
from util import inputs, rel_err_cols

pytestmark = pytest.mark.gpu


def _both(torch, x, pad=0):
    """(batch-major tensor, instance-major view of the same values with ld = rows + pad)"""
    bm = torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
    rows, B = x.shape
    store = torch.full((B, rows + pad), -7.0, dtype=torch.float64, device="cuda:0")
    store[:, :rows] = bm.t()
    return bm, store[:, :rows].t()


@pytest.mark.parametrize("pad", [0, 3])
def test_rnea_derivatives_instance_major_equals_batch_major(pad):
    import torch
    model = S.quadruped()
    pool = parallel.ModelPool(model, 1)
    B = 2 * 128448 + 4097 + 37        # three column chunks (1 GiB of copies each), ragged
    q, v, a = inputs(model, B, 5)
    (qb, qi), (vb, vi), (ab, ai) = (_both(torch, x, pad) for x in (q, v, a))
    ref = parallel.computeRNEADerivatives(pool, qb, vb, ab, want_tau=True)
    got = parallel.computeRNEADerivatives(pool, qi, vi, ai, want_tau=True)
    for r, g in zip(ref, got):
        assert g.stride(0) == 1 and torch.equal(g, r)
    # the reference's contract (outputs zeroed by the caller) gives the same matrices
    nn = model.nv * model.nv
    outs = [torch.zeros((B, nn), dtype=torch.float64, device="cuda:0").t() for _ in range(3)]
    parallel.computeRNEADerivatives(pool, qi, vi, ai, out=outs)
    for r, g in zip(ref, outs):
        assert torch.equal(g, r)
    # sampled columns against the oracle (first / last chunk, chunk seams)
    idx = np.array([0, 63, 64, 87039, 87040, 87041, 174079, 174080, B - 65, B - 1])   # (3 equal chunks of 87040)
    O = oracle.Oracle(model)
    rq, rv, ra = O.rnea_derivatives(q[:, idx], v[:, idx], a[:, idx])[:3]
    it = torch.from_numpy(idx).to("cuda:0")
    assert rel_err_cols(got[0][:, it].cpu().numpy(), rq).max() < 1e-10
    assert rel_err_cols(got[1][:, it].cpu().numpy(), rv).max() < 1e-10
    assert rel_err_cols(got[2][:, it].cpu().numpy(), ra).max() < 1e-10
    pool.release()


def test_minverse_and_aba_derivatives_instance_major_equal_batch_major():
    import torch
    model = S.simple_humanoid()
    pool = parallel.ModelPool(model, 1)
    B = 4096 + 129
    q, v, tau = inputs(model, B, 6)
    (qb, qi), (vb, vi), (tb, ti) = (_both(torch, x) for x in (q, v, tau))
    assert torch.equal(parallel.computeMinverse(pool, qi), parallel.computeMinverse(pool, qb))
    ref = parallel.computeABADerivatives(pool, qb, vb, tb, want_ddq=True)
    got = parallel.computeABADerivatives(pool, qi, vi, ti, want_ddq=True)
    for r, g in zip(ref, got):
        assert torch.equal(g, r)
    got3 = parallel.computeABADerivatives(pool, qi, vi, ti)      # ddq not requested: internal temporary
    for r, g in zip(ref[:3], got3):
        assert torch.equal(g, r)
    pool.release()
