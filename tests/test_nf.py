"""RealNVP coupling stack (GLMCMC_NFs.py:51-61): CPU -- the oracle's restatement against an independent eager-PyTorch
implementation (tolerance: PyTorch's Linear sums in its own order); GPU -- the MFMA kernels against the oracle,
bit for bit, plus invertibility at full size."""
import ctypes as C

import numpy as np
import pytest
import torch

from glabcmcmc_amd import _capi as A
from glabcmcmc_amd.flows import RealNVP
from helpers import bits


def make_flow(n_couplings, seed, scale=0.3):
    torch.manual_seed(seed)
    flow = RealNVP(n_couplings)
    with torch.no_grad():
        for c in flow.couplings:                      # non-zero last layer, otherwise every coupling is the identity
            c.l3.weight.normal_(0, scale / 128 ** 0.5)
            c.l3.bias.normal_(0, 0.1)
        flow.q0.loc.copy_(torch.tensor([[0.2, -0.1]]))
        flow.q0.log_scale.copy_(torch.tensor([[0.1, -0.2]]))
    return flow


def host_descriptor(flow):
    blob = flow.packed_params().numpy().copy()
    f = flow.descriptor(torch.from_numpy(blob))
    f.params = blob.ctypes.data
    return f, blob


@pytest.mark.parametrize("n_couplings", [1, 2, 8])
def test_oracle_matches_eager_pytorch(oracle, n_couplings):
    flow = make_flow(n_couplings, 10 + n_couplings)
    f, blob = host_descriptor(flow)
    n = 777
    eps = torch.randn(n, 2)
    e = np.ascontiguousarray(eps.numpy().T)
    z = np.empty((2, n), np.float32)
    lq = np.empty(n, np.float32)
    assert oracle.oracle_nf_sample(C.byref(f), e.ctypes.data, 0, 0, n, z.ctypes.data, lq.ctypes.data) == 0
    with torch.no_grad():
        zt, lqt = flow.sample_torch(n, eps)
    assert np.allclose(z.T, zt.numpy(), rtol=2e-5, atol=2e-5)
    assert np.allclose(lq, lqt.numpy(), rtol=2e-5, atol=2e-5)
    # log_prob of the samples returns their log_q (forward / inverse consistency), in both implementations
    lp = np.empty(n, np.float32)
    assert oracle.oracle_nf_log_prob(C.byref(f), z.ctypes.data, n, lp.ctypes.data) == 0
    assert np.allclose(lp, lq, rtol=1e-4, atol=1e-4)
    with torch.no_grad():
        lpt = flow.log_prob_torch(zt)
    assert np.allclose(lp, lpt.numpy(), rtol=1e-4, atol=1e-4)
    # Philox base noise path
    assert oracle.oracle_nf_sample(C.byref(f), None, 5, 100, n, z.ctypes.data, lq.ctypes.data) == 0
    assert np.isfinite(z).all() and abs(z.mean()) < 1.0


def test_zero_initialised_flow_is_the_base(oracle):
    """init_zeros=True (GLMCMC_NFs.py:56): every coupling starts as the identity, so the flow is its base"""
    flow = RealNVP(4)
    f, blob = host_descriptor(flow)
    n = 64
    eps = np.random.default_rng(0).standard_normal((2, n)).astype(np.float32)
    z = np.empty((2, n), np.float32)
    lq = np.empty(n, np.float32)
    assert oracle.oracle_nf_sample(C.byref(f), eps.ctypes.data, 0, 0, n, z.ctypes.data, lq.ctypes.data) == 0
    assert np.array_equal(z, eps)                               # 4 swaps = identity, loc 0, scale 1
    assert np.allclose(lq, -np.log(2 * np.pi) - 0.5 * (eps ** 2).sum(0), atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("n_couplings,n", [(1, 100), (3, 513), (8, 4096), (32, 1000)])
def test_mfma_kernels_equal_oracle(hip, oracle, n_couplings, n):
    flow = make_flow(n_couplings, 20 + n_couplings)
    f, blob = host_descriptor(flow)
    eps = torch.randn(n, 2)
    e = np.ascontiguousarray(eps.numpy().T)
    z = np.empty((2, n), np.float32)
    lq = np.empty(n, np.float32)
    assert oracle.oracle_nf_sample(C.byref(f), e.ctypes.data, 0, 0, n, z.ctypes.data, lq.ctypes.data) == 0
    gflow = flow.cuda()
    zg, lqg = gflow.sample(n, eps=eps)
    torch.cuda.synchronize()
    assert np.array_equal(bits(zg.cpu().numpy()), bits(z.T))
    assert np.array_equal(bits(lqg.cpu().numpy()), bits(lq))
    lp = np.empty(n, np.float32)
    assert oracle.oracle_nf_log_prob(C.byref(f), z.ctypes.data, n, lp.ctypes.data) == 0
    lpg = gflow.log_prob(zg)
    assert np.array_equal(bits(lpg.cpu().numpy()), bits(lp))
    # Philox base noise: same stream on both sides
    assert oracle.oracle_nf_sample(C.byref(f), None, 77, 1 << 33, n, z.ctypes.data, lq.ctypes.data) == 0
    zg, lqg = gflow.sample(n, seed=77, row0=1 << 33)
    assert np.array_equal(bits(zg.cpu().numpy()), bits(z.T)) and np.array_equal(bits(lqg.cpu().numpy()), bits(lq))


@pytest.mark.gpu
def test_full_size_invertibility(hip):
    """BASELINE config 5's shape: 8 couplings, 65 536 chains x N=5 = 327 680 rows.  log_prob(sample) == log_q and the
    eager-PyTorch implementation on the same device agrees to float32 round-off."""
    flow = make_flow(8, 99).cuda()
    n = 327680
    z, lq = flow.sample(n, seed=3)
    lp = flow.log_prob(z)
    torch.cuda.synchronize()
    assert torch.isfinite(z).all() and torch.isfinite(lq).all()
    assert (lp - lq).abs().max().item() < 2e-4
    with torch.no_grad():
        lpt = flow.log_prob_torch(z[:20000])
    assert (lpt - lp[:20000]).abs().max().item() < 2e-4
