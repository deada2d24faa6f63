"""Proposal draws and simulator noise of a candidate must be independent (include/glabc_numerics.h stream layout).

With an odd theta_dim a Box-Muller pair used to straddle the proposal / simulator boundary, so a Uniform proposal's
last coordinate and the simulator's first normal came from the same Philox word: the simulated y' then depended on
theta' beyond p(y | theta') and the sampler targeted the wrong law.  The simulator's normals now start at the next even
word.  These tests would have caught it: at theta_dim 1 and 3 a Uniform-proposal run must reach the analytic posterior
moments of the |theta| + Gaussian-noise model just as a Gaussian-proposal run does.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib
from glabcmcmc_amd import _capi as A
from glabcmcmc_amd import distribution


def abs_gauss_model(d, eps, y_obs=1.5):
    """glabc_model of y = |theta| + N(0, 0.05 I), prior N(0, I), Gaussian kernel of width eps -- examples/Mixture.py in d dims"""
    m = A.Model()
    m.sim_kind, m.theta_dim, m.y_dim = A.SIM_ABS_GAUSS, d, d
    m.prior = distribution.DiagGaussian(d, torch.zeros(d), torch.zeros(d)).descriptor()
    m.noise = distribution.DiagGaussian(d, torch.zeros(d), torch.log(torch.full((d,), 0.05).sqrt())).descriptor()
    for j in range(d):
        m.y_obs[j] = y_obs
    k = distribution.DiagGaussian(1, torch.tensor([0.0]), torch.log(torch.tensor([eps]))).descriptor()
    m.kern_log_scale, m.kern_scale, m.kern_c0 = k.p1[0], k.p2[0], k.c0
    m.epsilon = float(np.float32(eps))
    return m


def analytic(eps, y_obs=1.5):
    v = 0.05 + eps * eps                       # the Gaussian kernel on the L2 distance factorises per coordinate
    mu, s2 = y_obs / (1.0 + v), v / (1.0 + v)
    return mu, mu * mu + s2                    # E|theta_j|, E theta_j^2 (truncation at 0 is > 4 sigma away)


CASES = [(1, "uniform"), (1, "gauss"), (3, "uniform"), (3, "gauss")]


def proposals(d, kind):
    local = distribution.DiagGaussian(d, torch.zeros(d), torch.log(torch.full((d,), 0.4)))
    if kind == "uniform":
        glob = distribution.Uniform(d, torch.full((d,), -3.0), torch.full((d,), 3.0))
    else:
        glob = distribution.DiagGaussian(d, torch.zeros(d), torch.log(torch.full((d,), 1.3)))
    return local.descriptor(), glob.descriptor()


def check_moments(d, s_abs_unused, mean_sq, n_eff_chains, eps):
    _, want_sq = analytic(eps)
    # chains are independent; the tolerance is 5 standard errors of the chain-mean of theta^2 (estimated from the chains)
    se = mean_sq.std(ddof=1) / np.sqrt(len(mean_sq))
    got = mean_sq.mean()
    assert abs(got - want_sq) < 5 * se + 2e-3 * want_sq, (d, got, want_sq, se)


@pytest.mark.parametrize("d,kind", CASES)
def test_oracle_reaches_analytic_moments(d, kind):
    L = oracle_lib.load()
    eps, n, burn, T = 0.3, 1024, 300, 1500
    model = abs_gauss_model(d, eps)
    lp, ip = proposals(d, kind)
    rng = np.random.default_rng(d)
    theta0 = np.full((n, d), 1.3, np.float32) * rng.choice([-1.0, 1.0], (n, d)).astype(np.float32)
    y0 = (np.abs(theta0) + 0.2236 * rng.standard_normal((n, d))).astype(np.float32)
    hc = oracle_lib.HostChains(theta0, y0)
    cs = hc.struct()
    assert L.oracle_init_weights(C.byref(model), C.byref(ip), C.byref(cs)) == 0
    run, keep = oracle_lib.make_run(seed=99 + d, step0=1, n_steps=burn, gf=0.7, batch=4)
    assert L.oracle_glmcmc_steps(C.byref(model), C.byref(lp), C.byref(ip), C.byref(cs), C.byref(run)) == 0
    mom = oracle_lib.HostMoments(n, d)
    run, keep = oracle_lib.make_run(seed=99 + d, step0=1 + burn, n_steps=T, gf=0.7, batch=4, moments=mom)
    assert L.oracle_glmcmc_steps(C.byref(model), C.byref(lp), C.byref(ip), C.byref(cs), C.byref(run)) == 0
    k = 0
    for a in range(d):                                           # diagonal entries of the upper triangle
        check_moments(d, None, mom.sum_outer[k] / T, n, eps)
        k += d - a


@pytest.mark.gpu
@pytest.mark.parametrize("d,kind", CASES)
def test_hip_reaches_analytic_moments(hip, d, kind):
    from glabcmcmc_amd import engine
    dev = torch.device("cuda", 0)
    eps, n, burn, T = 0.3, 16384, 300, 1500
    model = abs_gauss_model(d, eps)
    lp, ip = proposals(d, kind)
    g = torch.Generator().manual_seed(d)
    theta0 = 1.3 * (torch.randint(0, 2, (n, d), generator=g).float() * 2 - 1)
    y0 = theta0.abs() + 0.2236 * torch.randn(n, d, generator=g)
    chains = engine.ChainBatch(theta0, y0, dev)
    engine.init_weights(model, ip, chains)
    engine.run_steps("glabc_glmcmc_steps", model, lp, ip, chains, burn, 1, 1234 + d, 0.7, 4)
    mom = engine.Moments(n, d, dev)
    engine.run_steps("glabc_glmcmc_steps", model, lp, ip, chains, T, 1 + burn, 1234 + d, 0.7, 4, moments=mom)
    so = mom.sum_outer.cpu().numpy()
    k = 0
    for a in range(d):
        check_moments(d, None, so[k] / T, n, eps)
        k += d - a
