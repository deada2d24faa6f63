"""The SLP-vectorizer exposure of the offline-compiled kernels, enumerated instead of sampled.

ROCm 7.2's SLP vectorizer has miscompiled this code base once -- the run-time compiled sampler kernel for some
(simulator, batch size) pairs; tools/ubench/slp_check.hip is the standalone reproducer.  The product's offline kernels keep the
pass where it pays (the one-wavefront-per-SIMD sampler kernels), so

  (1) EVERY sampler instantiation the dispatcher can pick -- theta_dim 1..8 x batch size 1..16 x lanes per chain 1 / 2 / 4 x
      the unit-Gaussian and the generic variant x both instruction schedules, GlobalMCMC, the wide kernel's lane groups, the
      team geometries, the g-and-k shape -- is run for a few iterations, with simulator noise that reads both Box-Muller
      pairs of a candidate's Philox block, against the CPU checker, bit for bit;
  (2) a twin of the whole library built with -fno-slp-vectorize (csrc/libglabc_hip_noslp.so, test infrastructure) is run on
      the same seeds through the same C ABI and must return the same bits: samplers, split-phase kernels, flow, KDE.
GPU only.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import oracle_lib
from helpers import bits, make_dist

pytestmark = pytest.mark.gpu

N_CHAINS, T = 70, 4          # two wavefronts' worth (the second ragged) x a few iterations: every decision kind occurs


def _model(d, eps=0.3):
    from glabcmcmc_amd import _capi as A
    from glabcmcmc_amd import distribution
    prior = distribution.DiagGaussian(d, torch.zeros(d), torch.zeros(d)).descriptor()
    noise = distribution.DiagGaussian(d, torch.zeros(d), torch.log(torch.full((d,), 0.05).sqrt())).descriptor()
    kern = distribution.DiagGaussian(1, torch.tensor([0.0]), torch.log(torch.tensor([eps]))).descriptor()
    m = A.Model()
    m.sim_kind, m.theta_dim, m.y_dim = A.SIM_ABS_GAUSS, d, d
    m.prior, m.noise = prior, noise
    for j in range(d):
        m.y_obs[j] = 1.5 - 0.125 * j
    m.kern_log_scale, m.kern_scale, m.kern_c0, m.epsilon = kern.p1[0], kern.p2[0], kern.c0, eps
    return m


def _proposals(d, unit):
    """unit: prior == global == N(0, I) -> the branch-free VAR_GAUSS_UNIT kernels; else VAR_GENERIC"""
    local = make_dist(("gauss", [0.0] * d, [0.3] * d)).descriptor()
    glob = make_dist(("gauss", [0.0] * d, [1.0] * d) if unit else ("gauss", [0.1] * d, [1.25] * d)).descriptor()
    return local, glob


def _inputs(d, yd, seed):
    rng = np.random.default_rng(seed)
    theta0 = rng.standard_normal((N_CHAINS, d)).astype(np.float32)
    y0 = (np.abs(rng.standard_normal((N_CHAINS, yd))) + 0.2).astype(np.float32)
    return theta0, y0


def _run_lib(lib, algo, model, local, glob, theta0, y0, N, lanes, flags, seed, n_steps=T):
    """one launch through the C ABI of `lib` (a ctypes handle: the product or the twin) -> (history, theta, y, log_w, flags)"""
    from glabcmcmc_amd import _capi as A
    from glabcmcmc_amd import engine
    dev = torch.device("cuda", 0)
    chains = engine.ChainBatch(torch.from_numpy(theta0), torch.from_numpy(y0), dev, chain0=12345678901)
    cs = chains.struct()
    if algo == "glmcmc":
        assert lib.glabc_init_weights(C.byref(model), C.byref(glob), C.byref(cs), None) == 0
    hist = torch.empty(n_steps, chains.d, chains.n, dtype=torch.float32, device=dev)
    run = A.Run()
    run.seed, run.step0, run.n_steps, run.global_frequency, run.batch_size = seed, 1, n_steps, 0.7, N
    run.history, run.hist_stride, run.lanes_per_chain, run.debug_flags = hist.data_ptr(), chains.n, lanes, flags
    fn = lib.glabc_glmcmc_steps if algo == "glmcmc" else lib.glabc_globalmcmc_steps
    rc = fn(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run), None)
    assert rc == 0, (rc, algo, N, lanes, flags)
    torch.cuda.synchronize()
    return (hist.cpu().numpy(), chains.theta.cpu().numpy(), chains.y.cpu().numpy(), chains.log_w.cpu().numpy(),
            chains.flags.cpu().numpy())


def _run_oracle(oracle, algo, model, local, glob, theta0, y0, N, seed, n_steps=T):
    hc = oracle_lib.HostChains(theta0, y0, chain0=12345678901)
    hh = np.zeros((n_steps, theta0.shape[1], theta0.shape[0]), np.float32)
    run, keep = oracle_lib.make_run(seed=seed, step0=1, n_steps=n_steps, gf=0.7, batch=N, history=hh)
    cs = hc.struct()
    if algo == "glmcmc":
        assert oracle.oracle_init_weights(C.byref(model), C.byref(glob), C.byref(cs)) == 0
        assert oracle.oracle_glmcmc_steps(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run)) == 0
    else:
        assert oracle.oracle_globalmcmc_steps(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run)) == 0
    return hh, hc.theta, hc.y, hc.log_w, hc.flags


def _same(got, want, isir, what):
    assert np.array_equal(bits(got[0]), bits(want[0])), ("history", what)
    assert np.array_equal(bits(got[1]), bits(want[1])) and np.array_equal(bits(got[2]), bits(want[2])), ("state", what)
    if isir:
        assert np.array_equal(bits(got[3]), bits(want[3])), ("log_w", what)
        assert np.array_equal(got[4].astype(np.uint32), want[4].astype(np.uint32)), ("flags", what)


@pytest.fixture(scope="module")
def product():
    from glabcmcmc_amd import _capi
    return _capi.lib()


@pytest.fixture(scope="module")
def twin():
    from glabcmcmc_amd import _capi
    path = os.path.join(os.path.dirname(_capi.LIB_PATH), "libglabc_hip_noslp.so")
    assert os.path.exists(path), "the -fno-slp-vectorize twin is built by gl-abc-mcmc_amd/csrc/Makefile (python __graft_entry__.py build)"
    return _capi.bind(path)


# ---------------------------------------------------------------------------------- (1) every instantiation vs the checker
@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 8])
def test_every_register_kernel_instantiation_equals_the_checker(product, oracle, d):
    """sampler_kernel<ALGO, D, D, N, L, VAR, SCHED>: N 1..16 x L 1 / 2 / 4 x {VAR_GAUSS_UNIT, VAR_GENERIC} x {max-ilp, default}
    schedule (GLABC_DEBUG_DEFAULT_SCHEDULE reaches the default-schedule one-lane kernels with a small launch) + GlobalMCMC"""
    from glabcmcmc_amd import _capi as A
    model = _model(d)
    theta0, y0 = _inputs(d, d, 100 + d)
    walked = 0
    for unit in (True, False):
        local, glob = _proposals(d, unit)
        for N in range(1, 17):
            want = _run_oracle(oracle, "glmcmc", model, local, glob, theta0, y0, N, 7000 + N)
            for lanes, flags in ((1, A.DEBUG_NO_TEAM), (1, A.DEBUG_NO_TEAM | A.DEBUG_DEFAULT_SCHEDULE), (2, A.DEBUG_NO_TEAM),
                                 (4, A.DEBUG_NO_TEAM)):
                got = _run_lib(product, "glmcmc", model, local, glob, theta0, y0, N, lanes, flags, 7000 + N)
                _same(got, want, True, dict(d=d, N=N, lanes=lanes, unit=unit, flags=flags))
                walked += 1
        want = _run_oracle(oracle, "globalmcmc", model, local, glob, theta0, y0, 1, 99)
        for flags in (0, A.DEBUG_DEFAULT_SCHEDULE):
            got = _run_lib(product, "globalmcmc", model, local, glob, theta0, y0, 1, 0, flags, 99)
            _same(got, want, False, dict(d=d, algo="globalmcmc", unit=unit, flags=flags))
            walked += 1
    assert walked == 2 * (16 * 4 + 2)


@pytest.mark.parametrize("d", [1, 2, 3, 4])
def test_every_wide_and_team_instantiation_equals_the_checker(product, oracle, d, monkeypatch):
    """wide_kernel<D, D, L> for L 8 / 16 / 32 / 64 (batch sizes on both sides of one candidate per lane) and
    team_sampler_kernel<D, D, N, VAR, NW> for NW 2 / 3 / 4 x N 2..16 x both variants"""
    from glabcmcmc_amd import _capi as A
    model = _model(d)
    theta0, y0 = _inputs(d, d, 200 + d)
    for unit in (True, False):
        local, glob = _proposals(d, unit)
        for N in (17, 40, 100):
            want = _run_oracle(oracle, "glmcmc", model, local, glob, theta0, y0, N, 8000 + N)
            for lanes in (8, 16, 32, 64):
                got = _run_lib(product, "glmcmc", model, local, glob, theta0, y0, N, lanes, 0, 8000 + N)
                _same(got, want, True, dict(d=d, N=N, lanes=lanes, unit=unit, kernel="wide"))
        for N in range(2, 17):
            want = _run_oracle(oracle, "glmcmc", model, local, glob, theta0, y0, N, 9000 + N)
            for nw in (2, 3, 4):
                monkeypatch.setenv("GLABC_TEAM_WAVES", str(nw))
                got = _run_lib(product, "glmcmc", model, local, glob, theta0, y0, N, 0, A.DEBUG_TEAM, 9000 + N)
                _same(got, want, True, dict(d=d, N=N, team=nw, unit=unit))
        monkeypatch.delenv("GLABC_TEAM_WAVES", raising=False)


@pytest.mark.parametrize("d", [1, 2, 3, 4])
def test_every_gamma_instantiation_equals_the_checker(product, oracle, d, monkeypatch):
    """VAR_GAMMA (a Gamma importance proposal and a Gamma prior, distribution.py:90-137): sampler_kernel<GLMCMC, D, D, N, 1,
    VAR_GAMMA> for N 1..16, GlobalMCMC, and team_sampler_kernel<D, D, N, VAR_GAMMA, NW> for NW 2 / 3 x N 2..16 -- built without
    the SLP vectorizer, walked all the same"""
    from glabcmcmc_amd import _capi as A
    from glabcmcmc_amd import distribution
    model = _model(d)
    model.prior = distribution.Gamma(torch.full((d,), 2.0), torch.full((d,), 1.0)).descriptor()
    local = make_dist(("gauss", [0.0] * d, [0.3] * d)).descriptor()
    glob = distribution.Gamma(torch.full((d,), 4.0), torch.full((d,), 3.0)).descriptor()
    theta0, y0 = _inputs(d, d, 300 + d)
    theta0 = np.abs(theta0) + np.float32(0.2)                   # inside the Gamma prior's support
    for N in range(1, 17):
        want = _run_oracle(oracle, "glmcmc", model, local, glob, theta0, y0, N, 6000 + N)
        got = _run_lib(product, "glmcmc", model, local, glob, theta0, y0, N, 0, A.DEBUG_NO_TEAM, 6000 + N)
        _same(got, want, True, dict(d=d, N=N, kernel="gamma one lane"))
        if N >= 2:
            for nw in (2, 3):
                monkeypatch.setenv("GLABC_TEAM_WAVES", str(nw))
                got = _run_lib(product, "glmcmc", model, local, glob, theta0, y0, N, 0, A.DEBUG_TEAM, 6000 + N)
                _same(got, want, True, dict(d=d, N=N, team=nw, kernel="gamma"))
            monkeypatch.delenv("GLABC_TEAM_WAVES", raising=False)
    want = _run_oracle(oracle, "globalmcmc", model, local, glob, theta0, y0, 1, 98)
    got = _run_lib(product, "globalmcmc", model, local, glob, theta0, y0, 1, 0, 0, 98)
    _same(got, want, False, dict(d=d, algo="globalmcmc", kernel="gamma"))


def _gk(eps=0.6):
    from glabcmcmc_amd import distribution
    from glabcmcmc_amd.examples.GK import GK_set
    model = GK_set(eps).descriptor()
    local = distribution.DiagGaussian(4, torch.zeros(1, 4), torch.log(torch.tensor([0.15, 0.1, 0.2, 0.1]))).descriptor()
    glob = distribution.Uniform(4, torch.zeros(4), torch.full((4,), 10.0)).descriptor()
    rng = np.random.default_rng(5)
    theta0 = (rng.random((N_CHAINS, 4)) * 10).astype(np.float32)
    y0 = np.sort(rng.standard_normal((N_CHAINS, 8)) * 2 + 3, axis=1).astype(np.float32)
    return model, local, glob, theta0, y0


def test_every_gk_instantiation_equals_the_checker(product, oracle, monkeypatch):
    """the g-and-k shape (theta_dim 4, y_dim 8): register kernels N 1..16 x L 1 / 2 / 4, team N 2..16, wide"""
    from glabcmcmc_amd import _capi as A
    model, local, glob, theta0, y0 = _gk()
    for N in list(range(1, 17)) + [20, 40]:
        want = _run_oracle(oracle, "glmcmc", model, local, glob, theta0, y0, N, 300 + N)
        cases = [(lanes, A.DEBUG_NO_TEAM) for lanes in ((1, 2, 4) if N <= 16 else (8, 16))]
        if 2 <= N <= 16:
            cases.append((0, A.DEBUG_TEAM))
        for lanes, flags in cases:
            got = _run_lib(product, "glmcmc", model, local, glob, theta0, y0, N, lanes, flags, 300 + N)
            _same(got, want, True, dict(gk=True, N=N, lanes=lanes, flags=flags))
    want = _run_oracle(oracle, "globalmcmc", model, local, glob, theta0, y0, 1, 31)
    _same(_run_lib(product, "globalmcmc", model, local, glob, theta0, y0, 1, 0, 0, 31), want, False, dict(gk=True, algo="globalmcmc"))


# ---------------------------------------------------------------------------------- (2) the -fno-slp-vectorize twin
@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 8])
def test_twin_samplers_give_the_same_bits(product, twin, d):
    """the product and its -fno-slp-vectorize twin, same seeds, same C ABI: equal histories, states, weights -- longer runs
    than (1) (60 iterations), every lanes-per-chain value, both variants, both schedules, GlobalMCMC, wide"""
    from glabcmcmc_amd import _capi as A
    model = _model(d)
    theta0, y0 = _inputs(d, d, 400 + d)
    for unit in (True, False):
        local, glob = _proposals(d, unit)
        for N in (1, 2, 3, 5, 8, 12, 16):
            for lanes, flags in ((1, A.DEBUG_NO_TEAM), (1, A.DEBUG_NO_TEAM | A.DEBUG_DEFAULT_SCHEDULE), (2, A.DEBUG_NO_TEAM),
                                 (4, A.DEBUG_NO_TEAM)):
                a = _run_lib(product, "glmcmc", model, local, glob, theta0, y0, N, lanes, flags, 77 + N, n_steps=60)
                b = _run_lib(twin, "glmcmc", model, local, glob, theta0, y0, N, lanes, flags, 77 + N, n_steps=60)
                _same(a, b, True, dict(d=d, N=N, lanes=lanes, unit=unit, flags=flags, twin=True))
        for flags in (0, A.DEBUG_DEFAULT_SCHEDULE):
            a = _run_lib(product, "globalmcmc", model, local, glob, theta0, y0, 1, 0, flags, 5, n_steps=60)
            b = _run_lib(twin, "globalmcmc", model, local, glob, theta0, y0, 1, 0, flags, 5, n_steps=60)
            _same(a, b, False, dict(d=d, algo="globalmcmc", unit=unit, twin=True))
        if d <= 4:
            for N, lanes in ((24, 8), (70, 16), (130, 32), (300, 64)):
                a = _run_lib(product, "glmcmc", model, local, glob, theta0, y0, N, lanes, 0, 9 + N, n_steps=12)
                b = _run_lib(twin, "glmcmc", model, local, glob, theta0, y0, N, lanes, 0, 9 + N, n_steps=12)
                _same(a, b, True, dict(d=d, N=N, lanes=lanes, unit=unit, kernel="wide", twin=True))


def test_twin_gk_flow_kde_and_split_phase_give_the_same_bits(product, twin):
    """the rest of the SLP-compiled kernels through both libraries: g-and-k sampler, RealNVP sample / log_prob (MFMA kernels),
    KernelDensity.log_prob, glabc_propose / glabc_select, the row-wise Model callbacks"""
    from glabcmcmc_amd import _capi as A
    from glabcmcmc_amd import KernelDensity, engine
    from glabcmcmc_amd.flows import RealNVP
    dev = torch.device("cuda", 0)
    model, local, glob, theta0, y0 = _gk()
    for N, lanes in ((1, 1), (5, 1), (5, 2), (16, 4), (40, 8)):
        a = _run_lib(product, "glmcmc", model, local, glob, theta0, y0, N, lanes, A.DEBUG_NO_TEAM, 3 + N, n_steps=30)
        b = _run_lib(twin, "glmcmc", model, local, glob, theta0, y0, N, lanes, A.DEBUG_NO_TEAM, 3 + N, n_steps=30)
        _same(a, b, True, dict(gk=True, N=N, lanes=lanes, twin=True))
    # flow
    torch.manual_seed(1)
    flow = RealNVP(8)
    with torch.no_grad():
        for c in flow.couplings:
            c.l3.weight.normal_(0, 0.3 / 128 ** 0.5)
            c.l3.bias.normal_(0, 0.1)
    flow = flow.cuda()
    blob = flow.packed_params()
    f = flow.descriptor(blob)
    for rows in (100, 4096, 70001):
        outs = []
        for lib in (product, twin):
            z = torch.empty(2, rows, dtype=torch.float32, device=dev)
            lq = torch.empty(rows, dtype=torch.float32, device=dev)
            lp = torch.empty(rows, dtype=torch.float32, device=dev)
            assert lib.glabc_nf_sample(C.byref(f), None, 1234, 5, rows, z.data_ptr(), lq.data_ptr(), None) == 0
            assert lib.glabc_nf_log_prob(C.byref(f), z.data_ptr(), rows, lp.data_ptr(), None) == 0
            torch.cuda.synchronize()
            outs.append((z.cpu().numpy(), lq.cpu().numpy(), lp.cpu().numpy()))
        for x, y in zip(*outs):
            assert np.array_equal(bits(x), bits(y)), ("flow", rows)
    # KDE
    kde = KernelDensity(device="cuda", seed=1).fit(torch.randn(3000, 2), torch.rand(3000))
    k = kde.descriptor()
    pts = torch.randn(2, 5000, device=dev)
    outs = []
    for lib in (product, twin):
        out = torch.empty(5000, dtype=torch.float32, device=dev)
        assert lib.glabc_kde_log_prob(C.byref(k), pts.data_ptr(), 5000, out.data_ptr(), None) == 0
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    assert np.array_equal(bits(outs[0]), bits(outs[1]))
    # split-phase kernels + row-wise callbacks (theta_dim 3)
    m3 = _model(3)
    l3, g3 = _proposals(3, False)
    N = 6
    th0, yy0 = _inputs(3, 3, 9)
    outs = []
    for lib in (product, twin):
        chains = engine.ChainBatch(torch.from_numpy(th0), torch.from_numpy(yy0), dev)
        cs = chains.struct()
        R = N * chains.n
        f32 = dict(dtype=torch.float32, device=dev)
        buf = dict(theta_prop=torch.zeros(R, 3, **f32), log_q=torch.zeros(R, **f32), noise=torch.zeros(R, 3, **f32),
                   log_u=torch.zeros(chains.n, **f32), u_res=torch.zeros(chains.n, dtype=torch.float64, device=dev),
                   is_global=torch.zeros(chains.n, dtype=torch.int32, device=dev), y=torch.zeros(R, 3, **f32),
                   prior=torch.zeros(R, **f32), kern=torch.zeros(R, **f32), prior_cur=torch.zeros(chains.n, **f32),
                   kern_cur=torch.zeros(chains.n, **f32))
        io = A.StepIO(N, 3, 3, 3, buf["theta_prop"].data_ptr(), buf["log_q"].data_ptr(), buf["noise"].data_ptr(),
                      buf["log_u"].data_ptr(), buf["u_res"].data_ptr(), buf["is_global"].data_ptr(), buf["y"].data_ptr(),
                      buf["prior"].data_ptr(), buf["kern"].data_ptr(), buf["prior_cur"].data_ptr(), buf["kern_cur"].data_ptr(), None)
        run = A.Run()
        run.seed, run.step0, run.n_steps, run.global_frequency, run.batch_size = 21, 1, 1, 0.6, N
        hist = torch.zeros(3, chains.n, **f32)
        run.history, run.hist_stride = hist.data_ptr(), chains.n
        assert lib.glabc_model_prior_log_prob(C.byref(m3), chains.theta.t().contiguous().data_ptr(), chains.n, buf["prior_cur"].data_ptr(), None) == 0
        assert lib.glabc_model_log_kernel(C.byref(m3), chains.y.t().contiguous().data_ptr(), chains.n, buf["kern_cur"].data_ptr(), None) == 0
        assert lib.glabc_propose(A.ALGO_GLMCMC, C.byref(l3), C.byref(g3), C.byref(cs), C.byref(run), C.byref(io), None) == 0
        assert lib.glabc_model_simulate(C.byref(m3), buf["theta_prop"].data_ptr(), buf["noise"].data_ptr(), R, 0, 0, buf["y"].data_ptr(), None) == 0
        assert lib.glabc_model_prior_log_prob(C.byref(m3), buf["theta_prop"].data_ptr(), R, buf["prior"].data_ptr(), None) == 0
        assert lib.glabc_model_log_kernel(C.byref(m3), buf["y"].data_ptr(), R, buf["kern"].data_ptr(), None) == 0
        assert lib.glabc_select(A.ALGO_GLMCMC, C.byref(g3), C.byref(cs), C.byref(run), C.byref(io), None) == 0
        torch.cuda.synchronize()
        outs.append([v.cpu().numpy() for v in (buf["theta_prop"], buf["log_q"], buf["noise"], buf["y"], buf["prior"], buf["kern"], hist,
                                               chains.theta, chains.log_w)])
    for x, y in zip(*outs):
        assert np.array_equal(bits(x.astype(np.float32)), bits(y.astype(np.float32)))
