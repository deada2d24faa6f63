"""The split-phase path for Models given as callbacks (glabc_propose -> Model callbacks -> glabc_select; generic.py).

CPU part (oracle only): the split-phase restatement in oracle/ -- the reference's iteration cut where the loop calls the
Model -- reproduces the REFERENCE's golden chains bit for bit and equals the one-piece restatement.
GPU part: a Model object with nothing but the reference's protocol (no descriptor()) runs through the HIP kernels and the
Python loop of generic.py and visits the reference's chains bit for bit when its callbacks are the build's row-wise
kernels; plain-torch Models (CUDA-tensor and CPU-tensor flavours) reach the analytic posterior; the prior-sentinel redraw
of GLMCMC.py:92-93, callback proposals, free theta_dim / batch_size.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

import oracle_lib
from helpers import SAMPLER_GOLDENS, bits, descriptors, load_golden, make_dist
from glabcmcmc_amd import _capi as A

PHILOX_GOLDENS = [n for n in SAMPLER_GOLDENS if "philox" in n]
ALGO = {"glmcmc": A.ALGO_GLMCMC, "globalmcmc": A.ALGO_GLOBALMCMC}
SENTINEL = np.float32(7 * math.log(1e-10))


# ----------------------------------------------------------------------------------------------- oracle, split-phase
def oracle_split_phase(L, algo, model, local, glob, theta0, y0, T, seed, gf, N, chain0=0, moments=None, redraw_prior=None):
    """T iterations of oracle_propose -> oracle_model_* callbacks -> oracle_select; returns (history [T][d][n], HostChains).
    redraw_prior: optional numpy function theta(n,d) -> prior (n,) replacing the Model's prior (sentinel tests)."""
    n, d = theta0.shape
    yd, nd = y0.shape[1], y0.shape[1]
    Np = N if algo == A.ALGO_GLMCMC else 1
    R = Np * n
    hc = oracle_lib.HostChains(theta0, y0, chain0=chain0)
    cs = hc.struct()
    buf = dict(theta_prop=np.zeros((R, d), np.float32), log_q=np.zeros(R, np.float32), sim_noise=np.zeros((R, nd), np.float32),
               log_u=np.zeros(n, np.float32), u_res=np.zeros(n, np.float64), is_global=np.zeros(n, np.int32),
               y_prop=np.zeros((R, yd), np.float32), prior_prop=np.zeros(R, np.float32), kern_prop=np.zeros(R, np.float32),
               prior_cur=np.zeros(n, np.float32), kern_cur=np.zeros(n, np.float32))
    io = A.StepIO(Np, d, yd, nd, *[buf[k].ctypes.data for k in ("theta_prop", "log_q", "sim_noise", "log_u", "u_res", "is_global",
                                                                 "y_prop", "prior_prop", "kern_prop", "prior_cur", "kern_cur")], None)

    def prior_of(theta_rows, out):
        if redraw_prior is not None:
            out[:] = redraw_prior(theta_rows)
        else:
            assert L.oracle_model_prior_log_prob(C.byref(model), theta_rows.ctypes.data, len(theta_rows), out.ctypes.data) == 0

    th0 = np.ascontiguousarray(theta0, np.float32)
    prior_of(th0, buf["prior_cur"])
    yy0 = np.ascontiguousarray(y0, np.float32)
    assert L.oracle_model_log_kernel(C.byref(model), yy0.ctypes.data, n, buf["kern_cur"].ctypes.data) == 0
    hist = np.zeros((T, d, n), np.float32)
    n_red = np.zeros(1, np.int32)
    total_redraws = 0
    for t in range(T):
        row = hist[t]
        run, keep = oracle_lib.make_run(seed=seed, step0=1 + t, n_steps=1, gf=gf, batch=Np, history=row, moments=moments)
        assert L.oracle_propose(algo, C.byref(local), C.byref(glob), C.byref(cs), C.byref(run), C.byref(io)) == 0
        prior_of(buf["theta_prop"], buf["prior_prop"])
        if algo == A.ALGO_GLMCMC:
            for rnd in range(1, 1000):
                n_red[0] = 0
                assert L.oracle_propose_redraw(C.byref(local), C.byref(cs), C.byref(run), C.byref(io), rnd, n_red.ctypes.data) == 0
                if n_red[0] == 0:
                    break
                total_redraws += int(n_red[0])
                prior_of(buf["theta_prop"][:n], buf["prior_prop"][:n])
        assert L.oracle_model_simulate(C.byref(model), buf["theta_prop"].ctypes.data, buf["sim_noise"].ctypes.data, R,
                                       buf["y_prop"].ctypes.data) == 0
        assert L.oracle_model_log_kernel(C.byref(model), buf["y_prop"].ctypes.data, R, buf["kern_prop"].ctypes.data) == 0
        assert L.oracle_select(algo, C.byref(glob), C.byref(cs), C.byref(run), C.byref(io)) == 0
    hc.redraws = total_redraws
    return hist, hc


@pytest.mark.parametrize("name", PHILOX_GOLDENS)
def test_oracle_split_phase_reproduces_reference_chains(oracle, name):
    g = load_golden(name)
    cfg = g["cfg"]
    model, local, glob = descriptors(cfg, g)
    C_ = min(g["theta0"].shape[0], 12)                       # the per-iteration Python loop is slow; a dozen chains pin it
    T = min(cfg["T"], 400)
    hist, _ = oracle_split_phase(oracle, ALGO[str(g["algo"])], model, local, glob, g["theta0"][:C_], g["y0"][:C_], T,
                                 cfg["seed"], cfg["gf"], cfg["N"], chain0=cfg.get("chain0", 0))
    got = np.concatenate([g["theta0"][None, :C_], hist.transpose(0, 2, 1)], axis=0)
    same = bits(got) == bits(g["chains"][:T + 1, :C_])
    assert same.all(), "first mismatch at (t, chain, dim) = %s" % (np.argwhere(~same)[0],)


@pytest.mark.parametrize("algo,N,d", [("glmcmc", 5, 2), ("glmcmc", 16, 3), ("glmcmc", 1, 1), ("globalmcmc", 1, 2), ("glmcmc", 7, 4)])
def test_oracle_split_phase_equals_one_piece(oracle, algo, N, d):
    """state, flags, log_w, move counts and streaming sums after T iterations: split-phase == one-piece restatement"""
    from test_stream_independence import abs_gauss_model, proposals
    L = oracle
    rng = np.random.default_rng(7 + d + N)
    n, T, seed, gf = 64, 120, 4242, 0.6
    model = abs_gauss_model(d, 0.4)
    lp, ip = proposals(d, "uniform" if N == 7 else "gauss")
    theta0 = rng.standard_normal((n, d)).astype(np.float32)
    y0 = (np.abs(theta0) + 0.2236 * rng.standard_normal((n, d))).astype(np.float32)
    mom_a, mom_b = oracle_lib.HostMoments(n, d), oracle_lib.HostMoments(n, d)
    hist, hc = oracle_split_phase(L, ALGO[algo], model, lp, ip, theta0, y0, T, seed, gf, N, chain0=10 ** 10, moments=mom_a)
    ref = oracle_lib.HostChains(theta0, y0, chain0=10 ** 10)
    cs = ref.struct()
    hh = np.zeros((T, d, n), np.float32)
    run, keep = oracle_lib.make_run(seed=seed, step0=1, n_steps=T, gf=gf, batch=N, history=hh, moments=mom_b)
    if algo == "glmcmc":
        assert L.oracle_init_weights(C.byref(model), C.byref(ip), C.byref(cs)) == 0
        assert L.oracle_glmcmc_steps(C.byref(model), C.byref(lp), C.byref(ip), C.byref(cs), C.byref(run)) == 0
    else:
        assert L.oracle_globalmcmc_steps(C.byref(model), C.byref(lp), C.byref(ip), C.byref(cs), C.byref(run)) == 0
    assert np.array_equal(bits(hist), bits(hh))
    assert np.array_equal(hc.n_moves, ref.n_moves) and hc.n_moves.sum() > 0
    assert np.array_equal(bits(hc.y), bits(ref.y))
    for a, b in ((mom_a.sum_theta, mom_b.sum_theta), (mom_a.sum_outer, mom_b.sum_outer), (mom_a.sum_jump, mom_b.sum_jump)):
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    if algo == "glmcmc":
        # log_weight_old is only defined where the reference would have computed it: chains whose `local` flag is clear
        clear = (ref.flags & A.FLAG_LOCAL) == 0
        assert np.array_equal(hc.flags & A.FLAG_LOCAL, ref.flags & A.FLAG_LOCAL)
        assert np.array_equal(bits(hc.log_w[clear]), bits(ref.log_w[clear]))


def box_prior(lo, hi):
    """a hand-written Model prior in the reference's convention: the sentinel 7*log(1e-10) outside the support"""
    def f(theta):
        inside = np.all((theta >= lo) & (theta <= hi), axis=1)
        return np.where(inside, np.float32(-1.25), SENTINEL).astype(np.float32)
    return f


def test_oracle_sentinel_redraw_keeps_local_proposals_inside_the_support(oracle):
    from test_stream_independence import abs_gauss_model, proposals
    rng = np.random.default_rng(3)
    n, T, d = 256, 60, 2
    model = abs_gauss_model(d, 0.5)
    lp, ip = proposals(d, "uniform")
    ip = make_dist(("uniform", [0.9, 0.9], [2.0, 2.0])).descriptor()          # global candidates always inside the box
    theta0 = rng.uniform(1.0, 1.9, (n, d)).astype(np.float32)
    y0 = (np.abs(theta0) + 0.2236 * rng.standard_normal((n, d))).astype(np.float32)
    hist, hc = oracle_split_phase(oracle, A.ALGO_GLMCMC, model, lp, ip, theta0, y0, T, 5, 0.3, 3,
                                  redraw_prior=box_prior(0.9, 2.0))
    assert hc.redraws > 50                                   # sd-0.4 increments near a box edge do land outside
    assert hist.min() >= 0.9 and hist.max() <= 2.0           # ... and none of them was ever simulated or accepted
    assert hc.n_moves.sum() > 100


# ----------------------------------------------------------------------------------------------------- GPU
class FixedDescriptor:
    """a proposal object whose glabc_dist is the fixture's (constants of the machine that ran the reference)"""

    def __init__(self, desc):
        self._d = desc

    def descriptor(self):
        return self._d


class ProtocolModel:
    """Nothing but the reference's Model protocol (examples/Mixture.py:5-53) plus the optional Philox-noise hook; the
    callbacks are the build's row-wise HIP kernels for a fixed glabc_model.  Deliberately NO descriptor()."""

    def __init__(self, desc, noise_hook=True):
        self._m = desc
        self.theta_dim, self.y_dim, self.epsilon = desc.theta_dim, desc.y_dim, desc.epsilon
        self.y_obs = torch.tensor([list(desc.y_obs)[:desc.y_dim]])
        if noise_hook:
            self.noise_dim = desc.y_dim
            self.simulate_from_noise = self._simulate_from_noise

    def _simulate_from_noise(self, theta, eps):
        from glabcmcmc_amd.distribution import _launch_simulate
        return _launch_simulate(self._m, theta, eps)

    def generate_samples(self, theta, num_samples=1):
        theta = theta.reshape(-1, self.theta_dim)
        return self._simulate_from_noise(theta, torch.randn(theta.shape[0], self.y_dim, device=theta.device))

    def prior_log_prob(self, samples):
        from glabcmcmc_amd.distribution import _launch_rowwise
        return _launch_rowwise("glabc_model_prior_log_prob", self._m, samples.reshape(-1, self.theta_dim), "prior")

    def discrepancy(self, y):
        from glabcmcmc_amd.distribution import _launch_rowwise
        return _launch_rowwise("glabc_model_discrepancy", self._m, y.reshape(-1, self.y_dim), "discrepancy")

    def calculate_log_kernel(self, y, epsilon=None):
        from glabcmcmc_amd.distribution import _launch_rowwise
        return _launch_rowwise("glabc_model_log_kernel", self._m, y.reshape(-1, self.y_dim), "log_kernel")


@pytest.mark.gpu
@pytest.mark.parametrize("name", PHILOX_GOLDENS)
def test_hip_protocol_only_model_reproduces_reference_chains(hip, name):
    """VERDICT r1 'done' criterion: a Model class with only the reference protocol runs run_glmcmc / run_global_mcmc and,
    its callbacks being the build's row-wise kernels, walks the reference's golden chains bit for bit."""
    import glabcmcmc_amd as g_
    g = load_golden(name)
    cfg = g["cfg"]
    model, local, glob = descriptors(cfg, g)
    pm = ProtocolModel(model)
    assert not hasattr(pm, "descriptor")
    T = min(cfg["T"], 500)
    runner = g_.MCMCRunner(pm)
    th0, y0 = torch.from_numpy(g["theta0"]), torch.from_numpy(g["y0"])
    kw = dict(seed=cfg["seed"], chain0=cfg.get("chain0", 0), output_file=None, verbose=False)
    if str(g["algo"]) == "glmcmc":
        out = runner.run_glmcmc(T + 1, th0, y0, cfg["gf"], FixedDescriptor(local), FixedDescriptor(glob), cfg["N"], **kw)
    else:
        out = runner.run_global_mcmc(T + 1, th0, y0, cfg["gf"], FixedDescriptor(local), FixedDescriptor(glob), **kw)
    got = out.numpy()
    if got.ndim == 2:
        got = got[:, None, :]
    same = bits(got) == bits(g["chains"][:T + 1])
    assert same.all(), "first mismatch at (t, chain, dim) = %s" % (np.argwhere(~same)[0],)


@pytest.mark.gpu
def test_hip_generic_equals_fused_at_full_size(hip):
    """65 536 chains: generic path (forced on the build's own Mixture_set) == fused kernel, histories and sums, bit for bit"""
    from glabcmcmc_amd import GLMCMC, distribution, engine
    from glabcmcmc_amd.examples.Mixture import Mixture_set
    n, T, seed = 65536, 12, 77
    m = Mixture_set(0.05)
    lp = distribution.DiagGaussian(2, torch.zeros(1, 2), torch.log(torch.tensor([0.35, 0.35])))
    ip = distribution.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.0, 0.0]))
    g = torch.Generator().manual_seed(5)
    th0 = torch.randn(n, 2, generator=g)
    y0 = th0.abs() + 0.2236 * torch.randn(n, 2, generator=g)
    dev = torch.device("cuda", 0)
    outs = []
    for path in ("fused", "generic"):
        mom = engine.Moments(n, 2, dev)
        st = {}
        h = GLMCMC(m, T + 1, th0, y0, lp, None, 0.9, ip, 5, seed=seed, stats=mom, return_device=True, verbose=False, path=path,
                   state_out=st)
        outs.append((h.cpu().numpy(), mom.sum_jump.cpu().numpy(), mom.sum_outer.cpu().numpy(), st["chains"].n_moves.cpu().numpy()))
    assert np.array_equal(bits(outs[0][0]), bits(outs[1][0]))
    for k in (1, 2):
        assert np.array_equal(outs[0][k].view(np.uint64), outs[1][k].view(np.uint64))
    assert np.array_equal(outs[0][3], outs[1][3]) and outs[0][3].sum() > 1000


class TorchMixture:
    """A user's Model in plain torch, written for whatever device its inputs are on (d dimensions of examples/Mixture.py)"""

    def __init__(self, d, epsilon):
        self.theta_dim = self.y_dim = d
        self.epsilon = epsilon
        self.y_obs = torch.full((1, d), 1.5)

    def generate_samples(self, theta, num_samples=1):
        return theta.abs() + math.sqrt(0.05) * torch.randn_like(theta)

    def prior_log_prob(self, samples):
        return -0.5 * self.theta_dim * math.log(2 * math.pi) - 0.5 * (samples ** 2).sum(1)

    def discrepancy(self, y):
        return ((y - self.y_obs.to(y.device)) ** 2).sum(1).sqrt()

    def calculate_log_kernel(self, y, epsilon=None):
        e = self.discrepancy(y) / self.epsilon
        return -0.5 * math.log(2 * math.pi) - math.log(self.epsilon) - 0.5 * e * e


class CpuOnlyMixture(TorchMixture):
    """The same Model written the way the reference's example is: CPU constants mixed into the arithmetic
    (examples/Mixture.py:9,36) -- it raises on CUDA tensors, so the loop must hand it CPU copies."""

    def discrepancy(self, y):
        return ((y - self.y_obs) ** 2).sum(1).sqrt()


@pytest.mark.gpu
@pytest.mark.parametrize("cls,d,N,n", [(TorchMixture, 2, 5, 65536), (TorchMixture, 6, 40, 4096), (CpuOnlyMixture, 2, 5, 8192)])
def test_hip_plain_torch_model_reaches_analytic_moments(hip, cls, d, N, n):
    """north_star: posterior moments within 1e-3 -- a user Model in plain torch at 65 536 chains; theta_dim 6 and
    batch_size 40 are beyond anything the fused kernels are compiled for"""
    from glabcmcmc_amd import GLMCMC, distribution, engine
    from test_stream_independence import analytic
    eps = 0.3
    m = cls(d, eps)
    lp = distribution.DiagGaussian(d, torch.zeros(d), torch.log(torch.full((d,), 0.3)))
    ip = distribution.DiagGaussian(d, torch.zeros(d), torch.log(torch.full((d,), 1.4)))
    g = torch.Generator().manual_seed(1)
    th0 = 1.3 * (torch.randint(0, 2, (n, d), generator=g).float() * 2 - 1)
    y0 = th0.abs() + 0.2236 * torch.randn(n, d, generator=g)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    burn, T = 150, 400
    st = {}
    GLMCMC(m, burn + 1, th0, y0, lp, None, 0.6, ip, N, seed=3, record_history=False, verbose=False, state_out=st)
    ch = st["chains"]
    assert st["callback_device"] == ("cpu" if cls is CpuOnlyMixture else "cuda")
    mom = engine.Moments(n, d, dev)
    GLMCMC(m, T + 1, ch.theta.t().cpu(), ch.y.t().cpu(), lp, None, 0.6, ip, N, seed=4, record_history=False, stats=mom,
           verbose=False)
    _, want_sq = analytic(eps)
    so = mom.sum_outer.cpu().numpy()
    k = 0
    for a in range(d):
        per_chain = so[k] / T
        se = per_chain.std(ddof=1) / np.sqrt(n)
        assert abs(per_chain.mean() - want_sq) < 5 * se + 1e-3 * want_sq, (a, per_chain.mean(), want_sq, se)
        k += d - a


class BoxedModel(ProtocolModel):
    """hand-written prior in the reference's sentinel convention (GLMCMC.py:92-93)"""

    def prior_log_prob(self, samples):
        s = samples.reshape(-1, self.theta_dim)
        inside = ((s >= 0.9) & (s <= 2.0)).all(1)
        return torch.where(inside, torch.full_like(s[:, 0], -1.25), torch.full_like(s[:, 0], float(SENTINEL)))


@pytest.mark.gpu
def test_hip_sentinel_redraw_equals_the_oracle(hip, oracle):
    from glabcmcmc_amd import GLMCMC
    from test_stream_independence import abs_gauss_model, proposals
    rng = np.random.default_rng(3)
    n, T, d = 256, 60, 2
    model = abs_gauss_model(d, 0.5)
    lp, _ = proposals(d, "uniform")
    ip = make_dist(("uniform", [0.9, 0.9], [2.0, 2.0])).descriptor()
    theta0 = rng.uniform(1.0, 1.9, (n, d)).astype(np.float32)
    y0 = (np.abs(theta0) + 0.2236 * rng.standard_normal((n, d))).astype(np.float32)
    want, hc = oracle_split_phase(oracle, A.ALGO_GLMCMC, model, lp, ip, theta0, y0, T, 5, 0.3, 3, redraw_prior=box_prior(0.9, 2.0))
    out = GLMCMC(BoxedModel(model), T + 1, torch.from_numpy(theta0), torch.from_numpy(y0), FixedDescriptor(lp), None, 0.3,
                 FixedDescriptor(ip), 3, seed=5, verbose=False)
    got = out.numpy()[1:].transpose(0, 2, 1)
    assert hc.redraws > 50
    assert np.array_equal(bits(got), bits(want))
    off = GLMCMC(BoxedModel(model), T + 1, torch.from_numpy(theta0), torch.from_numpy(y0), FixedDescriptor(lp), None, 0.3,
                 FixedDescriptor(ip), 3, seed=5, verbose=False, sentinel_redraw=False)
    assert not np.array_equal(bits(off.numpy()[1:].transpose(0, 2, 1)), bits(want))       # the loop is what made them equal


class MyProposal:
    """a user's proposal class: no descriptor, torch's generator"""

    def __init__(self, d, scale):
        self.d, self.scale = d, scale

    def forward(self, n):
        eps = torch.randn(n, self.d)
        return self.scale * eps, self.log_prob(self.scale * eps)

    def sample(self, n):
        return self.forward(n)[0]

    def log_prob(self, z):
        e = z / self.scale
        return -0.5 * self.d * math.log(2 * math.pi) - self.d * math.log(self.scale) - 0.5 * (e * e).sum(1)


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["glmcmc", "globalmcmc"])
def test_hip_callback_proposals(hip, algo):
    """proposal objects without a descriptor (a user's class; Gamma) are callbacks: forward / sample / log_prob"""
    from glabcmcmc_amd import GLMCMC, GlobalMCMC, distribution, engine
    from test_stream_independence import analytic
    d, n, eps, T = 2, 16384, 0.3, 500
    m = TorchMixture(d, eps)
    g = torch.Generator().manual_seed(1)
    th0 = 1.3 * (torch.randint(0, 2, (n, d), generator=g).float() * 2 - 1)
    y0 = th0.abs() + 0.2236 * torch.randn(n, d, generator=g)
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    mom = engine.Moments(n, d, dev)
    lp, ip = MyProposal(d, 0.3), MyProposal(d, 1.4)
    if algo == "glmcmc":
        GLMCMC(m, T + 1, th0, y0, lp, None, 0.6, ip, 4, seed=3, record_history=False, stats=mom, verbose=False)
    else:
        GlobalMCMC(m, T + 1, th0, y0, ip, None, 0.5, lp, seed=3, record_history=False, stats=mom, verbose=False)
    _, want_sq = analytic(eps)
    per_chain = mom.sum_outer[0].cpu().numpy() / T
    se = per_chain.std(ddof=1) / np.sqrt(n)
    assert abs(per_chain.mean() - want_sq) < 5 * se + 2e-3 * want_sq, (per_chain.mean(), want_sq, se)


def test_dispatch_without_a_device():
    """no GPU here: a descriptor-less Model no longer raises TypeError -- it reaches the generic path, which (like the
    fused one) refuses to run without a HIP device; path='fused' keeps the loud refusal"""
    import glabcmcmc_amd as g_
    from glabcmcmc_amd import generic
    if torch.cuda.is_available():
        pytest.skip("checks the CPU-only behaviour")
    m = TorchMixture(2, 0.3)
    dg = g_.DiagGaussian(2, torch.zeros(2), torch.zeros(2))
    assert not generic.fused_supported(m, (dg, dg), 5)
    assert generic.fused_supported(g_.examples.Mixture.Mixture_set(0.05), (dg, dg), 5) if hasattr(g_, "examples") else True
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g_.GLMCMC(m, 10, torch.zeros(2), torch.zeros(1, 2), dg, None, 0.5, dg, 5)
    with pytest.raises(TypeError, match="descriptor"):
        g_.GLMCMC(m, 10, torch.zeros(2), torch.zeros(1, 2), dg, None, 0.5, dg, 5, path="fused")
    ga = g_.Gamma(torch.tensor([2.0, 3.0]), torch.tensor([1.0, 2.0]))
    assert generic.dist_descriptor(ga, 2) is None and generic.dist_descriptor(dg, 2) is not None
    assert not generic.fused_supported(g_.examples.Mixture.Mixture_set(0.05), (dg, dg), 17)


@pytest.mark.gpu
@pytest.mark.parametrize("model_kind", ["torch", "protocol"])
def test_hip_generic_glmala_has_the_law_of_the_fused_kernel(hip, model_kind):
    """run_glmala with the Model as callbacks (iSIR through propose / select, MALA gradient through the Model's
    generate_samples / discrepancy) against the fused GLMALA kernel: same pooled second moments, move rate and mean squared
    jump within the Monte-Carlo error of 16 384 chains (the two use different random streams for the MALA move)."""
    import glabcmcmc_amd as g_
    from glabcmcmc_amd import distribution, engine
    from glabcmcmc_amd.examples.Mixture import Mixture_set
    n, T, eps, tau, num_grad, gf, N = 16384, 150, 0.3, 0.3, 20, 0.5, 4
    ip = distribution.DiagGaussian(2, torch.zeros(2), torch.zeros(2))
    g = torch.Generator().manual_seed(2)
    th0 = 1.3 * (torch.randint(0, 2, (n, 2), generator=g).float() * 2 - 1)
    y0 = th0.abs() + 0.2236 * torch.randn(n, 2, generator=g)
    dev = torch.device("cuda", 0)
    res = []
    for which in ("fused", "generic"):
        if which == "fused":
            m = Mixture_set(eps)
        else:
            m = TorchMixture(2, eps) if model_kind == "torch" else ProtocolModel(Mixture_set(eps).descriptor())
        mom = engine.Moments(n, 2, dev)
        st = {}
        torch.manual_seed(11)
        g_.MCMCRunner(m).run_glmala(T + 1, th0, y0, gf, ip, N, tau, num_grad, output_file=None, seed=9, record_history=False,
                                    stats=mom, verbose=False, state_out=st)
        sq = (mom.sum_outer[0] + mom.sum_outer[2]).cpu().numpy() / T
        jump = (mom.sum_jump[0] + mom.sum_jump[2]).cpu().numpy() / T
        moves = st["chains"].n_moves.cpu().numpy().astype(np.float64) / T
        res.append((sq, jump, moves))
        if which == "generic":
            # the reference's precision switch (GLMALA.py:43,197-198): Theta_old is a float64 tensor from a chain's first
            # accepted MALA move on; its float32 cast is the state Theta_Re records
            th64, wide = st["theta64"], st["th64"]
            assert th64.dtype == torch.float64 and 0.05 < float(wide.float().mean()) <= 1.0
            assert torch.equal(th64.float(), st["chains"].theta.t())
            narrow = th64[~wide]
            assert torch.equal(narrow, narrow.float().double())                 # float32-exact until the switch
            assert not torch.equal(th64[wide], th64[wide].float().double())     # genuinely double after it
    for a, b, name in zip(res[0], res[1], ("theta^2", "squared jump", "move rate")):
        se = math.sqrt(a.var(ddof=1) / n + b.var(ddof=1) / n)
        assert abs(a.mean() - b.mean()) < 5 * se, (name, a.mean(), b.mean(), se)
        assert a.mean() > 0


@pytest.mark.gpu
def test_hip_rowsum_of_any_length_matches_torch_sum(hip):
    """aten_rowsum_rt on the device against torch.sum's values (tests/golden/primitives.npz rowsum_*), n = 1 ... 4099:
    the four scalar lanes, the 8-wide vectors over four accumulators, ATen's cascade levels from 512 elements on"""
    from test_oracle_golden import ROWSUM_LENGTHS
    p = load_golden("primitives")
    for n in ROWSUM_LENGTHS:
        x = torch.from_numpy(np.ascontiguousarray(p["rowsum_%d_x" % n])).cuda()
        out = torch.empty(x.shape[0], dtype=torch.float32, device="cuda")
        assert hip.glabc_selftest_rowsum(x.data_ptr(), x.shape[0], n, out.data_ptr(), None) == 0
        assert np.array_equal(bits(out.cpu().numpy()), bits(p["rowsum_%d_sum" % n])), n


class WideSummaryModel:
    """theta_dim 12 > GLABC_MAX_DIM, y_dim 3 != theta_dim: sizes only the split-phase path with callback proposals handles"""

    def __init__(self):
        self.theta_dim, self.y_dim, self.epsilon = 12, 3, 0.5
        self.y_obs = torch.tensor([[1.0, 0.5, 2.0]])

    def generate_samples(self, theta, num_samples=1):
        t = theta.reshape(-1, 12)
        s = torch.stack([t[:, :4].abs().mean(1), t[:, 4:8].mean(1), (t[:, 8:] ** 2).mean(1)], dim=1)
        return s + 0.1 * torch.randn_like(s)

    def prior_log_prob(self, samples):
        return -0.5 * (samples.reshape(-1, 12) ** 2).sum(1)

    def discrepancy(self, y):
        return ((y.reshape(-1, 3) - self.y_obs.to(y.device)) ** 2).sum(1).sqrt()

    def calculate_log_kernel(self, y, epsilon=None):
        e = self.discrepancy(y) / self.epsilon
        return -0.5 * e * e


@pytest.mark.gpu
def test_hip_theta_dim_beyond_max_dim_and_other_y_dim(hip):
    from glabcmcmc_amd import GLMCMC, GlobalMCMC
    m = WideSummaryModel()
    n = 2048
    th0 = torch.randn(n, 12)
    y0 = m.generate_samples(th0)
    lp, ip = MyProposal(12, 0.15), MyProposal(12, 1.0)
    torch.manual_seed(0)
    out = GLMCMC(m, 80, th0, y0, lp, None, 0.5, ip, 7, seed=1, verbose=False, return_device=True)
    assert out.shape == (80, n, 12) and torch.isfinite(out).all()
    d_first = m.discrepancy(m.generate_samples(out[1])).mean().item()
    d_last = m.discrepancy(m.generate_samples(out[-1])).mean().item()
    assert d_last < 0.7 * d_first                                     # the chains move toward the observation
    out2 = GlobalMCMC(m, 40, th0, y0, ip, None, 0.3, lp, seed=2, verbose=False, return_device=True)
    assert out2.shape == (40, n, 12) and (out2[1:] != out2[:-1]).any()


@pytest.mark.gpu
def test_hip_split_phase_entry_points_validate(hip):
    """NULL pointers, bad sizes, zero chains: status codes, no launch"""
    from glabcmcmc_amd import engine
    dev = torch.device("cuda", 0)
    chains = engine.ChainBatch(torch.zeros(4, 2), torch.zeros(4, 2), dev)
    cs = chains.struct()
    run = A.Run()
    run.seed, run.step0, run.n_steps, run.global_frequency, run.batch_size = 1, 1, 1, 0.5, 3
    io = A.StepIO()
    io.n_prop, io.theta_dim, io.y_dim = 3, 2, 2
    g = make_dist(("gauss", [0, 0], [1, 1])).descriptor()
    assert hip.glabc_propose(A.ALGO_GLMCMC, C.byref(g), C.byref(g), C.byref(cs), C.byref(run), C.byref(io), None) == -1   # NULL buffers
    assert hip.glabc_select(A.ALGO_GLMCMC, C.byref(g), C.byref(cs), C.byref(run), C.byref(io), None) == -1
    assert hip.glabc_propose(7, C.byref(g), C.byref(g), C.byref(cs), C.byref(run), C.byref(io), None) == -3                 # unknown algo
    run.n_steps = 2
    assert hip.glabc_propose(A.ALGO_GLMCMC, C.byref(g), C.byref(g), C.byref(cs), C.byref(run), C.byref(io), None) == -4     # one iteration per call
    run.n_steps = 1
    io.n_prop = 2
    assert hip.glabc_propose(A.ALGO_GLOBALMCMC, C.byref(g), C.byref(g), C.byref(cs), C.byref(run), C.byref(io), None) == -4  # GlobalMCMC: one candidate
    empty = engine.ChainBatch(torch.zeros(4, 2), torch.zeros(4, 2), dev)
    es = empty.struct()
    es.n_chains = 0
    buf = torch.zeros(64, device=dev)
    io = A.StepIO(3, 2, 2, 0, *([buf.data_ptr()] * 2), None, *([buf.data_ptr()] * 8), None)
    assert hip.glabc_propose(A.ALGO_GLMCMC, C.byref(g), C.byref(g), C.byref(es), C.byref(run), C.byref(io), None) == 0       # zero chains: nothing to do
    assert hip.glabc_select(A.ALGO_GLMCMC, C.byref(g), C.byref(es), C.byref(run), C.byref(io), None) == 0
    assert hip.glabc_model_simulate(None, buf.data_ptr(), None, 4, 0, 0, buf.data_ptr(), None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["globalmcmc_philox_bench", "glmcmc_philox_n8"])
def test_hip_graph_replay_walks_the_reference_chains(hip, name):
    """One captured hipGraph of an iteration (the two HIP kernels with the iteration index in device memory + the Model's
    kernels), replayed: same chains as launching every iteration -- i.e. the reference's golden chains, bit for bit."""
    import glabcmcmc_amd as g_
    g = load_golden(name)
    cfg = g["cfg"]
    model, local, glob = descriptors(cfg, g)
    T = min(cfg["T"], 300)
    th0, y0 = torch.from_numpy(g["theta0"]), torch.from_numpy(g["y0"])
    st = {}
    kw = dict(seed=cfg["seed"], chain0=cfg.get("chain0", 0), verbose=False, graph=True, state_out=st)
    if str(g["algo"]) == "glmcmc":
        out = g_.GLMCMC(ProtocolModel(model), T + 1, th0, y0, FixedDescriptor(local), None, cfg["gf"], FixedDescriptor(glob),
                        cfg["N"], sentinel_redraw=False, **kw)
    else:
        out = g_.GlobalMCMC(ProtocolModel(model), T + 1, th0, y0, FixedDescriptor(glob), None, cfg["gf"], FixedDescriptor(local), **kw)
    assert st.get("graph") is True
    same = bits(out.numpy()) == bits(g["chains"][:T + 1])
    assert same.all(), "first mismatch at (t, chain, dim) = %s" % (np.argwhere(~same)[0],)
    assert (st["chains"].n_moves.cpu().numpy() > 0).any()


@pytest.mark.gpu
def test_hip_theta_dim_beyond_the_descriptor_limit(hip):
    """theta_dim 12 (> GLABC_MAX_DIM = 8: no glabc_dist can describe the proposals, so they are callbacks as well): GLMCMC
    and GlobalMCMC run through the split-phase path, chains move, and E theta_j^2 heads for the analytic value of every
    coordinate (loose: a 12-dimensional random walk mixes slowly; the point is that nothing about the path is dimension-bound)"""
    from glabcmcmc_amd import GLMCMC, GlobalMCMC, distribution, engine
    d, n, eps = 12, 4096, 1.0
    m = TorchMixture(d, eps)
    lp = distribution.DiagGaussian(d, torch.zeros(d), torch.log(torch.full((d,), 0.25)))
    ip = distribution.DiagGaussian(d, torch.zeros(d), torch.log(torch.full((d,), 1.0)))
    g = torch.Generator().manual_seed(1)
    th0 = 0.8 * (torch.randint(0, 2, (n, d), generator=g).float() * 2 - 1)
    y0 = th0.abs() + 0.2236 * torch.randn(n, d, generator=g)
    torch.manual_seed(0)
    st = {}
    mom = engine.Moments(n, d, torch.device("cuda", 0))
    out = GLMCMC(m, 301, th0, y0, lp, None, 0.3, ip, 16, seed=3, stats=mom, verbose=False, state_out=st, return_device=True)
    assert out.shape == (301, n, d) and torch.isfinite(out).all()
    assert st["chains"].n_moves.float().mean() > 20                         # the random walk accepts
    # exact E theta_j^2: |theta_j| is N(mu, s^2) truncated to (0, inf), mu = 1.5/(1+v), s^2 = v/(1+v), v = 0.05 + eps^2 (at
    # eps = 1 the truncation matters: 1.194, where the untruncated formula of `analytic` gives 1.048)
    from scipy.stats import norm
    v = 0.05 + eps ** 2
    mu, sd = 1.5 / (1 + v), math.sqrt(v / (1 + v))
    lam = norm.pdf(-mu / sd) / norm.cdf(mu / sd)
    mean1 = mu + sd * lam
    want_sq = sd ** 2 * (1 + (-mu / sd) * lam - lam ** 2) + mean1 ** 2
    late = (out[150:] ** 2).mean(dim=(0, 1)).cpu().numpy()
    assert np.all(np.abs(late - want_sq) < 0.03 * want_sq), (late, want_sq)
    out2 = GlobalMCMC(m, 51, th0, y0, ip, None, 0.2, lp, seed=4, verbose=False)
    assert out2.shape == (51, n, d) and torch.isfinite(out2).all()


@pytest.mark.gpu
def test_hip_speculative_graph_replay_with_the_sentinel_check(hip, oracle):
    """GLMCMC's default (sentinel check on) replays a captured iteration speculatively: (a) a prior that never returns the
    sentinel -- the whole run is replayed and equals the eager loop bit for bit over several 64-iteration segments; (b) a prior
    that does -- the first hit rolls the segment back and the iteration is captured again with the reference's redraw loop
    (GLMCMC.py:92-93) as a bounded number of rounds INSIDE the graph, doubled until a segment passes: the run stays a replayed
    graph and still gives the checker's chains; (c) with fewer rounds allowed in a graph than the prior needs, the rest of the run
    goes through the eager loop: the same chains again."""
    import glabcmcmc_amd as g_
    from test_stream_independence import abs_gauss_model, proposals
    g = load_golden("glmcmc_philox_bench")
    cfg = g["cfg"]
    model, local, glob = descriptors(cfg, g)
    n = g["theta0"].shape[0]
    th0, y0 = torch.from_numpy(g["theta0"]), torch.from_numpy(g["y0"])
    T = 200
    outs = {}
    for mode in ("auto", False):
        st = {}
        outs[mode] = g_.GLMCMC(ProtocolModel(model), T + 1, th0, y0, FixedDescriptor(local), None, cfg["gf"], FixedDescriptor(glob),
                               cfg["N"], seed=cfg["seed"], chain0=cfg.get("chain0", 0), verbose=False, graph=mode, state_out=st)
        assert bool(st.get("graph")) == (mode == "auto") and "graph_rolled_back_at" not in st
    assert np.array_equal(bits(outs["auto"].numpy()), bits(outs[False].numpy()))
    Tg = min(T, cfg["T"])
    assert np.array_equal(bits(outs["auto"].numpy()[:Tg + 1]), bits(g["chains"][:Tg + 1]))       # ... and the reference's
    # (b)
    rng = np.random.default_rng(3)
    n, T, d = 256, 150, 2
    bm = abs_gauss_model(d, 0.5)
    lp, _ = proposals(d, "uniform")
    ip = make_dist(("uniform", [0.9, 0.9], [2.0, 2.0])).descriptor()
    theta0 = rng.uniform(1.0, 1.9, (n, d)).astype(np.float32)
    y0b = (np.abs(theta0) + 0.2236 * rng.standard_normal((n, d))).astype(np.float32)
    want, hc = oracle_split_phase(oracle, A.ALGO_GLMCMC, bm, lp, ip, theta0, y0b, T, 5, 0.3, 3, redraw_prior=box_prior(0.9, 2.0))
    st = {}
    out = g_.GLMCMC(BoxedModel(bm), T + 1, torch.from_numpy(theta0), torch.from_numpy(y0b), FixedDescriptor(lp), None, 0.3,
                    FixedDescriptor(ip), 3, seed=5, verbose=False, state_out=st)
    assert st.get("graph_rolled_back_at") == 4 and st.get("graph") and 2 <= st["graph_redraw_rounds"] <= 32
    assert np.array_equal(bits(out.numpy()[1:].transpose(0, 2, 1)), bits(want)) and hc.redraws > 100
    eager = g_.GLMCMC(BoxedModel(bm), T + 1, torch.from_numpy(theta0), torch.from_numpy(y0b), FixedDescriptor(lp), None, 0.3,
                      FixedDescriptor(ip), 3, seed=5, verbose=False, graph=False)
    assert np.array_equal(bits(out.numpy()), bits(eager.numpy()))
    # (c)
    st = {}
    out = g_.GLMCMC(BoxedModel(bm), T + 1, torch.from_numpy(theta0), torch.from_numpy(y0b), FixedDescriptor(lp), None, 0.3,
                    FixedDescriptor(ip), 3, seed=5, verbose=False, state_out=st, max_graph_rounds=1)
    assert st.get("graph_rolled_back_at") == 4 and not st.get("graph")
    assert np.array_equal(bits(out.numpy()), bits(eager.numpy()))


@pytest.mark.gpu
@pytest.mark.parametrize("d,nd", [(12, 3), (9, 5), (11, 1), (8, 7), (3, 6)])
def test_hip_propose_noise_rows_beyond_eight_proposal_words(hip, oracle, d, nd):
    """glabc_propose with callback proposals (no descriptor: theta_dim may exceed GLABC_MAX_DIM) and a Model that takes its
    simulator noise from the stream (noise_dim > 0): Philox words 8 .. DP-1 of a candidate are proposal words, not simulator
    normals -- an earlier propose_kernel wrote them in front of the candidate's noise row.  Every row equals the CPU
    checker's, and the guard bands around the buffer stay untouched."""
    from glabcmcmc_amd import engine
    dev = torch.device("cuda", 0)
    n, N, seed = 77, 4, 991
    chains = engine.ChainBatch(torch.zeros(n, d), torch.zeros(n, 2), dev, chain0=5)
    cs = chains.struct()
    R, G = N * n, 64
    noise = torch.full((R * nd + 2 * G,), 7.25, dtype=torch.float32, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    buf = dict(theta_prop=torch.zeros(R, d, **f32), log_q=torch.zeros(R, **f32), log_u=torch.zeros(n, **f32),
               u_res=torch.zeros(n, dtype=torch.float64, device=dev), is_global=torch.zeros(n, dtype=torch.int32, device=dev))
    io = A.StepIO(N, d, 2, nd, buf["theta_prop"].data_ptr(), buf["log_q"].data_ptr(), noise.data_ptr() + 4 * G,
                  buf["log_u"].data_ptr(), buf["u_res"].data_ptr(), buf["is_global"].data_ptr(), None, None, None, None, None, None)
    run = A.Run()
    run.seed, run.step0, run.n_steps, run.global_frequency, run.batch_size = seed, 3, 1, 0.6, N
    assert hip.glabc_propose(A.ALGO_GLMCMC, None, None, C.byref(cs), C.byref(run), C.byref(io), None) == 0
    torch.cuda.synchronize()
    got = noise.cpu().numpy()
    assert (got[:G] == 7.25).all() and (got[-G:] == 7.25).all(), "glabc_propose wrote outside sim_noise"
    hc = oracle_lib.HostChains(np.zeros((n, d), np.float32), np.zeros((n, 2), np.float32), chain0=5)
    hcs = hc.struct()
    h = dict(theta_prop=np.zeros((R, d), np.float32), log_q=np.zeros(R, np.float32), noise=np.zeros((R, nd), np.float32),
             log_u=np.zeros(n, np.float32), u_res=np.zeros(n, np.float64), is_global=np.zeros(n, np.int32))
    hio = A.StepIO(N, d, 2, nd, h["theta_prop"].ctypes.data, h["log_q"].ctypes.data, h["noise"].ctypes.data, h["log_u"].ctypes.data,
                   h["u_res"].ctypes.data, h["is_global"].ctypes.data, None, None, None, None, None, None)
    hrun, keep = oracle_lib.make_run(seed=seed, step0=3, n_steps=1, gf=0.6, batch=N)
    assert oracle.oracle_propose(A.ALGO_GLMCMC, None, None, C.byref(hcs), C.byref(hrun), C.byref(hio)) == 0
    assert np.array_equal(bits(got[G:-G].reshape(R, nd)), bits(h["noise"]))
    assert np.array_equal(buf["is_global"].cpu().numpy(), h["is_global"])
    assert np.array_equal(buf["u_res"].cpu().numpy(), h["u_res"])


def _reference_isir_with_nan_rows(lw_old, theta_prop, prior_fn, kern_fn, log_q, u_res):
    """GLMCMC.py:66-84 literally, in torch on the CPU, for ONE chain whose proposal returned `theta_prop` (N, d) and `log_q`:
    rows with a NaN coordinate dropped (:67-70), weights of the survivors, torch.sum, weight_sampling's Python loop.  Returns
    (index into the COMPACTED list incl. the current state at 0, or None; number of survivors)."""
    has_nans = torch.isnan(theta_prop)
    keep = torch.all(~has_nans, dim=1)
    th, lq = theta_prop[keep].clone(), log_q[keep].clone()
    log_weight0 = prior_fn(th) + kern_fn(th) - lq
    log_weight = torch.cat((lw_old.view(-1), log_weight0))
    weight = torch.exp(log_weight)
    weight[torch.isnan(weight)] = 0.0
    weight = weight / torch.sum(weight)
    s, ind = 0, None
    for j, w in enumerate(weight.tolist()):
        s = s + w
        if u_res < s:
            ind = j
            break
    return ind, int(keep.sum())


def test_oracle_select_drops_nan_proposals_like_the_reference(oracle):
    """GLMCMC.py:67-70 (glabc_step_io.n_valid): a callback proposal hands back rows with NaN coordinates; the caller compacts
    the chain's candidates and glabc_select / oracle_select work on the survivors -- the index equals the one the reference's
    lines give on the same numbers (torch.sum over the shorter vector, the same running sum), chain by chain."""
    rng = np.random.default_rng(5)
    n, N, d = 300, 7, 2
    C_ = n
    theta_prop = rng.standard_normal((N, n, d)).astype(np.float32)
    bad = rng.random((N, n)) < 0.2
    bad[:, :5] = False
    bad[:, 5] = True                                                           # a chain with every proposal NaN: stays
    theta_prop[bad, rng.integers(0, d, bad.sum())] = np.nan
    prior = (-0.5 * (np.nan_to_num(theta_prop) ** 2).sum(-1)).astype(np.float32)
    kern = (-rng.random((N, n)) * 3).astype(np.float32)
    log_q = (-rng.random((N, n)) * 2).astype(np.float32)
    u_res = rng.random(n)
    lw_old = (-rng.random(n) * 2).astype(np.float32)
    # what the caller does (generic.run): survivors first, in their order; NaN rows behind them
    order = np.argsort(bad.astype(np.uint8), axis=0, kind="stable")
    take = lambda a: np.take_along_axis(a, order if a.ndim == 2 else order[..., None], axis=0)      # noqa: E731
    tp_c, pr_c, kn_c, lq_c = take(theta_prop), take(prior), take(kern), take(log_q)
    n_valid = (N - bad.sum(0)).astype(np.int32)
    hc = oracle_lib.HostChains(np.zeros((n, d), np.float32), np.zeros((n, 2), np.float32))
    hc.log_w[:] = lw_old
    hc.flags[:] = 0
    cs = hc.struct()
    flat = lambda a: np.ascontiguousarray(a.reshape(N * n, -1) if a.ndim == 3 else a.reshape(N * n))   # noqa: E731
    buf = dict(theta_prop=flat(tp_c), log_q=flat(lq_c), y=np.zeros((N * n, 2), np.float32), prior=flat(pr_c), kern=flat(kn_c),
               log_u=np.zeros(n, np.float32), u_res=u_res.copy(), is_global=np.ones(n, np.int32),
               prior_cur=np.zeros(n, np.float32), kern_cur=np.zeros(n, np.float32), q_cur=np.zeros(n, np.float32))
    io = A.StepIO(N, d, 2, 0, buf["theta_prop"].ctypes.data, buf["log_q"].ctypes.data, None, buf["log_u"].ctypes.data,
                  buf["u_res"].ctypes.data, buf["is_global"].ctypes.data, buf["y"].ctypes.data, buf["prior"].ctypes.data,
                  buf["kern"].ctypes.data, buf["prior_cur"].ctypes.data, buf["kern_cur"].ctypes.data, buf["q_cur"].ctypes.data,
                  n_valid.ctypes.data)
    run, keep = oracle_lib.make_run(seed=1, step0=1, n_steps=1, gf=1.0, batch=N)
    assert oracle.oracle_select(A.ALGO_GLMCMC, None, C.byref(cs), C.byref(run), C.byref(io)) == 0
    moved = (buf["is_global"] & 2) != 0
    checked = 0
    for c in range(n):
        th = torch.from_numpy(theta_prop[:, c, :])
        pr_rows, kn_rows = torch.from_numpy(prior[:, c]), torch.from_numpy(kern[:, c])
        keep_rows = ~torch.isnan(th).any(1)
        ind, nv = _reference_isir_with_nan_rows(torch.tensor([lw_old[c]]), th, lambda t: pr_rows[keep_rows], lambda t: kn_rows[keep_rows],
                                                torch.from_numpy(log_q[:, c]), float(u_res[c]))
        assert nv == n_valid[c]
        want_move = ind is not None and ind != 0
        assert bool(moved[c]) == want_move, c
        if want_move:
            survivors = np.flatnonzero(~bad[:, c])
            assert np.array_equal(bits(hc.theta[:, c]), bits(theta_prop[survivors[ind - 1], c]))
            checked += 1
    assert checked > 50 and not moved[5] and (n_valid < N).sum() > 100


@pytest.mark.gpu
def test_hip_select_drops_nan_proposals(hip, oracle):
    """glabc_select with glabc_step_io.n_valid == the CPU checker (which the test above holds to the reference's lines), and the
    package's GLMCMC with a callback proposal that returns NaN rows runs, never moves to one, and reaches the posterior"""
    import glabcmcmc_amd as g_
    from glabcmcmc_amd import distribution, engine
    rng = np.random.default_rng(6)
    n, N, d = 1000, 6, 2
    dev = torch.device("cuda", 0)
    theta_prop = rng.standard_normal((N * n, d)).astype(np.float32)
    n_valid = rng.integers(0, N + 1, n).astype(np.int32)
    for c in range(n):
        theta_prop[np.arange(n_valid[c], N) * n + c] = np.nan
    prior = (-0.5 * (np.nan_to_num(theta_prop) ** 2).sum(-1)).astype(np.float32)
    prior[np.isnan(theta_prop).any(1)] = np.nan
    kern, log_q = (-rng.random(N * n) * 3).astype(np.float32), (-rng.random(N * n) * 2).astype(np.float32)
    u_res, lw_old = rng.random(n), (-rng.random(n) * 2).astype(np.float32)
    outs = []
    for side in ("hip", "oracle"):
        if side == "hip":
            chains = engine.ChainBatch(torch.zeros(n, d), torch.zeros(n, 2), dev)
            chains.log_w.copy_(torch.from_numpy(lw_old))
            chains.flags.zero_()
            cs = chains.struct()
            t = {k: torch.from_numpy(v).to(dev) for k, v in dict(theta_prop=theta_prop, log_q=log_q, y=np.zeros((N * n, 2), np.float32),
                 prior=prior, kern=kern, log_u=np.zeros(n, np.float32), u_res=u_res, is_global=np.ones(n, np.int32),
                 prior_cur=np.zeros(n, np.float32), kern_cur=np.zeros(n, np.float32), q_cur=np.zeros(n, np.float32), n_valid=n_valid).items()}
            p = lambda k: t[k].data_ptr()                                      # noqa: E731
        else:
            hc = oracle_lib.HostChains(np.zeros((n, d), np.float32), np.zeros((n, 2), np.float32))
            hc.log_w[:] = lw_old
            hc.flags[:] = 0
            cs = hc.struct()
            t = dict(theta_prop=theta_prop.copy(), log_q=log_q.copy(), y=np.zeros((N * n, 2), np.float32), prior=prior.copy(), kern=kern.copy(),
                     log_u=np.zeros(n, np.float32), u_res=u_res.copy(), is_global=np.ones(n, np.int32), prior_cur=np.zeros(n, np.float32),
                     kern_cur=np.zeros(n, np.float32), q_cur=np.zeros(n, np.float32), n_valid=n_valid.copy())
            p = lambda k: t[k].ctypes.data                                     # noqa: E731
        io = A.StepIO(N, d, 2, 0, p("theta_prop"), p("log_q"), None, p("log_u"), p("u_res"), p("is_global"), p("y"), p("prior"), p("kern"),
                      p("prior_cur"), p("kern_cur"), p("q_cur"), p("n_valid"))
        if side == "hip":
            run = A.Run()
            run.seed, run.step0, run.n_steps, run.global_frequency, run.batch_size = 1, 1, 1, 1.0, N
            assert hip.glabc_select(A.ALGO_GLMCMC, None, C.byref(cs), C.byref(run), C.byref(io), None) == 0
            torch.cuda.synchronize()
            outs.append((chains.theta.cpu().numpy(), chains.log_w.cpu().numpy(), t["is_global"].cpu().numpy()))
        else:
            run, keep = oracle_lib.make_run(seed=1, step0=1, n_steps=1, gf=1.0, batch=N)
            assert oracle.oracle_select(A.ALGO_GLMCMC, None, C.byref(cs), C.byref(run), C.byref(io)) == 0
            outs.append((hc.theta, hc.log_w, t["is_global"]))
    for x, y in zip(*outs):
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
    assert not np.isnan(outs[0][0]).any() and ((outs[0][2] & 2) != 0).sum() > 100

    class NanProposal:
        """an importance proposal object (callbacks only) that returns a NaN coordinate in a fifth of its rows"""

        def __init__(self):
            self.base = distribution.DiagGaussian(2, torch.zeros(2), torch.zeros(2))

        def forward(self, num_samples=1):
            z, lq = self.base.forward(num_samples)
            z = z.clone()
            z[torch.rand(num_samples) < 0.2, 0] = float("nan")
            return z, lq

        def log_prob(self, z):
            return self.base.log_prob(z.cpu()).to(z.device)

    torch.manual_seed(3)
    m = TorchMixture(2, 0.3)
    lp = distribution.DiagGaussian(2, torch.zeros(2), torch.log(torch.tensor([0.3, 0.3])))
    nch, T = 4096, 200
    th0 = torch.full((nch, 2), 1.3)
    mom = engine.Moments(nch, 2, dev)
    out = g_.GLMCMC(m, T + 1, th0, th0 + 0.2 * torch.randn(nch, 2), lp, None, 0.8, NanProposal(), 6, seed=4, stats=mom, verbose=False,
                    return_device=True)
    assert torch.isfinite(out).all()
    sq = float(mom.second_moment().diagonal(dim1=1, dim2=2).mean())
    assert 1.7 < sq < 2.5, sq
