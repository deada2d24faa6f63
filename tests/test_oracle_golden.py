"""The CPU oracle (oracle/glabc_oracle.c) against the golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from glabcmcmc_amd import _capi as A
from helpers import GLMALA_GOLDENS_EXACT, GLMALA_GOLDENS_MKL, SAMPLER_GOLDENS, bits, descriptors, load_golden, mala_params

# The oracle's exp/log are those of include/glabc_numerics.h, ATen's are its own vectorised
# ones; densities agree to a few float32 ulp of the largest term, not bit for bit.
LOGP_RTOL = 4e-7


def _close(a, b, scale=None):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    assert np.array_equal(a[~fin], b[~fin])
    s = np.maximum(np.abs(b[fin]), 1.0) if scale is None else scale
    assert np.all(np.abs(a[fin] - b[fin]) <= LOGP_RTOL * s), np.max(np.abs(a[fin] - b[fin]) / s)


def _dist(kind, p0, p1, p2, c0):
    d = A.Dist()
    d.kind, d.dim = kind, len(p0)
    for i in range(len(p0)):
        d.p0[i], d.p1[i], d.p2[i] = float(p0[i]), float(p1[i]), float(p2[i])
    d.c0 = float(c0)
    return d


@pytest.fixture(scope="module")
def prim():
    return load_golden("primitives")


@pytest.mark.parametrize("tag", ["std", "lp", "gen", "d1", "d4", "d7", "d8"])
def test_diag_gaussian(oracle, prim, tag):
    loc, ls, sc = prim["dg_%s_loc" % tag], prim["dg_%s_log_scale" % tag], prim["dg_%s_scale" % tag]
    d = len(loc)
    dist = _dist(A.DIST_DIAG_GAUSS, loc, ls, sc, np.float32(-0.5 * d * np.log(2 * np.pi)))
    z = prim["dg_%s_z" % tag]
    out = np.empty(len(z), np.float32)
    assert oracle.oracle_dist_log_prob(C.byref(dist), z.ctypes.data, len(z), out.ctypes.data) == 0
    # no transcendental inside log_prob: every op is IEEE + - * /, so this is bit-exact
    assert np.array_equal(bits(out), bits(prim["dg_%s_log_prob" % tag]))
    eps = prim["dg_%s_eps" % tag]
    zz = np.empty_like(eps)
    lp = np.empty(len(eps), np.float32)
    assert oracle.oracle_dist_forward(C.byref(dist), eps.ctypes.data, len(eps), zz.ctypes.data, lp.ctypes.data) == 0
    assert np.array_equal(bits(zz), bits(prim["dg_%s_fwd_z" % tag]))
    assert np.array_equal(bits(lp), bits(prim["dg_%s_fwd_log_p" % tag]))


@pytest.mark.parametrize("tag", ["box", "inc", "d4"])
def test_uniform(oracle, prim, tag):
    low, high = prim["un_%s_low" % tag], prim["un_%s_high" % tag]
    dist = _dist(A.DIST_UNIFORM, low, high, high - low, prim["un_%s_log_prob_val" % tag])
    z = prim["un_%s_z" % tag]
    out = np.empty(len(z), np.float32)
    assert oracle.oracle_dist_log_prob(C.byref(dist), z.ctypes.data, len(z), out.ctypes.data) == 0
    assert np.array_equal(bits(out), bits(prim["un_%s_log_prob" % tag]))
    assert np.isfinite(out[0]) and np.isfinite(out[1])          # closed interval (distribution.py:83)
    u = prim["un_%s_u" % tag]
    zz = np.empty_like(u)
    lp = np.empty(len(u), np.float32)
    assert oracle.oracle_dist_forward(C.byref(dist), u.ctypes.data, len(u), zz.ctypes.data, lp.ctypes.data) == 0
    assert np.array_equal(bits(zz), bits(prim["un_%s_fwd_z" % tag]))
    assert np.array_equal(bits(lp), bits(prim["un_%s_fwd_log_p" % tag]))


def test_uniform_default_normaliser(prim):
    # distribution.py:55,71: default bounds have shape (1,), so log_prob_val is -log(4) for any dim
    assert prim["un_default_log_prob_val"] == np.float32(-np.log(4.0))


@pytest.mark.parametrize("eps", [0.05, 0.3])
def test_mixture_callbacks(oracle, prim, eps):
    from glabcmcmc_amd.examples.Mixture import Mixture_set
    tag = "mix_%g" % eps
    m = Mixture_set(eps).descriptor()
    # the descriptor's host-computed constants are the reference's own float32 values
    assert np.array_equal(bits(np.array(m.noise.p2[:2])), bits(prim[tag + "_noise_scale"]))
    assert np.array_equal(bits(np.array(m.noise.p1[:2])), bits(prim[tag + "_noise_log_scale"]))
    assert bits(np.float32(m.kern_scale)) == bits(prim[tag + "_kern_scale"])[0]
    assert bits(np.float32(m.kern_log_scale)) == bits(prim[tag + "_kern_log_scale"])[0]
    theta, noise, y = prim[tag + "_theta"], prim[tag + "_noise"], prim[tag + "_y"]
    n = len(theta)
    out = np.empty_like(y)
    assert oracle.oracle_model_simulate(C.byref(m), theta.ctypes.data, noise.ctypes.data, n, out.ctypes.data) == 0
    assert np.array_equal(bits(out), bits(y))
    o = np.empty(n, np.float32)
    assert oracle.oracle_model_prior_log_prob(C.byref(m), theta.ctypes.data, n, o.ctypes.data) == 0
    assert np.array_equal(bits(o), bits(prim[tag + "_prior"]))
    # discrepancy ends in a sqrt (IEEE here, ATen's AVX path is 1 ulp off for ~0.7 % of inputs)
    assert oracle.oracle_model_discrepancy(C.byref(m), y.ctypes.data, n, o.ctypes.data) == 0
    dis = prim[tag + "_dis"]
    assert np.all(np.abs(o.astype(np.float64) - dis) <= np.spacing(dis))
    assert np.mean(bits(o) == bits(dis)) > 0.98
    assert oracle.oracle_model_log_kernel(C.byref(m), y.ctypes.data, n, o.ctypes.data) == 0
    _close(o, prim[tag + "_logk"])


ROWSUM_LENGTHS = list(range(1, 18)) + [18, 31, 32, 33, 63, 64, 65, 100, 128, 255, 256, 257, 511, 512, 513, 1000, 2049, 4099]


def test_rowsum_order(oracle, prim):
    for n in ROWSUM_LENGTHS:
        x = prim["rowsum_%d_x" % n]
        got = np.array([oracle.oracle_aten_rowsum_f32(r.ctypes.data, n) for r in np.ascontiguousarray(x)], np.float32)
        assert np.array_equal(bits(got), bits(prim["rowsum_%d_sum" % n])), n


def test_esjd(oracle, prim):
    assert prim["esjd_known"] == np.float32(0.75)
    i = 0
    while "esjd_chain_%d" % i in prim:
        x = prim["esjd_chain_%d" % i]
        T, d = x.shape
        hist = np.ascontiguousarray(x.reshape(T, d, 1))
        out = np.empty(1, np.float32)
        assert oracle.oracle_esjd(hist.ctypes.data, T, d, 1, 1, out.ctypes.data) == 0
        ref = prim["esjd_value_%d" % i]
        assert abs(out[0] - ref) <= 2e-5 * abs(ref), (i, out[0], ref)
        i += 1


def run_oracle(oracle, g, philox_chain0=None):
    """Run the oracle on a sampler golden's configuration; returns the (T+1, C, d) chains."""
    cfg = g["cfg"]
    model, local, glob = descriptors(cfg, g)
    C_, T, d = g["theta0"].shape[0], cfg["T"], g["theta0"].shape[1]
    ch = oracle_lib.HostChains(g["theta0"], g["y0"], chain0=cfg.get("chain0", 0))
    hist = np.zeros((T, d, C_), np.float32)
    tape = None
    if str(g["mode"]) == "tape":
        tape = (np.ascontiguousarray(g["tape_u"]), np.ascontiguousarray(g["tape_r"]),
                np.ascontiguousarray(g["tape_z"]), g["tape_z"].shape[2])
    run, keep = oracle_lib.make_run(seed=cfg["seed"], step0=1, n_steps=T, gf=cfg["gf"], batch=cfg["N"],
                                    history=hist, tape=tape)
    cs = ch.struct()
    if str(g["algo"]) == "glmcmc":
        assert oracle.oracle_init_weights(C.byref(model), C.byref(glob), C.byref(cs)) == 0
        rc = oracle.oracle_glmcmc_steps(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run))
    else:
        rc = oracle.oracle_globalmcmc_steps(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run))
    assert rc == 0
    chains = np.concatenate([g["theta0"][None], hist.transpose(0, 2, 1)], axis=0)
    return chains, ch


@pytest.mark.parametrize("name", SAMPLER_GOLDENS)
def test_sampler_chains_bit_exact(oracle, name):
    """The unmodified reference loop and the oracle, fed the same random numbers, visit the
    same float32 states at every iteration of every chain."""
    g = load_golden(name)
    chains, ch = run_oracle(oracle, g)
    ref = g["chains"]
    assert chains.shape == ref.shape
    same = bits(chains) == bits(ref)
    assert same.all(), "first mismatch at (t, chain, dim) = %s" % (np.argwhere(~same)[0],)
    moves = (np.diff(ref, axis=0) != 0).any(-1).sum(0)
    assert np.array_equal(ch.n_moves, moves.astype(np.uint32))
    assert moves.sum() > 0


def run_oracle_glmala(oracle, g):
    cfg = g["cfg"]
    model, _, glob = descriptors(cfg, g)
    mala = mala_params(cfg)
    C_, T, d = g["theta0"].shape[0], cfg["T"], 2
    ch = oracle_lib.HostChains(g["theta0"], g["y0"], chain0=cfg.get("chain0", 0)).add_mala_state()
    hist = np.zeros((T, d, C_), np.float32)
    run, keep = oracle_lib.make_run(seed=cfg["seed"], step0=1, n_steps=T, gf=cfg["gf"], batch=cfg["N"], history=hist)
    cs = ch.struct()
    assert oracle.oracle_glmala_init(C.byref(model), C.byref(cs)) == 0
    assert oracle.oracle_glmala_steps(C.byref(model), C.byref(glob), C.byref(mala), C.byref(cs), C.byref(run)) == 0
    chains = np.concatenate([g["theta0"][None], hist.transpose(0, 2, 1)], axis=0)
    return chains, ch


@pytest.mark.parametrize("name", GLMALA_GOLDENS_EXACT)
def test_glmala_chains_bit_exact(oracle, name):
    """GLMALA.py:150-200 (iSIR + MALA with the common-random-number finite-difference gradient, the
    float32 -> float64 switch of the state, the stale iSIR weight) replayed on the same random
    numbers, with the reference's torch.sqrt correctly rounded: every recorded float32 Theta_Re row
    of every chain equals the reference's."""
    g = load_golden(name)
    chains, ch = run_oracle_glmala(oracle, g)
    ref = g["chains"]
    same = bits(chains) == bits(ref)
    assert same.all(), "first mismatch at (t, chain, dim) = %s of %d" % (np.argwhere(~same)[0], (~same).sum())
    moves = (np.diff(ref, axis=0) != 0).any(-1).sum(0)
    assert np.array_equal(ch.n_moves, moves.astype(np.uint32))
    if g["cfg"]["gf"] < 1.0:
        assert (ch.flags & A.FLAG_TH64).any()          # some chain really switched to float64


def divergence_profile(chains, ref):
    """per chain: index of the first differing row (T+1 if none) and the size of that first difference in ulp"""
    same = (bits(chains) == bits(ref)).all(-1)
    Tn, Cn = same.shape
    first = np.where(same.all(0), Tn, np.argmin(same, axis=0))
    ulp = np.zeros(Cn)
    for c in range(Cn):
        if first[c] < Tn:
            a, b = chains[first[c], c].astype(np.float64), ref[first[c], c].astype(np.float64)
            ulp[c] = (np.abs(a - b) / np.spacing(np.abs(ref[first[c], c])).astype(np.float64)).max()
    return first, ulp


@pytest.mark.parametrize("name", GLMALA_GOLDENS_MKL)
def test_glmala_vs_unpatched_reference(oracle, name):
    """Against the reference exactly as it runs on this torch build (MKL VML sqrt, not correctly
    rounded): a chain follows the reference bit for bit until its first sqrt-ulp event; that first
    difference is a few float32 ulp, i.e. ~1e-6 relative (afterwards the float32 finite-difference prior gradient,
    GLMALA.py:84-85, amplifies it and the trajectories separate), and the number of accepted moves
    stays within 1 %."""
    g = load_golden(name)
    chains, ch = run_oracle_glmala(oracle, g)
    ref = g["chains"]
    first, ulp = divergence_profile(chains, ref)
    Tn = ref.shape[0]
    assert (first >= 1).all()
    assert ulp.max() <= 16.0 and np.median(ulp[ulp > 0]) <= 2.0, ulp
    if name == "glmala_philox_bench":                   # BASELINE config 3: few accepted MALA moves per chain
        assert (first == Tn).mean() >= 0.75
    moves_ref = (np.diff(ref, axis=0) != 0).any(-1).sum()
    assert abs(int(ch.n_moves.sum()) - int(moves_ref)) <= 0.01 * moves_ref + 1


def test_glmala_gradient_matches_reference(oracle):
    """numberical_gradient_logABC (GLMALA.py:46-95) on golden inputs: reference values were produced with
    the correctly rounded sqrt; the float64 result agrees to ~1e-13 (torch.mean / torch.var / torch.log use
    their own summation cascades and MKL's log)."""
    g = load_golden("glmala_gradient")
    cfg = g["cfg"]
    model, _, _ = descriptors(cfg, g)
    mala = mala_params(cfg)
    out = np.zeros(2)
    worst = 0.0
    for i in range(g["theta"].shape[0]):
        th = np.ascontiguousarray(g["theta"][i])
        assert oracle.oracle_numerical_gradient(C.byref(model), C.byref(mala), th.ctypes.data, cfg["seed"], int(g["chain"][i]),
                                                int(g["step"][i]), 1, out.ctypes.data) == 0
        worst = max(worst, np.max(np.abs(out - g["grad"][i]) / np.maximum(1.0, np.abs(g["grad"][i]))))
    assert worst < 1e-11, worst


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_gamma_log_prob(oracle, prim, tag):
    """Gamma.log_prob (distribution.py:123-137, SciPy float64): same -inf pattern (negative z, pdf underflow, z = 0),
    1e-12 where the pdf is a normal double; in the subnormal tail log(pdf) itself only has a few significant bits."""
    import torch
    from glabcmcmc_amd import distribution
    g = distribution.Gamma(torch.from_numpy(prim["gm_%s_shape" % tag]), torch.from_numpy(prim["gm_%s_rate" % tag]))
    d = g.gamma_descriptor()
    z, ref = prim["gm_%s_z" % tag], prim["gm_%s_log_prob" % tag]
    out = np.empty(len(z))
    assert oracle.oracle_gamma_log_prob(C.byref(d), z.ctypes.data, len(z), out.ctypes.data) == 0
    assert np.array_equal(np.isneginf(out), np.isneginf(ref)) and np.array_equal(np.isposinf(out), np.isposinf(ref))
    fin = np.isfinite(ref)
    normal = fin & (ref > -650)
    assert normal.sum() > 200
    assert np.max(np.abs(out[normal] - ref[normal]) / np.maximum(1, np.abs(ref[normal]))) < 1e-12
    assert np.max(np.abs(out[fin] - ref[fin]) / np.maximum(1, np.abs(ref[fin]))) < 1e-6
    # host mirror on CPU tensors = the reference's own SciPy path
    assert np.array_equal(g.log_prob(torch.from_numpy(z)).numpy(), ref)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_gamma_forward(oracle, prim, tag):
    """Gamma.forward (distribution.py:106-121): the specified Philox / Marsaglia-Tsang draw is stable (same variates as when
    the fixture was made) and the log_p it returns equals the REFERENCE's forward() on those variates (scipy's pdf, float64)."""
    import torch
    from glabcmcmc_amd import distribution
    g = distribution.Gamma(torch.from_numpy(prim["gf_%s_shape" % tag]), torch.from_numpy(prim["gf_%s_rate" % tag]))
    d = g.gamma_descriptor()
    seed, row0 = (int(v) for v in prim["gf_%s_seed_row0" % tag])
    z_ref, lp_ref = prim["gf_%s_z" % tag], prim["gf_%s_log_p" % tag]
    z, lp = np.empty_like(z_ref), np.empty_like(lp_ref)
    assert oracle.oracle_gamma_forward(C.byref(d), len(z), seed, row0, z.ctypes.data, lp.ctypes.data) == 0
    assert np.array_equal(z.view(np.uint64), z_ref.view(np.uint64))
    assert (z > 0).all() and np.isfinite(lp_ref).all()
    assert np.max(np.abs(lp - lp_ref) / np.maximum(1, np.abs(lp_ref))) < 1e-12


# ------------------------------------------------------------------ the ATen-level restatement (oracle/aten_loop.py)
@pytest.mark.parametrize("name", ["glmcmc_tape_small", "globalmcmc_tape_small"])
def test_aten_restatement_replays_the_reference_tapes(name):
    """oracle/aten_loop.py (one chain, the reference's ATen operation sequence: what bench.py times as the reference-like
    CPU baseline) fed with the stored tapes gives the chains the reference's own loops produced, bit for bit.
    (The host constants log(sqrt(0.05)), exp(log eps) ... are recomputed by torch as in the reference: the comparison is
    exact on the machine type that generated the fixture, i.e. in the build container, where the CPU suite runs.)"""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import aten_loop
    g = load_golden(name)
    cfg = g["cfg"]
    consts_here = torch.exp(torch.log(torch.tensor([0.05, 0.05]).sqrt())).numpy()
    if not np.array_equal(bits(consts_here), bits(g["c_noise_scale"])):
        pytest.skip("this host's torch rounds exp(log(sqrt(0.05))) differently from the machine that wrote the fixture")
    model = aten_loop.Mixture(cfg["epsilon"])
    local, glob = aten_loop.make_distribution(cfg["local"]), aten_loop.make_distribution(cfg["global"])
    T = cfg["T"]
    for c in range(g["theta0"].shape[0]):
        draws = aten_loop.TapeDraws(g["tape_u"][c], g["tape_r"][c], g["tape_z"][c], 2)
        th0, y0 = torch.from_numpy(g["theta0"][c].copy()), torch.from_numpy(g["y0"][c].copy())
        if str(g["algo"]) == "glmcmc":
            out = aten_loop.glmcmc(model, T, th0, y0, local, glob, cfg["gf"], cfg["N"], draws)
        else:
            out = aten_loop.globalmcmc(model, T, th0, y0, glob, local, cfg["gf"], draws)
        assert np.array_equal(bits(out.numpy()), bits(g["chains"][:, c, :])), "chain %d" % c


# ---------------------------------------------------------------------------------- Gamma inside the samplers
def _gamma_cases():
    import ast
    g = load_golden("gamma_candidates")
    return g, ast.literal_eval(str(g["cases"]))


def test_gamma_candidates_match_the_reference_forward(oracle):
    """GLABC_DIST_GAMMA as the importance proposal (include/glabc.h): the checker's double variates are the ones the fixture
    fed to the reference's Gamma.forward (distribution.py:106-121), and its float64 log q' is the reference's log_prob of
    them to 1e-12 (scipy's log / exp / gammaln against the specified ones); through oracle_propose the candidates are those
    variates rounded to float32 once."""
    import torch
    from glabcmcmc_amd import _capi as A, distribution
    g, cases = _gamma_cases()
    oracle.oracle_gamma_candidates.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    for i, seed, chain, step, N in cases:
        shape, rate = g["gc_%d_shape" % i], g["gc_%d_rate" % i]
        d = distribution.Gamma(torch.from_numpy(shape), torch.from_numpy(rate)).descriptor()
        assert d.kind == A.DIST_GAMMA and d.dim == len(shape)
        k = d.dim
        z, lp = np.empty((N, k)), np.empty(N)
        assert oracle.oracle_gamma_candidates(C.byref(d), seed, chain, step, N, z.ctypes.data, lp.ctypes.data) == 0
        assert np.array_equal(z.view(np.uint64), g["gc_%d_z" % i].view(np.uint64)), i
        ref = g["gc_%d_log_p" % i]
        assert np.all(np.abs(lp - ref) <= 1e-12 * np.maximum(1.0, np.abs(ref))), (i, np.abs(lp - ref).max())
        assert (z > 0).all() and np.isfinite(lp).all()
        mean = z.mean(0)
        want = shape.astype(np.float64) / rate
        assert np.all(np.abs(mean - want) < 0.6 * want), (mean, want)               # the right distribution (few draws)
        # the split-phase draw: theta' = (float) z, log q' = (float) log_p
        hc = oracle_lib.HostChains(np.zeros((1, k), np.float32), np.zeros((1, k), np.float32), chain0=chain)
        cs = hc.struct()
        h = dict(theta_prop=np.zeros((N, k), np.float32), log_q=np.zeros(N, np.float32), log_u=np.zeros(1, np.float32),
                 u_res=np.zeros(1, np.float64), is_global=np.zeros(1, np.int32))
        io = A.StepIO(N, k, k, 0, h["theta_prop"].ctypes.data, h["log_q"].ctypes.data, None, h["log_u"].ctypes.data,
                      h["u_res"].ctypes.data, h["is_global"].ctypes.data, None, None, None, None, None, None)
        run, keep = oracle_lib.make_run(seed=seed, step0=step, n_steps=1, gf=1.0, batch=N)
        assert oracle.oracle_propose(A.ALGO_GLMCMC, None, C.byref(d), C.byref(cs), C.byref(run), C.byref(io)) == 0
        assert np.array_equal(bits(h["theta_prop"]), bits(z.astype(np.float32)))
        assert np.array_equal(bits(h["log_q"]), bits(lp.astype(np.float32)))


def test_gamma_log_prob_at_float32_points_matches_the_reference(oracle):
    """Gamma.log_prob (distribution.py:123-137) as a prior is evaluated: float64 at the float32 point, -inf outside the support
    and where the pdf underflows (B7), rounded to float32 once -- within one float32 ulp of the reference's float64 sum."""
    import torch
    from glabcmcmc_amd import distribution
    g, cases = _gamma_cases()
    for i, *_ in cases:
        d = distribution.Gamma(torch.from_numpy(g["gc_%d_shape" % i]), torch.from_numpy(g["gc_%d_rate" % i])).descriptor()
        pts, ref = g["gc_%d_pts" % i], g["gc_%d_pts_log_prob" % i]
        out = np.empty(len(pts), np.float32)
        assert oracle.oracle_dist_log_prob(C.byref(d), pts.ctypes.data, len(pts), out.ctypes.data) == 0
        inf = np.isinf(ref)
        assert np.array_equal(np.isinf(out), inf) and np.array_equal(out[inf], ref[inf].astype(np.float32)) and inf.sum() >= 2, i
        # (shape < 1 at z = 0: the pdf itself is +inf, distribution.py:133-136 returns log(inf) = +inf)
        want = ref[~inf].astype(np.float32)
        ulp = np.spacing(np.abs(want))
        assert np.all(np.abs(out[~inf].astype(np.float64) - want.astype(np.float64)) <= ulp), i
