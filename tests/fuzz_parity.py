"""Randomised differential test, gfx950 kernels vs the CPU checker (not collected by pytest; run on the GPU box):

    python tests/fuzz_parity.py [seconds] [seed] [rtc | nfgrad | mala]

Random theta_dim, batch size (every fourth GLMCMC case beyond 16: the wide kernel, up to 1200 proposals), epsilon (1e-4 .. 10), global_frequency, Gaussian / Uniform proposals with random
parameters, y_obs (also near zero), lanes per chain, iterations per launch, chain id offsets -- GLMCMC and GlobalMCMC
histories, final states and streaming sums must agree with the oracle bit for bit; every fourth case is GLMALA
(random tau, num_grad, float64 state, gradients), every eighth the g-and-k Model, every sixteenth a random user simulator
compiled into the kernel at run time (all of them with a third argument `rtc`), every sixty-fourth the flow's hand-written
gradient against the checker's (tolerance; `nfgrad` for only those).
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "gl-abc-mcmc_amd")]
import oracle_lib                                   # noqa: E402
from glabcmcmc_amd import _capi, engine             # noqa: E402
from glabcmcmc_amd.examples.Mixture import Mixture_set   # noqa: E402
from helpers import bits, make_dist                 # noqa: E402


def random_dist(rng, d, local):
    if rng.random() < 0.65:
        loc = np.zeros(d) if (local or rng.random() < 0.5) else rng.normal(0, 0.5, d)
        scale = np.ones(d) if (not local and rng.random() < 0.5) else np.exp(rng.normal(-0.5 if local else 0.2, 0.6, d))
        return ("gauss", [float(v) for v in loc], [float(v) for v in scale])
    w = np.exp(rng.normal(-1.0 if local else 1.0, 0.5, d))
    c = np.zeros(d) if local else rng.normal(0, 0.3, d)
    return ("uniform", [float(v) for v in c - w], [float(v) for v in c + w])


def one_case(rng, oracle, k):
    d = int(rng.integers(1, 9))                                          # 5 .. 8: the default-schedule objects
    algo = "glmcmc" if rng.random() < 0.7 else "globalmcmc"
    N = int(rng.integers(1, 17)) if algo == "glmcmc" else 1
    wide = d <= 4 and algo == "glmcmc" and rng.random() < 0.25           # glabc_wide.hip: lane groups share a chain's proposals
    if wide:
        N = int(rng.choice([17, 31, 32, 33, 64, 65, int(rng.integers(17, 300)), int(rng.integers(300, 1200))]))
    eps = float(np.exp(rng.uniform(np.log(1e-4), np.log(10))))
    gf = float(rng.choice([0.0, 1.0, rng.random()]))
    lspec, gspec = random_dist(rng, d, True), random_dist(rng, d, False)
    model = Mixture_set(eps).descriptor()             # kernel constants for this epsilon; the rest is filled per dimension
    model.theta_dim = model.y_dim = d
    pspec = ("gauss", [0.0] * d, [1.0] * d) if rng.random() < 0.6 else \
        ("gauss", [float(v) for v in rng.normal(0, 0.3, d)], [float(v) for v in np.exp(rng.normal(0.2, 0.4, d))])
    if rng.random() < 0.15:
        pspec = ("uniform", [-4.0] * d, [4.0] * d)
    if d <= 4 and rng.random() < 0.12:                # GLABC_DIST_GAMMA as importance / global proposal and / or prior (VAR_GAMMA)
        gam = lambda: ("gamma", [float(v) for v in np.exp(rng.normal(0.7, 0.8, d))], [float(v) for v in np.exp(rng.normal(0.5, 0.5, d))])  # noqa: E731
        which = int(rng.integers(0, 3))
        if which != 1:
            gspec = gam()
        if which != 0:
            pspec = gam()
    model.prior = make_dist(pspec).descriptor()
    model.noise = make_dist(("gauss", [0.0] * d, [float(v) for v in np.exp(rng.normal(-1.5, 0.3, d))])).descriptor()
    y_obs = [float(v) for v in rng.choice([0.0, 1e-3, 1.5, float(rng.normal(1, 1))], d)]
    for j in range(d):
        model.y_obs[j] = y_obs[j]
    local, glob = make_dist(lspec).descriptor(), make_dist(gspec).descriptor()
    n, T = int(rng.integers(1, 700)), int(rng.integers(1, 60))
    lanes = int(rng.choice([0, 1, 2, 4])) if algo == "glmcmc" else 0
    if wide:
        n, T = int(rng.integers(1, 200 if N < 300 else 40)), int(rng.integers(1, 25))
        lanes = int(rng.choice([0, 8, 16, 32, 64]))
    spl = int(rng.integers(1, T + 1))
    seed, chain0 = int(rng.integers(0, 2 ** 63)), int(rng.integers(0, 2 ** 40))
    theta0 = rng.normal(0, 1, (n, d)).astype(np.float32)
    y0 = (np.abs(theta0) + 0.2236068 * rng.normal(0, 1, (n, d))).astype(np.float32)
    dev = torch.device("cuda", 0)
    chains = engine.ChainBatch(torch.from_numpy(theta0), torch.from_numpy(y0), dev, chain0=chain0)
    if algo == "glmcmc":
        engine.init_weights(model, glob, chains)
    hist = torch.empty(T, d, n, device=dev)
    mom = engine.Moments(n, d, dev)
    entry = "glabc_glmcmc_steps" if algo == "glmcmc" else "glabc_globalmcmc_steps"
    # launch geometry: one in four GLMCMC cases forces the team kernels (glabc_team.h: 2 .. 4 wavefronts per 64 chains; the
    # library falls back to fewer wavefronts, or to sampler_kernel, where the configuration has no such team)
    team = 0
    if algo == "glmcmc" and not wide and rng.random() < 0.25:
        team = int(rng.integers(2, 5))
        os.environ["GLABC_TEAM_WAVES"] = str(team)
    engine.run_steps(entry, model, local, glob, chains, T, 1, seed, gf, N, history=hist, moments=mom, steps_per_launch=spl,
                     lanes_per_chain=0 if team else lanes, debug_flags=_capi.DEBUG_TEAM if team else 0)
    os.environ.pop("GLABC_TEAM_WAVES", None)
    torch.cuda.synchronize()
    hc = oracle_lib.HostChains(theta0, y0, chain0=chain0)
    hh = np.zeros((T, d, n), np.float32)
    hm = oracle_lib.HostMoments(n, d)
    run, keep = oracle_lib.make_run(seed=seed, step0=1, n_steps=T, gf=gf, batch=N, history=hh, moments=hm)
    cs = hc.struct()
    if algo == "glmcmc":
        assert oracle.oracle_init_weights(C.byref(model), C.byref(glob), C.byref(cs)) == 0
        rc = oracle.oracle_glmcmc_steps(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run))
    else:
        rc = oracle.oracle_globalmcmc_steps(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run))
    assert rc == 0
    desc = dict(case=k, algo=algo, d=d, N=N, eps=eps, gf=gf, local=lspec, glob=gspec, prior=pspec, y_obs=y_obs,
                n=n, T=T, lanes=lanes, spl=spl, team=team)
    ok = np.array_equal(bits(hist.cpu().numpy()), bits(hh)) and np.array_equal(bits(chains.theta.cpu().numpy()), bits(hc.theta)) \
        and np.array_equal(bits(chains.y.cpu().numpy()), bits(hc.y)) and np.array_equal(mom.sum_jump.cpu().numpy(), hm.sum_jump)
    if algo == "glmcmc":
        ok = ok and np.array_equal(bits(chains.log_w.cpu().numpy()), bits(hc.log_w)) \
            and np.array_equal(chains.flags.cpu().numpy().astype(np.uint32), hc.flags)
    return ok, desc, int(hc.n_moves.sum())


def one_case_mala(rng, oracle, k):
    """GLMALA: float64 state after the first accepted MALA move, wave-cooperative gradient, split fixed-point sums."""
    d = int(rng.integers(1, 5))
    N = int(rng.integers(1, 9))
    eps = float(np.exp(rng.uniform(np.log(0.02), np.log(3))))
    gf = float(rng.choice([0.0, 1.0, rng.random()]))
    tau = float(np.exp(rng.uniform(np.log(0.05), np.log(0.6))))
    num = int(rng.integers(2, 60))
    gspec = random_dist(rng, d, False)
    model = Mixture_set(eps).descriptor()
    model.theta_dim = model.y_dim = d
    model.prior = make_dist(("gauss", [0.0] * d, [1.0] * d)).descriptor()
    model.noise = make_dist(("gauss", [0.0] * d, [float(v) for v in np.exp(rng.normal(-1.5, 0.3, d))])).descriptor()
    y_obs = [float(v) for v in rng.choice([0.0, 1e-3, 1.5, float(rng.normal(1, 1))], d)]
    for j in range(d):
        model.y_obs[j] = y_obs[j]
    glob = make_dist(gspec).descriptor()
    mala = _capi.Mala(tau, tau ** 2, eps ** 2, num, 0)
    n, T = int(rng.integers(1, 200)), int(rng.integers(1, 40))
    spl = int(rng.integers(1, T + 1))
    lanes = int(rng.integers(0, 3))             # launch geometry: the library's choice / one / two wavefronts per 64 chains
    seed, chain0 = int(rng.integers(0, 2 ** 63)), int(rng.integers(0, 2 ** 40))
    theta0 = rng.normal(0, 1, (n, d)).astype(np.float32)
    y0 = (np.abs(theta0) + 0.2236068 * rng.normal(0, 1, (n, d))).astype(np.float32)
    dev = torch.device("cuda", 0)
    chains = engine.ChainBatch(torch.from_numpy(theta0), torch.from_numpy(y0), dev, chain0=chain0).add_mala_state()
    engine.glmala_init(model, chains)
    hist = torch.empty(T, d, n, device=dev)
    engine.run_glmala_steps(model, glob, mala, chains, T, 1, seed, gf, N, history=hist, steps_per_launch=spl,
                            lanes_per_chain=lanes)
    torch.cuda.synchronize()
    hc = oracle_lib.HostChains(theta0, y0, chain0=chain0).add_mala_state()
    hh = np.zeros((T, d, n), np.float32)
    run, keep = oracle_lib.make_run(seed=seed, step0=1, n_steps=T, gf=gf, batch=N, history=hh)
    cs = hc.struct()
    assert oracle.oracle_glmala_init(C.byref(model), C.byref(cs)) == 0
    assert oracle.oracle_glmala_steps(C.byref(model), C.byref(glob), C.byref(mala), C.byref(cs), C.byref(run)) == 0
    desc = dict(case=k, algo="glmala", d=d, N=N, eps=eps, gf=gf, tau=tau, num=num, glob=gspec, y_obs=y_obs, n=n, T=T, spl=spl,
                lanes=lanes)
    ok = np.array_equal(bits(hist.cpu().numpy()), bits(hh)) and np.array_equal(chains.theta64.cpu().numpy(), hc.theta64) \
        and np.array_equal(chains.y64.cpu().numpy(), hc.y64) and np.array_equal(chains.log_w64.cpu().numpy(), hc.log_w64) \
        and np.array_equal(chains.grad.cpu().numpy(), hc.grad) \
        and np.array_equal(chains.flags.cpu().numpy().astype(np.uint32), hc.flags)
    return ok, desc, int(hc.n_moves.sum())


def one_case_gk(rng, oracle, k):
    """GLMCMC / GlobalMCMC on the g-and-k Model (theta_dim 4, y_dim 8: exp / tanh / pow per variate, sorted data)."""
    from glabcmcmc_amd.examples.GK import GK_set
    algo = "glmcmc" if rng.random() < 0.8 else "globalmcmc"
    N = int(rng.integers(1, 9)) if algo == "glmcmc" else 1
    eps = float(np.exp(rng.uniform(np.log(0.1), np.log(5))))
    gf = float(rng.choice([0.0, 1.0, rng.random()]))
    model = GK_set(eps).descriptor()
    lspec = ("gauss", [0.0] * 4, [float(v) for v in np.exp(rng.normal(-1.8, 0.4, 4))]) if rng.random() < 0.7 else \
        ("uniform", [-0.3] * 4, [0.3] * 4)
    gspec = ("uniform", [0.0] * 4, [10.0] * 4) if rng.random() < 0.6 else \
        ("gauss", [3.0, 1.5, 2.0, 1.0], [float(v) for v in np.exp(rng.normal(0.3, 0.3, 4))])
    local, glob = make_dist(lspec).descriptor(), make_dist(gspec).descriptor()
    n, T = int(rng.integers(1, 300)), int(rng.integers(1, 40))
    lanes = int(rng.choice([0, 1, 2, 4])) if algo == "glmcmc" else 0
    seed, chain0 = int(rng.integers(0, 2 ** 63)), int(rng.integers(0, 2 ** 40))
    theta0 = rng.uniform(0.5, 5, (n, 4)).astype(np.float32)
    y0 = np.sort(rng.normal(3, 2, (n, 8)), axis=1).astype(np.float32)
    dev = torch.device("cuda", 0)
    chains = engine.ChainBatch(torch.from_numpy(theta0), torch.from_numpy(y0), dev, chain0=chain0)
    if algo == "glmcmc":
        engine.init_weights(model, glob, chains)
    hist = torch.empty(T, 4, n, device=dev)
    entry = "glabc_glmcmc_steps" if algo == "glmcmc" else "glabc_globalmcmc_steps"
    engine.run_steps(entry, model, local, glob, chains, T, 1, seed, gf, N, history=hist, lanes_per_chain=lanes)
    torch.cuda.synchronize()
    hc = oracle_lib.HostChains(theta0, y0, chain0=chain0)
    hh = np.zeros((T, 4, n), np.float32)
    run, keep = oracle_lib.make_run(seed=seed, step0=1, n_steps=T, gf=gf, batch=N, history=hh)
    cs = hc.struct()
    if algo == "glmcmc":
        assert oracle.oracle_init_weights(C.byref(model), C.byref(glob), C.byref(cs)) == 0
        rc = oracle.oracle_glmcmc_steps(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run))
    else:
        rc = oracle.oracle_globalmcmc_steps(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run))
    assert rc == 0
    desc = dict(case=k, algo=algo + "-gk", N=N, eps=eps, gf=gf, local=lspec, glob=gspec, n=n, T=T, lanes=lanes)
    ok = np.array_equal(bits(hist.cpu().numpy()), bits(hh)) and np.array_equal(bits(chains.theta.cpu().numpy()), bits(hc.theta)) \
        and np.array_equal(bits(chains.y.cpu().numpy()), bits(hc.y))
    return ok, desc, int(hc.n_moves.sum())


def one_case_nf(rng, oracle, k):
    """The RealNVP coupling stack on the matrix cores (tile mode up to 8192 rows, pair mode above) vs the oracle."""
    from glabcmcmc_amd.flows import RealNVP
    nc = int(rng.integers(1, 5))
    n = int(rng.choice([rng.integers(1, 400), rng.integers(8100, 8300), rng.integers(8193, 30000), rng.integers(65000, 70000)]))
    torch.manual_seed(int(rng.integers(0, 2 ** 31)))
    flow = RealNVP(nc)
    with torch.no_grad():
        for c in flow.couplings:
            c.l3.weight.normal_(0, float(rng.uniform(0.05, 0.5)) / 128 ** 0.5)
            c.l3.bias.normal_(0, 0.1)
        flow.q0.loc.copy_(torch.tensor([[float(rng.normal(0, 0.3)), float(rng.normal(0, 0.3))]]))
        flow.q0.log_scale.copy_(torch.tensor([[float(rng.normal(0, 0.2)), float(rng.normal(0, 0.2))]]))
    blob = flow.packed_params().numpy().copy()
    f = flow.descriptor(torch.from_numpy(blob))
    f.params = blob.ctypes.data
    seed, row0 = int(rng.integers(0, 2 ** 63)), int(rng.integers(0, 2 ** 40))
    z, lq, lp = np.empty((2, n), np.float32), np.empty(n, np.float32), np.empty(n, np.float32)
    assert oracle.oracle_nf_sample(C.byref(f), None, seed, row0, n, z.ctypes.data, lq.ctypes.data) == 0
    assert oracle.oracle_nf_log_prob(C.byref(f), z.ctypes.data, n, lp.ctypes.data) == 0
    g = flow.cuda()
    zg, lqg = g.sample(n, seed=seed, row0=row0)
    lpg = g.log_prob(zg)
    torch.cuda.synchronize()
    ok = np.array_equal(bits(zg.cpu().numpy()), bits(z.T)) and np.array_equal(bits(lqg.cpu().numpy()), bits(lq)) \
        and np.array_equal(bits(lpg.cpu().numpy()), bits(lp))
    return ok, dict(case=k, algo="nf", couplings=nc, rows=n), 0


def random_simulator(rng, d, yd, nd):
    """C source of a random simulator theta[d], eps[nd] -> y[yd]: each summary a random mix of exp / log / sqrt / fma / abs of
    random components times random noise components (which eps are read, and how many, varies from case to case)"""
    lines = []
    for j in range(yd):
        a, b, c = (int(rng.integers(0, d)) for _ in range(3))
        e1, e2 = int(rng.integers(0, nd)), int(rng.integers(0, nd))
        f = ["fabsf(theta[%d])" % a, "glabc_expf(-0.5f * fabsf(theta[%d]))" % a, "sqrtf(theta[%d] * theta[%d] + 0.25f)" % (a, a),
             "glabc_logf(1.0f + theta[%d] * theta[%d])" % (a, a), "fmaf(theta[%d], 0.5f, theta[%d])" % (a, b)][int(rng.integers(0, 5))]
        g = ["%sf * eps[%d]" % (repr(round(float(rng.uniform(0.05, 0.5)), 3)), e1),
             "%sf * eps[%d] * glabc_expf(0.1f * eps[%d])" % (repr(round(float(rng.uniform(0.05, 0.5)), 3)), e1, e2),
             "0.2f * eps[%d] + 0.1f * eps[%d] * theta[%d]" % (e1, e2, c)][int(rng.integers(0, 3))]
        lines.append("    y[%d] = %s + %s;" % (j, f, g))
    return "GLABC_SIMULATOR void glabc_user_simulate(const float* theta, const float* eps, float* y)\n{\n%s\n}\n" % "\n".join(lines)


def random_hooks(rng, d, yd):
    """C source of random replacements of the Model's other callbacks (include/glabc.h, "The Model's OTHER callbacks as user
    source"): each of prior / discrepancy / kernel is announced with probability 1/2"""
    src = ""
    if rng.random() < 0.5:
        s = repr(round(float(rng.uniform(0.6, 2.5)), 3))
        terms = " + ".join(["fabsf(theta[%d])" % j if rng.random() < 0.5 else "0.5f * (theta[%d] * theta[%d])" % (j, j) for j in range(d)])
        src += ("#define GLABC_USER_PRIOR 1\nGLABC_SIMULATOR float glabc_user_prior_log_prob(const float* theta)\n"
                "{\n    return -((%s) / %sf) - 1.25f;\n}\n" % (terms, s))
    if rng.random() < 0.5:
        kind = int(rng.integers(0, 3))
        if kind == 0:                                                  # weighted L1
            body = "    float s = 0.0f;\n" + "".join("    s += %sf * fabsf(y[%d] - y_obs[%d]);\n" % (repr(round(float(rng.uniform(0.3, 2.0)), 2)), j, j)
                                                     for j in range(yd)) + "    return s;\n"
        elif kind == 1:                                                # L-infinity
            body = "    float s = 0.0f;\n" + "".join("    s = fmaxf(s, fabsf(y[%d] - y_obs[%d]));\n" % (j, j) for j in range(yd)) + "    return s;\n"
        else:                                                          # Euclidean in another summation order
            body = "    float s = 0.0f;\n" + "".join("    s = fmaf(y[%d] - y_obs[%d], y[%d] - y_obs[%d], s);\n" % (j, j, j, j)
                                                     for j in reversed(range(yd))) + "    return sqrtf(s);\n"
        src += "#define GLABC_USER_DISCREPANCY 1\nGLABC_SIMULATOR float glabc_user_discrepancy(const float* y, const float* y_obs)\n{\n%s}\n" % body
    if rng.random() < 0.5:
        body = ["    return -dis / scale;\n",
                "    const float u = dis / (3.0f * scale);\n    return u < 0.9f ? glabc_logf(1.0f - u * u) : -1.6607312f - 40.0f * (u - 0.9f);\n",
                "    const float u = dis / scale;\n    return -glabc_logf(1.0f + u * u);\n"][int(rng.integers(0, 3))]
        src += "#define GLABC_USER_KERNEL 1\nGLABC_SIMULATOR float glabc_user_log_kernel(float dis, float scale)\n{\n%s}\n" % body
    return src


def one_case_rtc(rng, oracle, k):
    """A random user simulator compiled into the fused kernel at run time (glabc_rtc_compile) vs the checker calling the same
    source through gcc; theta_dim / y_dim / noise_dim 1..8, N 1..16; every other case replaces a random subset of the Model's prior / discrepancy /
    kernel by user source too."""
    import glabcmcmc_amd as g_
    from test_rtc import host_simulator
    d, yd, nd = int(rng.integers(1, 9)), int(rng.integers(1, 9)), int(rng.integers(1, 9))
    if rng.random() < 0.4:
        yd = d                                   # theta_dim == y_dim: the unit-Gaussian instantiation exists as well
    algo = "glmcmc" if rng.random() < 0.8 else "globalmcmc"
    N = int(rng.integers(1, 17)) if algo == "glmcmc" else 1
    src = random_simulator(rng, d, yd, nd)
    if rng.random() < 0.5:
        src += random_hooks(rng, d, yd)           # the Model's prior / discrepancy / kernel as user source as well
    from test_rtc import host_hooks
    keep_lib, fn = host_simulator(src, d, yd, nd)
    oracle.oracle_set_user_simulator(fn)
    oracle.oracle_set_user_model(*host_hooks(keep_lib))
    eps = float(np.exp(rng.uniform(np.log(0.05), np.log(5))))
    gf = float(rng.choice([0.0, 1.0, rng.random()]))
    lspec, gspec = random_dist(rng, d, True), random_dist(rng, d, False)
    unit = rng.random() < 0.5                     # unit prior + unit global proposal + y_obs away from 0 -> VAR_GAUSS_UNIT is launched
    prior = make_dist(("gauss", [0.0] * d, [1.0] * d if unit else [float(v) for v in np.exp(rng.normal(0.2, 0.3, d))]))
    if unit:
        gspec = ("gauss", [0.0] * d, [1.0] * d)
    cm = g_.CompiledModel(d, yd, src, prior, [float(v) for v in rng.normal(0.8, 0.3, yd)], eps, noise_dim=nd)
    model = cm.descriptor()
    local, glob = make_dist(lspec).descriptor(), make_dist(gspec).descriptor()
    n, T = int(rng.integers(1, 400)), int(rng.integers(1, 30))
    seed, chain0 = int(rng.integers(0, 2 ** 63)), int(rng.integers(0, 2 ** 40))
    theta0 = rng.normal(0, 1, (n, d)).astype(np.float32)
    y0 = rng.normal(0.8, 0.5, (n, yd)).astype(np.float32)
    dev = torch.device("cuda", 0)
    chains = engine.ChainBatch(torch.from_numpy(theta0), torch.from_numpy(y0), dev, chain0=chain0)
    hist = torch.empty(T, d, n, device=dev)       # (no init_weights: the chains start `local`, GLMCMC.py:50, so the first global move computes it)
    mom = engine.Moments(n, d, dev)
    a = _capi.ALGO_GLMCMC if algo == "glmcmc" else _capi.ALGO_GLOBALMCMC
    engine.run_steps(None, model, local, glob, chains, T, 1, seed, gf, N, history=hist, moments=mom, rtc_program=cm.program(a, N))
    torch.cuda.synchronize()
    hc = oracle_lib.HostChains(theta0, y0, chain0=chain0)
    hh = np.zeros((T, d, n), np.float32)
    hm = oracle_lib.HostMoments(n, d)
    run, keep = oracle_lib.make_run(seed=seed, step0=1, n_steps=T, gf=gf, batch=N, history=hh, moments=hm)
    cs = hc.struct()
    if algo == "glmcmc":
        rc = oracle.oracle_glmcmc_steps(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run))
    else:
        rc = oracle.oracle_globalmcmc_steps(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run))
    assert rc == 0
    desc = dict(case=k, algo=algo + "-rtc", d=d, yd=yd, nd=nd, N=N, eps=eps, gf=gf, local=lspec, glob=gspec, n=n, T=T, source=src)
    ok = np.array_equal(bits(hist.cpu().numpy()), bits(hh)) and np.array_equal(bits(chains.theta.cpu().numpy()), bits(hc.theta)) \
        and np.array_equal(bits(chains.y.cpu().numpy()), bits(hc.y)) and np.array_equal(mom.sum_jump.cpu().numpy(), hm.sum_jump)
    if algo == "glmcmc":
        ok = ok and np.array_equal(bits(chains.log_w.cpu().numpy()), bits(hc.log_w))
    del cm
    oracle.oracle_set_user_model(None, None, None)
    return ok, desc, int(hc.n_moves.sum())


def one_case_nf_grad(rng, oracle, k):
    """The hand-written backward of the coupling stack (glabc_nf_grad) vs the checker's double-precision gradient: every
    tensor within 2e-4 of its largest entry, the loss within 2e-6 relative (floating-point tolerance, see tests/test_nf_train.py)."""
    from glabcmcmc_amd.flows import HipAdam, RealNVP
    nc = int(rng.integers(1, 7))
    n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 3000), rng.integers(3000, 12000)]))
    torch.manual_seed(int(rng.integers(0, 2 ** 31)))
    flow = RealNVP(nc)
    with torch.no_grad():
        for c in flow.couplings:
            c.l3.weight.normal_(0, float(rng.uniform(0.05, 0.5)) / 128 ** 0.5)
            c.l3.bias.normal_(0, 0.1)
            c.l1.bias.normal_(0, float(rng.uniform(0.0, 0.5)))
            c.l2.bias.normal_(0, 0.1)
        flow.q0.loc.copy_(torch.tensor([[float(rng.normal(0, 0.3)), float(rng.normal(0, 0.3))]]))
        flow.q0.log_scale.copy_(torch.tensor([[float(rng.normal(0, 0.2)), float(rng.normal(0, 0.2))]]))
    blob = flow.packed_params().numpy().copy()
    f = flow.descriptor(torch.from_numpy(blob))
    f.params = blob.ctypes.data
    x = (rng.standard_normal((2, n)) * float(rng.uniform(0.5, 2.0)) + rng.normal(0, 0.5, (2, 1))).astype(np.float32)
    gp, gb, loss = np.zeros_like(blob), np.zeros(4, np.float32), np.zeros(1, np.float32)
    assert oracle.oracle_nf_grad(C.byref(f), x.ctypes.data, n, gp.ctypes.data, gb.ctypes.data, loss.ctypes.data) == 0
    opt = HipAdam(flow.cuda())
    lh, gph, gbh = opt.gradient(torch.from_numpy(x).cuda(), chain_major=True)
    torch.cuda.synchronize()
    gph, gbh, lh = gph.cpu().numpy(), gbh.cpu().numpy(), float(lh)
    H = 128
    ok = abs(lh - float(loss[0])) <= 2e-6 * abs(float(loss[0])) + 1e-7 and np.allclose(gbh, gb, rtol=2e-4, atol=2e-6)
    v4g, v4h = gp[:, H * H + 2 * H:H * H + 6 * H].reshape(nc, H, 4), gph[:, H * H + 2 * H:H * H + 6 * H].reshape(nc, H, 4)
    worst = 0.0
    for a, b in ((gp[:, :H * H], gph[:, :H * H]), (gp[:, H * H:H * H + H], gph[:, H * H:H * H + H]),
                 (gp[:, H * H + H:H * H + 2 * H], gph[:, H * H + H:H * H + 2 * H]), (v4g[:, :, 0], v4h[:, :, 0]),
                 (v4g[:, :, 1:3], v4h[:, :, 1:3]), (gp[:, H * H + 6 * H:H * H + 6 * H + 2], gph[:, H * H + 6 * H:H * H + 6 * H + 2])):
        scale = np.abs(a).max()
        ok = ok and bool(np.isfinite(b).all())
        worst = max(worst, float(np.abs(a - b).max() / (scale + 1e-30)))
    ok = ok and worst <= 2e-4         # (both sides open the ReLU gates of the float32 evaluation: no jumps at the kinks)
    return ok, dict(case=k, algo="nf-grad", couplings=nc, rows=n, loss=float(loss[0]), worst=worst), 0



def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    oracle = oracle_lib.load()
    _capi.lib()
    RTC_ONLY = len(sys.argv) > 3 and sys.argv[3] == "rtc"
    NF_ONLY = len(sys.argv) > 3 and sys.argv[3] == "nfgrad"
    MALA_ONLY = len(sys.argv) > 3 and sys.argv[3] == "mala"
    t0, k, moves, bad = time.time(), 0, 0, []
    while time.time() - t0 < budget:
        fn = one_case_mala if MALA_ONLY else one_case_nf_grad if (NF_ONLY or k % 64 == 17) else one_case_rtc if (RTC_ONLY or k % 16 == 6) else one_case_mala if k % 4 == 3 else one_case_gk if k % 8 == 5 else \
            one_case_nf if k % 32 == 9 else one_case
        ok, desc, mv = fn(rng, oracle, k)
        moves += mv
        if not ok:
            bad.append(desc)
            print("MISMATCH", desc, flush=True)
        k += 1
        if k % 50 == 0:
            print("%d cases, %d moves, %d mismatches" % (k, moves, len(bad)), flush=True)
    print("done: %d cases, %d accepted moves, %d mismatches" % (k, moves, len(bad)))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
