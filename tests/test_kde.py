"""KernelDensity + AGLMCMC (reference kernel_density.py, AGLMCMC.py).

CPU: the oracle's restatement against what the reference's KernelDensity computed (tests/golden/kde.npz, written by
make_golden.py from the reference on the CPU).  Tolerance, not bits: the reference sums in float32 with torch's
cascade and uses torch's exp/log; the restatement specifies its own sums (float64 fixed order / exact fixed point).
GPU: the gfx950 kernels against the oracle, bit for bit; the KernelDensity class against the golden numbers; AGLMCMC's
posterior against the plain GLMCMC sampler's.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from glabcmcmc_amd import _capi as A
from helpers import bits, descriptors, load_golden

RTOL = 2e-5          # float32 summation-order differences (bandwidth, weights)
LP_ATOL = 2e-4       # log density: |log p| reaches ~1e3 in the far tail, float32 ulp there is 6e-5


def _cases(g):
    return eval(str(g["cases"]))


def oracle_fit(oracle, X, w, h, bw_fixed):
    n, d = X.shape
    xs = np.ascontiguousarray(X.T)
    weights, log_w = np.empty(n, np.float32), np.empty(n, np.float32)
    wq, consts = np.empty(n, np.int64), np.empty(d + 2, np.float32)
    bwp = None if bw_fixed is None else np.ascontiguousarray(bw_fixed, np.float32).ctypes.data
    rc = oracle.oracle_kde_fit(xs.ctypes.data, None if w is None else w.ctypes.data, n, d, float(h), bwp, weights.ctypes.data,
                               log_w.ctypes.data, wq.ctypes.data, consts.ctypes.data)
    assert rc == 0
    return xs, weights, log_w, wq, consts


def kde_struct(xs, log_w, cum_q, consts, d, n):
    k = A.Kde()
    k.dim, k.n_samples = d, n
    k.x, k.log_w = xs.ctypes.data, log_w.ctypes.data
    k.cum_q = None if cum_q is None else cum_q.ctypes.data
    for j in range(d):
        k.bandwidth[j] = float(consts[j])
    k.sum_log_bw, k.c_2pi = float(consts[d]), float(consts[d + 1])
    return k


def rule(kind, n, d):
    return (n * (d + 2) / 4.) ** (-1. / (d + 4)) if kind == "silverman" else n ** (-1. / (d + 4))


def case_inputs(g, tag, d, n, weighted, bw):
    X = g[tag + "_X"]
    w = g[tag + "_w"] if weighted else None
    if bw == "fixed":
        b = g[tag + "_bw_in"]
        return X, w, 0.0, (np.repeat(b, d) if b.size == 1 else b).astype(np.float32)
    return X, w, rule(bw, n, d), None


def test_oracle_kde_matches_reference():
    import oracle_lib
    oracle = oracle_lib.load()
    g = load_golden("kde")
    for tag, d, n, weighted, bw in _cases(g):
        X, w, h, bw_fixed = case_inputs(g, tag, d, n, weighted, bw)
        xs, weights, log_w, wq, consts = oracle_fit(oracle, X, w, h, bw_fixed)
        np.testing.assert_allclose(weights, g[tag + "_weights"], rtol=RTOL, atol=1e-12)
        np.testing.assert_allclose(consts[:d], g[tag + "_bandwidth"], rtol=RTOL)
        assert abs(int(wq.sum()) / 2.0 ** 40 - 1) < 1e-6                               # the integer weights sum to 1 (float32 weights)
        k = kde_struct(xs, log_w, None, consts, d, n)
        pts = g[tag + "_pts"]
        ps = np.ascontiguousarray(pts.T)
        out = np.empty(len(pts), np.float32)
        assert oracle.oracle_kde_log_prob(C.byref(k), ps.ctypes.data, len(pts), out.ctypes.data) == 0
        ref = g[tag + "_log_prob"]
        np.testing.assert_allclose(out, ref, rtol=RTOL, atol=LP_ATOL)
        assert np.isfinite(out).all()


def test_oracle_train_weights_match_reference(oracle):
    g = load_golden("kde")
    cfg = eval(str(g["cfg"]))
    theta, dis, logq = g["tw_theta"], g["tw_dis"], g["tw_logq"]
    ts = np.ascontiguousarray(theta.T)
    for j in range(3):
        model, _, _ = descriptors(cfg, g)
        eps, ls, sc = g["tw%d_consts" % j]
        model.kern_log_scale, model.kern_scale = float(ls), float(sc)
        w = np.empty(len(dis), np.float32)
        assert oracle.oracle_kde_train_weights(C.byref(model), ts.ctypes.data, dis.ctypes.data, logq.ctypes.data, len(dis),
                                               w.ctypes.data) == 0
        ref = g["tw%d" % j]
        np.testing.assert_allclose(w, ref, rtol=3e-6, atol=1e-37)


def test_oracle_kde_sample_is_a_draw_from_the_mixture(oracle):
    """inverse-CDF index on the integer prefix sums + bandwidth*normal: index frequencies follow the weights, the
    residuals are N(0, bw^2)"""
    rng = np.random.default_rng(5)
    n, d = 5, 2
    X = (rng.standard_normal((n, d)) * 10).astype(np.float32)
    w = np.array([0.5, 0.1, 0.0, 0.3, 0.1], np.float32)
    xs, weights, log_w, wq, consts = oracle_fit(oracle, X, w, 0.0, np.array([0.05, 0.2], np.float32))
    cum = np.cumsum(wq)
    k = kde_struct(xs, log_w, cum, consts, d, n)
    m = 40000
    out = np.empty((d, m), np.float32)
    assert oracle.oracle_kde_sample(C.byref(k), m, 123, 0, out.ctypes.data) == 0
    idx = np.argmin(((out.T[:, None, :] - X[None]) ** 2).sum(-1), 1)
    freq = np.bincount(idx, minlength=n) / m
    assert np.abs(freq - w).max() < 0.01 and freq[2] == 0
    res = out.T - X[idx]
    assert np.allclose(res.std(0), [0.05, 0.2], rtol=0.03) and np.abs(res.mean(0)).max() < 0.005
    # another row offset continues the same stream
    out2 = np.empty((d, 100), np.float32)
    assert oracle.oracle_kde_sample(C.byref(k), 100, 123, 500, out2.ctypes.data) == 0
    assert np.array_equal(bits(out2), bits(out[:, 500:600]))


def test_oracle_dist_forward_philox(oracle):
    from helpers import make_dist
    for spec in (("gauss", [0.5, -1, 2], [0.3, 1.0, 2.0]), ("uniform", [-2, 0], [2, 5])):
        dist = make_dist(spec).descriptor()
        d, n = dist.dim, 20000
        z, lp = np.empty((d, n), np.float32), np.empty(n, np.float32)
        assert oracle.oracle_dist_forward_philox(C.byref(dist), n, 9, 1 << 33, z.ctypes.data, lp.ctypes.data) == 0
        back = np.empty(n, np.float32)
        zr = np.ascontiguousarray(z.T)
        assert oracle.oracle_dist_log_prob(C.byref(dist), zr.ctypes.data, n, back.ctypes.data) == 0
        np.testing.assert_allclose(lp, back, rtol=0, atol=2e-5)                    # forward's log_p == log_prob(z) up to rounding
        if spec[0] == "gauss":
            assert np.allclose(z.mean(1), spec[1], atol=0.05) and np.allclose(z.std(1), spec[2], rtol=0.03)
        else:
            assert (z.min(1) >= spec[1]).all() and (z.max(1) <= spec[2]).all()


# ----------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_hip_kde_equals_oracle(hip, oracle):
    """glabc_kde_fit / _log_prob / _sample / _train_weights and glabc_dist_forward against the oracle, bit for bit."""
    g = load_golden("kde")
    s = None
    for tag, d, n, weighted, bw in _cases(g):
        X, w, h, bw_fixed = case_inputs(g, tag, d, n, weighted, bw)
        xs, weights, log_w, wq, consts = oracle_fit(oracle, X, w, h, bw_fixed)
        xg = torch.from_numpy(xs).cuda()
        wg = None if w is None else torch.from_numpy(w).cuda()
        weights_g, log_w_g = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
        wq_g, consts_g = torch.empty(n, dtype=torch.int64, device="cuda"), torch.empty(d + 2, device="cuda")
        bwp = None if bw_fixed is None else (C.c_float * d)(*[float(v) for v in bw_fixed])
        rc = hip.glabc_kde_fit(xg.data_ptr(), None if wg is None else wg.data_ptr(), n, d, float(h), bwp, weights_g.data_ptr(),
                               log_w_g.data_ptr(), wq_g.data_ptr(), consts_g.data_ptr(), s)
        assert rc == 0
        torch.cuda.synchronize()
        assert np.array_equal(bits(weights_g.cpu().numpy()), bits(weights))
        assert np.array_equal(bits(log_w_g.cpu().numpy()), bits(log_w))
        assert np.array_equal(wq_g.cpu().numpy(), wq)
        assert np.array_equal(bits(consts_g.cpu().numpy()), bits(consts))
        cum = np.cumsum(wq)
        cum_g = torch.cumsum(wq_g, 0)
        assert np.array_equal(cum_g.cpu().numpy(), cum)
        k = kde_struct(xs, log_w, cum, consts, d, n)
        kg = A.Kde()
        C.memmove(C.byref(kg), C.byref(k), C.sizeof(k))
        kg.x, kg.log_w, kg.cum_q = xg.data_ptr(), log_w_g.data_ptr(), cum_g.data_ptr()
        rng = np.random.default_rng(3)
        pts = np.concatenate([g[tag + "_pts"], (rng.standard_normal((777, d)) * 2).astype(np.float32)])
        ps = np.ascontiguousarray(pts.T)
        ref = np.empty(len(pts), np.float32)
        assert oracle.oracle_kde_log_prob(C.byref(k), ps.ctypes.data, len(pts), ref.ctypes.data) == 0
        pg, og = torch.from_numpy(ps).cuda(), torch.empty(len(pts), device="cuda")
        assert hip.glabc_kde_log_prob(C.byref(kg), pg.data_ptr(), len(pts), og.data_ptr(), s) == 0
        torch.cuda.synchronize()
        assert np.array_equal(bits(og.cpu().numpy()), bits(ref)), tag
        m = 5000
        ref_s = np.empty((d, m), np.float32)
        assert oracle.oracle_kde_sample(C.byref(k), m, 77, 1 << 34, ref_s.ctypes.data) == 0
        sg = torch.empty(d, m, device="cuda")
        assert hip.glabc_kde_sample(C.byref(kg), m, 77, 1 << 34, sg.data_ptr(), s) == 0
        torch.cuda.synchronize()
        assert np.array_equal(bits(sg.cpu().numpy()), bits(ref_s)), tag
    # training weights
    cfg = eval(str(g["cfg"]))
    theta, dis, logq = g["tw_theta"], g["tw_dis"], g["tw_logq"]
    ts = np.ascontiguousarray(theta.T)
    for j in range(3):
        model, _, _ = descriptors(cfg, g)
        eps, ls, sc = g["tw%d_consts" % j]
        model.kern_log_scale, model.kern_scale = float(ls), float(sc)
        w = np.empty(len(dis), np.float32)
        assert oracle.oracle_kde_train_weights(C.byref(model), ts.ctypes.data, dis.ctypes.data, logq.ctypes.data, len(dis),
                                               w.ctypes.data) == 0
        tg, dg, lg = torch.from_numpy(ts).cuda(), torch.from_numpy(dis).cuda(), torch.from_numpy(logq).cuda()
        wg = torch.empty(len(dis), device="cuda")
        assert hip.glabc_kde_train_weights(C.byref(model), tg.data_ptr(), dg.data_ptr(), lg.data_ptr(), len(dis), wg.data_ptr(), s) == 0
        torch.cuda.synchronize()
        assert np.array_equal(bits(wg.cpu().numpy()), bits(w))
    # forward() draws
    from helpers import make_dist
    for spec in (("gauss", [0.5, -1, 2], [0.3, 1.0, 2.0]), ("uniform", [-2, 0], [2, 5]), ("gauss", [0], [1]),
                 ("gauss", [0, 1, 2, 3], [1, 2, 3, 4])):
        dist = make_dist(spec).descriptor()
        d, n = dist.dim, 3001
        z, lp = np.empty((d, n), np.float32), np.empty(n, np.float32)
        assert oracle.oracle_dist_forward_philox(C.byref(dist), n, 9, 1 << 33, z.ctypes.data, lp.ctypes.data) == 0
        zg, lg = torch.empty(d, n, device="cuda"), torch.empty(n, device="cuda")
        assert hip.glabc_dist_forward(C.byref(dist), n, 9, 1 << 33, zg.data_ptr(), lg.data_ptr(), s) == 0
        torch.cuda.synchronize()
        assert np.array_equal(bits(zg.cpu().numpy()), bits(z)) and np.array_equal(bits(lg.cpu().numpy()), bits(lp))


@pytest.mark.gpu
def test_kernel_density_class_matches_reference(hip):
    """The drop-in class (fit / log_prob / sample / forward) against the reference's numbers."""
    from glabcmcmc_amd import KernelDensity
    g = load_golden("kde")
    for tag, d, n, weighted, bw in _cases(g):
        bw_arg = bw if bw != "fixed" else (float(g[tag + "_bw_in"][0]) if g[tag + "_bw_in"].size == 1 else torch.from_numpy(g[tag + "_bw_in"]))
        k = KernelDensity(bandwidth=bw_arg, device="cuda", seed=5)
        k.fit(torch.from_numpy(g[tag + "_X"]), torch.from_numpy(g[tag + "_w"]) if weighted else None)
        np.testing.assert_allclose(k.bandwidth.cpu().numpy(), g[tag + "_bandwidth"], rtol=RTOL)
        np.testing.assert_allclose(k.weights.cpu().numpy(), g[tag + "_weights"], rtol=RTOL, atol=1e-12)
        lp = k.log_prob(torch.from_numpy(g[tag + "_pts"]))
        np.testing.assert_allclose(lp.cpu().numpy(), g[tag + "_log_prob"], rtol=RTOL, atol=LP_ATOL)
        z, lq = k.forward(2000)
        assert z.shape == (2000, d) and lq.shape == (2000,)
        np.testing.assert_array_equal(k.log_prob(z).cpu().numpy(), lq.cpu().numpy())
        z2 = k.sample(2000)
        assert not torch.equal(z, z2)                                         # the stream advances between calls
        # draws follow the estimator: mean of the draws ~ weighted mean of the centres
        wm = (k.weights[:, None] * k.X).sum(0)
        big = k.sample(200000)
        sd = torch.sqrt((k.weights[:, None] * (k.X - wm) ** 2).sum(0) + k.bandwidth ** 2)
        assert torch.all((big.mean(0) - wm).abs() < 5 * sd / 200000 ** 0.5 + 1e-3)
    with pytest.raises(RuntimeError):
        KernelDensity(device="cuda").log_prob(torch.zeros(1, 2))


@pytest.mark.gpu
def test_aglmcmc_end_to_end(hip, tmp_path):
    """AGLMCMC on the example Model: returns Theta_Re (num_ite, d), anneals hat_eps to its target, refits the KDE, and its
    pooled posterior matches the plain GLMCMC sampler's (same target, epsilon 0.3) within Monte-Carlo error."""
    from glabcmcmc_amd import AGLMCMC, GLMCMC, MCMCRunner, distribution
    from glabcmcmc_amd.examples.Mixture import Mixture_set
    torch.manual_seed(0)
    Model = Mixture_set(0.3)
    lp = distribution.DiagGaussian(2, loc=torch.zeros(1, 2), log_scale=torch.log(torch.tensor([0.35, 0.35])))
    ip = distribution.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.5, 0.5]))
    # single chain through the runner, CSV written, more than the reference's hard-wired 10 000 rows would be fine too
    theta0 = torch.tensor([1.5, 1.5])
    y0 = Model.generate_samples(theta0)
    runner = MCMCRunner(Model, str(tmp_path))
    st = {}
    out = runner.run_aglmcmc(num_iterations=1500, initial_theta=theta0, initial_y=y0, global_frequency=0.6, local_proposal=lp,
                             Initial_ISIR_prop=ip, batch_size=5, step_size=40, alpha=0.8, hat_eps_T=0.5, seed=3, verbose=False,
                             state_out=st)
    assert out.shape == (1500, 2) and out.dtype == torch.float32 and torch.isfinite(out).all()
    assert st["num_train"] >= 3 and st["hat_eps"] == 0.5 and st["kde"] is not None
    rows = open(tmp_path / "glmcmc_results.csv").read().strip().split("\n")
    assert len(rows) == 1500
    # batched: 512 chains share the adaptive proposal
    n = 512
    th0 = torch.zeros(n, 2) + 1.5
    y0 = Model.generate_samples(th0)
    a = AGLMCMC(Model, 1200, th0, y0, lp, ip, None, 0.6, 30, 5, 0.8, 0.5, seed=4, verbose=False, state_out=st,
                check_density_cache=True)          # the per-chain proposal-density cache == a full evaluation, every iteration
    b = GLMCMC(Model, 1200, th0, y0, lp, None, 0.6, ip, 5, seed=5, verbose=False)
    assert a.shape == (1200, n, 2)
    pa, pb = a[400:].abs().reshape(-1, 2), b[400:].abs().reshape(-1, 2)
    assert torch.allclose(pa.mean(0), pb.mean(0), atol=0.03), (pa.mean(0), pb.mean(0))
    assert torch.allclose(pa.std(0), pb.std(0), atol=0.03), (pa.std(0), pb.std(0))
    assert st["num_train"] >= 3


@pytest.mark.gpu
def test_aglmcmc_follows_the_reference_chain(hip):
    """tests/golden/aglmcmc_philox.npz: the reference's AGLMCMC (unmodified loop, 1500 iterations, 22 KDE refits) driven by
    the build's Philox streams (make_golden.AGTape).  The GPU sampler must walk the same chain.  Not bit for bit: after
    the first refit the pool entries carry the KDE bandwidth, whose float32 sums the reference takes in torch's order
    (1e-7 relative), so states agree to ~1e-6 -- and every accept / resample / centre-index decision must agree."""
    from glabcmcmc_amd import AGLMCMC, distribution
    from glabcmcmc_amd.examples.Mixture import Mixture_set
    g = load_golden("aglmcmc_philox")
    cfg = eval(str(g["cfg"]))
    Model = Mixture_set(cfg["epsilon"])
    lp = distribution.DiagGaussian(2, loc=torch.zeros(2), log_scale=torch.log(torch.tensor([0.35, 0.35])))
    ip = distribution.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.5, 0.5]))
    st = {}
    out = AGLMCMC(Model, cfg["T"], torch.from_numpy(g["theta0"]), torch.from_numpy(g["y0"]), lp, ip, None, cfg["gf"],
                  cfg["step_size"], cfg["batch_size"], cfg["alpha"], cfg["hat_eps_T"], seed=cfg["seed"], verbose=False,
                  state_out=st)
    ref = g["chain"]
    got = out.numpy()
    assert got.shape == ref.shape
    diff = np.abs(got - ref).max(1)
    first_bad = int(np.argmax(diff > 1e-4)) if (diff > 1e-4).any() else -1
    assert first_bad == -1, "chains part at iteration %d: %s vs %s" % (first_bad, got[first_bad], ref[first_bad])
    assert st["num_train"] == int(g["kde_refits"])
    assert np.array_equal(bits(got[:40]), bits(ref[:40]))                  # before the first refit: bit for bit


@pytest.mark.gpu
def test_aglmcmc_with_a_callback_model(hip, tmp_path):
    """run_aglmcmc with a user's plain-torch Model (no descriptor, no calculate_log_kernel_dis: the kernel of a discrepancy is
    taken from calculate_log_kernel(y, epsilon)) goes through generic.run_aglmcmc: reference shapes and CSV, hat_eps annealed
    to its target, KDE refits, and the pooled posterior agrees with the fused AGLMCMC on the built-in Model."""
    from glabcmcmc_amd import AGLMCMC, MCMCRunner, distribution
    from glabcmcmc_amd.examples.Mixture import Mixture_set
    from glabcmcmc_amd.examples.UserModel import TorchMixture
    torch.manual_seed(0)
    user = TorchMixture(2, 0.3)
    lp = distribution.DiagGaussian(2, loc=torch.zeros(1, 2), log_scale=torch.log(torch.tensor([0.35, 0.35])))
    ip = distribution.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.5, 0.5]))
    theta0 = torch.tensor([1.5, 1.5])
    st = {}
    out = MCMCRunner(user, str(tmp_path)).run_aglmcmc(num_iterations=600, initial_theta=theta0, initial_y=user.generate_samples(theta0),
                                                      global_frequency=0.6, local_proposal=lp, Initial_ISIR_prop=ip, batch_size=5,
                                                      step_size=40, alpha=0.8, hat_eps_T=0.5, seed=3, verbose=False, state_out=st)
    assert out.shape == (600, 2) and out.dtype == torch.float32 and torch.isfinite(out).all()
    assert st["num_train"] >= 2 and st["kde"] is not None and "callback_device" in st
    assert len(open(tmp_path / "glmcmc_results.csv").read().strip().split("\n")) == 600
    n = 512
    th0 = torch.zeros(n, 2) + 1.5
    a = AGLMCMC(user, 1000, th0, user.generate_samples(th0), lp, ip, None, 0.6, 30, 5, 0.8, 0.5, seed=4, verbose=False, state_out=st)
    b = AGLMCMC(Mixture_set(0.3), 1000, th0, user.generate_samples(th0), lp, ip, None, 0.6, 30, 5, 0.8, 0.5, seed=5, verbose=False)
    assert a.shape == (1000, n, 2) and st["hat_eps"] == 0.5 and st["num_train"] >= 3
    pa, pb = a[400:].abs().reshape(-1, 2), b[400:].abs().reshape(-1, 2)
    assert torch.allclose(pa.mean(0), pb.mean(0), atol=0.03), (pa.mean(0), pb.mean(0))
    assert torch.allclose(pa.std(0), pb.std(0), atol=0.03), (pa.std(0), pb.std(0))


@pytest.mark.gpu
@pytest.mark.parametrize("d,n", [(2, 8192), (2, 5000), (1, 8192), (3, 3000)])
def test_hip_kde_log_prob_at_aglmcmc_size_equals_oracle(hip, oracle, d, n):
    """KDEs of AGLMCMC's size (up to 8192 centres) against the checker, bit for bit -- zero-weight centres, far tails and NaN
    points included -- also through the indexed entry point (listed points only, count on the device, the rest untouched)."""
    rng = np.random.default_rng(n + d)
    X = (rng.standard_normal((n, d)) * 1.5).astype(np.float32)
    w = rng.random(n).astype(np.float32)
    w[rng.random(n) < 0.1] = 0.0
    xs, weights, log_w, wq, consts = oracle_fit(oracle, X, w, rule("silverman", n, d), None)
    k = kde_struct(xs, log_w, None, consts, d, n)
    pts = (rng.standard_normal((1500, d)) * 2.5).astype(np.float32)
    pts[7] = 60.0                                                       # far tail
    pts[11, 0] = np.nan
    ps = np.ascontiguousarray(pts.T)
    ref = np.empty(len(pts), np.float32)
    assert oracle.oracle_kde_log_prob(C.byref(k), ps.ctypes.data, len(pts), ref.ctypes.data) == 0
    xg, lg = torch.from_numpy(xs).cuda(), torch.from_numpy(log_w).cuda()
    kg = A.Kde()
    C.memmove(C.byref(kg), C.byref(k), C.sizeof(k))
    kg.x, kg.log_w, kg.cum_q = xg.data_ptr(), lg.data_ptr(), None
    pg, og = torch.from_numpy(ps).cuda(), torch.empty(len(pts), device="cuda")
    assert hip.glabc_kde_log_prob(C.byref(kg), pg.data_ptr(), len(pts), og.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(bits(og.cpu().numpy()), bits(ref))
    # indexed: a third of the points, in a shuffled order, count on the device
    idx = torch.from_numpy(rng.permutation(len(pts))[:700].astype(np.int32)).cuda()
    cnt = torch.tensor([700], dtype=torch.int32, device="cuda")
    out = torch.full((len(pts),), -7.0, device="cuda")
    assert hip.glabc_kde_log_prob_indexed(C.byref(kg), pg.data_ptr(), len(pts), idx.data_ptr(), cnt.data_ptr(), len(pts),
                                          out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    sel = idx.cpu().numpy()
    assert np.array_equal(bits(got[sel]), bits(ref[sel]))
    rest = np.setdiff1d(np.arange(len(pts)), sel)
    assert (got[rest] == -7.0).all()
