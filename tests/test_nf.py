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
@pytest.mark.parametrize("n_couplings,n", [(1, 100), (3, 513), (8, 4096), (32, 1000), (2, 8193), (3, 20000), (2, 32768), (2, 32800),
                                           (2, 70001), (1, 140000)])
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


def _mix_descs(eps=0.3):
    from helpers import descriptors
    return descriptors(dict(epsilon=eps, local=("gauss", [0, 0], [0.35, 0.35]), **{"global": ("gauss", [0, 0], [1, 1])}))


@pytest.mark.gpu
def test_pool_weights_and_nf_step_equal_oracle(hip, oracle):
    """glabc_pool_weights and glabc_glmcmc_nf_step (GLMCMC_NFs.py:73-111,141-152) against the oracle, bit for bit."""
    import oracle_lib
    model, local, _ = _mix_descs()
    rng = np.random.default_rng(0)
    n_chains, N, step_size = 300, 5, 3
    P = N * step_size
    rows = P * n_chains
    theta = (rng.standard_normal((2, rows)) * 1.5).astype(np.float32)
    log_q = (rng.standard_normal(rows) - 3).astype(np.float32)
    x = np.empty((2, rows), np.float32)
    w = np.empty(rows, np.float32)
    assert oracle.oracle_pool_weights(C.byref(model), theta.ctypes.data, log_q.ctypes.data, rows, 11, 1 << 35, x.ctypes.data,
                                      w.ctypes.data) == 0
    tg, lg = torch.from_numpy(theta).cuda(), torch.from_numpy(log_q).cuda()
    xg, wg = torch.empty(2, rows, device="cuda"), torch.empty(rows, device="cuda")
    assert hip.glabc_pool_weights(C.byref(model), tg.data_ptr(), lg.data_ptr(), rows, 11, 1 << 35, xg.data_ptr(), wg.data_ptr(),
                                  None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(bits(xg.cpu().numpy()), bits(x)) and np.array_equal(bits(wg.cpu().numpy()), bits(w))
    assert (w > 0).any()
    # iterations against the pool (more global steps than slices: the exhausted-pool guard is exercised too)
    theta0 = rng.standard_normal((n_chains, 2)).astype(np.float32)
    y0 = np.abs(theta0).astype(np.float32)
    hc = oracle_lib.HostChains(theta0, y0, chain0=7, with_isir=False)
    kk_h = np.zeros(n_chains, np.int32)
    from glabcmcmc_amd import engine
    gc = engine.ChainBatch(torch.from_numpy(theta0), torch.from_numpy(y0), torch.device("cuda", 0), chain0=7)
    kk_g = torch.zeros(n_chains, dtype=torch.int32, device="cuda")
    for it in range(1, 8):
        lqo = (rng.standard_normal(n_chains) - 2).astype(np.float32)
        hrow = np.zeros((2, n_chains), np.float32)
        pool_h = A.Pool(theta.ctypes.data, x.ctypes.data, w.ctypes.data, lqo.ctypes.data, kk_h.ctypes.data, step_size, 0)
        run_h, keep = oracle_lib.make_run(seed=5, step0=it, n_steps=1, gf=0.7, batch=N, history=hrow)
        cs = hc.struct()
        assert oracle.oracle_glmcmc_nf_step(C.byref(model), C.byref(local), C.byref(pool_h), C.byref(cs), C.byref(run_h)) == 0
        lg2 = torch.from_numpy(lqo).cuda()
        hg = torch.empty(2, n_chains, device="cuda")
        pool_g = A.Pool(tg.data_ptr(), xg.data_ptr(), wg.data_ptr(), lg2.data_ptr(), kk_g.data_ptr(), step_size, 0)
        run_g = A.Run()
        run_g.seed, run_g.step0, run_g.n_steps, run_g.global_frequency, run_g.batch_size = 5, it, 1, 0.7, N
        run_g.history, run_g.hist_stride = hg.data_ptr(), n_chains
        csg = gc.struct()
        assert hip.glabc_glmcmc_nf_step(C.byref(model), C.byref(local), C.byref(pool_g), C.byref(csg), C.byref(run_g), None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(bits(hg.cpu().numpy()), bits(hrow)), it
        assert np.array_equal(kk_g.cpu().numpy(), kk_h)
        assert np.array_equal(bits(gc.y.cpu().numpy()), bits(hc.y))
    assert hc.n_moves.sum() > 0 and np.array_equal(gc.n_moves.cpu().numpy().astype(np.uint32), hc.n_moves)


@pytest.mark.gpu
def test_glmcmc_nf_end_to_end(hip):
    """GLMCMC_NF on the GPU: shapes of the reference API, the flow trains (forward-KL drops), chains reach the posterior."""
    import glabcmcmc_amd as g
    from glabcmcmc_amd.examples.Mixture import Mixture_set
    m = Mixture_set(0.3)
    lp = g.DiagGaussian(2, torch.zeros(1, 2), torch.log(torch.tensor([0.35, 0.35])))
    st = {}
    out = g.GLMCMC_NF(m, 120, torch.zeros(256, 2), torch.full((256, 2), 1.5), lp, None, 0.7, 4, 5, None, 10,
                      num_layers=4, seed=1, state_out=st, lr=5e-3)
    assert out.shape == (120, 256, 2) and torch.isfinite(out).all()
    assert st["num_train"] >= 5 and np.isfinite(st["loss_hist"]).all()
    assert st["loss_hist"][-1] < st["loss_hist"][0]                      # the proposal moves toward the weighted pool
    late = out[60:].abs().mean().item()
    assert 1.0 < late < 1.9                                              # |theta| near 1.4-1.5 (prior gives 0.8)
    one = g.GLMCMC_NF(m, 30, torch.tensor([0.0, 0.0]), torch.tensor([[1.5, 1.5]]), lp, None, 0.5, 3, 5, None, 2,
                      num_layers=2, seed=2, verbose=False)
    assert one.shape == (30, 2) and one.device.type == "cpu"


@pytest.mark.gpu
def test_log_prob_cache_follows_the_moved_chains(hip, oracle):
    """GLMCMC_NF keeps NF_model.log_prob(Theta_old) (GLMCMC_NFs.py:96-98) per chain and refreshes it only for the chains
    glabc_glmcmc_nf_step reports as moved (glabc_nf_log_prob_indexed, count read on the device): after every iteration the
    cache must equal, bit for bit, a from-scratch evaluation of all current states, and the moved list must be the oracle's."""
    import oracle_lib
    from glabcmcmc_amd import engine
    model, local, _ = _mix_descs()
    flow = make_flow(3, 21)
    fh, blob_h = host_descriptor(flow)
    fg_blob = flow.packed_params().cuda()
    fg = flow.descriptor(fg_blob)
    rng = np.random.default_rng(4)
    n, N, step_size, gf = 1500, 4, 6, 0.7
    rows = N * step_size * n
    theta = (rng.standard_normal((2, rows)) * 1.3).astype(np.float32)
    x = (np.abs(theta) + 0.2236 * rng.standard_normal((2, rows))).astype(np.float32)
    w = np.exp(rng.standard_normal(rows)).astype(np.float32)
    tg, xg, wg = (torch.from_numpy(a).cuda() for a in (theta, x, w))
    theta0 = rng.standard_normal((n, 2)).astype(np.float32)
    y0 = np.abs(theta0).astype(np.float32)
    hc = oracle_lib.HostChains(theta0, y0, chain0=3, with_isir=False)
    gc = engine.ChainBatch(torch.from_numpy(theta0), torch.from_numpy(y0), torch.device("cuda", 0), chain0=3)
    kk_h, kk_g = np.zeros(n, np.int32), torch.zeros(n, dtype=torch.int32, device="cuda")
    moved_h, n_moved_h = np.zeros(n, np.int32), np.zeros(1, np.int32)
    moved_g = torch.zeros(n, dtype=torch.int32, device="cuda")
    n_moved_g = torch.zeros(2, dtype=torch.int32, device="cuda")

    def full_log_prob(theta_cm):
        out = np.empty(n, np.float32)
        assert oracle.oracle_nf_log_prob(C.byref(fh), np.ascontiguousarray(theta_cm).ctypes.data, n, out.ctypes.data) == 0
        return out

    cache = torch.empty(n, dtype=torch.float32, device="cuda")
    assert hip.glabc_nf_log_prob(C.byref(fg), gc.theta.data_ptr(), n, cache.data_ptr(), None) == 0
    total = 0
    for it in range(1, step_size + 3):
        lqo = full_log_prob(hc.theta)
        assert np.array_equal(bits(cache.cpu().numpy()), bits(lqo)), it
        n_moved_h[0] = 0
        pool_h = A.Pool(theta.ctypes.data, x.ctypes.data, w.ctypes.data, lqo.ctypes.data, kk_h.ctypes.data, step_size, 0,
                        moved_h.ctypes.data, n_moved_h.ctypes.data, None)
        run_h, keep = oracle_lib.make_run(seed=8, step0=it, n_steps=1, gf=gf, batch=N)
        cs = hc.struct()
        assert oracle.oracle_glmcmc_nf_step(C.byref(model), C.byref(local), C.byref(pool_h), C.byref(cs), C.byref(run_h)) == 0
        cur, nxt = n_moved_g[it & 1:], n_moved_g[(it + 1) & 1:]
        pool_g = A.Pool(tg.data_ptr(), xg.data_ptr(), wg.data_ptr(), cache.data_ptr(), kk_g.data_ptr(), step_size, 0,
                        moved_g.data_ptr(), cur.data_ptr(), nxt.data_ptr())
        run_g = A.Run()
        run_g.seed, run_g.step0, run_g.n_steps, run_g.global_frequency, run_g.batch_size = 8, it, 1, gf, N
        csg = gc.struct()
        assert hip.glabc_glmcmc_nf_step(C.byref(model), C.byref(local), C.byref(pool_g), C.byref(csg), C.byref(run_g), None) == 0
        assert hip.glabc_nf_log_prob_indexed(C.byref(fg), gc.theta.data_ptr(), n, moved_g.data_ptr(), cur.data_ptr(), n,
                                             cache.data_ptr(), None) == 0
        k = int(cur[0].item())
        assert k == int(n_moved_h[0]) and int(nxt[0].item()) == 0
        assert np.array_equal(np.sort(moved_g[:k].cpu().numpy()), np.sort(moved_h[:k]))
        assert np.array_equal(bits(gc.theta.cpu().numpy()), bits(hc.theta))
        total += k
    assert total > n // 2
    assert np.array_equal(bits(cache.cpu().numpy()), bits(full_log_prob(hc.theta)))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["torch", "compiled"])
def test_glmcmc_nf_with_a_callback_model_has_the_law_of_the_fused_path(hip, kind):
    """run_glmcmc_nf with a Model that has no built-in descriptor -- a user's plain-torch class, and a CompiledModel (C
    simulator) -- goes through generic.run_glmcmc_nf (pools and local moves evaluated by the Model's own methods, selection by
    glabc_propose / glabc_select, the flow and its training by the same kernels): same shapes / side effects, the flow
    trains, and pooled E|theta|, E theta^2 and the move rate agree with the fused path within Monte-Carlo error."""
    import glabcmcmc_amd as g
    from glabcmcmc_amd.examples.Mixture import Mixture_set
    from glabcmcmc_amd.examples.UserModel import TorchMixture
    eps, n, T, gf, S, N, train = 0.3, 4096, 100, 0.7, 4, 5, 6
    lp = g.DiagGaussian(2, torch.zeros(1, 2), torch.log(torch.tensor([0.35, 0.35])))
    if kind == "torch":
        user = TorchMixture(2, eps)
    else:
        src = ("GLABC_SIMULATOR void glabc_user_simulate(const float* theta, const float* eps, float* y)\n"
               "{ for (int j = 0; j < 2; ++j) y[j] = fabsf(theta[j]) + 0.2236068f * eps[j]; }\n")
        user = g.CompiledModel(2, 2, src, g.DiagGaussian(2, torch.zeros(2), torch.zeros(2)), [1.5, 1.5], eps)
    gen = torch.Generator().manual_seed(3)
    th0 = 1.3 * (torch.randint(0, 2, (n, 2), generator=gen).float() * 2 - 1)
    y0 = th0.abs() + 0.2236 * torch.randn(n, 2, generator=gen)
    res = {}
    for name, m in (("fused", Mixture_set(eps)), ("generic", user)):
        st = {}
        torch.manual_seed(5)
        out = g.MCMCRunner(m).run_glmcmc_nf(T + 1, th0, y0, gf, lp, None, N, S, train, output_file=None, num_layers=4, seed=7,
                                            state_out=st, lr=5e-3, verbose=False, return_device=True)
        assert out.shape == (T + 1, n, 2) and torch.isfinite(out).all()
        assert st["num_train"] == train and np.isfinite(st["loss_hist"]).all() and st["loss_hist"][-1] < st["loss_hist"][0]
        late = out[T // 2:]
        res[name] = (late.abs().mean(dim=(0, 2)).cpu().numpy(), (late ** 2).mean(dim=(0, 2)).cpu().numpy(),
                     st["chains"].n_moves.cpu().numpy().astype(np.float64) / T)
    assert "callback_device" in st                                             # the second run took the callback path
    for a, b, what in zip(res["fused"], res["generic"], ("E|theta|", "E theta^2", "move rate")):
        se = np.sqrt(a.var(ddof=1) / n + b.var(ddof=1) / n)
        assert abs(a.mean() - b.mean()) < 5 * se, (what, a.mean(), b.mean(), se)
    one = g.GLMCMC_NF(user, 30, torch.tensor([0.0, 0.0]), torch.tensor([[1.5, 1.5]]), lp, None, 0.5, 3, 5, None, 2,
                      num_layers=2, seed=2, verbose=False)
    assert one.shape == (30, 2) and one.device.type == "cpu"


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["fused", "generic"])
def test_sharded_glmcmc_nf_keeps_one_flow(hip, kind):
    """chains sharded over two ranks (gloo, both on this GPU) that share the flow: the ranks refresh their pools together,
    average their gradients (row-weighted) and end with bit-identical flows and the same loss history"""
    import json
    import os
    import subprocess
    import sys
    from conftest import PKG_PARENT
    env = dict(os.environ, PYTHONPATH=PKG_PARENT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "nf_sharded.py")
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29547" if kind == "fused" else "29548", script, kind],
                         env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    rows = sorted([json.loads(l) for l in run.stdout.split("\n") if l.startswith("{")], key=lambda r: r["rank"])
    assert [r["rank"] for r in rows] == [0, 1] and rows[0]["n"] + rows[1]["n"] == 1000
    assert rows[0]["flow_sha"] == rows[1]["flow_sha"] and rows[0]["loss"] == rows[1]["loss"]
    assert rows[0]["num_train"] == rows[1]["num_train"] == 4 and rows[0]["pools"] == rows[1]["pools"]
    assert all(r["finite"] and 1.0 < r["mean_abs"] < 1.9 for r in rows)
