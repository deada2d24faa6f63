"""The training step of GLMCMC_NF's flow (GLMCMC_NFs.py:63,112-124: loss = forward_kld = -mean(log_prob), Adam with L2 weight
decay).  The reference differentiates with autograd; the build has a hand-written backward on the matrix cores.

CPU: the checker's gradient (oracle_nf_grad: the float32 evaluation's states and ReLU gates, the chain rule in double) against
     torch autograd in float64 of the same model (cases without a pre-activation within rounding of zero); oracle_adam_step
     against torch.optim.Adam.
GPU: glabc_nf_grad against the checker (floating-point tolerance: every gradient tensor within 2e-4 of its largest entry,
     the loss within 2e-6 relative -- float32 matrix-core sums over the rows against exact sums) and against float32 torch
     autograd on the device; bit-reproducibility from run to run; glabc_adam_step == checker bit for bit; whole training
     steps (HipAdam) against torch autograd + torch.optim.Adam.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from glabcmcmc_amd.flows import RealNVP
from test_nf import host_descriptor

H = 128


def trained_looking_flow(n_couplings, seed):
    """every parameter away from its initial value (the reference's init has W3 = 0: no gradient flows into the MLP)"""
    torch.manual_seed(seed)
    flow = RealNVP(n_couplings)
    with torch.no_grad():
        for c in flow.couplings:
            c.l3.weight.normal_(0, 0.4 / H ** 0.5)
            c.l3.bias.normal_(0, 0.1)
            c.l1.bias.normal_(0, 0.3)
            c.l2.bias.normal_(0, 0.1)
        flow.q0.loc.copy_(torch.tensor([[0.2, -0.1]]))
        flow.q0.log_scale.copy_(torch.tensor([[0.1, -0.2]]))
    return flow


def autograd_gradient(flow, x, dtype):
    """(loss, grad blob in the packed layout, grad base) by torch autograd"""
    import copy
    g = copy.deepcopy(flow).to(dtype)
    loss = g.forward_kld(x.to(dtype))
    loss.backward()
    blocks = []
    for c in g.couplings:
        v4 = torch.stack([c.l2.bias.grad, c.l3.weight.grad[0], c.l3.weight.grad[1], torch.zeros_like(c.l2.bias)], dim=1).reshape(-1)
        blocks.append(torch.cat([c.l2.weight.grad.t().reshape(-1), c.l1.weight.grad[:, 0], c.l1.bias.grad, v4,
                                 c.l3.bias.grad, torch.zeros(2, dtype=dtype, device=v4.device)]))
    return float(loss.detach()), torch.stack(blocks), torch.cat([g.q0.loc.grad.reshape(-1), g.q0.log_scale.grad.reshape(-1)])


def sections(blob):
    """the tensors of a packed gradient, by name (a tolerance per tensor, not per blob)"""
    b = np.asarray(blob).reshape(-1, blob.shape[-1])
    v4 = b[:, H * H + 2 * H:H * H + 6 * H].reshape(-1, H, 4)
    return {"W2": b[:, :H * H], "W1": b[:, H * H:H * H + H], "b1": b[:, H * H + H:H * H + 2 * H], "b2": v4[:, :, 0],
            "W3": v4[:, :, 1:3], "b3": b[:, H * H + 6 * H:H * H + 6 * H + 2], "pad": np.concatenate([v4[:, :, 3], b[:, H * H + 6 * H + 2:]], 1)}


def assert_close(got, want, tol, what):
    for name, w in sections(want).items():
        g = sections(got)[name]
        scale = np.abs(w).max()
        if name == "pad":
            assert not g.any() and not w.any(), what
            continue
        assert scale > 0, (what, name)
        err = np.abs(g - w).max()
        assert err <= tol * scale, "%s %s: max |diff| %.3g vs scale %.3g" % (what, name, err, scale)


def oracle_gradient(oracle, flow, x_rows):
    f, blob = host_descriptor(flow)
    xx = np.ascontiguousarray(x_rows.numpy().T.astype(np.float32))
    gp = np.zeros_like(blob)
    gb = np.zeros(4, np.float32)
    loss = np.zeros(1, np.float32)
    assert oracle.oracle_nf_grad(C.byref(f), xx.ctypes.data, xx.shape[1], gp.ctypes.data, gb.ctypes.data, loss.ctypes.data) == 0
    return float(loss[0]), gp, gb


@pytest.mark.parametrize("n_couplings,n", [(1, 50), (3, 301), (8, 64)])
def test_oracle_gradient_equals_autograd_float64(oracle, n_couplings, n):
    flow = trained_looking_flow(n_couplings, 3 + n_couplings)
    x = torch.randn(n, 2, generator=torch.Generator().manual_seed(n)) * 1.3
    loss_o, gp, gb = oracle_gradient(oracle, flow, x)
    loss_t, gpt, gbt = autograd_gradient(flow, x, torch.float64)
    assert abs(loss_o - loss_t) <= 2e-6 * abs(loss_t)
    assert_close(gp, gpt.numpy(), 2e-6, "oracle vs autograd f64")          # the oracle returns float32
    assert np.allclose(gb, gbt.numpy(), rtol=2e-6, atol=1e-8)


def test_oracle_gradient_of_the_reference_initialisation(oracle):
    """init_zeros=True (GLMCMC_NFs.py:56): W3 = 0, so only W3 / b3 and the base receive gradient at the first step"""
    flow = RealNVP(4)
    x = torch.randn(200, 2, generator=torch.Generator().manual_seed(1))
    loss_o, gp, gb = oracle_gradient(oracle, flow, x)
    loss_t, gpt, gbt = autograd_gradient(flow, x, torch.float64)
    s, t = sections(gp), sections(gpt.numpy())
    for name in ("W2", "W1", "b1", "b2"):
        assert not s[name].any() and not t[name].any()
    assert np.allclose(s["W3"], t["W3"], rtol=1e-5, atol=1e-9) and np.allclose(s["b3"], t["b3"], rtol=1e-5, atol=1e-9)
    assert np.allclose(gb, gbt.numpy(), rtol=1e-5, atol=1e-8) and abs(loss_o - loss_t) < 1e-6


def test_oracle_adam_equals_torch_adam(oracle):
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal(5000).astype(np.float32)
    p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([p], lr=5e-4, weight_decay=1e-5)               # GLMCMC_NFs.py:63
    q, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for step in range(1, 6):
        g = (rng.standard_normal(5000) * 10.0 ** rng.uniform(-6, 1, 5000)).astype(np.float32)
        p.grad = torch.from_numpy(g.copy())
        opt.step()
        assert oracle.oracle_adam_step(q.ctypes.data, g.ctypes.data, m.ctypes.data, v.ctypes.data, q.size, 5e-4, 0.9, 0.999, 1e-8,
                                       1e-5, step) == 0
        assert np.allclose(q, p.detach().numpy(), rtol=0, atol=2e-7 * 5e-4 + 1e-7 * np.abs(p0).max())
    assert np.abs(q - p0).max() > 1e-3                                    # it moved


# ---------------------------------------------------------------------------------------------------------------- GPU
def hip_gradient(flow_dev, x_rows):
    from glabcmcmc_amd.flows import HipAdam
    opt = HipAdam(flow_dev)
    loss, gp, gb = opt.gradient(x_rows.cuda())
    torch.cuda.synchronize()
    return float(loss), gp.cpu().numpy().copy(), gb.cpu().numpy().copy(), opt


@pytest.mark.gpu
@pytest.mark.parametrize("n_couplings,n", [(1, 100), (3, 4096 + 37), (8, 1000), (2, 70001), (32, 300)])
def test_hip_gradient_matches_the_checker(hip, oracle, n_couplings, n):
    flow = trained_looking_flow(n_couplings, 11 + n_couplings)
    x = torch.randn(n, 2, generator=torch.Generator().manual_seed(n)) * 1.3
    loss_o, gp, gb = oracle_gradient(oracle, flow, x)
    loss_h, gph, gbh, opt = hip_gradient(flow.cuda(), x)
    assert abs(loss_h - loss_o) <= 2e-6 * abs(loss_o), (loss_h, loss_o)
    assert_close(gph, gp, 2e-4, "hip vs checker")
    assert np.allclose(gbh, gb, rtol=2e-4, atol=2e-6)
    # the same call again: the same bits (fixed-order reduction over workgroups)
    loss2, gp2, gb2 = opt.gradient(x.cuda())
    torch.cuda.synchronize()
    assert np.array_equal(gp2.cpu().numpy().view(np.uint32), gph.view(np.uint32)) and float(loss2) == loss_h


@pytest.mark.gpu
def test_hip_gradient_matches_float32_autograd_on_the_device(hip):
    flow = trained_looking_flow(8, 5).cuda()
    x = (torch.randn(20000, 2, generator=torch.Generator().manual_seed(2)) * 1.3).cuda()
    loss_t, gpt, gbt = autograd_gradient(flow, x, torch.float32)
    loss_h, gph, gbh, _ = hip_gradient(flow, x.cpu())
    assert abs(loss_h - loss_t) <= 1e-5 * abs(loss_t)
    assert_close(gph, gpt.cpu().numpy(), 1e-3, "hip vs torch autograd f32")   # two float32 summation orders
    assert np.allclose(gbh, gbt.cpu().numpy(), rtol=1e-3, atol=1e-5)


@pytest.mark.gpu
def test_hip_adam_equals_the_checker_bit_for_bit(hip, oracle):
    from glabcmcmc_amd import _capi
    rng = np.random.default_rng(1)
    n = 100003
    p0 = rng.standard_normal(n).astype(np.float32)
    q, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    dp, dm, dv = torch.from_numpy(p0.copy()).cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 5):
        g = (rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 1, n)).astype(np.float32)
        dg = torch.from_numpy(g).cuda()
        stream = torch.cuda.current_stream().cuda_stream
        _capi.check(hip.glabc_adam_step(dp.data_ptr(), dg.data_ptr(), dm.data_ptr(), dv.data_ptr(), n, 5e-4, 0.9, 0.999, 1e-8, 1e-5,
                                        step, C.c_void_p(stream)), "glabc_adam_step")
        assert oracle.oracle_adam_step(q.ctypes.data, g.ctypes.data, m.ctypes.data, v.ctypes.data, n, 5e-4, 0.9, 0.999, 1e-8, 1e-5,
                                       step) == 0
        assert np.array_equal(dp.cpu().numpy().view(np.uint32), q.view(np.uint32))
        assert np.array_equal(dv.cpu().numpy().view(np.uint32), v.view(np.uint32))
    assert hip.glabc_adam_step(dp.data_ptr(), dg.data_ptr(), dm.data_ptr(), dv.data_ptr(), n, 5e-4, 0.9, 0.999, 1e-8, 1e-5, 0, None) == -4


@pytest.mark.gpu
def test_hip_training_steps_follow_torch_training_steps(hip):
    """GLMCMC_NFs.py:114-124 three times: HipAdam.step against forward_kld + backward + torch.optim.Adam.step on a copy"""
    import copy
    from glabcmcmc_amd.flows import HipAdam
    flow = trained_looking_flow(4, 9).cuda()
    ref = copy.deepcopy(flow)
    topt = torch.optim.Adam(ref.parameters(), lr=5e-4, weight_decay=1e-5)
    opt = HipAdam(flow, lr=5e-4, weight_decay=1e-5)
    gen = torch.Generator().manual_seed(4)
    losses = []
    for k in range(3):
        x = (torch.randn(30000, 2, generator=gen) * 0.7 + torch.tensor([0.5, -0.3])).cuda()
        topt.zero_grad()
        lt = ref.forward_kld(x)
        lt.backward()
        topt.step()
        lh = opt.step(x)
        losses.append((lh, float(lt.detach())))
        assert abs(lh - float(lt.detach())) <= 2e-5 * abs(float(lt.detach()))
    for a, b in zip(flow.parameters(), ref.parameters()):
        # one Adam step moves a parameter by ~lr whatever the size of its gradient: entries whose gradient is at the noise
        # level of the float32 sums may differ by a fraction of lr, everything else by far less
        d = (a.detach() - b.detach()).abs()
        assert float(d.max()) <= 3 * 5e-4 and float(d.mean()) <= 2e-5, (float(d.max()), float(d.mean()))
    assert opt.steps == 3 and torch.equal(flow.packed_params(), opt.blob)
    # a NaN loss changes nothing (the reference skips backward and its optimizer.step() finds no gradients)
    before = opt.blob.clone()
    bad = x.clone()
    bad[0, 0] = float("nan")
    assert opt.step(bad) != opt.step(bad) and opt.steps == 3 and torch.equal(opt.blob, before)


@pytest.mark.gpu
def test_hip_training_lowers_the_forward_kl(hip):
    """from the reference's initialisation (identity flow) towards a shifted, correlated target: the loss falls"""
    from glabcmcmc_amd.flows import HipAdam
    torch.manual_seed(0)
    flow = RealNVP(8).cuda()
    opt = HipAdam(flow, lr=2e-3, weight_decay=1e-5)
    gen = torch.Generator().manual_seed(5)
    base = torch.randn(40000, 2, generator=gen)
    x = torch.stack([0.8 * base[:, 0] + 1.0, 0.5 * base[:, 1] + 0.6 * base[:, 0] - 0.5], dim=1).cuda()
    first = opt.step(x)
    for _ in range(60):
        last = opt.step(x)
    assert last < first - 0.3, (first, last)
    # and the flow that the samplers use has the trained parameters
    assert abs(float(-flow.log_prob(x).mean()) - float(opt.gradient(x)[0])) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["1", "4"])
def test_hip_gradient_other_kernel_forms(hip, variant):
    """GLABC_NF_BW selects the form of the backward kernel once per process (8 = default; 4 = the sign-bit kernel as two 4-wave
    workgroups per CU; 1 = the form that stages a2 itself): the measuring knobs pass the same gradient check, in a child
    process of their own"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GLABC_NF_BW=variant)
    here = os.path.abspath(__file__)
    run = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-x", "-m", "gpu", "-k",
                          "test_hip_gradient_matches_the_checker and (3-4133 or 2-70001)"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert run.returncode == 0 and "2 passed" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]
