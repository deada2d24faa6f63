"""RealNVP proposal of GLMCMC_NF (reference: GLMCMC_NFs.py:51-61) without the normflows dependency.

The reference builds ``nf.NormalizingFlow(base, [AffineCouplingBlock(MLP([1,128,128,2], init_zeros=True)),
Permute(2, 'swap')] * num_layers)`` from the third-party ``normflows`` package (not vendored, not installable
offline).  ``RealNVP`` restates that model's published semantics as a plain ``torch.nn.Module``:

* ``sample(n)`` and ``log_prob(x)`` on a CUDA flow run the hand-written matrix-core kernels behind
  ``glabc_nf_sample`` / ``glabc_nf_log_prob`` (exact-float32 MFMA, include/glabc.h); ``sample_torch`` /
  ``log_prob_torch`` are the same maps in eager PyTorch (CPU use, cross-checks);
* the training step of GLMCMC_NFs.py:112-124 on a CUDA flow is ``HipAdam.step(x)``: loss and gradient by the hand-written
  backward kernels (``glabc_nf_grad``) and ``glabc_adam_step`` on the packed parameters; ``forward_kld`` + autograd on stock
  PyTorch ops is the cross-check of the tests (and what a CPU flow uses).
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _capi

HIDDEN = _capi.NF_HIDDEN


class BaseDiagGaussian(nn.Module):
    """nf.distributions.base.DiagGaussian(2): trainable loc / log_scale of shape (1, d), zeros."""

    def __init__(self, d=2):
        super().__init__()
        self.d = d
        self.loc = nn.Parameter(torch.zeros(1, d))
        self.log_scale = nn.Parameter(torch.zeros(1, d))

    def forward(self, num_samples=1):
        eps = torch.randn(num_samples, self.d, dtype=self.loc.dtype, device=self.loc.device)
        z = self.loc + torch.exp(self.log_scale) * eps
        log_p = -0.5 * self.d * np.log(2 * np.pi) - torch.sum(self.log_scale + 0.5 * torch.pow(eps, 2), 1)
        return z, log_p

    def log_prob(self, z):
        return -0.5 * self.d * np.log(2 * np.pi) - torch.sum(
            self.log_scale + 0.5 * torch.pow((z - self.loc) / torch.exp(self.log_scale), 2), 1)


class Coupling(nn.Module):
    """AffineCouplingBlock(MLP([1, 128, 128, 2], init_zeros=True)) for theta_dim = 2."""

    def __init__(self):
        super().__init__()
        self.l1, self.l2, self.l3 = nn.Linear(1, HIDDEN), nn.Linear(HIDDEN, HIDDEN), nn.Linear(HIDDEN, 2)
        nn.init.zeros_(self.l3.weight)                  # init_zeros=True, GLMCMC_NFs.py:56
        nn.init.zeros_(self.l3.bias)

    def params(self, z0):
        p = self.l3(torch.relu(self.l2(torch.relu(self.l1(z0)))))      # LeakyReLU(0.0) == ReLU
        return p[:, 0:1], p[:, 1:2]                     # shift = param[:, 0::2], log-scale = param[:, 1::2]


class RealNVP(nn.Module):
    def __init__(self, num_layers=32, base=None):
        super().__init__()
        self.q0 = base if base is not None else BaseDiagGaussian(2)
        self.couplings = nn.ModuleList([Coupling() for _ in range(num_layers)])

    # ---- eager PyTorch (autograd) ------------------------------------------------------------------
    def sample_torch(self, num_samples=1, eps=None):
        if eps is None:
            z, log_q = self.q0(num_samples)
        else:
            z = self.q0.loc + torch.exp(self.q0.log_scale) * eps
            log_q = -0.5 * 2 * np.log(2 * np.pi) - torch.sum(self.q0.log_scale + 0.5 * torch.pow(eps, 2), 1)
        for c in self.couplings:
            z0, z1 = z[:, 0:1], z[:, 1:2]
            shift, log_s = c.params(z0)
            z1 = z1 * torch.exp(log_s) + shift
            log_q = log_q - log_s[:, 0]
            z = torch.cat([z1, z0], dim=1)              # Permute(2, 'swap')
        return z, log_q

    def log_prob_torch(self, x):
        z = x
        log_q = torch.zeros(x.shape[0], dtype=x.dtype, device=x.device)
        for c in reversed(self.couplings):
            z0, z1 = z[:, 1:2], z[:, 0:1]               # undo the swap
            shift, log_s = c.params(z0)
            z1 = (z1 - shift) * torch.exp(-log_s)
            log_q = log_q - log_s[:, 0]
            z = torch.cat([z0, z1], dim=1)
        return log_q + self.q0.log_prob(z)

    def forward_kld(self, x):
        return -torch.mean(self.log_prob_torch(x))

    # ---- matrix-core kernels -----------------------------------------------------------------------
    def packed_params(self):
        """[n_couplings][GLABC_NF_COUPLING_FLOATS] float32 blob in the layout of include/glabc.h"""
        blocks = []
        for c in self.couplings:
            w2t = c.l2.weight.detach().t().contiguous().reshape(-1)            # W2^T [k][i]
            v4 = torch.stack([c.l2.bias.detach(), c.l3.weight.detach()[0], c.l3.weight.detach()[1],
                              torch.zeros_like(c.l2.bias)], dim=1).reshape(-1)
            b3 = torch.cat([c.l3.bias.detach(), torch.zeros(2, device=c.l3.bias.device)])
            blocks.append(torch.cat([w2t, c.l1.weight.detach()[:, 0], c.l1.bias.detach(), v4, b3]))
        blob = torch.stack(blocks).to(torch.float32).contiguous()
        assert blob.shape[1] == _capi.NF_COUPLING_FLOATS
        return blob

    @torch.no_grad()
    def load_packed(self, blob, base=None):
        """the inverse of packed_params: parameters <- blob [n_couplings][GLABC_NF_COUPLING_FLOATS] (and base = loc0, loc1,
        log_scale0, log_scale1)"""
        H = HIDDEN
        for c, blk in zip(self.couplings, blob):
            c.l2.weight.copy_(blk[:H * H].view(H, H).t())
            c.l1.weight.copy_(blk[H * H:H * H + H].view(H, 1))
            c.l1.bias.copy_(blk[H * H + H:H * H + 2 * H])
            v4 = blk[H * H + 2 * H:H * H + 6 * H].view(H, 4)
            c.l2.bias.copy_(v4[:, 0])
            c.l3.weight.copy_(v4[:, 1:3].t())
            c.l3.bias.copy_(blk[H * H + 6 * H:H * H + 6 * H + 2])
        if base is not None:
            self.q0.loc.copy_(base[0:2].view(1, 2))
            self.q0.log_scale.copy_(base[2:4].view(1, 2))

    def descriptor(self, blob):
        f = _capi.Flow()
        f.n_couplings, f.hidden, f.params = len(self.couplings), HIDDEN, blob.data_ptr()
        loc = self.q0.loc.detach().reshape(-1).float().cpu()
        ls = self.q0.log_scale.detach().reshape(-1).float().cpu()
        sc = torch.exp(ls)
        for j in range(2):
            f.base_loc[j], f.base_log_scale[j], f.base_scale[j] = float(loc[j]), float(ls[j]), float(sc[j])
        f.base_c0 = float(np.float32(-0.5 * 2 * np.log(2 * np.pi)))
        return f

    def _device(self):
        return next(self.parameters()).device

    @torch.no_grad()
    def sample(self, num_samples=1, eps=None, seed=0, row0=0):
        """NF_model.sample(n) -> (z (n, 2), log_q (n,)).  On CUDA: glabc_nf_sample (MFMA); eps (n, 2) optional
        base noise, otherwise Philox(seed, row0 + r)."""
        dev = self._device()
        if dev.type != "cuda":
            return self.sample_torch(num_samples, eps)
        blob = self.packed_params()
        f = self.descriptor(blob)
        n = int(num_samples)
        e = None if eps is None else eps.detach().to(dev, torch.float32).t().contiguous()
        z = torch.empty(2, n, dtype=torch.float32, device=dev)
        lq = torch.empty(n, dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _capi.check(_capi.lib().glabc_nf_sample(C.byref(f), None if e is None else e.data_ptr(), int(seed), int(row0), n,
                                                    z.data_ptr(), lq.data_ptr(), C.c_void_p(stream)), "glabc_nf_sample")
        return z.t(), lq

    @torch.no_grad()
    def log_prob(self, x):
        dev = self._device()
        if dev.type != "cuda":
            return self.log_prob_torch(x)
        blob = self.packed_params()
        f = self.descriptor(blob)
        xx = x.detach().to(dev, torch.float32).reshape(-1, 2).t().contiguous()
        lq = torch.empty(xx.shape[1], dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _capi.check(_capi.lib().glabc_nf_log_prob(C.byref(f), xx.data_ptr(), xx.shape[1], lq.data_ptr(),
                                                      C.c_void_p(stream)), "glabc_nf_log_prob")
        return lq


class HipAdam:
    """torch.optim.Adam(NF_model.parameters(), lr, weight_decay) of GLMCMC_NFs.py:63 with the step of :112-124 on the device:
    ``step(x)`` = zero_grad; loss = forward_kld(x); backward unless the loss is NaN / inf; optimizer.step() -- loss and
    gradient from ``glabc_nf_grad`` (hand-written backward on the matrix cores), the update from ``glabc_adam_step`` on the
    packed parameter blob; the module's parameters are refreshed from the blob afterwards.  (With a NaN / inf loss the
    reference's ``optimizer.step()`` finds no gradients and changes nothing; neither does this.)"""

    def __init__(self, flow, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5):
        dev = flow._device()
        if dev.type != "cuda":
            raise RuntimeError("HipAdam drives the HIP training kernels: the flow must be on a cuda (HIP) device")
        self.flow, self.lr, self.betas, self.eps, self.weight_decay = flow, lr, betas, eps, weight_decay
        self.blob = flow.packed_params()
        self.base = torch.cat([flow.q0.loc.detach().reshape(-1), flow.q0.log_scale.detach().reshape(-1)]).float().contiguous()
        self.state = [torch.zeros_like(t) for t in (self.blob, self.blob, self.base, self.base)]    # exp_avg, exp_avg_sq x 2
        self.grad_blob, self.grad_base = torch.zeros_like(self.blob), torch.zeros_like(self.base)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.steps = 0
        self._ws = None

    def gradient(self, x, chain_major=False):
        """x (n, 2) -- or (2, n) with chain_major -- -> (loss tensor, grad blob, grad base) on the device; nothing is updated"""
        flow, dev = self.flow, self.flow._device()
        xx = x.detach().to(dev, torch.float32)
        xx = (xx if chain_major else xx.reshape(-1, 2).t()).contiguous()
        n = xx.shape[1]
        f = flow.descriptor(self.blob)
        b = torch.cat([self.base, torch.exp(self.base[2:4])]).tolist()        # loc0, loc1, log_scale0, log_scale1, scale0, scale1
        f.base_loc[0], f.base_loc[1], f.base_log_scale[0], f.base_log_scale[1], f.base_scale[0], f.base_scale[1] = b
        need = C.c_int64()
        _capi.check(_capi.lib().glabc_nf_grad_workspace(f.n_couplings, n, C.byref(need)), "glabc_nf_grad_workspace")
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _capi.check(_capi.lib().glabc_nf_grad(C.byref(f), xx.data_ptr(), n, self._ws.data_ptr(), self._ws.numel(),
                                                  self.grad_blob.data_ptr(), self.grad_base.data_ptr(), self.loss.data_ptr(),
                                                  C.c_void_p(stream)), "glabc_nf_grad")
        return self.loss, self.grad_blob, self.grad_base

    def step(self, x, chain_major=False, group=None, via=None):
        """group: a torch.distributed process group (or True for the default one) whose ranks share this flow -- the
        gradient and the loss become the row-weighted means over all ranks' batches before the update (parallel.py)"""
        loss, gb, gq = self.gradient(x, chain_major)
        if group is not None:
            from .parallel import average_gradients
            n_local = x.shape[1] if chain_major else x.reshape(-1, 2).shape[0]
            average_gradients([gb, gq, loss], n_local, None if group is True else group, via)
        value = float(loss)                                      # one synchronisation per training step (GLMCMC_NFs.py:121)
        if value != value or value in (float("inf"), float("-inf")):
            return value
        self.steps += 1
        dev = self.flow._device()
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            for p, g, m, v in ((self.blob, gb, self.state[0], self.state[1]), (self.base, gq, self.state[2], self.state[3])):
                _capi.check(_capi.lib().glabc_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), self.lr,
                                                        self.betas[0], self.betas[1], self.eps, self.weight_decay, self.steps,
                                                        C.c_void_p(stream)), "glabc_adam_step")
        self.flow.load_packed(self.blob, self.base)
        return value
