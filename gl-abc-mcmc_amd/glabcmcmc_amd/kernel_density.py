"""KernelDensity -- weighted Gaussian KDE, the adaptive proposal of AGLMCMC (reference: kernel_density.py:4-177).

Same class surface as the reference (``fit / log_prob / sample / forward``, ``bandwidth`` = 'silverman' | 'scott' |
float | tensor), executed by the gfx950 kernels of csrc/glabc_kde.hip through the C ABI (include/glabc.h):

* ``fit``       -> ``glabc_kde_fit``      weights, weighted std, bandwidth, log-weights (one workgroup, float64 sums)
* ``log_prob``  -> ``glabc_kde_log_prob`` the dense (points x centres) logsumexp, one wavefront per point
* ``sample``    -> ``glabc_kde_sample``   inverse-CDF centre index + bandwidth * normal from the Philox stream

There is no PyTorch evaluation path: the estimator lives on a HIP device.  What torch does here is plumbing --
transposes into the kernels' dimension-major layout and one integer prefix sum.
"""
import ctypes as C

import torch

from . import _capi, engine


class KernelDensity:
    def __init__(self, bandwidth='silverman', device=None, seed=None):
        self.bandwidth = bandwidth
        self.device = device
        self.X = None
        self.weights = None
        self.n_samples = 0
        self.dim = None
        self._fitted = False
        self._seed = seed
        self._rows_drawn = 0
        self._desc = None

    def _rule_factor(self, n):
        """kernel_density.py:24-33 (Python floats)"""
        if self.bandwidth == 'silverman':
            return (n * (self.dim + 2) / 4.) ** (-1. / (self.dim + 4))
        if self.bandwidth == 'scott':
            return n ** (-1. / (self.dim + 4))
        raise ValueError("bandwidth should be 'silverman', 'scott' or a float")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def fit(self, X, weights=None):
        """X (n_samples, n_features), optional weights (n_samples,) -- kernel_density.py:70-94."""
        dev = self.device = engine.require_device(self.device)
        X = torch.as_tensor(X, dtype=torch.float32).detach().to(dev)
        self.n_samples, self.dim = X.shape
        if not 1 <= self.dim <= _capi.MAX_DIM:
            raise ValueError("KernelDensity supports 1..%d features" % _capi.MAX_DIM)
        n, d = self.n_samples, self.dim
        self.X = X
        self._x = X.t().contiguous()                                           # [dim][n]
        w_raw = None if weights is None else torch.as_tensor(weights, dtype=torch.float32).detach().to(dev).contiguous()
        if w_raw is not None and w_raw.numel() != n:
            raise ValueError("weights has %d entries for %d samples" % (w_raw.numel(), n))
        h, bw_fixed = 0.0, None
        if isinstance(self.bandwidth, str):
            h = self._rule_factor(n)
        else:
            bw = torch.as_tensor(self.bandwidth, dtype=torch.float32).detach().cpu().reshape(-1)
            bw = bw.expand(d) if bw.numel() == 1 else bw
            bw_fixed = (C.c_float * d)(*[float(v) for v in bw])
        self.weights = torch.empty(n, dtype=torch.float32, device=dev)
        self._log_w = torch.empty(n, dtype=torch.float32, device=dev)
        wq = torch.empty(n, dtype=torch.int64, device=dev)
        consts = torch.empty(d + 2, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _capi.check(_capi.lib().glabc_kde_fit(self._x.data_ptr(), None if w_raw is None else w_raw.data_ptr(), n, d, float(h),
                                                  bw_fixed, self.weights.data_ptr(), self._log_w.data_ptr(), wq.data_ptr(),
                                                  consts.data_ptr(), self._stream()), "glabc_kde_fit")
        self._cum_q = torch.cumsum(wq, 0)                                      # integers: exact in any order
        c = consts.cpu()
        self._consts = [float(v) for v in c]
        self.bandwidth = c[:d].to(dev)                                         # kernel_density.py:90-92
        self._fitted = True
        self._desc = None
        return self

    def descriptor(self):
        """glabc_kde (include/glabc.h)"""
        if not self._fitted:
            raise RuntimeError("Must call fit() before computing probabilities")
        if self._desc is not None:
            return self._desc
        k = _capi.Kde()
        k.dim, k.n_samples = self.dim, self.n_samples
        k.x, k.log_w, k.cum_q = self._x.data_ptr(), self._log_w.data_ptr(), self._cum_q.data_ptr()
        for j in range(self.dim):
            k.bandwidth[j] = self._consts[j]
        k.sum_log_bw, k.c_2pi = self._consts[self.dim], self._consts[self.dim + 1]
        self._desc = k
        return k

    def log_prob_soa(self, pts):
        """pts dimension-major [dim][n_points] on the device (the samplers' chain-major state) -> (n_points,)"""
        k = self.descriptor()
        n_points = pts.shape[1]
        out = torch.empty(n_points, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _capi.check(_capi.lib().glabc_kde_log_prob(C.byref(k), pts.data_ptr(), n_points, out.data_ptr(), self._stream()),
                        "glabc_kde_log_prob")
        return out

    def log_prob_soa_indexed(self, pts, idx, n_dev, out):
        """the same for the points idx[0 .. *n_dev - 1] of pts only (idx, n_dev: int32 device tensors; the count never visits
        the host), results written to out[idx]; the other entries of `out` are left as they are"""
        k = self.descriptor()
        with torch.cuda.device(self.device):
            _capi.check(_capi.lib().glabc_kde_log_prob_indexed(C.byref(k), pts.data_ptr(), pts.shape[1], idx.data_ptr(), n_dev.data_ptr(),
                                                               pts.shape[1], out.data_ptr(), self._stream()), "glabc_kde_log_prob_indexed")
        return out

    def log_prob(self, x):
        """x (n_points, n_features) -> (n_points,) log densities -- kernel_density.py:96-128."""
        if not self._fitted:
            raise RuntimeError("Must call fit() before computing probabilities")
        x = torch.as_tensor(x, dtype=torch.float32).detach().to(self.device).reshape(-1, self.dim)
        return self.log_prob_soa(x.t().contiguous())

    def sample_soa(self, n_samples, seed=None, row0=None):
        """-> dimension-major [dim][n_samples] draws"""
        k = self.descriptor()
        if seed is None:
            if self._seed is None:
                self._seed = engine.draw_seed(None)
            seed = self._seed
        if row0 is None:
            row0 = self._rows_drawn
            self._rows_drawn += int(n_samples)
        out = torch.empty(self.dim, int(n_samples), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _capi.check(_capi.lib().glabc_kde_sample(C.byref(k), int(n_samples), seed, int(row0), out.data_ptr(), self._stream()),
                        "glabc_kde_sample")
        return out

    def sample(self, n_samples=1, return_log_prob=False, seed=None, row0=None):
        """kernel_density.py:130-156"""
        if not self._fitted:
            raise RuntimeError("Must call fit() before sampling")
        soa = self.sample_soa(n_samples, seed, row0)
        samples = soa.t().contiguous()
        if return_log_prob:
            return samples, self.log_prob_soa(soa)
        return samples

    def forward(self, n_samples=1):
        """kernel_density.py:158-177 (the reference scales the noise by exp(log(bandwidth)); here by the bandwidth)"""
        return self.sample(n_samples, return_log_prob=True)
