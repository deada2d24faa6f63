"""Host mirror of the reference's ``glabcmcmc.distribution`` (distribution.py:7-293).

Same class names, constructor arguments and ``forward / log_prob / sample``
protocol (distribution.py:16-48).  What is new:

* ``descriptor()`` -- the ``glabc_dist`` struct (include/glabc.h) that the fused
  sampler kernels take, so a proposal / prior object crosses the C ABI as plain
  numbers;
* on CUDA tensors ``log_prob`` runs the hand-written gfx950 kernel
  (``glabc_dist_log_prob``); there is no PyTorch path for device tensors.  On CPU
  tensors the methods evaluate the reference's formulas with torch ops, so user
  scripts that build ``y0`` or evaluate a density on the host keep working.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi


def _shape_tuple(shape):
    if isinstance(shape, int):
        return (shape,)
    return tuple(shape)


def _expand(param, shape):
    """Broadcast a parameter tensor to `shape` and flatten (float32, CPU)."""
    p = torch.as_tensor(param, dtype=torch.float32).detach().cpu()
    return (torch.zeros(shape, dtype=torch.float32) + p).reshape(-1)


def _fill(arr, values):
    for i, v in enumerate(values):
        arr[i] = float(v)


def _launch_rowwise(fn_name, desc, z, what):
    """Run a rows-in / scalars-out C-ABI kernel on a CUDA tensor."""
    lib = _capi.lib()
    z2 = z.detach().to(torch.float32).reshape(z.shape[0], -1).contiguous()
    out = torch.empty(z2.shape[0], dtype=torch.float32, device=z.device)
    stream = torch.cuda.current_stream(z.device).cuda_stream
    with torch.cuda.device(z.device):
        _capi.check(getattr(lib, fn_name)(C.byref(desc), z2.data_ptr(), z2.shape[0], out.data_ptr(),
                                          C.c_void_p(stream)), what)
    return out


def _launch_simulate(desc, theta, eps):
    """glabc_model_simulate on CUDA tensors: theta (n, theta_dim), eps (n, y_dim) standard normals -> y (n, y_dim)"""
    lib = _capi.lib()
    th = theta.detach().to(torch.float32).reshape(-1, desc.theta_dim).contiguous()
    ee = eps.detach().to(torch.float32).reshape(th.shape[0], desc.y_dim).contiguous()
    y = torch.empty(th.shape[0], desc.y_dim, dtype=torch.float32, device=theta.device)
    stream = torch.cuda.current_stream(theta.device).cuda_stream
    with torch.cuda.device(theta.device):
        _capi.check(lib.glabc_model_simulate(C.byref(desc), th.data_ptr(), ee.data_ptr(), th.shape[0], 0, 0, y.data_ptr(),
                                             C.c_void_p(stream)), "glabc_model_simulate")
    return y


class BaseDistribution:
    """distribution.py:7-48"""

    def __init__(self):
        super().__init__()

    def forward(self, num_samples=1):
        raise NotImplementedError

    def log_prob(self, z):
        raise NotImplementedError

    def sample(self, num_samples=1, **kwargs):
        z, _ = self.forward(num_samples, **kwargs)
        return z

    def descriptor(self):
        raise NotImplementedError("%s has no glabc_dist descriptor: it cannot be used inside the fused "
                                  "HIP samplers" % type(self).__name__)


class Uniform(BaseDistribution):
    """Multivariate uniform distribution (distribution.py:50-86)."""

    def __init__(self, shape, low=torch.tensor([-2.0]), high=torch.tensor([2.0])):
        super().__init__()
        self.shape = _shape_tuple(shape)
        self.low = low
        self.high = high
        # distribution.py:71 -- note the product runs over the parameter's own shape, so the
        # default (1,)-shaped bounds give -log(4) whatever `shape` is (reference behaviour).
        self.log_prob_val = -torch.log(torch.prod(self.high - self.low))

    def forward(self, num_samples=1, context=None):
        eps = torch.rand((num_samples,) + self.shape, dtype=self.low.dtype, device=self.low.device)
        z = self.low + (self.high - self.low) * eps
        log_p = self.log_prob_val * torch.ones(num_samples, device=self.low.device)
        return z, log_p

    def log_prob(self, z, context=None):
        if z.is_cuda:
            return _launch_rowwise("glabc_dist_log_prob", self.descriptor(), z, "Uniform.log_prob")
        log_p = self.log_prob_val * torch.ones(z.shape[0], device=z.device)
        out_range = torch.logical_or(z < self.low, z > self.high)
        ind_inf = torch.any(torch.reshape(out_range, (z.shape[0], -1)), dim=-1)
        log_p[ind_inf] = -np.inf
        return log_p

    def descriptor(self):
        d = int(np.prod(self.shape))
        if d > _capi.MAX_DIM:
            raise ValueError("Uniform dim %d > GLABC_MAX_DIM" % d)
        low, high = _expand(self.low, self.shape), _expand(self.high, self.shape)
        desc = _capi.Dist()
        desc.kind, desc.dim = _capi.DIST_UNIFORM, d
        _fill(desc.p0, low)
        _fill(desc.p1, high)
        _fill(desc.p2, high - low)                 # distribution.py:77 (high - low) in float32
        desc.c0 = float(torch.as_tensor(self.log_prob_val, dtype=torch.float32))
        return desc


class Gamma(BaseDistribution):
    """Multivariate independent Gamma distribution (distribution.py:90-137); float64, SciPy."""

    def __init__(self, Shape, Rate, device=None, seed=None):
        super().__init__()
        self.Shape = Shape.numpy()
        self.Rate = Rate.numpy()
        # device / seed are additions: with a CUDA device forward() draws on the GPU (glabc_gamma_forward: Marsaglia-Tsang
        # in double on the Philox stream of `seed`, a fresh block of rows per call); without, the reference's SciPy draw
        self.device = None if device is None else torch.device(device)
        self.seed = seed
        self._rows_drawn = 0

    def forward(self, num_samples=1, context=None, device=None, seed=None, row0=None):
        dev = self.device if device is None else torch.device(device)
        if dev is not None and dev.type == "cuda":
            desc = self.gamma_descriptor()
            if seed is None:
                if self.seed is None:
                    self.seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
                seed = self.seed
            if row0 is None:
                row0 = self._rows_drawn
                self._rows_drawn += int(num_samples)
            z = torch.empty(num_samples, desc.dim, dtype=torch.float64, device=dev)
            log_p = torch.empty(num_samples, dtype=torch.float64, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            with torch.cuda.device(dev):
                _capi.check(_capi.lib().glabc_gamma_forward(C.byref(desc), int(num_samples), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                            int(row0), z.data_ptr(), log_p.data_ptr(), C.c_void_p(stream)),
                            "Gamma.forward")
            return z.view((num_samples,) + tuple(self.Shape.shape)), log_p
        from scipy.stats import gamma
        size = (num_samples,) + tuple(self.Shape.shape)
        z = torch.tensor(gamma.rvs(self.Shape, scale=1 / self.Rate, size=size))
        return z, self.log_prob(z)

    def descriptor(self):
        """glabc_dist of kind GLABC_DIST_GAMMA (include/glabc.h): what the samplers take -- Gamma as the importance / global
        proposal or as a Model's prior inside the fused kernels and glabc_propose / glabc_select.  p0 = shape, p1 = rate,
        p2 = 1/rate as distribution.py:118,133 forms it (float32 division), p3 = scipy.special.gammaln(shape) in the dtype of
        the float32 Shape array (distribution.py:103)."""
        from scipy.special import gammaln
        shape = np.asarray(self.Shape, dtype=np.float32).reshape(-1)
        rate = np.asarray(self.Rate, dtype=np.float32).reshape(-1)
        scale = (np.float32(1) / rate).astype(np.float32)
        gl = np.asarray(gammaln(shape), dtype=np.float32).reshape(-1)
        if shape.size > _capi.MAX_DIM or rate.size != shape.size:
            raise ValueError("Gamma descriptor: 1..%d coordinates with one shape and one rate each" % _capi.MAX_DIM)
        d = _capi.Dist()
        d.kind, d.dim = _capi.DIST_GAMMA, int(shape.size)
        for j in range(d.dim):
            d.p0[j], d.p1[j], d.p2[j], d.p3[j] = float(shape[j]), float(rate[j]), float(scale[j]), float(gl[j])
        return d

    def gamma_descriptor(self):
        """glabc_gamma (include/glabc.h), the float64 entry points glabc_gamma_log_prob / glabc_gamma_forward: float64 shape,
        scale = 1/rate as distribution.py:133 forms it, gammaln(shape)"""
        from scipy.special import gammaln
        shape = np.asarray(self.Shape).reshape(-1)
        scale = np.asarray(1 / self.Rate).reshape(-1)
        d = _capi.GammaDesc()
        d.dim = int(shape.size)
        if d.dim > 3:
            raise ValueError("Gamma on the GPU supports up to 3 dimensions")
        # scipy evaluates gammaln in the dtype of the shape parameter: the reference stores Shape as the float32
        # array Shape.numpy() (distribution.py:103), so gammaln(shape) is a float32 number there
        gl = np.asarray(gammaln(self.Shape)).reshape(-1)
        for j in range(d.dim):
            d.shape[j], d.scale[j], d.gammaln[j] = float(shape[j]), float(scale[j]), float(gl[j])
        return d

    def log_prob(self, z, context=None):
        if z.is_cuda:
            desc = self.gamma_descriptor()
            zz = z.detach().to(torch.float64).reshape(z.shape[0], -1).contiguous()
            out = torch.empty(zz.shape[0], dtype=torch.float64, device=z.device)
            stream = torch.cuda.current_stream(z.device).cuda_stream
            with torch.cuda.device(z.device):
                _capi.check(_capi.lib().glabc_gamma_log_prob(C.byref(desc), zz.data_ptr(), zz.shape[0], out.data_ptr(),
                                                             C.c_void_p(stream)), "Gamma.log_prob")
            return out
        from scipy.stats import gamma
        p = gamma.pdf(z.cpu().numpy(), self.Shape, scale=1 / self.Rate)
        with np.errstate(divide="ignore"):
            log_p = np.where(p > 0, np.log(p), -np.inf)
        return torch.sum(torch.tensor(log_p), dim=1)


class DiagGaussian(BaseDistribution):
    """Multivariate Gaussian with diagonal covariance (distribution.py:143-203)."""

    def __init__(self, shape, loc, log_scale):
        super().__init__()
        self.shape = _shape_tuple(shape)
        self.n_dim = len(self.shape)
        self.d = np.prod(self.shape)
        self.loc = loc
        self.log_scale = log_scale

    def forward(self, num_samples=1, context=None):
        eps = torch.randn((num_samples,) + self.shape, dtype=self.loc.dtype, device=self.loc.device)
        z = self.loc + torch.exp(self.log_scale) * eps
        log_p = -0.5 * self.d * np.log(2 * np.pi) - torch.sum(
            self.log_scale + 0.5 * torch.pow(eps, 2), list(range(1, self.n_dim + 1)))
        return z, log_p

    def log_prob(self, z, context=None):
        if z.is_cuda:
            return _launch_rowwise("glabc_dist_log_prob", self.descriptor(), z, "DiagGaussian.log_prob")
        return -0.5 * self.d * np.log(2 * np.pi) - torch.sum(
            self.log_scale + 0.5 * torch.pow((z - self.loc) / torch.exp(self.log_scale), 2),
            list(range(1, self.n_dim + 1)))

    def cdf(self, z):
        from torch.distributions import Normal
        return torch.prod(Normal(self.loc, torch.exp(self.log_scale)).cdf(z), dim=-1)

    def register_buffer(self, param, param1):
        pass

    def descriptor(self):
        d = int(self.d)
        if d > _capi.MAX_DIM:
            raise ValueError("DiagGaussian dim %d > GLABC_MAX_DIM" % d)
        loc = _expand(self.loc, self.shape)
        log_scale = _expand(self.log_scale, self.shape)
        desc = _capi.Dist()
        desc.kind, desc.dim = _capi.DIST_DIAG_GAUSS, d
        _fill(desc.p0, loc)
        _fill(desc.p1, log_scale)
        # exp(log_scale) exactly as torch's float32 exp returns it: distribution.py:170,178
        _fill(desc.p2, torch.exp(log_scale))
        desc.c0 = float(np.float32(-0.5 * d * np.log(2 * np.pi)))      # distribution.py:171,177
        return desc


class GaussianMixture(BaseDistribution):
    """Mixture of diagonal Gaussians (distribution.py:206-293); API surface only, torch ops."""

    def __init__(self, n_modes, dim, loc=None, scale=None, weights=None):
        super().__init__()
        self.n_modes = n_modes
        self.dim = dim
        loc = np.random.randn(n_modes, dim) if loc is None else loc
        loc = np.array(loc)[None, ...]
        scale = np.ones((n_modes, dim)) if scale is None else scale
        scale = np.array(scale)[None, ...]
        weights = np.ones(n_modes) if weights is None else weights
        weights = np.array(weights, dtype=float)[None, ...]
        weights /= weights.sum(1)
        self.loc = torch.nn.Parameter(torch.tensor(1.0 * loc))
        self.log_scale = torch.nn.Parameter(torch.tensor(np.log(1.0 * scale)))
        self.weight_scores = torch.nn.Parameter(torch.tensor(np.log(1.0 * weights)))

    def _mode_log_p(self, z):
        weights = torch.softmax(self.weight_scores, 1)
        eps = (z[:, None, :] - self.loc) / torch.exp(self.log_scale)
        return (-0.5 * self.dim * np.log(2 * np.pi) + torch.log(weights)
                - 0.5 * torch.sum(torch.pow(eps, 2), 2) - torch.sum(self.log_scale, 2))

    def forward(self, num_samples=1):
        weights = torch.softmax(self.weight_scores, 1)
        mode = torch.multinomial(weights[0, :], num_samples, replacement=True)
        mode_1h = torch.nn.functional.one_hot(mode, self.n_modes)[..., None]
        eps_ = torch.randn(num_samples, self.dim, dtype=self.loc.dtype, device=self.loc.device)
        scale_sample = torch.sum(torch.exp(self.log_scale) * mode_1h, 1)
        loc_sample = torch.sum(self.loc * mode_1h, 1)
        z = eps_ * scale_sample + loc_sample
        return z, torch.logsumexp(self._mode_log_p(z), 1)

    def log_prob(self, z):
        if self.dim == 1 and z.dim() == 1:
            z = z[:, None]
        return torch.logsumexp(self._mode_log_p(z), 1)
