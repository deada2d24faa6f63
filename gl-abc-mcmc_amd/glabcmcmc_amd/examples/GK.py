"""g-and-k Model for BASELINE.json config 4 ("GLMCMC on g-and-k style simulator, dim=4").

The reference tree contains no such model (its only Model is examples/Mixture.py), so this one is the build's own,
written against the reference's duck-typed Model protocol (examples/Mixture.py:5-53):

    theta = (A, B, g, k),  prior Uniform(0, 10)^4
    one simulation = the ORDER STATISTICS of y_dim = 8 draws of the g-and-k distribution,
        y_j = A + B (1 + c tanh(g z_j / 2)) (1 + z_j^2)^k z_j,   z_j ~ N(0, 1),   c = 0.8
    y_obs = that quantile function for (A, B, g, k) = (3, 1, 2, 0.5) at the normal scores of (j - 0.5)/8
    discrepancy = Euclidean distance between sorted simulation and y_obs; Gaussian ABC kernel of width epsilon

``descriptor()`` hands the fused gfx950 kernels the same model as a ``glabc_model`` with ``GLABC_SIM_GK``.
CPU tensors evaluate the torch formulas below (also the Model that tests/golden/make_golden.py drives through the
reference's own GLMCMC loop).
"""
import numpy as np
import torch

from .. import _capi
from .. import distribution
from ..distribution import _fill, _launch_rowwise, _launch_simulate

Y_DIM = 8
TRUE_THETA = (3.0, 1.0, 2.0, 0.5)


def gk_quantile(z, A, B, g, k, c=0.8):
    return A + B * (1 + c * torch.tanh(g * z / 2)) * (1 + z ** 2) ** k * z


class GK_set:
    def __init__(self, epsilon, c=0.8):
        self.epsilon = epsilon
        self.c = c
        self.theta_dim = 4
        self.y_dim = Y_DIM
        p = (torch.arange(1, Y_DIM + 1, dtype=torch.float64) - 0.5) / Y_DIM
        z = torch.special.ndtri(p).to(torch.float32)
        self.y_obs = gk_quantile(z, *TRUE_THETA, c=c).view(1, -1)

    def _prior(self):
        return distribution.Uniform(4, torch.zeros(4), torch.full((4,), 10.0))

    def _kernel(self, epsilon):
        return distribution.DiagGaussian(1, loc=torch.tensor([0.0]), log_scale=torch.log(torch.tensor([epsilon])))

    def generate_samples(self, theta, num_samples=1):
        theta = theta.reshape(-1, 4)
        n = theta.shape[0] * (num_samples if theta.shape[0] == 1 else 1) if num_samples > 1 and theta.shape[0] == 1 else theta.shape[0]
        th = theta.expand(n, 4) if theta.shape[0] == 1 else theta
        z = torch.randn((n, self.y_dim), dtype=torch.float32).to(theta.device)
        y = gk_quantile(z, th[:, 0:1], th[:, 1:2], th[:, 2:3], th[:, 3:4], self.c)
        return torch.sort(y, dim=1).values

    noise_dim = Y_DIM

    def simulate_from_noise(self, theta, eps):
        """generate_samples(theta, 1) with the y_dim standard normals supplied"""
        if theta.is_cuda:
            return _launch_simulate(self.descriptor(), theta, eps)
        th = theta.reshape(-1, 4)
        return torch.sort(gk_quantile(eps, th[:, 0:1], th[:, 1:2], th[:, 2:3], th[:, 3:4], self.c), dim=1).values

    def prior_log_prob(self, samples):
        samples = samples.view(-1, self.theta_dim)
        if samples.is_cuda:
            return _launch_rowwise("glabc_model_prior_log_prob", self.descriptor(), samples, "prior_log_prob")
        return self._prior().log_prob(samples)

    def discrepancy(self, y):
        y = y.view(-1, self.y_dim)
        if y.is_cuda:
            return _launch_rowwise("glabc_model_discrepancy", self.descriptor(), y, "discrepancy")
        return torch.sqrt(torch.sum((y - self.y_obs) ** 2, dim=1))

    def calculate_log_kernel(self, y, epsilon=None):
        if epsilon is None:
            epsilon = self.epsilon
        if y.is_cuda:
            return _launch_rowwise("glabc_model_log_kernel", self.descriptor(epsilon), y.view(-1, self.y_dim),
                                   "calculate_log_kernel")
        return self._kernel(epsilon).log_prob(self.discrepancy(y).view(-1, 1))

    def descriptor(self, epsilon=None):
        if epsilon is None:
            epsilon = self.epsilon
        m = _capi.Model()
        m.sim_kind = _capi.SIM_GK
        m.theta_dim, m.y_dim, m.gk_c = self.theta_dim, self.y_dim, float(np.float32(self.c))
        m.prior = self._prior().descriptor()
        m.noise = distribution.DiagGaussian(Y_DIM, torch.zeros(Y_DIM), torch.zeros(Y_DIM)).descriptor()    # unused by GK
        _fill(m.y_obs, self.y_obs.reshape(-1))
        kern = self._kernel(epsilon).descriptor()
        m.kern_log_scale, m.kern_scale, m.kern_c0 = kern.p1[0], kern.p2[0], kern.c0
        m.epsilon = float(np.float32(epsilon))
        return m
