"""Host mirror of the reference's example Model (examples/Mixture.py:5-53).

``Mixture_set`` keeps the duck-typed Model protocol the samplers call --
``theta_dim, y_dim, y_obs, epsilon`` and ``generate_samples / prior_log_prob /
discrepancy / calculate_log_kernel[_dis]`` -- and adds ``descriptor()``: the
``glabc_model`` struct through which the fused gfx950 kernels evaluate the same
callbacks in registers.  The callbacks run the HIP kernels on CUDA tensors and
the reference's torch formulas on CPU tensors.
"""
import numpy as np
import torch

from .. import _capi
from .. import distribution
from ..distribution import _fill, _launch_rowwise, _launch_simulate


class Mixture_set:
    def __init__(self, epsilon, prior=None):
        self.epsilon = epsilon
        self.theta_dim = 2
        self.y_obs = torch.tensor([[1.5, 1.5]])
        self.y_dim = self.y_obs.shape[1]
        # an addition: another prior than the reference's N(0, I) (Mixture.py:30), any distribution.* object with a descriptor --
        # e.g. distribution.Gamma, which the fused kernels evaluate in double (include/glabc.h GLABC_DIST_GAMMA)
        self._prior_override = prior

    # the simulator's noise and the prior, as the reference constructs them per call
    def _likelihood(self):                      # Mixture.py:19
        return distribution.DiagGaussian(self.theta_dim, torch.tensor([0.0, 0.0]),
                                         torch.log(torch.tensor([0.05, 0.05]).sqrt()))

    def _prior(self):                           # Mixture.py:30
        if self._prior_override is not None:
            return self._prior_override
        return distribution.DiagGaussian(self.theta_dim, torch.tensor([0.0, 0.0]), torch.tensor([0.0, 0.0]))

    def _kernel(self, epsilon):                 # Mixture.py:42-43
        return distribution.DiagGaussian(1, loc=torch.tensor([0.0]), log_scale=torch.log(torch.tensor([epsilon])))

    def generate_samples(self, theta, num_samples=1):
        """Mixture.py:13-26: |theta| + N(0, 0.05 I); three shape regimes."""
        num_theta = 1 if theta.dim() == 1 else theta.shape[0]
        lik = self._likelihood()
        dev = theta.device
        if num_theta == 1:
            return torch.abs(theta) + lik.sample(num_samples).to(dev)
        if num_samples == 1:
            return torch.abs(theta) + lik.sample(num_theta).to(dev)
        dim_theta = theta.shape[1]
        noise = lik.sample(num_samples * num_theta).view(num_theta, num_samples, dim_theta).to(dev)
        return torch.abs(theta).unsqueeze(1).repeat(1, num_samples, 1) + noise

    @property
    def noise_dim(self):
        """standard normals one simulation consumes (generic.py hands them over from the run's Philox stream)"""
        return self.y_dim

    def simulate_from_noise(self, theta, eps):
        """generate_samples(theta, 1) with the simulator's standard normals supplied: |theta| + (loc + scale*eps)"""
        if theta.is_cuda:
            return _launch_simulate(self.descriptor(), theta, eps)
        lik = self._likelihood()
        return torch.abs(theta) + (lik.loc + torch.exp(lik.log_scale) * eps)

    def prior_log_prob(self, samples):
        samples = samples.view(-1, self.theta_dim)
        if samples.is_cuda:
            return _launch_rowwise("glabc_model_prior_log_prob", self.descriptor(), samples, "prior_log_prob")
        return self._prior().log_prob(samples)

    def discrepancy(self, y):
        y = y.view(-1, self.y_dim)
        if y.is_cuda:
            return _launch_rowwise("glabc_model_discrepancy", self.descriptor(), y, "discrepancy")
        self.y_obs = self.y_obs.view(-1, self.y_dim)
        return torch.sqrt(torch.sum((y - self.y_obs) ** 2, dim=1))

    def calculate_log_kernel(self, y, epsilon=None):
        if epsilon is None:
            epsilon = self.epsilon
        if y.is_cuda:
            return _launch_rowwise("glabc_model_log_kernel", self.descriptor(epsilon), y.view(-1, self.y_dim),
                                   "calculate_log_kernel")
        return self._kernel(epsilon).log_prob(self.discrepancy(y).view(-1, 1))

    def calculate_log_kernel_dis(self, dis, epsilon=None):
        if epsilon is None:
            epsilon = self.epsilon
        return self._kernel(epsilon).log_prob(dis.cpu().view(-1, 1)).to(dis.device)

    def descriptor(self, epsilon=None):
        """glabc_model (include/glabc.h) with every constant computed the way the reference
        computes it on the host (float32 torch ops), so the kernels reuse the same bits."""
        if epsilon is None:
            epsilon = self.epsilon
        m = _capi.Model()
        m.sim_kind = _capi.SIM_ABS_GAUSS
        m.theta_dim, m.y_dim = self.theta_dim, self.y_dim
        m.prior = self._prior().descriptor()
        m.noise = self._likelihood().descriptor()
        _fill(m.y_obs, self.y_obs.reshape(-1))
        kern = self._kernel(epsilon).descriptor()
        m.kern_log_scale, m.kern_scale, m.kern_c0 = kern.p1[0], kern.p2[0], kern.c0
        m.epsilon = float(np.float32(epsilon))
        return m


def main(num_ite=1000, output_dir='./', verbose=True):
    """The reference's example script (examples/Mixture.py:55-85), every sampler enabled -- the reference keeps all but
    run_glmcmc commented out.  Returns the five chains."""
    from ..MCMCRunner import MCMCRunner
    from ..ESJD import esjd
    from ..flows import BaseDiagGaussian
    Model = Mixture_set(epsilon=0.05)
    torch.manual_seed(0)
    np.random.seed(0)
    theta0 = torch.tensor([0.0, 0.0])
    y0 = Model.generate_samples(theta0)
    lp = distribution.DiagGaussian(2, loc=torch.zeros(1, 2), log_scale=torch.log(torch.tensor([0.35, 0.35])))
    ip = distribution.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.0, 0.0]))
    gp = distribution.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.0, 0.0]))
    gp_base = BaseDiagGaussian(2)                                   # nf.distributions.base.DiagGaussian(2), Mixture.py:71
    runner = MCMCRunner(Model, output_dir=output_dir)
    kw = dict(verbose=verbose)
    chains = {
        "global": runner.run_global_mcmc(num_ite, theta0, y0, 0.5, lp, gp, output_file='global_mcmc_results.csv', **kw),
        "glmcmc": runner.run_glmcmc(num_ite, theta0, y0, 0.9, lp, ip, 5, output_file='glmcmc_results.csv', **kw),
        "aglmcmc": runner.run_aglmcmc(num_ite, theta0, y0, 1, lp, ip, 5, 200, 0.8, 0.2, output_file='aglmcmc_results.csv', **kw),
        "glmala": runner.run_glmala(num_ite, theta0, y0, 0.8, ip, 5, 0.3, 100, output_file='glmala_results.csv', **kw),
        "glmcmc_nf": runner.run_glmcmc_nf(num_ite, theta0, y0, 0.5, lp, gp_base, 5, 200, 50,
                                          output_file='glmcmc_nf_results.csv', **kw),
    }
    if verbose:
        for name, chain in chains.items():
            print(name, "ESJD", esjd(chain))
    return chains


if __name__ == "__main__":
    main()
