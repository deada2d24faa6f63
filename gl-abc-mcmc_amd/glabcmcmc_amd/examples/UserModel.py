"""A user's own Model: nothing but the reference's duck-typed protocol (examples/Mixture.py:5-53), written in plain torch.

No ``descriptor()``: the samplers cannot compile it into the fused kernels and run it through the split-phase path
(``generic.py``: ``glabc_propose`` -> these callbacks on one (batch_size * n_chains, dim) batch -> ``glabc_select``).
The arithmetic is the d-dimensional version of the reference's example -- y = |theta| + N(0, 0.05 I), prior N(0, I),
Gaussian ABC kernel on the Euclidean distance to y_obs = (1.5, ..., 1.5) -- so its posterior is known in closed form
 and ``bench.py --workload callback`` / the tests can check what the loop produces.

    python -m glabcmcmc_amd.examples.UserModel            # 65 536 chains through MCMCRunner.run_glmcmc
"""
import math

import torch


class TorchMixture:
    def __init__(self, theta_dim=2, epsilon=0.05):
        self.theta_dim = self.y_dim = theta_dim
        self.epsilon = epsilon
        self.y_obs = torch.full((1, theta_dim), 1.5)
        self._y_obs_on = {}                      # y_obs per device: no host -> device copy inside the sampler loop

    def generate_samples(self, theta, num_samples=1):
        theta = theta.reshape(-1, self.theta_dim)
        return theta.abs() + math.sqrt(0.05) * torch.randn_like(theta)

    def prior_log_prob(self, samples):
        samples = samples.reshape(-1, self.theta_dim)
        return -0.5 * self.theta_dim * math.log(2 * math.pi) - 0.5 * (samples ** 2).sum(1)

    def discrepancy(self, y):
        y = y.reshape(-1, self.y_dim)
        if y.device not in self._y_obs_on:
            self._y_obs_on[y.device] = self.y_obs.to(y.device)
        return ((y - self._y_obs_on[y.device]) ** 2).sum(1).sqrt()

    def calculate_log_kernel(self, y, epsilon=None):
        e = self.discrepancy(y) / (self.epsilon if epsilon is None else epsilon)
        return -0.5 * math.log(2 * math.pi) - math.log(self.epsilon if epsilon is None else epsilon) - 0.5 * e * e

    def analytic_second_moment(self):
        v = 0.05 + self.epsilon ** 2
        return (1.5 / (1 + v)) ** 2 + v / (1 + v)


def main(n_chains=65536, num_ite=300):
    from .. import distribution, engine
    from ..MCMCRunner import MCMCRunner
    model = TorchMixture(2, 0.05)
    lp = distribution.DiagGaussian(2, torch.zeros(2), torch.log(torch.tensor([0.35, 0.35])))
    ip = distribution.DiagGaussian(2, torch.zeros(2), torch.zeros(2))
    theta0 = torch.zeros(n_chains, 2)
    y0 = model.generate_samples(theta0)
    stats = engine.Moments(n_chains, 2, engine.require_device())
    MCMCRunner(model).run_glmcmc(num_ite, theta0, y0, 0.9, lp, ip, 5, output_file=None, record_history=False, stats=stats,
                                 verbose=False)
    print("E theta^2 = %s   (stationary value %.4f)" % (stats.second_moment().diagonal(dim1=1, dim2=2).mean(0).tolist(),
                                                      model.analytic_second_moment()))


if __name__ == "__main__":
    main()
